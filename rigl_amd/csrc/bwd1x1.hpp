// K1, single-pass backward of the big-M 1x1 convs (included inside namespace rigl::k1 of conv.hip).
//
// The 56x56 1x1 layers of ResNet-50 (64 <-> 256 channels, 401 408 pixels at batch 128) are HBM-bound, and their backward
// reads the output gradient TWICE: once for dX = dY W^T, once for dW = X^T dY -- 205 MB each time for the 64->256 layers,
// and the shared launch of the igemm / tr bodies takes exactly the sum of the two streams (84 us ~ 39 + 48).  Here a
// workgroup walks a range of pixels once: every 32-pixel K-tile of dY and X is brought into LDS by LDS-DMA, multiplied
// by the register-resident W for dX (v_mfma_f32_16x16x32_bf16, D = W-fragment x dY-fragment so a lane holds 4 consecutive
// input channels of one pixel) and by X^T for the workgroup's private dW accumulators (v_mfma_f32_32x32x16_bf16 on
// ds_read_b64_tr_b16 fragments), dX goes out tile by tile (+ the shortcut gradient), the dW partial once at the end into
// the workgroup's split-K slab (rigl::k1::launch_wgrad_reduce sums the slabs in a fixed order).  dY is read once.
// The dY tile serves both a row-major ds_read_b128 (k = output channel) and a transposing read (k = pixel): its 16-byte
// chunks are XORed with a row function found by exhaustive search over XOR-linear maps (tools/probes/swizzle_search.py)
// that is conflict-free for BOTH access patterns -- 512-byte rows: ((r & 1) << 1) | ((r >> 1 & 1) << 2) | ((r ^ r >> 2) & 1) << 3,
// 128-byte rows: ((r >> 1) & 1) << 1 | ((r >> 1 ^ r >> 2) & 1) << 2 -- applied on the DMA's source side.
// Memory-bound by ~3x (512 MFMA cycles per wave per 20 KB K-tile), so the loop is the plain ring: counted vmcnt, one
// barrier per K-tile, two 256-thread workgroups per CU.  Measured at batch 128 (gpurun r3p-r3s): 64->256 84 -> 70 us,
// 64->64 39 -> 32 us, 256->64 110 -> 109 us (level: off by default).  What mattered, in order: workgroups take every
// splits-th K-tile instead of a contiguous range (77 -> 70 us: the grid streams one contiguous window instead of 512
// streams 400 KB apart); dX leaves through LDS as whole 16-byte row segments; the stores of tile kt are issued at the top
// of iteration kt + 1 (stores and loads share vmcnt on gfx9) and the wait counts LOADS only.
// Reference: the autodiff of layers.masked_conv2d (pruning_layers.py:139-157; sparse_optimizers_base.py:478-485).
#pragma once

struct Bwd1x1Args {
  const uint16_t* X;    // [M][CI] bf16
  const uint16_t* DY;   // [M][CO] bf16
  const uint16_t* W;    // [CI][CO] bf16 (the HWIO shadow of a 1x1 kernel)
  const uint16_t* ADD;  // [M][CI] bf16 or NULL: added to dX (bf16(bf16(dgrad) + addend), like the dgrad epilogue)
  uint16_t* DX;         // [M][CI] bf16
  float* SLAB;          // [splits][CI][CO] fp32 partial dW (NULL with DO_W = false)
  int M, splits, interleave;
  uint32_t x_bytes, dy_bytes;
};

template <int ROWB>
__device__ __forceinline__ int dual_swz(int row) {
  if (ROWB == 512) return ((row & 1) << 1) | (((row >> 1) & 1) << 2) | (((row ^ (row >> 2)) & 1) << 3);
  return (((row >> 1) & 1) << 1) | ((((row >> 1) ^ (row >> 2)) & 1) << 2);
}

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int CI, int CO, bool DO_W, int NST>
struct Bwd1x1Smem { static constexpr int BYTES = NST * 32 * ((DO_W ? CI : 0) + CO) * 2 + (CI <= 64 ? 2 : 1) * 32 * (CI * 2 + 16); };
template <int CI, int CO, bool DO_W, int NST>
__global__ __launch_bounds__(THREADS) void k_bwd1x1(Bwd1x1Args P) {
  constexpr int PX = 32;
  constexpr int YROWB = CO * 2, XROWB = CI * 2;
  constexpr int Y_BYTES = PX * YROWB, X_BYTES = DO_W ? PX * XROWB : 0, STAGE = Y_BYTES + X_BYTES;
  constexpr int YI = YROWB / 32, XI = XROWB / 32;             // DMA wave-instructions per tile (1 KB each)
  constexpr int YPW = YI / 4 > 0 ? YI / 4 : 1, XPW = XI / 4 > 0 ? XI / 4 : 1;   // per wave
  static_assert(YI % 4 == 0 && XI % 4 == 0, "tiles are whole rounds of four waves");
  constexpr int L = YPW + (DO_W ? XPW : 0);                  // DMA instructions per thread per K-tile
  constexpr int CF = CI / 16, CFW = CF / 2;                  // dgrad: 16-channel fragments, per wave (2 pixel fragments x 2 wave columns)
  constexpr int KSD = CO / 32;                               // dgrad k-steps (output channels, 32 per MFMA)
  constexpr int NFI = CI / 32, NFO = CO / 32;                // wgrad: 32x32 fragments of dW
  // wgrad fragments per wave: the longer side is split over wave pairs
  constexpr bool WIDE_O = NFO >= NFI;
  constexpr int WF = NFI * NFO / 4 > 0 ? NFI * NFO / 4 : 1;  // fragments per wave
  static_assert(CFW * KSD * 4 <= 96, "W fragments stay in registers");
  // the dX tile of a K-tile is staged in LDS ([32 pixels][CI] bf16, rows padded by 16 bytes) so that it leaves as whole
  // 16-byte-per-lane row segments: straight from the accumulators a lane owns 8 bytes of 16 different pixels' rows
  // The tile of K-tile kt is written after its MFMAs and stored at the top of iteration kt + 1, behind that iteration's
  // barrier: on gfx9 stores count in vmcnt like loads, so stores issued just before the DMA wait would have to retire
  // (or part of the NEXT stage's DMA land) before the wait is satisfied.  Two staging tiles where they are small
  // (CI = 64); one tile and a second barrier per iteration for CI = 256.
  constexpr bool DXDB = CI <= 64;
  constexpr int NDX = DXDB ? 2 : 1;
  constexpr int DXROWB = XROWB + 16, DX_BYTES = PX * DXROWB;
  static_assert(NST * STAGE + NDX * DX_BYTES == Bwd1x1Smem<CI, CO, DO_W, NST>::BYTES, "host and device agree on the LDS size");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const dxs = smem + NST * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int KT_all = (P.M + PX - 1) / PX;
  // K-tiles of this workgroup: every splits-th one (interleave: at any moment the grid streams ONE contiguous window
  // of the tensors, splits x 16-20 KB, instead of `splits` streams a fixed 400 KB apart) or a contiguous range
  const int kt_begin = P.interleave ? split : (int)((int64_t)KT_all * split / P.splits);
  const int kt_step = P.interleave ? P.splits : 1;
  const int KT = P.interleave ? (split < KT_all ? (KT_all - split + P.splits - 1) / P.splits : 0)
                              : (int)((int64_t)KT_all * (split + 1) / P.splits) - kt_begin;
  const __amdgpu_buffer_rsrc_t rsrcY = make_rsrc(P.DY, P.dy_bytes), rsrcX = make_rsrc(P.X, P.x_bytes);
  const u32x4 rsrcY4 = make_rsrc4(P.DY, P.dy_bytes), rsrcX4 = make_rsrc4(P.X, P.x_bytes);
  (void)rsrcY4; (void)rsrcX4; (void)rsrcY; (void)rsrcX;

  // ---- DMA lanes ------------------------------------------------------------------------------------------------------
  // wave-instruction i of a tile with ROWB-byte rows fills rows i * (1024 / ROWB) ..; lane l: row + l / (ROWB / 16),
  // 16-byte slot l % (ROWB / 16), fetching the logical chunk slot ^ swizzle(row)
  int y_row[YPW], y_col[YPW], x_row[XPW], x_col[XPW];
#pragma unroll
  for (int q = 0; q < YPW; ++q) {
    const int i = q * 4 + wave, row = i * (1024 / YROWB) + lane / (YROWB / 16), slot = lane % (YROWB / 16);
    y_row[q] = row; y_col[q] = (slot ^ dual_swz<YROWB>(row)) * 8;
  }
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int i = q * 4 + wave, row = i * (1024 / XROWB) + lane / (XROWB / 16), slot = lane % (XROWB / 16);
    x_row[q] = row; x_col[q] = (slot ^ dual_swz<XROWB>(row)) * 8;
  }
// both streams (dY rows, X rows) are read once: non-temporal (-DRIGL_B1_NO_NT: without; in the step conv_bwd 3.601 / 3.598 ->
// 3.542 / 3.544 ms, 10.409 / 10.392 -> 10.344 / 10.354 ms)
#if !defined(RIGL_B1_NO_NT) && !defined(RIGL_DMA_BUILTIN)
#define B1_DMA16(r_, lds_, off_) lds_dma16_nt(r_##4, lds_, off_)
#else
#define B1_DMA16(r_, lds_, off_) RIGL_DMA16(r_, lds_, off_)
#endif
#define B1_ISSUE(kt_, stage_)                                                                            \
  {                                                                                                      \
    const int p0_ = (kt_begin + (kt_) * kt_step) * PX;                                                             \
    _Pragma("unroll") for (int q = 0; q < YPW; ++q) {                                                    \
      const int p_ = p0_ + y_row[q];                                                                     \
      const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * CO + y_col[q]) * 2u) : (int)OOB;                 \
      B1_DMA16(rsrcY, smem + (stage_) * STAGE + (q * 4 + wave) * 1024, off_); \
    }                                                                                                    \
    if (DO_W) {                                                                                          \
      _Pragma("unroll") for (int q = 0; q < XPW; ++q) {                                                  \
        const int p_ = p0_ + x_row[q];                                                                   \
        const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * CI + x_col[q]) * 2u) : (int)OOB;               \
        B1_DMA16(rsrcX, smem + (stage_) * STAGE + Y_BYTES + (q * 4 + wave) * 1024, off_); \
      }                                                                                                  \
    }                                                                                                    \
  }

  // ---- dgrad: W fragments in registers ------------------------------------------------------------------------------------
  // wave (pf = wave & 1, half = wave >> 1): pixel fragment pf (16 pixels), channel fragments half * CFW .. + CFW
  const int pf = wave & 1, chalf = wave >> 1;
  bf16x8 wfr[CFW][KSD];
#pragma unroll
  for (int c = 0; c < CFW; ++c)
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      const int ci = (chalf * CFW + c) * 16 + (lane & 15), co = ks * 32 + (lane >> 4) * 8;
      wfr[c][ks] = *reinterpret_cast<const bf16x8*>(P.W + (int64_t)ci * CO + co);
    }
  // dY fragment of k-step ks: lane l = row 16 * pf + (l & 15), logical chunk 4 * ks + (l >> 4)
  const int d_row = 16 * pf + (lane & 15);
  const int d_base = d_row * YROWB, d_swz = dual_swz<YROWB>(d_row), d_hi = lane >> 4;

  // ---- wgrad: transposing fragment reads -------------------------------------------------------------------------------------
  const int g = lane >> 4, j16 = lane & 15;
  const int t_row = 8 * (g >> 1) + (j16 >> 2);                      // + 16 * ks2 (+ 4 for the second half of the 8 pixels)
  const int t_low = 2 * (g & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
  // fragment assignment: the wave's WF fragments of the NFI x NFO grid
  //   WIDE_O: input fragment wave % NFI... (CI = 64: fi = wave & 1), output fragments (wave / NFI) * WF ..
  //   else  : output fragment wave % NFO, input fragments (wave / NFO) * WF ..
  f32x16 acc2[DO_W ? WF : 1];
  if (DO_W) {
#pragma unroll
    for (int f = 0; f < WF; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[f][e] = 0.f;
  }
  const int w_fix = WIDE_O ? wave % NFI : wave % NFO;               // the fragment index on the short side
  const int w_var0 = WIDE_O ? (wave / NFI) * WF : (wave / NFO) * WF;   // first fragment index on the long side
  // byte offsets (inside a tile) of the two 4-pixel halves of a fragment's rows, for k-step 0
#define B1_TR_OFF(ROWB_, chunk_, plus4_) \
  ((t_row + (plus4_)) * (ROWB_) + ((((chunk_) + t_low) ^ dual_swz<ROWB_>(t_row + (plus4_))) << 4) + t_half)

  constexpr int DXP = PX * (CI / 8) / THREADS;               // 16-byte pieces of a dX tile per thread
  static_assert(PX * (CI / 8) % THREADS == 0, "whole pieces per thread");
  // LOADS issued after stage kt's DMA when iteration kt waits for it (loads return in order among themselves; stores
  // may retire earlier, so only loads may be counted on): the DMA of NST - 2 later stages and, with a shortcut
  // gradient, NST - 1 addend tiles
  constexpr int W_N = L * (NST - 2), W_A = W_N + DXP * (NST - 1);
  uint4 addv[DXP];                                            // the shortcut gradient of the tile stored next
  // stores tile ktp (staged in LDS by every wave before the barrier this is called behind)
#define B1_FLUSH(ktp_, second_barrier_)                                                                    \
  {                                                                                                        \
    const int p0_ = (kt_begin + (ktp_) * kt_step) * PX;                                                              \
    const unsigned char* src_ = dxs + ((ktp_) & (NDX - 1)) * DX_BYTES;                                     \
    uint4 v_[DXP];                                                                                         \
    _Pragma("unroll") for (int q = 0; q < DXP; ++q) {                                                      \
      const int idx = q * THREADS + tid, row = idx / (CI / 8), ch = idx % (CI / 8);                        \
      v_[q] = *reinterpret_cast<const uint4*>(src_ + row * DXROWB + ch * 16);                              \
    }                                                                                                      \
    if (second_barrier_) __syncthreads();                                                                  \
    _Pragma("unroll") for (int q = 0; q < DXP; ++q) {                                                      \
      const int idx = q * THREADS + tid, row = idx / (CI / 8), ch = idx % (CI / 8);                        \
      if (p0_ + row < P.M) {                                                                               \
        uint4 v = v_[q];                                                                                   \
        if (P.ADD) {                                                                                       \
          const uint4 a = addv[q];                                                                         \
          v.x = add_bf16x2(v.x, a.x); v.y = add_bf16x2(v.y, a.y); v.z = add_bf16x2(v.z, a.z); v.w = add_bf16x2(v.w, a.w); \
        }                                                                                                  \
        *reinterpret_cast<uint4*>(P.DX + (int64_t)(p0_ + row) * CI + ch * 8) = v;                          \
      }                                                                                                    \
    }                                                                                                      \
  }
  for (int t = 0; t < NST - 1; ++t)
    if (t < KT) B1_ISSUE(t, t);
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + NST - 1 <= KT) {
      if (P.ADD && kt >= NST - 1) wait_vmcnt<W_A>();     // (the first NST - 1 iterations have fewer addend loads behind them)
      else wait_vmcnt<W_N>();
    } else {
      wait_vmcnt<0>();
    }
    // this wave's dX tile of the previous iteration (ds_write, read by OTHER waves in B1_FLUSH below) must have reached
    // the LDS before the barrier releases them: a raw s_barrier does not wait for LDS writes (round 4: dX wrong in ~1 run
    // of 10 at batch 128 -- the __syncthreads() this loop avoids, to keep the DMA in flight, had implied this wait)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // The counted waits above assume THIS program order of the vector-memory operations of an iteration: stage DMA, the
    // stores of the previous tile, the addend loads of this tile.  Nothing else orders them for the compiler (round 4:
    // with the branches of a removed knob gone it hoisted the addend loads above the DMA issue, the wait then left one
    // DMA piece of the tile being read in flight, and dX + addend was wrong in 1 run of 4) -- so the order is pinned.
    if (kt + NST - 1 < KT) B1_ISSUE(kt + NST - 1, (kt + NST - 1) % NST);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) B1_FLUSH(kt - 1, !DXDB);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* Ys = smem + (kt % NST) * STAGE;
    const unsigned char* Xs = Ys + Y_BYTES;
    if (P.ADD) {
      const int p0 = (kt_begin + kt * kt_step) * PX;
#pragma unroll
      for (int q = 0; q < DXP; ++q) {
        const int idx = q * THREADS + tid, row = idx / (CI / 8), ch = idx % (CI / 8);
        addv[q] = p0 + row < P.M ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)(p0 + row) * CI + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- dX tile: D[ci][px] = W-fragment x dY-fragment --------------------------------------------------------------------
    f32x4 acc1[CFW];
#pragma unroll
    for (int c = 0; c < CFW; ++c) acc1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {
      const bf16x8 yf = *reinterpret_cast<const bf16x8*>(Ys + d_base + (((4 * ks + d_hi) ^ d_swz) << 4));
#pragma unroll
      for (int c = 0; c < CFW; ++c) acc1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[c][ks], yf, acc1[c], 0, 0, 0);
    }
    // ---- dW partial: D2[ci][co] += X^T-fragment x dY^T-fragment, two k-steps of 16 pixels --------------------------------------
    if (DO_W) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 fa[WIDE_O ? 1 : WF], fb[WIDE_O ? WF : 1];
        if (WIDE_O) {
          fa[0] = lds_read_tr_pair(Xs + k2 * 16 * XROWB + B1_TR_OFF(XROWB, w_fix * 4, 0), Xs + k2 * 16 * XROWB + B1_TR_OFF(XROWB, w_fix * 4, 4));
#pragma unroll
          for (int f = 0; f < WF; ++f)
            fb[f] = lds_read_tr_pair(Ys + k2 * 16 * YROWB + B1_TR_OFF(YROWB, (w_var0 + f) * 4, 0),
                                     Ys + k2 * 16 * YROWB + B1_TR_OFF(YROWB, (w_var0 + f) * 4, 4));
#pragma unroll
          for (int f = 0; f < WF; ++f) acc2[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[f], acc2[f], 0, 0, 0);
        } else {
          fb[0] = lds_read_tr_pair(Ys + k2 * 16 * YROWB + B1_TR_OFF(YROWB, w_fix * 4, 0), Ys + k2 * 16 * YROWB + B1_TR_OFF(YROWB, w_fix * 4, 4));
#pragma unroll
          for (int f = 0; f < WF; ++f)
            fa[f] = lds_read_tr_pair(Xs + k2 * 16 * XROWB + B1_TR_OFF(XROWB, (w_var0 + f) * 4, 0),
                                     Xs + k2 * 16 * XROWB + B1_TR_OFF(XROWB, (w_var0 + f) * 4, 4));
#pragma unroll
          for (int f = 0; f < WF; ++f) acc2[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f], fb[0], acc2[f], 0, 0, 0);
        }
      }
    }
    // ---- dX tile -> LDS (lane (l & 15) = pixel, 4 consecutive channels (l >> 4) * 4 .. of each fragment) -------------------------
    {
      const int prow = 16 * pf + (lane & 15);
      unsigned char* const dst = dxs + (kt & (NDX - 1)) * DX_BYTES;
#pragma unroll
      for (int c = 0; c < CFW; ++c) {
        const f32x2 lo = {acc1[c][0], acc1[c][1]}, hi2 = {acc1[c][2], acc1[c][3]};
        uint2 pk;
        pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
        pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
        *reinterpret_cast<uint2*>(dst + prow * DXROWB + ((chalf * CFW + c) * 16 + (lane >> 4) * 4) * 2) = pk;
      }
    }
  }
  if (KT > 0) {
    __syncthreads();
    B1_FLUSH(KT - 1, false);
  }
#undef B1_FLUSH
#undef B1_ISSUE
#undef B1_TR_OFF
  // ---- the workgroup's dW partial -> its slab: D2 row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) = ci, column lane & 31 = co ------
  if (DO_W) {
    float* const out = P.SLAB + (int64_t)split * CI * CO;
#pragma unroll
    for (int f = 0; f < WF; ++f) {
      const int fi = WIDE_O ? w_fix : w_var0 + f, fo = WIDE_O ? w_var0 + f : w_fix;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = fi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), co = fo * 32 + (lane & 31);
        out[(int64_t)ci * CO + co] = acc2[f][e];
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Legal: 1x1, stride 1, no padding, (cin, cout) one of the instantiated pairs, a long pixel axis.  "bwd1x1" = 0 turns the
// kernel off (a layer's dX then comes from the igemm body again, in both entry points).
static inline int bwd1x1_kind(const RiglConvDesc* d) {
  if (d->kh != 1 || d->kw != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_top || d->pad_left) return 0;
  if (d->ho != d->h || d->wo != d->w) return 0;            // (a cropped output grid: the generic bodies walk ho x wo)
  if ((int64_t)d->n * d->h * d->w < 65536) return 0;
  if (RIGL_TUNE("bwd1x1", 1) == 0) return 0;
  if (d->cin == 64 && d->cout == 256) return 1;
  // (256 -> 64 measured level with the fused igemm launch, 109 vs 110 us at batch 128: not instantiated)
  if (d->cin == 64 && d->cout == 64) return 3;
  return 0;
}
// one round of workgroups: two per CU, a 3-deep ring each (one per CU with a 7-deep ring measured slower: 12.40 vs 12.37 ms per step)
static inline int bwd1x1_splits() { return 2 * num_cus(); }
static inline size_t bwd1x1_workspace(const RiglConvDesc* d) {
  return bwd1x1_kind(d) ? (size_t)2 * num_cus() * d->cin * d->cout * 4 : 0;
}
template <int CI, int CO, bool DO_W, int NST>
static bool bwd1x1_ready_i() {
  constexpr int SMEM = Bwd1x1Smem<CI, CO, DO_W, NST>::BYTES;
  static_assert(SMEM <= 160 * 1024 / 2, "the rings of the two workgroups resident on a CU fit its LDS");
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd1x1<CI, CO, DO_W, NST>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
  return ready;
}
// Will the single-pass kernel (with its weight-gradient half) launch for this layer?  Asked BEFORE a caller commits to its
// path (workspace layout, reduce), so a refused dynamic-LDS opt-in falls through to the generic bodies cleanly.
static bool bwd1x1_ready(const RiglConvDesc* d) {
  switch (bwd1x1_kind(d)) {
    case 1: return bwd1x1_ready_i<64, 256, true, 3>();
    case 3: return bwd1x1_ready_i<64, 64, true, 3>();
    default: return false;
  }
}
template <int CI, int CO, bool DO_W, int NST>
static bool launch_bwd1x1_i(const Bwd1x1Args& a, hipStream_t st) {
  constexpr int SMEM = Bwd1x1Smem<CI, CO, DO_W, NST>::BYTES;
  if (!bwd1x1_ready_i<CI, CO, DO_W, NST>()) return false;
  RIGL_K_LAUNCH((k_bwd1x1<CI, CO, DO_W, NST>), dim3((unsigned)a.splits), dim3(THREADS), SMEM, st, a);
  return true;
}
template <int CI, int CO>
static bool launch_bwd1x1_t(const Bwd1x1Args& a, bool w, hipStream_t st) {
  return w ? launch_bwd1x1_i<CI, CO, true, 3>(a, st) : launch_bwd1x1_i<CI, CO, false, 3>(a, st);
}
static bool launch_bwd1x1(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                          const rigl_bf16* addend, rigl_bf16* dx, float* slab, hipStream_t st) {
  Bwd1x1Args a = {};
  a.X = x; a.DY = dy; a.W = w_hwio; a.ADD = addend; a.DX = dx; a.SLAB = slab;
  a.M = d->n * d->h * d->w; a.splits = bwd1x1_splits(); a.interleave = RIGL_TUNE("bwd1x1_il", 1);
  a.x_bytes = (uint32_t)((size_t)a.M * d->cin * 2); a.dy_bytes = (uint32_t)((size_t)a.M * d->cout * 2);
  const bool w = slab != nullptr;
  switch (bwd1x1_kind(d)) {
    case 1: return launch_bwd1x1_t<64, 256>(a, w, st);
    case 3: return launch_bwd1x1_t<64, 64>(a, w, st);
    default: return false;
  }
}
