// K1, single-pass backward of the 1x1 / stride-1 "reduce" convs with MANY input channels (included inside namespace
// rigl::k1 of conv.hip): cin a multiple of 128 (512, 1024, 256 ...), cout 128 or 256 -- the layers bwd1x1.hpp's
// "W in registers, cin = 64" form cannot take and that ran as two GEMMs sharing one grid (k_bwd_fused: dY read twice, X
// and dY staged by two different bodies, 2-3.5x over their bound).  Here the input-channel axis is cut into slices of
// SC = 128 channels and a workgroup (8 waves, one per CU) owns ONE slice for ONE group of pixel rows:
//   * its W slice [128][cout] never leaves the registers: wave (cf = wave & 3) holds the 32 x cout fragment of its 32 input
//     channels as MFMA "A" operands (cout / 4 VGPRs);
//   * it walks its 32-pixel K-tiles once (tile kt of row group g = tile g + kt * G of the tensor: at any moment the grid
//     streams ONE window): dY rows [32][cout] and the X slice [32][128] come in by LDS-DMA through an NST-deep ring;
//   * dX[32][slice] = dY x Wslice^T on v_mfma_f32_32x32x16_bf16 (operands swapped: a lane holds 4 consecutive input
//     channels of one pixel) by the four waves of one half of the workgroup -- the halves alternate tile by tile, and waves
//     w and w + 4 share a SIMD, so every SIMD issues the same MFMA count per tile; the tile leaves through an LDS
//     staging tile as whole 16-byte row segments (+ the shortcut gradient: bf16(bf16(acc) + addend));
//   * dW[slice][cout] += X^T dY on transposing LDS reads (ds_read_b64_tr_b16) of the SAME two tiles, 2 x 2 (cout = 256)
//     or 1 x 2 (cout = 128) 32x32 fragments per wave, accumulated over ALL tiles of the workgroup; one partial per
//     workgroup at the end into slab[g][slice rows][cout] (launch_wgrad_reduce sums the G slabs in a fixed order:
//     deterministic dW).
// dY is read from HBM once per XCD (the `slices` workgroups of a row group sit on one XCD and run in step), X, the addend
// and dX once.  The dY tile serves a row-major ds_read_b128 (k = output channel) and the transposing read (k = pixel);
// its 16-byte chunks are XORed with ((row & 3) << 2) | ((row >> 2) & 3) on the DMA's source side: the sixteen rows of a
// ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) then sit in sixteen different 16-byte bank
// slots, and the four rows of a transposing 32-lane pass in four different 64-byte bank quarters (256- and 512-byte rows
// alike: a row is a whole number of 256-byte bank windows).  tools/experiments/emu/bs_emu.py restates the index arithmetic per lane.
// Loop: one raw barrier per K-tile; the half that multiplies tile kt waits (counted vmcnt, LOADS only: bwd1x1.hpp) for the
// pieces it issued three iterations earlier, the other half issues tile kt + 3 and stores dX tile kt - 1.
// How it got here (gpurun r6a-r6d, 14x14 1024 -> 256 at batch 128, one-call backward alone from HBM, us; the shared-launch
// bodies it replaces: 71.7): first version -- every wave issues its share of the DMA, stores its piece of the previous dX
// tile and loads its piece of the addend into registers in every iteration -- 75.4 (phase stamps, -DRIGL_BS_TRACE + tools/
// bs_trace.py: 30-36 % of a wave's time in the flush, 22-29 % waiting: loads return in order, so waiting for the addend
// piece requested one iteration ago drains the three-tile DMA queue behind it, and every wave's MFMAs wait behind its own
// vector-memory issue); addend through the DMA ring and added at staging time + the halves alternating duty: 60.3 (28x28
// 512 -> 128: 95.0 -> 78.2, 512 -> 256: 126.1 -> 96.8, 56x56 256 -> 128: 170.5 -> 145.7).  The kernel then moves its 198 MB
// (X, addend, dX, dY, 32 slabs) in ~41 us = 4.8 TB/s of mixed reads and writes: s_setprio for the multiplying half
// measured level (removed).
// Reference: the autodiff of layers.masked_conv2d for the bottleneck's first conv (pruning_layers.py:139-157,
// resnet_model.py:456-470; sparse_optimizers_base.py:478-485 for the dense dW).
#pragma once

struct BsArgs {
  const uint16_t* X;    // [M][CI] bf16
  const uint16_t* DY;   // [M][CO] bf16
  const uint16_t* W;    // [CI][CO] bf16 (the HWIO shadow of a 1x1 kernel)
  const uint16_t* ADD;  // [M][CI] bf16 or NULL; with add_sh > 0 the gradient of a SUBSAMPLED view: one row per pixel with
                        // h % add_sh == 0 and w % add_sw == 0, laid out [image][add_ho][add_wo][CI] (rigl_masked_conv2d_bwd_sub)
  const uint8_t* ABITS; // or NULL: 1 bit per element of ADD ([M][CI / 8] bytes, bit j of a byte = channel 8 b + j): the addend
                        // counts only where its bit is set -- the shortcut gradient handed over UNMASKED with the ReLU bits of
                        // relu(bn3 + shortcut) (rigl_masked_conv2d_bwd_masked: the masked copy is never written)
  uint16_t* DX;         // [M][CI] bf16
  float* SLAB;          // [G][CI][CO] fp32 partial dW (unused with DO_W = false)
  int M, CI, slices, G;
  uint32_t x_bytes, dy_bytes, add_bytes;
  int add_sh, add_sw, add_ho, add_wo, IH, IW;
  FastDiv fd_w, fd_h;
  unsigned long long* TRACE;   // development (-DRIGL_BS_TRACE): [grid][2 waves (0 and 4)][8] s_memtime ticks per phase
};
#ifdef RIGL_BS_TRACE
#define BS_STAMP(i_) { if (P.TRACE) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tr_acc[i_] += n_ - tr_last; tr_last = n_; } }
#else
#define BS_STAMP(i_) { }
#endif

constexpr int BS_THREADS = 512;
// LDS-DMA through inline assembly (lds_dma16 in conv.hip: the compiler then puts no vmcnt(0) of its own between a wave's DMA
// issue and its next LDS read); -DRIGL_DMA_BUILTIN restores the builtin for A/B runs
#ifdef RIGL_DMA_BUILTIN
#define BS_DMA16(r_, lds_, off_) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (__attribute__((address_space(3))) void*)(lds_), 16, off_, 0, 0, 0)
#define BS_DMA4(r_, lds_, off_) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (__attribute__((address_space(3))) void*)(lds_), 4, off_, 0, 0, 0)
#else
#define BS_DMA16(r_, lds_, off_) lds_dma16(r_##4, lds_, off_)
#define BS_DMA4(r_, lds_, off_) lds_dma4(r_##4, lds_, off_)
#endif
// the once-read streams (X slice, addend slice) carry the non-temporal hint (-DRIGL_BS_NO_NT: without; alone 56x56 256->128
// 134.7 -> 127.3 us, 14x14 1024->512 87.0 -> 82.8, the others level; in the step -0.005 ms in both alternating pairs)
#if !defined(RIGL_BS_NO_NT) && !defined(RIGL_DMA_BUILTIN)
#define BS_DMA16_ONCE(r_, lds_, off_) lds_dma16_nt(r_##4, lds_, off_)
#else
#define BS_DMA16_ONCE(r_, lds_, off_) BS_DMA16(r_, lds_, off_)
#endif

__device__ __forceinline__ int bs_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// bf16(a + b) per half on the hardware converter (round to nearest even)
__device__ __forceinline__ uint32_t bs_add_bf16x2(uint32_t a, uint32_t b) {
  const f32x2 s2 = {__uint_as_float(a << 16) + __uint_as_float(b << 16),
                    __uint_as_float(a & 0xFFFF0000u) + __uint_as_float(b & 0xFFFF0000u)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(s2, bf16x2));
}

template <int CO, bool DO_W>
struct BsGeom {
#ifndef RIGL_BS_NST128
#define RIGL_BS_NST128 4   // (5 -- four 24 KB tiles in flight -- measured level in the step and 3 us slower alone: 77.5 -> 80.4 us)
#endif
  static constexpr int SC = 128, PX = 32;
  static constexpr int NST = CO == 128 ? RIGL_BS_NST128 : 4;   // ring depth
  static constexpr int YROWB = CO * 2, XROWB = SC * 2;
  static constexpr int Y_BYTES = PX * YROWB, X_BYTES = DO_W ? PX * XROWB : 0, A_BYTES = PX * XROWB;
  static constexpr int B_BYTES = PX * SC / 8;                  // the ReLU bits of the shortcut-gradient slice (16 bytes per row)
  static constexpr int STAGE = Y_BYTES + X_BYTES + A_BYTES + B_BYTES;   // dY rows, the X slice, the slice of the shortcut gradient (+ bits)
  static constexpr int DXROWB = XROWB + 8, DX_BYTES = PX * DXROWB;     // 8 bytes of padding: conflict-free ds_write_b64 (rowstream.hpp)
  static constexpr int SMEM = NST * STAGE + 2 * DX_BYTES;
};

template <int CO, bool DO_W>
__global__ __launch_bounds__(BS_THREADS) void k_bwdslice(BsArgs P) {
  using G = BsGeom<CO, DO_W>;
  constexpr int SC = G::SC, PX = G::PX, NST = G::NST, YROWB = G::YROWB, XROWB = G::XROWB;
  constexpr int Y_BYTES = G::Y_BYTES, X_BYTES = G::X_BYTES, STAGE = G::STAGE, DXROWB = G::DXROWB, DX_BYTES = G::DX_BYTES;
  // a tile's DMA pieces (1 KB wave-instructions) are issued by the FOUR waves of one half: per wave
  constexpr int YPW = Y_BYTES / 1024 / 4, XPW = DO_W ? 2 : 0, APW = 2;
  static_assert(Y_BYTES % 4096 == 0 && PX * XROWB == 8192, "whole rounds of four waves");
  constexpr int PW_N = YPW + XPW, PW_A = PW_N + APW;           // DMA instructions per issuing thread per K-tile (without / with addend)
  constexpr int KS = CO / 16;                                  // dgrad k-steps
  constexpr int NFO = CO / 32;                                 // wgrad: 32x32 fragments of dW[128][CO] along cout (4 along cin)
  constexpr int TI = NFO >= 8 ? 2 : 1, TO = 2;                 // fragments per wave: TI x TO
  static_assert((4 / TI) * (NFO / TO) == 8, "eight waves cover the slice's dW");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_bs[];
  unsigned char* const dxs = smem_bs + NST * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, r31 = lane & 31;
  // workgroup -> (channel slice, row group): the slices of one row group share an XCD (block b runs on XCD b % 8)
  const int xcd = (int)(blockIdx.x & 7u), idx = (int)(blockIdx.x >> 3);
  const int slice = idx % P.slices, g = xcd + 8 * (idx / P.slices);
  const int KT_all = (P.M + PX - 1) / PX;
  const int KT = g < KT_all ? (KT_all - g + P.G - 1) / P.G : 0;
  const __amdgpu_buffer_rsrc_t rsrcY = make_rsrc(P.DY, P.dy_bytes), rsrcX = make_rsrc(P.X, P.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.ADD ? P.ADD : P.DY, P.ADD ? P.add_bytes : 0u);
  const bool has_add = P.ADD != nullptr, has_bits = P.ABITS != nullptr;
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(has_bits ? (const void*)P.ABITS : (const void*)P.DY, has_bits ? P.add_bytes / 16u : 0u);
  const u32x4 rsrcY4 = make_rsrc4(P.DY, P.dy_bytes), rsrcX4 = make_rsrc4(P.X, P.x_bytes);
  const u32x4 rsrcA4 = make_rsrc4(P.ADD ? P.ADD : P.DY, P.ADD ? P.add_bytes : 0u);
  const u32x4 rsrcB4 = make_rsrc4(has_bits ? (const void*)P.ABITS : (const void*)P.DY, has_bits ? P.add_bytes / 16u : 0u);
  (void)rsrcY; (void)rsrcX; (void)rsrcA; (void)rsrcB; (void)rsrcY4; (void)rsrcX4; (void)rsrcA4; (void)rsrcB4;
  constexpr int A_BYTES = G::A_BYTES;
  // the two halves of the workgroup: cf = the wave's 32-channel fragment of the slice, par = the parity of the tiles whose dX it
  // computes (and whose DMA it issued); waves w and w + 4 share a SIMD
  const int cf = wave & 3, par = wave >> 2;

  // ---- DMA lanes: wave-instruction i of a tile with ROWB-byte rows fills rows i * (1024 / ROWB) ..; lane l: row + l / (ROWB / 16),
  // 16-byte slot l % (ROWB / 16), fetching the logical chunk slot ^ swizzle(row); piece i = q * 4 + cf of the issuing half
  int y_row[YPW], y_col[YPW], x_row[2], x_col[2];
#pragma unroll
  for (int q = 0; q < YPW; ++q) {
    const int i = q * 4 + cf, row = i * (1024 / YROWB) + lane / (YROWB / 16), slot = lane % (YROWB / 16);
    y_row[q] = row; y_col[q] = (slot ^ bs_swz(row)) * 8;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = q * 4 + cf, row = i * (1024 / XROWB) + lane / (XROWB / 16);
    x_row[q] = row; x_col[q] = slice * SC + (((lane % (XROWB / 16)) ^ bs_swz(row)) * 8);
  }
#define BS_ISSUE(kt_, stage_)                                                                            \
  {                                                                                                      \
    const int p0_ = (g + (kt_) * P.G) * PX;                                                              \
    unsigned char* const st_ = smem_bs + (stage_) * STAGE;                                               \
    _Pragma("unroll") for (int q = 0; q < YPW; ++q) {                                                    \
      const int p_ = p0_ + y_row[q];                                                                     \
      const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * CO + y_col[q]) * 2u) : (int)OOB;                 \
      BS_DMA16(rsrcY, st_ + (q * 4 + cf) * 1024, off_); \
    }                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
      const int p_ = p0_ + x_row[q];                                                                     \
      const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * P.CI + x_col[q]) * 2u) : (int)OOB;               \
      if (DO_W)                                                                                          \
        BS_DMA16_ONCE(rsrcX, st_ + Y_BYTES + (q * 4 + cf) * 1024, off_); \
      if (has_add) {                                                                                     \
        int offa_ = off_;                                                                                \
        if (P.add_sh) {      /* the addend row of pixel p_ (if it has one: else zeros) */                 \
          const int t_ = fdiv(p_, P.fd_w), wi_ = p_ - t_ * P.IW, im_ = fdiv(t_, P.fd_h), hi_ = t_ - im_ * P.IH; \
          const int qh_ = hi_ / P.add_sh, qw_ = wi_ / P.add_sw;                                          \
          const bool on_ = p_ < P.M && qh_ * P.add_sh == hi_ && qw_ * P.add_sw == wi_;                   \
          offa_ = on_ ? (int)((uint32_t)(((im_ * P.add_ho + qh_) * P.add_wo + qw_) * P.CI + x_col[q]) * 2u) : (int)OOB; \
        }                                                                                                \
        BS_DMA16_ONCE(rsrcA, st_ + Y_BYTES + X_BYTES + (q * 4 + cf) * 1024, offa_); \
      }                                                                                                  \
    }                                                                                                    \
    if (has_bits && cf < 2) {      /* 32 rows x 16 bytes of ReLU bits: two wave-instructions of 4 bytes per lane */ \
      const int ib_ = cf * 64 + lane, p_ = p0_ + (ib_ >> 2);                                             \
      const int offb_ = p_ < P.M ? (int)(((uint32_t)(p_ * P.CI + slice * SC) >> 3) + (uint32_t)((ib_ & 3) * 4)) : (int)OOB; \
      BS_DMA4(rsrcB, st_ + Y_BYTES + X_BYTES + A_BYTES + cf * 256, offb_); \
    }                                                                                                    \
  }

  // ---- dgrad: the wave's W fragment (32 input channels x CO) in registers
  bf16x8 wfr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ci = slice * SC + cf * 32 + r31, co = ks * 16 + hi * 8;
    wfr[ks] = *reinterpret_cast<const bf16x8*>(P.W + (int64_t)ci * CO + co);
  }
  // dY fragment of k-step ks: lane l = row l & 31, logical chunk 2 * ks + (l >> 5)
  const int d_swz = bs_swz(r31);

  // ---- wgrad: transposing fragment reads (lane geometry of bwd1x1.hpp / wgrad_tr_body)
  const int gq = lane >> 4, j16 = lane & 15;
  const int t_row = 8 * (gq >> 1) + (j16 >> 2);                     // + 16 * k2 (+ 4 for the second half of the 8 pixels)
  const int t_low = 2 * (gq & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
  const int fi0 = TI == 2 ? (wave & 1) * 2 : (wave & 3);            // first cin fragment of the wave
  const int fo0 = TI == 2 ? (wave >> 1) * 2 : (wave >> 2) * 2;      // first cout fragment
  f32x16 acc2[DO_W ? TI : 1][TO];
  if (DO_W) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TO; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
#define BS_TR_OFF(ROWB_, chunk_, plus4_) \
  ((t_row + (plus4_)) * (ROWB_) + ((((chunk_) + t_low) ^ bs_swz(t_row + (plus4_))) << 4) + t_half)

  // a dX tile is 32 rows x 16 pieces of 16 bytes, stored by the four waves of one half: two pieces per thread
  const int ft = cf * 64 + lane;
  // stores the staged tile ktp (written by this half before the barrier this is called behind)
#define BS_FLUSH(ktp_)                                                                                   \
  {                                                                                                      \
    uint4 v_[2];                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
      const int pc_ = q * 256 + ft, row_ = pc_ >> 4, ch_ = pc_ & 15;                                     \
      const unsigned char* src_ = dxs + ((ktp_) & 1) * DX_BYTES + row_ * DXROWB + ch_ * 16;              \
      const uint2 lo8_ = *reinterpret_cast<const uint2*>(src_), hi8_ = *reinterpret_cast<const uint2*>(src_ + 8); \
      v_[q] = make_uint4(lo8_.x, lo8_.y, hi8_.x, hi8_.y);                                                \
    }                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
      const int pc_ = q * 256 + ft, row_ = pc_ >> 4, ch_ = pc_ & 15;                                     \
      const int p_ = (g + (ktp_) * P.G) * PX + row_;                                                     \
      if (p_ < P.M) *reinterpret_cast<uint4*>(P.DX + (int64_t)p_ * P.CI + slice * SC + ch_ * 8) = v_[q]; \
    }                                                                                                    \
  }
#ifdef RIGL_BS_TRACE
  unsigned long long tr_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tr_last = __builtin_amdgcn_s_memtime();
#endif
  // Tile t is issued NST - 1 iterations before it is multiplied, by the half that is NOT multiplying then: half
  // (t + NST) & 1.  Prologue: the first NST - 1 tiles, each by its half.
  constexpr int ISS = NST & 1;              // par of the half that issues the even tiles
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (((t & 1) ^ ISS) == par && t < KT) BS_ISSUE(t, t);
  BS_STAMP(7);
  for (int kt = 0; kt < KT; ++kt) {
    const bool mine = (kt & 1) == par;
    if ((((kt & 1) ^ ISS) == par)) {
      // this half issued tile kt; the only LOADS it has issued behind it are the pieces of tile kt + 2 (NST <= 5; stores are
      // not counted on: bwd1x1.hpp)
      static_assert(NST == 4 || NST == 5, "one younger tile of the same half in flight behind the awaited one");
      if (kt + 2 < KT) { if (has_add) wait_vmcnt<PW_A>(); else wait_vmcnt<PW_N>(); }
      else wait_vmcnt<0>();
    }
    // (a raw s_barrier does not wait for this wave's ds_writes of the previous dX tile: bwd1x1.hpp)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    BS_STAMP(0);
    const unsigned char* Ys = smem_bs + (kt % NST) * STAGE;
    const unsigned char* Xs = Ys + Y_BYTES;
    const unsigned char* As = Xs + X_BYTES;
    f32x16 a0, a1;
    uint2 av[4];
    if (!mine) {
      // the other half's tile: bring in tile kt + 3 (ours) and store our dX tile kt - 1 -- the vector-memory issue of the
      // workgroup runs under the MFMAs of the half that shares our SIMDs
      if (kt + NST - 1 < KT) BS_ISSUE(kt + NST - 1, (kt + NST - 1) % NST);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BS_STAMP(1);
      if (kt > 0) BS_FLUSH(kt - 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BS_STAMP(2);
    } else {
      // ---- dX tile: D[ci][px] = W-fragment x dY-fragment, two accumulators (even / odd k-steps: no back-to-back dependent MFMAs)
      if (has_add) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          av[q] = *reinterpret_cast<const uint2*>(As + r31 * XROWB + (((cf * 4 + q) ^ d_swz) << 4) + hi * 8);
        if (has_bits) {
          // byte q of this word = the bits of channels cf * 32 + 8 q ..: this lane's four are bits 4 hi .. 4 hi + 3
          const uint32_t bw = *reinterpret_cast<const uint32_t*>(As + A_BYTES + r31 * 16 + cf * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t nib = (bw >> (8 * q + 4 * hi)) & 0xFu;
            av[q].x &= ((nib & 1u) ? 0x0000FFFFu : 0u) | ((nib & 2u) ? 0xFFFF0000u : 0u);
            av[q].y &= ((nib & 4u) ? 0x0000FFFFu : 0u) | ((nib & 8u) ? 0xFFFF0000u : 0u);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) a0[e] = a1[e] = 0.f;
      {
        // (opaque copies: the fragment addresses are recomputed per tile instead of living in sixteen registers; the reads of
        // k-steps ks + 2, ks + 3 are issued before the MFMAs of ks, ks + 1: no exposed LDS latency between MFMAs)
        int r_ = r31, h_ = hi;
        asm volatile("" : "+v"(r_), "+v"(h_));
        const unsigned char* const yb_ = Ys + r_ * YROWB;
        const int ds_ = bs_swz(r_);
        bf16x8 yq[2][2];
        yq[0][0] = *reinterpret_cast<const bf16x8*>(yb_ + ((h_ ^ ds_) << 4));
        yq[0][1] = *reinterpret_cast<const bf16x8*>(yb_ + (((2 + h_) ^ ds_) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
          const int cur = (ks >> 1) & 1;
          if (ks + 2 < KS) {
            yq[cur ^ 1][0] = *reinterpret_cast<const bf16x8*>(yb_ + (((2 * ks + 4 + h_) ^ ds_) << 4));
            yq[cur ^ 1][1] = *reinterpret_cast<const bf16x8*>(yb_ + (((2 * ks + 6 + h_) ^ ds_) << 4));
          }
          __builtin_amdgcn_sched_barrier(0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[ks], yq[cur][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[ks + 1], yq[cur][1], a1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      BS_STAMP(4);
    }
    // ---- dW partial: D2[ci][co] += X^T-fragment x dY^T-fragment, two k-steps of 16 pixels
    if (DO_W) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 fa[TI], fb[TO];
#pragma unroll
        for (int i = 0; i < TI; ++i)
          fa[i] = lds_read_tr_pair(Xs + k2 * 16 * XROWB + BS_TR_OFF(XROWB, (fi0 + i) * 4, 0), Xs + k2 * 16 * XROWB + BS_TR_OFF(XROWB, (fi0 + i) * 4, 4));
#pragma unroll
        for (int j = 0; j < TO; ++j)
          fb[j] = lds_read_tr_pair(Ys + k2 * 16 * YROWB + BS_TR_OFF(YROWB, (fo0 + j) * 4, 0), Ys + k2 * 16 * YROWB + BS_TR_OFF(YROWB, (fo0 + j) * 4, 4));
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TO; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc2[i][j], 0, 0, 0);
      }
    }
    BS_STAMP(5);
    // ---- dX tile (+ the shortcut gradient: bf16(bf16(acc) + addend)) -> LDS: D row (e & 3) + 8 * (e >> 2) + 4 * hi = input
    // channel of the fragment, column lane & 31 = pixel
    if (mine) {
      unsigned char* const dst = dxs + (kt & 1) * DX_BYTES + r31 * DXROWB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 lo = {a0[4 * q] + a1[4 * q], a0[4 * q + 1] + a1[4 * q + 1]};
        const f32x2 hi2 = {a0[4 * q + 2] + a1[4 * q + 2], a0[4 * q + 3] + a1[4 * q + 3]};
        uint2 pk;
        pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
        pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
        if (has_add) { pk.x = bs_add_bf16x2(pk.x, av[q].x); pk.y = bs_add_bf16x2(pk.y, av[q].y); }
        *reinterpret_cast<uint2*>(dst + (cf * 32 + 8 * q + 4 * hi) * 2) = pk;
      }
    }
    BS_STAMP(6);
  }
  if (KT > 0) {
    __syncthreads();
    if (((KT - 1) & 1) == par) BS_FLUSH(KT - 1);
  }
#undef BS_FLUSH
#undef BS_ISSUE
#undef BS_TR_OFF
  // ---- the workgroup's dW partial -> slab g, rows of its slice
  if (DO_W) {
    float* const out = P.SLAB + ((int64_t)g * P.CI + slice * SC) * CO;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TO; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = (fi0 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi, co = (fo0 + j) * 32 + r31;
          out[(int64_t)ci * CO + co] = acc2[i][j][e];
        }
  }
#ifdef RIGL_BS_TRACE
  if (P.TRACE && lane == 0 && (wave & 3) == 0) {
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();
    tr_acc[7] += n_ - tr_last;
    for (int i = 0; i < 8; ++i) P.TRACE[((int64_t)blockIdx.x * 2 + (wave >> 2)) * 8 + i] = tr_acc[i];
  }
#endif
}

// ---- cout = 512: slices of 64 input channels ----------------------------------------------------------------------------
// With 512 output channels the W fragment (32 x 512: 128 registers) and the slice's dW accumulators (128 x 512 fp32 over eight
// waves: 128 registers) do not fit a wave together.  Here a slice is SC = 64 channels: dW[64][512] is 64 registers per wave
// (2 x 2 fragments of 32 x 32, wave w = the cout fragments 2 w, 2 w + 1), and the dgrad of a tile is split over the FOUR waves
// of the multiplying half as (cf = channel fragment of 32) x (kh = half of the output channels): a wave holds the 32 x 256 W
// fragment of its (cf, kh) (64 registers), multiplies its half of the reduction, and the two halves of one channel fragment
// meet through LDS: the kh = 1 wave leaves its fp32 partial ([4][64 lanes] x 16 bytes: conflict-free) at the end of its
// multiplying iteration, the kh = 0 wave adds it to its own in its NEXT iteration -- the one in which its half has the
// vector-memory duty -- converts, adds the shortcut gradient it picked up from the ring while multiplying, and stores the
// [32 pixels][32 channels] block through a wave-private staging tile (no second barrier; 64-byte row segments).  In that
// iteration the four waves of the half issue the DMA of tile kt + 2 (ten instructions each; issued by two waves, twenty each,
// the issue was as long as the dgrad phase).  Ring: three stages of 40.25 KB (dY rows [32][512], X slice [32][64], addend slice
// + bits): the issuing waves wait for their latest tile with vmcnt(0) -- nothing younger of theirs is in flight, the other
// half's tile is.  128-byte rows (X, addend) hold two rows per 256-byte bank window: the X
// tile, read only by the transposing reads (4 rows x 64 bytes per 32-lane pass), flips chunk bit 2 with row bit 1; the
// addend tile, read row-per-lane, spreads rows 0 .. 15 over the 16 (row parity, chunk ^ (row >> 1)) slots.
// Summation order of a dX element: (even k-steps + odd k-steps of kh = 0) + (the same of kh = 1), identical with and without
// the weight-gradient half.  tools/experiments/emu/bs_emu.py restates the index arithmetic (run_workgroup64).
template <int CO, bool DO_W>
struct Bs64Geom {
  static constexpr int SC = 64, PX = 32, NST = 3;
  static constexpr int YROWB = CO * 2, XROWB = SC * 2;
  static constexpr int Y_BYTES = PX * YROWB, X_BYTES = DO_W ? PX * XROWB : 0, A_BYTES = PX * XROWB, B_BYTES = PX * SC / 8;
  static constexpr int STAGE = Y_BYTES + X_BYTES + A_BYTES + B_BYTES;
  static constexpr int PART_BYTES = 64 * 16 * 4;                 // one wave's fp32 dgrad partial
  static constexpr int SROWB = 32 * 2 + 8, STG_BYTES = PX * SROWB;   // a wave's staging tile: [32 pixels][32 channels] + 8 bytes of padding
  static constexpr int SMEM = NST * STAGE + 4 * PART_BYTES + 4 * STG_BYTES;
};
__device__ __forceinline__ int bs64_swz_x(int row) { return ((row >> 1) & 1) << 2; }
__device__ __forceinline__ int bs64_swz_a(int row) { return (row >> 1) & 7; }

template <int CO, bool DO_W>
__global__ __launch_bounds__(BS_THREADS) void k_bwdslice64(BsArgs P) {
  using G = Bs64Geom<CO, DO_W>;
  constexpr int SC = G::SC, PX = G::PX, NST = G::NST, YROWB = G::YROWB, XROWB = G::XROWB;
  constexpr int Y_BYTES = G::Y_BYTES, X_BYTES = G::X_BYTES, A_BYTES = G::A_BYTES, STAGE = G::STAGE;
  constexpr int YPW = PX / 4;                                  // dY rows (1 KB wave-instructions) per issuing wave
  static_assert(YROWB == 1024 && XROWB == 128 && NST == 3, "one DMA instruction per dY row, eight X rows per instruction");
  constexpr int KSH = CO / 32;                                 // dgrad k-steps of one wave (half of the output channels)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_bs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, r31 = lane & 31;
  const int xcd = (int)(blockIdx.x & 7u), idx = (int)(blockIdx.x >> 3);
  const int slice = idx % P.slices, g = xcd + 8 * (idx / P.slices);
  const int KT_all = (P.M + PX - 1) / PX;
  const int KT = g < KT_all ? (KT_all - g + P.G - 1) / P.G : 0;
  const __amdgpu_buffer_rsrc_t rsrcY = make_rsrc(P.DY, P.dy_bytes), rsrcX = make_rsrc(P.X, P.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.ADD ? P.ADD : P.DY, P.ADD ? P.add_bytes : 0u);
  const bool has_add = P.ADD != nullptr, has_bits = P.ABITS != nullptr;
  const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(has_bits ? (const void*)P.ABITS : (const void*)P.DY, has_bits ? P.add_bytes / 16u : 0u);
  const u32x4 rsrcY4 = make_rsrc4(P.DY, P.dy_bytes), rsrcX4 = make_rsrc4(P.X, P.x_bytes);
  const u32x4 rsrcA4 = make_rsrc4(P.ADD ? P.ADD : P.DY, P.ADD ? P.add_bytes : 0u);
  const u32x4 rsrcB4 = make_rsrc4(has_bits ? (const void*)P.ABITS : (const void*)P.DY, has_bits ? P.add_bytes / 16u : 0u);
  (void)rsrcY; (void)rsrcX; (void)rsrcA; (void)rsrcB; (void)rsrcY4; (void)rsrcX4; (void)rsrcA4; (void)rsrcB4;
  // par = the parity of the tiles this wave multiplies for dX, kh = its half of the output channels, cf = its channel fragment
  const int par = wave >> 2, kh = (wave >> 1) & 1, cf = wave & 1;
  unsigned char* const part = smem_bs + NST * STAGE + (par * 2 + cf) * G::PART_BYTES;
  unsigned char* const stg = smem_bs + NST * STAGE + 4 * G::PART_BYTES + (par * 2 + cf) * G::STG_BYTES;

  // ---- DMA (the four waves of a half, w4 = wave & 3): a dY row per instruction (row q * 4 + w4), lane = 16-byte slot; X / addend:
  // eight rows per instruction (piece w4), lane l: row + l / 8, slot l % 8
  const int w4 = wave & 3;
  const int x_row = w4 * 8 + lane / 8;
  const int x_col = slice * SC + (((lane % 8) ^ bs64_swz_x(x_row)) * 8), a_col = slice * SC + (((lane % 8) ^ bs64_swz_a(x_row)) * 8);
#define BS64_ISSUE(kt_, stage_)                                                                          \
  {                                                                                                      \
    const int p0_ = (g + (kt_) * P.G) * PX;                                                              \
    unsigned char* const st_ = smem_bs + (stage_) * STAGE;                                               \
    int ln_ = lane;      /* (opaque: the sixteen per-row offsets are recomputed here instead of living in sixteen registers) */ \
    asm volatile("" : "+v"(ln_));                                                                        \
    _Pragma("unroll") for (int q = 0; q < YPW; ++q) {                                                    \
      const int row_ = q * 4 + w4, p_ = p0_ + row_;                                                      \
      const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * CO + ((ln_ ^ bs_swz(row_)) * 8)) * 2u) : (int)OOB; \
      BS_DMA16(rsrcY, st_ + row_ * 1024, off_);        \
    }                                                                                                    \
    {                                                                                                    \
      const int p_ = p0_ + x_row;                                                                        \
      if (DO_W) {                                                                                        \
        const int off_ = p_ < P.M ? (int)((uint32_t)(p_ * P.CI + x_col) * 2u) : (int)OOB;                \
        BS_DMA16_ONCE(rsrcX, st_ + Y_BYTES + w4 * 1024, off_);                                                \
      }                                                                                                  \
      if (has_add) {                                                                                     \
        int offa_ = p_ < P.M ? (int)((uint32_t)(p_ * P.CI + a_col) * 2u) : (int)OOB;                     \
        if (P.add_sh) {      /* the addend row of pixel p_ (if it has one: else zeros) */                 \
          const int t_ = fdiv(p_, P.fd_w), wi_ = p_ - t_ * P.IW, im_ = fdiv(t_, P.fd_h), hi_ = t_ - im_ * P.IH; \
          const int qh_ = hi_ / P.add_sh, qw_ = wi_ / P.add_sw;                                          \
          const bool on_ = p_ < P.M && qh_ * P.add_sh == hi_ && qw_ * P.add_sw == wi_;                   \
          offa_ = on_ ? (int)((uint32_t)(((im_ * P.add_ho + qh_) * P.add_wo + qw_) * P.CI + a_col) * 2u) : (int)OOB; \
        }                                                                                                \
        BS_DMA16_ONCE(rsrcA, st_ + Y_BYTES + X_BYTES + w4 * 1024, offa_);                                     \
      }                                                                                                  \
    }                                                                                                    \
    if (has_bits && w4 == 0) {     /* 32 rows x 8 bytes of ReLU bits: one wave-instruction of 4 bytes per lane */ \
      const int p_ = p0_ + (lane >> 1);                                                                  \
      const int offb_ = p_ < P.M ? (int)(((uint32_t)(p_ * P.CI + slice * SC) >> 3) + (uint32_t)((lane & 1) * 4)) : (int)OOB; \
      BS_DMA4(rsrcB, st_ + Y_BYTES + X_BYTES + A_BYTES, offb_); \
    }                                                                                                    \
  }

  // ---- dgrad: the wave's W fragment (32 input channels x its 256 output channels) in registers
  bf16x8 wfr[KSH];
#pragma unroll
  for (int ks = 0; ks < KSH; ++ks) {
    const int ci = slice * SC + cf * 32 + r31, co = (kh * KSH + ks) * 16 + hi * 8;
    wfr[ks] = *reinterpret_cast<const bf16x8*>(P.W + (int64_t)ci * CO + co);
  }
  const int a_swz = bs64_swz_a(r31);

  // ---- wgrad: transposing fragment reads; wave w: both channel fragments x the output-channel fragments 2 w, 2 w + 1
  const int gq = lane >> 4, j16 = lane & 15;
  const int t_row = 8 * (gq >> 1) + (j16 >> 2);
  const int t_low = 2 * (gq & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
  const int fo0 = wave * 2;
  f32x16 acc2[DO_W ? 2 : 1][2];
  if (DO_W) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
  // (tr_, tl_, th_: opaque per-tile copies of t_row, t_low, t_half -- the eight fragment offsets are recomputed per tile instead
  // of living in eight registers)
#define BS64_TR_Y(chunk_, plus4_) ((tr_ + (plus4_)) * YROWB + ((((chunk_) + tl_) ^ bs_swz(tr_ + (plus4_))) << 4) + th_)
#define BS64_TR_X(chunk_, plus4_) ((tr_ + (plus4_)) * XROWB + ((((chunk_) + tl_) ^ bs64_swz_x(tr_ + (plus4_))) << 4) + th_)

  f32x16 a0;                 // the wave's dgrad partial of the tile it multiplied last (kh = 0: until it is combined)
  uint2 av[4];               // kh = 0: the shortcut-gradient fragment of that tile
#pragma unroll
  for (int e = 0; e < 16; ++e) a0[e] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) av[q] = make_uint2(0u, 0u);
  // kh = 0 of the half that multiplied tile ktp: own partial + the partner's, bf16, + the shortcut gradient, staged, stored
#define BS64_COMBINE(ktp_)                                                                               \
  {                                                                                                      \
    rs_f32x4 oq_[4];       /* (the four reads first: one wait instead of read -> wait -> write four times) */ \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) oq_[q] = *reinterpret_cast<const rs_f32x4*>(part + q * 1024 + lane * 16); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
      const rs_f32x4 o_ = oq_[q];                                                                        \
      const f32x2 lo_ = {a0[4 * q] + o_[0], a0[4 * q + 1] + o_[1]}, hi_ = {a0[4 * q + 2] + o_[2], a0[4 * q + 3] + o_[3]}; \
      uint2 pk_;                                                                                         \
      pk_.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo_, bf16x2));                        \
      pk_.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi_, bf16x2));                        \
      if (has_add) { pk_.x = bs_add_bf16x2(pk_.x, av[q].x); pk_.y = bs_add_bf16x2(pk_.y, av[q].y); }     \
      *reinterpret_cast<uint2*>(stg + r31 * G::SROWB + (8 * q + 4 * hi) * 2) = pk_;                      \
    }                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
      const int pc_ = q * 64 + lane, row_ = pc_ >> 2, ch_ = pc_ & 3;                                     \
      const unsigned char* src_ = stg + row_ * G::SROWB + ch_ * 16;                                      \
      const uint2 lo8_ = *reinterpret_cast<const uint2*>(src_), hi8_ = *reinterpret_cast<const uint2*>(src_ + 8); \
      const int p_ = (g + (ktp_) * P.G) * PX + row_;                                                     \
      if (p_ < P.M)                                                                                      \
        *reinterpret_cast<uint4*>(P.DX + (int64_t)p_ * P.CI + slice * SC + cf * 32 + ch_ * 8) = make_uint4(lo8_.x, lo8_.y, hi8_.x, hi8_.y); \
    }                                                                                                    \
  }

#ifdef RIGL_BS_TRACE
  unsigned long long tr_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tr_last = __builtin_amdgcn_s_memtime();
#endif
  // tile t is issued two iterations before it is multiplied, by the half that is NOT multiplying then (par = (t & 1) ^ 1);
  // prologue: tiles 0 and 1
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (((t & 1) ^ 1) == par && t < KT) BS64_ISSUE(t, t);
  BS_STAMP(7);
  for (int kt = 0; kt < KT; ++kt) {
    const bool mine = (kt & 1) == par;
    if (!mine) wait_vmcnt<0>();                   // this wave issued its share of tile kt: its latest loads (and its dX stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    BS_STAMP(0);
    const unsigned char* Ys = smem_bs + (kt % NST) * STAGE;
    const unsigned char* Xs = Ys + Y_BYTES;
    const unsigned char* As = Xs + X_BYTES;
    if (!mine) {
      if (kt + NST - 1 < KT) BS64_ISSUE(kt + NST - 1, (kt + NST - 1) % NST);
      if (kh == 0 && kt > 0) BS64_COMBINE(kt - 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BS_STAMP(1 + (kh == 0));
    } else {
      if (kh == 0 && has_add) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          av[q] = *reinterpret_cast<const uint2*>(As + r31 * XROWB + (((cf * 4 + q) ^ a_swz) << 4) + hi * 8);
        if (has_bits) {
          // byte q of this word = the bits of channels cf * 32 + 8 q ..: this lane's four are bits 4 hi .. 4 hi + 3
          const uint32_t bw = *reinterpret_cast<const uint32_t*>(As + A_BYTES + r31 * 8 + cf * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t nib = (bw >> (8 * q + 4 * hi)) & 0xFu;
            av[q].x &= ((nib & 1u) ? 0x0000FFFFu : 0u) | ((nib & 2u) ? 0xFFFF0000u : 0u);
            av[q].y &= ((nib & 4u) ? 0x0000FFFFu : 0u) | ((nib & 8u) ? 0xFFFF0000u : 0u);
          }
        }
      }
      BS_STAMP(3);
      f32x16 a1;
#pragma unroll
      for (int e = 0; e < 16; ++e) a0[e] = a1[e] = 0.f;
      // (opaque copies: the sixteen fragment addresses are recomputed per tile instead of living in sixteen registers)
      int r_ = r31, h_ = hi;
      asm volatile("" : "+v"(r_), "+v"(h_));
      const unsigned char* const yb_ = Ys + r_ * YROWB;
      const int ds_ = bs_swz(r_), cb_ = 2 * kh * KSH + h_;
      // (the reads of k-steps ks + 2, ks + 3 are issued before the MFMAs of ks, ks + 1: no exposed LDS latency between MFMAs)
      bf16x8 yq[2][2];
      yq[0][0] = *reinterpret_cast<const bf16x8*>(yb_ + ((cb_ ^ ds_) << 4));
      yq[0][1] = *reinterpret_cast<const bf16x8*>(yb_ + (((cb_ + 2) ^ ds_) << 4));
#pragma unroll
      for (int ks = 0; ks < KSH; ks += 2) {
        const int cur = (ks >> 1) & 1;
        if (ks + 2 < KSH) {
          yq[cur ^ 1][0] = *reinterpret_cast<const bf16x8*>(yb_ + (((cb_ + 2 * ks + 4) ^ ds_) << 4));
          yq[cur ^ 1][1] = *reinterpret_cast<const bf16x8*>(yb_ + (((cb_ + 2 * ks + 6) ^ ds_) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[ks], yq[cur][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[ks + 1], yq[cur][1], a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) a0[e] += a1[e];
      if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const rs_f32x4 o = {a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]};
          *reinterpret_cast<rs_f32x4*>(part + q * 1024 + lane * 16) = o;
        }
      }
      BS_STAMP(4);
    }
    // ---- dW partial: D2[ci][co] += X^T-fragment x dY^T-fragment, two k-steps of 16 pixels
    if (DO_W) {
      int tr_ = t_row, tl_ = t_low, th_ = t_half;
      asm volatile("" : "+v"(tr_), "+v"(tl_), "+v"(th_));
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fa[i] = lds_read_tr_pair(Xs + k2 * 16 * XROWB + BS64_TR_X(i * 4, 0), Xs + k2 * 16 * XROWB + BS64_TR_X(i * 4, 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          fb[j] = lds_read_tr_pair(Ys + k2 * 16 * YROWB + BS64_TR_Y((fo0 + j) * 4, 0), Ys + k2 * 16 * YROWB + BS64_TR_Y((fo0 + j) * 4, 4));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc2[i][j], 0, 0, 0);
      }
    }
    BS_STAMP(5);
  }
  if (KT > 0) {
    __syncthreads();
    if (((KT - 1) & 1) == par && kh == 0) BS64_COMBINE(KT - 1);
  }
#undef BS64_COMBINE
#undef BS64_ISSUE
#undef BS64_TR_X
#undef BS64_TR_Y
  if (DO_W) {
    float* const out = P.SLAB + ((int64_t)g * P.CI + slice * SC) * CO;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi, co = (fo0 + j) * 32 + r31;
          out[(int64_t)ci * CO + co] = acc2[i][j][e];
        }
  }
#ifdef RIGL_BS_TRACE
  if (P.TRACE && lane == 0 && (wave == 0 || wave == 2)) {     // the combining and the issuing wave of half 0, channel fragment 0
    const unsigned long long n_ = __builtin_amdgcn_s_memtime();
    tr_acc[7] += n_ - tr_last;
    for (int i = 0; i < 8; ++i) P.TRACE[((int64_t)blockIdx.x * 2 + (wave >> 1)) * 8 + i] = tr_acc[i];
  }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Legal: 1x1, stride 1, no padding, same output grid; cout 128 or 256 (slices of 128 input channels) or 512 (slices of 64:
// k_bwdslice64, knob "bwdslice512"); cin a multiple of the slice with cin / slice a power of two <= 32; enough 32-pixel tiles
// for every row group.  "bwdslice": 0 = off, 1 = on.
struct BsPlan { int co, sc, slices, G; };
static bool bs_plan(const RiglConvDesc* d, BsPlan& p) {
  if (d->kh != 1 || d->kw != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_top || d->pad_left) return false;
  if (d->ho != d->h || d->wo != d->w) return false;
  if (d->cout != 128 && d->cout != 256 && d->cout != 512) return false;
  p.sc = d->cout == 512 ? 64 : 128;
  if (d->cin % p.sc) return false;
  p.co = d->cout; p.slices = d->cin / p.sc;
  if (p.slices > 32 || (p.slices & (p.slices - 1))) return false;
  const int64_t M = (int64_t)d->n * d->h * d->w;
  if (M * d->cin * 2 >= (1ll << 31)) return false;
  int groups = num_cus() / (8 * p.slices);
  if (groups < 1) groups = 1;
  p.G = 8 * groups;
  if ((M + 31) / 32 < 4ll * p.G) return false;       // at least four tiles per row group
  return true;
}
template <int CO, bool DO_W>
static bool bs_ready_i() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwdslice<CO, DO_W>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, BsGeom<CO, DO_W>::SMEM) == hipSuccess;
  return ready;
}
template <int CO, bool DO_W>
static bool bs64_ready_i() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwdslice64<CO, DO_W>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, Bs64Geom<CO, DO_W>::SMEM) == hipSuccess;
  return ready;
}
static bool bs_use(const RiglConvDesc* d, BsPlan* out = nullptr) {
  BsPlan p;
  if (!bs_plan(d, p)) return false;
  const int knob = RIGL_TUNE("bwdslice", 1);
  if (knob == 0) return false;
  if (p.co == 512 && !RIGL_TUNE("bwdslice512", 1)) return false;
  const bool ready = p.co == 512 ? (bs64_ready_i<512, true>() && bs64_ready_i<512, false>())
                     : p.co == 256 ? (bs_ready_i<256, true>() && bs_ready_i<256, false>())
                                   : (bs_ready_i<128, true>() && bs_ready_i<128, false>());
  if (ready && out) *out = p;
  return ready;
}
static size_t bs_workspace(const RiglConvDesc* d) {
  BsPlan p;
  return bs_plan(d, p) ? (size_t)p.G * d->cin * d->cout * 4 : 0;
}
template <int CO, bool DO_W>
static void launch_bs_i(const BsArgs& a, hipStream_t st) {
  RIGL_K_LAUNCH((k_bwdslice<CO, DO_W>), dim3((unsigned)(a.slices * a.G)), dim3(BS_THREADS), (unsigned)(BsGeom<CO, DO_W>::SMEM), st, a);
}
// slab == NULL: dX only (the same bits as with the weight-gradient half)
static void launch_bs(const RiglConvDesc* d, const BsPlan& p, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                      const rigl_bf16* addend, rigl_bf16* dx, float* slab, hipStream_t st, int sub_h = 1, int sub_w = 1,
                      const uint8_t* addend_bits = nullptr) {
  BsArgs a = {};
#ifdef RIGL_BS_TRACE
  { const char* e = getenv("RIGL_BS_TRACE_PTR"); a.TRACE = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  a.X = x; a.DY = dy; a.W = w_hwio; a.ADD = addend; a.DX = dx; a.SLAB = slab;
  a.ABITS = addend ? addend_bits : nullptr;
  a.M = d->n * d->h * d->w; a.CI = d->cin; a.slices = p.slices; a.G = p.G;
  a.x_bytes = (uint32_t)((size_t)a.M * d->cin * 2); a.dy_bytes = (uint32_t)((size_t)a.M * d->cout * 2);
  a.add_bytes = a.x_bytes; a.IH = d->h; a.IW = d->w;
  if (addend && (sub_h > 1 || sub_w > 1)) {
    a.add_sh = sub_h; a.add_sw = sub_w; a.add_ho = (d->h + sub_h - 1) / sub_h; a.add_wo = (d->w + sub_w - 1) / sub_w;
    a.add_bytes = (uint32_t)((size_t)d->n * a.add_ho * a.add_wo * d->cin * 2);
    a.fd_w = make_fastdiv(d->w); a.fd_h = make_fastdiv(d->h);
  }
  if (p.co == 512) {
    const dim3 grid((unsigned)(a.slices * a.G)), blk(BS_THREADS);
    if (slab) { RIGL_K_LAUNCH((k_bwdslice64<512, true>), grid, blk, (unsigned)(Bs64Geom<512, true>::SMEM), st, a); }
    else { RIGL_K_LAUNCH((k_bwdslice64<512, false>), grid, blk, (unsigned)(Bs64Geom<512, false>::SMEM), st, a); }
  } else if (p.co == 256) { if (slab) launch_bs_i<256, true>(a, st); else launch_bs_i<256, false>(a, st); }
  else { if (slab) launch_bs_i<128, true>(a, st); else launch_bs_i<128, false>(a, st); }
}
