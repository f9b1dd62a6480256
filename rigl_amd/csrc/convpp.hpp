// K1 "ping-pong" implicit-GEMM body for the long-reduction layers (3x3 convs, 1x1 convs with >= 512 reduction
// channels): 512 threads = 8 waves, TWO waves per SIMD that alternate roles every phase -- while one wave of a SIMD
// runs a cluster of MFMAs, its partner issues the LDS fragment reads and the LDS-DMA loads of later K-tiles -- in the
// spirit of the 8-phase schedule of cdna_hip_programming.md section 5 (raw s_barrier twice per phase, counted vmcnt,
// no vmcnt(0) in steady state, DMA loads in flight across four phases).  Included by conv.hip (same namespace, same
// IgemmArgs: reference call site rigl/imagenet_resnet/pruning_layers.py:139-157 fwd, autodiff dX
// sparse_optimizers_base.py:478-485).
//
// Geometry (template): WM x WN waves (= 8), each wave TM x TN MFMA tiles of 32x32 (v_mfma_f32_32x32x16_bf16, operands
// swapped like igemm_body so a lane holds 4 consecutive output channels), BK = 64 (one filter tap x 64 channels):
//   <2,4,4,2> 256x256 tile, 128x64 per wave   (N % 256 == 0)
//   <2,4,2,2> 128x256 tile,  64x64 per wave   (N % 256 == 0, layers with few rows)
//   <4,2,2,2> 256x128 tile,  64x64 per wave   (N % 128 == 0)
//   <4,2,4,2> 512x128 tile, 128x64 per wave   (N % 128 == 0, many rows)
//
// A K-tile lives in LDS as four PIECES: A0 / A1 = the first / second half of every wave's rows, B0 / B1 = the first /
// second half of every wave's columns ([row][64] bf16 = 128-byte rows, XOR-swizzled 16-byte chunks; the swizzle is
// applied on the DMA's SOURCE side because `buffer_load ... lds` writes lane-linear).  A wave multiplies a K-tile
// quadrant by quadrant, (A0,B0) (A0,B1) (A1,B1) (A1,B0), in PH phases (template; the one-quadrant-per-phase form
// PH = 4 of round 3 measured 5-35 % slower on every tile and is gone):
//   PH = 2  two quadrants per phase (16 MFMAs between barriers on the 128x64 wave tiles), two stages:
//           tile t phase A (reads A0 B0 B1): A1(t+1)      phase B (reads A1): A0(t+2) B0(t+2) B1(t+2)
//   PH = 1  the whole K-tile per phase (for the 64x64 wave tiles), three stages: tile t re-fills the stage of tile t-1
//           with tile t+2.
// A piece is issued in the phase right after its slot's last read, whose reading segment ends with
// `s_waitcnt lgkmcnt(0)` BEFORE its barrier -- the two wave groups run one barrier apart,
// so a read and a re-fill one barrier apart would race otherwise.  The covering `s_waitcnt vmcnt(N)` sits at the end of
// the read segment of the phase BEFORE the reading phase (every wave's wait precedes, by a barrier, every wave's read --
// the stagger makes "wait and read in the same phase" a race), N = the loads of the (at most four) younger pieces.
#pragma once
// (included inside namespace rigl::k1 of conv.hip)

template <int WM, int WN, int TM, int TN, int PH>
struct PPGeom {
  static_assert(PH == 2 || PH == 1, "phases per K-tile");
  static constexpr int NST = PH == 1 ? 3 : 2;                      // K-tile stages in LDS
  static_assert(WM * WN == 8, "eight waves");
  static_assert(TM % 2 == 0 && TN % 2 == 0, "a wave tile splits into 2 x 2 quadrants");
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int AH_ROWS = BM / 2, BH_ROWS = BN / 2;
  static constexpr int AH_BYTES = AH_ROWS * 128, BH_BYTES = BH_ROWS * 128;
  static constexpr int STAGE = 2 * AH_BYTES + 2 * BH_BYTES;
  static constexpr int NA = AH_ROWS / 64, NB = BH_ROWS / 64;       // LDS-DMA instructions per thread per piece
  static_assert(AH_ROWS % 64 == 0 && BH_ROWS % 64 == 0, "a piece is whole rounds of 8 waves x 8 rows");
  static constexpr int QM = TM / 2, QN = TN / 2;                   // MFMA tiles per quadrant
  static constexpr int EPI_ROWB = TN * 64 + 16;                    // bytes per staged output row of a wave (+16: bank spread)
  static constexpr int EPI_WAVE = QM * 32 * EPI_ROWB;              // one half of a wave's rows at a time
  static constexpr int EPI_STATS = 8 * EPI_WAVE;                   // [8 waves][2][TN*32] floats behind the staging areas
  static constexpr int EPI_ALL = EPI_STATS + 8 * 2 * TN * 32 * 4;
  static constexpr int SMEM = (NST * STAGE > EPI_ALL) ? NST * STAGE : EPI_ALL;
  static_assert(SMEM <= 160 * 1024, "LDS per CU");
};

#define PP_BARRIER()                                                                                     \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    asm volatile("" ::: "memory");                                                                       \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
// read segments that end with this wait retire their fragment reads BEFORE the barrier: the slot may be re-filled by
// the other wave group in the very next barrier interval
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

struct PPCursor { int tap, cb, r, s; };

// CLS (dgrad of a strided conv): rows are enumerated per stride-parity class (h % sh, w % sw) exactly as igemm_body does --
// a tile never mixes classes, so it only visits the filter taps that can reach its pixels (3x3 / 2: 4, 2, 2 or 1 taps;
// 1x1 / 2: three classes have none and just write zeros), and inside a class the gathered dY pixel moves by -1 per
// visited tap, like a stride-1 dgrad.
template <int WM, int WN, int TM, int TN, int PH, int MODE /*0 fwd, 1 dgrad*/, bool CLS = false>
__device__ __forceinline__ void pp_igemm_body(const IgemmArgs& P, unsigned char* const smem, uint32_t bid, uint32_t nblk) {
  static_assert(!CLS || MODE == 1, "parity classes are a dgrad notion");
  using G = PPGeom<WM, WN, TM, TN, PH>;
  constexpr int BM = G::BM, BN = G::BN, NA = G::NA, NB = G::NB, QM = G::QM, QN = G::QN;
  constexpr int STAGE = G::STAGE, AH_BYTES = G::AH_BYTES, BH_BYTES = G::BH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int grp = wave >> 2;                 // waves w and w + 4 share a SIMD: one of each group per SIMD
  // K split (forward only): logical block 2 t + h = half h of tile t's reduction
  const bool ksp = MODE == 0 && !CLS && P.ksplit == 2;
  const uint32_t lblk = xcd_remap(bid, nblk);
  const uint32_t tile = ksp ? lblk >> 1 : lblk;
  const int khalf = ksp ? (int)(lblk & 1u) : 0;
  const int tile_m = (int)(tile / (uint32_t)P.tiles_n);
  const int n0 = (int)(tile % (uint32_t)P.tiles_n) * BN;
  int m0 = tile_m * BM;
  int c_ph = 0, c_pw = 0, c_cnt = P.M, c_hc = 1, c_wc = 1;
  FastDiv c_fwc = {0u, 0u}, c_fhc = {0u, 0u};
  if (CLS) {
    // classes interleaved round-robin over tile_m (the XCD remap hands each XCD a contiguous range of tile_m and the
    // classes differ in work), the left-over tiles class by class -- the igemm_body map, on this body's tile height
    int c, local;
    const int il = P.cls_interleave * P.cls_n;
    if (tile_m < il) {
      c = P.cls_ids[tile_m % P.cls_n]; local = tile_m / P.cls_n;
    } else {
      int rest = tile_m - il;
      c = 0; local = 0;
      for (int k = 0; k < 4; ++k) {
        const int tc = P.cls_tile_begin[k + 1] - P.cls_tile_begin[k];
        const int have = P.cls_cnt[k] > 0 ? (tc > P.cls_interleave ? tc - P.cls_interleave : 0) : tc;
        if (rest < have) { c = k; local = (P.cls_cnt[k] > 0 ? P.cls_interleave : 0) + rest; break; }
        rest -= have;
      }
    }
    m0 = local * BM;
    c_ph = c / P.sw; c_pw = c % P.sw;
    c_cnt = P.cls_cnt[c]; c_hc = P.cls_hc[c]; c_wc = P.cls_wc[c];
    c_fwc = P.fd_cwc[c]; c_fhc = P.fd_chc[c];
  }
  // visited taps: (r0 + i * r_step, s0 + j * s_step), i < n_r, j < n_s
  const int r0 = CLS ? (c_ph + P.ph) % P.sh : 0, s0 = CLS ? (c_pw + P.pw) % P.sw : 0;
  const int r_step = CLS ? P.sh : 1, s_step = CLS ? P.sw : 1;
  const int n_r = CLS ? (r0 < P.KH ? (P.KH - r0 + r_step - 1) / r_step : 0) : P.KH;
  const int n_s = CLS ? (s0 < P.KW ? (P.KW - s0 + s_step - 1) / s_step : 0) : P.KW;
  // row index (class-local for CLS) -> output pixel (n, rh, rw)
#define PP_ROW_PIXEL(mm_, n_, rh_, rw_)                                                                  \
  {                                                                                                      \
    if (CLS) {                                                                                           \
      const int t_ = fdiv(mm_, c_fwc), w2_ = (mm_) - t_ * c_wc;                                          \
      n_ = fdiv(t_, c_fhc);                                                                              \
      rw_ = w2_ * P.sw + c_pw; rh_ = (t_ - n_ * c_hc) * P.sh + c_ph;                                     \
    } else {                                                                                             \
      const int t_ = fdiv(mm_, P.fd_rw);                                                                 \
      rw_ = (mm_) - t_ * P.RW; n_ = fdiv(t_, P.fd_rh); rh_ = t_ - n_ * P.RH;                             \
    }                                                                                                    \
  }

  // ---- per-thread DMA rows (fixed for the whole K loop) -----------------------------------------------------------
  // Piece h, instruction j of wave w fills piece rows (j*8 + w)*8 .. +7; lane l: row +(l >> 3), 16-byte slot l & 7,
  // fetching source chunk slot ^ ((row >> 1) & 7).
  int a_base[2][NA];
  uint32_t a_mask[2][NA];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int lr = (j * 8 + wave) * 8 + (lane >> 3);
      const int wmr = lr / (QM * 32), rem = lr % (QM * 32);
      const int m = m0 + wmr * TM * 32 + h * QM * 32 + rem;
      const int dchunk = (lane & 7) ^ ((lr >> 1) & 7);
      const bool ok = m < c_cnt;
      const int mm = ok ? m : 0;
      int n, rh, rw;
      PP_ROW_PIXEL(mm, n, rh, rw);
      // first visited tap's gathered pixel (c0, c1); visited tap (i, j) reads (c0 +- i, c1 +- j)
      int c0, c1;
      if (MODE == 0) { c0 = rh * P.sh - P.ph; c1 = rw * P.sw - P.pw; }
      else if (CLS) { c0 = (rh + P.ph - r0) / P.sh; c1 = (rw + P.pw - s0) / P.sw; }     // exact inside a parity class
      else { c0 = rh + P.ph; c1 = rw + P.pw; }
      a_base[h][j] = ((n * P.GH + c0) * P.GW + c1) * P.a_pix_stride + dchunk * 8;
      // bit (i * n_s + j) = visited tap (i, j) reads inside the image: rows i with 0 <= c0 +- i < GH, columns likewise
      uint32_t mk = 0u;
      if (ok) {
        const int s_lo = MODE == 0 ? -c1 : c1 - P.GW + 1, s_hi = MODE == 0 ? P.GW - 1 - c1 : c1;
        const int r_lo = MODE == 0 ? -c0 : c0 - P.GH + 1, r_hi = MODE == 0 ? P.GH - 1 - c0 : c0;
        const int sl = s_lo > 0 ? s_lo : 0, sh_ = s_hi < n_s - 1 ? s_hi : n_s - 1;
        const uint32_t cm = sh_ >= sl ? ((2u << sh_) - 1u) & ~((1u << sl) - 1u) : 0u;
        for (int r = 0; r < n_r; ++r)
          if (r >= r_lo && r <= r_hi) mk |= cm << (r * n_s);
      }
      a_mask[h][j] = mk;
    }
  int b_base[2][NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int lr = (j * 8 + wave) * 8 + (lane >> 3);
      const int wnc = lr / (QN * 32), rem = lr % (QN * 32);
      const int nn = n0 + wnc * TN * 32 + h * QN * 32 + rem;
      const int dchunk = (lane & 7) ^ ((lr >> 1) & 7);
      b_base[h][j] = nn * P.b_row_stride + dchunk * 8;
    }
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);
  const int kcb = P.Cred >> 6;
  const int KT_all = n_r * n_s * kcb;
  const int kt_first = ksp && khalf ? KT_all / 2 : 0;                       // this workgroup's K-tiles [kt_first, kt_first + KT)
  const int KT = ksp ? (khalf ? KT_all - KT_all / 2 : KT_all / 2) : KT_all;
  const int a_row_step = (MODE == 0 ? P.GW : -P.GW) * P.a_pix_stride, a_col_step = (MODE == 0 ? 1 : -1) * P.a_pix_stride;

#define PP_CURSOR_T PPCursor
#define PP_CURSOR_ZERO(c_) { (c_).tap = kt_first / kcb; (c_).cb = kt_first % kcb; (c_).r = (c_).tap / n_s; (c_).s = (c_).tap % n_s; }
#define PP_NEXT(c_) { if (++(c_).cb == kcb) { (c_).cb = 0; ++(c_).tap; if (++(c_).s == n_s) { (c_).s = 0; ++(c_).r; } } }
// (-DRIGL_ABLATE_A3: timing experiment only -- the A pieces of the second and third tap of a filter row are not fetched)
#ifdef RIGL_ABLATE_A3
#define PP_ABLATE_SKIP(c_) if ((c_).s == 0)
#else
#define PP_ABLATE_SKIP(c_)
#endif
#define PP_ISSUE_A(h_, stage_, c_)                                                                       \
  PP_ABLATE_SKIP(c_) {                                                                                   \
    const int da_ = (c_).r * a_row_step + (c_).s * a_col_step + ((c_).cb << 6);                          \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                                     \
      const bool ok_ = ((a_mask[h_][j] >> (c_).tap) & 1u) != 0u;                                         \
      const int off_ = ok_ ? (int)((uint32_t)(a_base[h_][j] + da_) * 2u) : (int)OOB;                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcA, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + (h_) * AH_BYTES + (j * 8 + wave) * 1024), \
          16, off_, 0, 0, 0);                                                                            \
    }                                                                                                    \
  }
#define PP_ISSUE_B(h_, stage_, c_)                                                                       \
  {                                                                                                      \
    const int db_ = ((r0 + (c_).r * r_step) * P.KW + s0 + (c_).s * s_step) * P.b_tap_stride + ((c_).cb << 6); \
    _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                     \
      const int off_ = (int)((uint32_t)(b_base[h_][j] + db_) * 2u);                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcB, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + 2 * AH_BYTES + (h_) * BH_BYTES + (j * 8 + wave) * 1024), \
          16, off_, 0, 0, 0);                                                                            \
    }                                                                                                    \
  }

  // ---- fragment read addresses --------------------------------------------------------------------------------------
  // MFMA operand fragment: lane l holds row (l & 31), 8 consecutive k at chunk 2*ks + (l >> 5) of the 64-wide K-tile.
  const int sw = ((lane & 31) >> 1) & 7, hi = lane >> 5;
  int kof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kof[ks] = (((ks * 2 + hi) ^ sw) << 4);
  const int a_row_off = (wm * QM * 32 + (lane & 31)) * 128, b_row_off = (wn * QN * 32 + (lane & 31)) * 128;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int AF = PH == 1 ? TM : QM;                // A fragments held at a time (PH = 1 holds both halves)
  bf16x8 af[AF][4], b0[QN][4], b1[QN][4];

  // (ab_ = first af[] slot filled)
#define PP_READ_A(h_, stage_, ab_)                                                                       \
  _Pragma("unroll") for (int i = 0; i < QM; ++i)                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      af[(ab_) + i][ks] = *reinterpret_cast<const bf16x8*>(smem + (stage_) * STAGE + (h_) * AH_BYTES + a_row_off + i * 4096 + kof[ks]);
#define PP_READ_B(dst_, h_, stage_)                                                                      \
  _Pragma("unroll") for (int j = 0; j < QN; ++j)                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      dst_[j][ks] = *reinterpret_cast<const bf16x8*>(smem + (stage_) * STAGE + 2 * AH_BYTES + (h_) * BH_BYTES + b_row_off + j * 4096 + kof[ks]);
  // quadrant (ha_, hb_): rows ha_*QM.., columns hb_*QN.. of the wave tile; D = W-fragment x X-fragment (transposed tile);
  // ab_ = af[] slot of the quadrant's first A fragment
#define PP_QUAD(ha_, hb_, bsrc_, ab_)                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
    _Pragma("unroll") for (int i = 0; i < QM; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < QN; ++j)                                                     \
        acc[(ha_) * QM + i][(hb_) * QN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                   \
            bsrc_[j][ks], af[(ab_) + i][ks], acc[(ha_) * QM + i][(hb_) * QN + j], 0, 0, 0);
  // loads that may stay in flight behind a wait = the youngest pieces of the issue order (any four consecutive pieces
  // are two A and two B halves)
  constexpr int W4 = 2 * NA + 2 * NB, W1 = NA;

  if (KT > 0) {        // (a parity class without a reachable tap -- 1x1 / 2: three of four -- only writes zeros)
#include "convpp_loop.inc"
  }
#undef PP_READ_A
#undef PP_READ_B
#undef PP_QUAD
#undef PP_ISSUE_A
#undef PP_ISSUE_B
#undef PP_NEXT
#undef PP_CURSOR_T
#undef PP_CURSOR_ZERO

  // ---- K split: hand-off of the partial tile (cdna_hip_programming.md, split-K hand-off in its counter form) --------------
  if constexpr (MODE == 0 && !CLS) {
    if (ksp) {
      constexpr int SLAB = BM * BN;                                     // floats per partial tile
      const __amdgpu_buffer_rsrc_t rs_my = make_rsrc(P.KS_SLAB + ((int64_t)tile * 2 + khalf) * SLAB, SLAB * 4);
      const __amdgpu_buffer_rsrc_t rs_other = make_rsrc(P.KS_SLAB + ((int64_t)tile * 2 + (khalf ^ 1)) * SLAB, SLAB * 4);
      // thread-major image: store g of a thread = 16 bytes at (g * 512 + tid) * 16 -- whole 8 KB rows per instruction;
      // write-through (sc1) stores, read back by the partner with sc1 loads wherever it runs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16& a = acc[i][j];
            const u32x4 v = {__float_as_uint(a[4 * q]), __float_as_uint(a[4 * q + 1]), __float_as_uint(a[4 * q + 2]), __float_as_uint(a[4 * q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_my, ((((i * TN + j) * 4 + q) * 512) + tid) * 16, 0, 16);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      volatile uint32_t* const flag = reinterpret_cast<volatile uint32_t*>(smem);
      if (tid == 0) *flag = __hip_atomic_fetch_add(P.KS_CNT + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const uint32_t ticket = *flag;
      __syncthreads();                                                   // (the flag's word is the first wave's staging area)
      if (ticket == 0u) return;                                          // first to arrive: the partner finishes the tile
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_other, ((((i * TN + j) * 4 + q) * 512) + tid) * 16, 0, 16);
            acc[i][j][4 * q] += __uint_as_float(v.x); acc[i][j][4 * q + 1] += __uint_as_float(v.y);
            acc[i][j][4 * q + 2] += __uint_as_float(v.z); acc[i][j][4 * q + 3] += __uint_as_float(v.w);
          }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // Every wave stages its own tile (half of its rows at a time) in a private LDS area and stores full 128-byte row
  // segments: no workgroup barrier.  D layout: col = lane & 31 -> pixel, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  // -> channel.
  constexpr int ROWB = G::EPI_ROWB, WCOLS = TN * 32, CH = WCOLS / 8, RPI = 64 / CH, ITERS = QM * 32 / RPI;
  static_assert(CH == 8, "a wave's 64 columns = 8 chunks of 16 bytes per row");
  unsigned char* const stg = smem + wave * G::EPI_WAVE;
  uint16_t* const C = static_cast<uint16_t*>(P.C);
  const int ch = lane & 7, rsub = lane >> 3;
  const int ncol = n0 + wn * WCOLS + ch * 8;
  float sy[8], sq[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) sy[c] = sq[c] = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < QM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = i * 32 + (lane & 31), col = j * 32 + 8 * q + 4 * (lane >> 5);
          const f32x16& a = acc[h * QM + i][j];
          const f32x2 lo = {a[4 * q], a[4 * q + 1]}, hi2 = {a[4 * q + 2], a[4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          *reinterpret_cast<uint2*>(stg + row * ROWB + col * 2) = pk;
        }
    const int mrow0 = m0 + wm * TM * 32 + h * QM * 32;
    // output row of each staged row: the row index itself, or (class-major rows) the pixel it enumerates; -1 = beyond the end
    int mout[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int lm = mrow0 + it * RPI + rsub;
      if (CLS) {
        int n, rh, rw;
        const int mm = lm < c_cnt ? lm : 0;
        PP_ROW_PIXEL(mm, n, rh, rw);
        mout[it] = lm < c_cnt ? (n * P.RH + rh) * P.RW + rw : -1;
      } else {
        mout[it] = lm < P.M ? lm : -1;
      }
    }
    uint4 addv[ITERS];
    if (P.ADD) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
        addv[it] = mout[it] >= 0 ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)mout[it] * P.ldc + ncol) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = it * RPI + rsub, m = mout[it];
      uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ch * 16);
      if (MODE == 0 && P.STATS) {
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float f = __uint_as_float((c & 1) ? (vw[c >> 1] & 0xFFFF0000u) : (vw[c >> 1] << 16));
          sy[c] += f; sq[c] = fmaf(f, f, sq[c]);
        }
      }
      if (m >= 0) {
        if (P.ADD) {
          const uint4 q = addv[it];
          v.x = add_bf16x2(v.x, q.x); v.y = add_bf16x2(v.y, q.y); v.z = add_bf16x2(v.z, q.z); v.w = add_bf16x2(v.w, q.w);
        }
        store16(C + (int64_t)m * P.ldc + ncol, v);
      }
    }
  }
  if (MODE == 0 && P.STATS) {
    // Batch-norm statistics of the bf16 outputs, one partial row per 128 output rows (rigl_conv2d_stats_parts): a wave's
    // column sums over its rows (rows beyond M hold exact zeros), lanes combined by a fixed xor tree, then the waves
    // that share a 128-row unit in wave-row order -- deterministic.
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
      for (int c = 0; c < 8; ++c) { sy[c] += __shfl_xor(sy[c], off); sq[c] += __shfl_xor(sq[c], off); }
    float* const wst = reinterpret_cast<float*>(smem + G::EPI_STATS);   // [8 waves][2][WCOLS]
    if (lane < 8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) { wst[(wave * 2 + 0) * WCOLS + lane * 8 + c] = sy[c]; wst[(wave * 2 + 1) * WCOLS + lane * 8 + c] = sq[c]; }
    }
    __syncthreads();
    constexpr int WPU = 128 / (TM * 32) > 0 ? 128 / (TM * 32) : 1;      // wave rows per 128-row unit (1 when a wave owns 128 rows)
    constexpr int UNITS = BM / 128;
    static_assert(TM * 32 <= 128, "a wave's rows fit one statistics unit");
    for (int idx = tid; idx < UNITS * 2 * BN; idx += 512) {
      const int u = idx / (2 * BN), k = (idx / BN) & 1, col = idx % BN;
      const int wcol = col / WCOLS, cc = col % WCOLS;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WPU; ++w) s += wst[(((u * WPU + w) * WN + wcol) * 2 + k) * WCOLS + cc];
      const int64_t unit = (int64_t)tile_m * UNITS + u;
      if (unit * 128 < P.M) P.STATS[unit * 2 * P.N + (int64_t)k * P.N + n0 + col] = s;
    }
  }
}

#undef PP_ROW_PIXEL
template <int WM, int WN, int TM, int TN, int PH, int MODE, bool CLS = false>
__global__ __launch_bounds__(512) void k_igemm_pp(IgemmArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pp[];
  pp_igemm_body<WM, WN, TM, TN, PH, MODE, CLS>(P, smem_pp, blockIdx.x, gridDim.x);
}

// ---- weight gradient on the same skeleton ---------------------------------------------------------------------------------
// dW[tap][ci][co] = sum over output pixels p of X[pixel(p, tap)][ci] * dY[p][co]  (dense fp32, sparse_optimizers_base.py:478-485
// through autodiff): rows = 256 input channels, columns = 256 output channels of one filter tap, reduction = a range of
// output pixels (split-K: the workgroup writes its partial tile into slab `split`, rigl::k1::launch_wgrad_reduce sums the
// slabs in a fixed order).  Both operands are reduction(pixel)-major in memory, so the LDS pieces are [64 pixels][128
// channels] images exactly as the DMA writes them and the fragments come out of ds_read_b64_tr_b16 (the k_wgrad_tr recipe:
// lane j of a 16-lane group supplies the address of 4 channels of pixel j/4 and receives channel j for 4 pixels); 64-byte
// quads of a pixel row are XORed with (pixel & 3) on the DMA's source side so the 4 pixel rows of a 32-lane pass sit in 4
// different bank groups.  Piece A_h = channels {wave row 0, 1} x 64-channel half h of X, piece B_h = {wave column 0..3}
// x 32-channel half h of dY.
struct PPWCursor { int kt; int pa[2], pb[2]; };

template <int PH>
__device__ __forceinline__ void pp_wgrad_body(const WgradArgs& P, unsigned char* const smem, uint32_t bid, uint32_t nblk) {
  constexpr int WN = 4, TM = 4, TN = 2;
  using G = PPGeom<2, 4, 4, 2, PH>;
  constexpr int NA = G::NA, NB = G::NB, QM = G::QM, QN = G::QN;
  constexpr int STAGE = G::STAGE, AH_BYTES = G::AH_BYTES, BH_BYTES = G::BH_BYTES;
  static_assert(NA == 2 && NB == 2 && QM == 2 && QN == 1, "256 x 256 tile, 64-pixel K-tiles");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int grp = wave >> 2;
  uint32_t b = xcd_remap(bid, nblk);
  const int tco = (int)(b % (uint32_t)P.tiles_co); b /= (uint32_t)P.tiles_co;
  const int tci = (int)(b % (uint32_t)P.tiles_ci); b /= (uint32_t)P.tiles_ci;
  const int taps = P.KH * P.KW;
  const int tap = (int)(b % (uint32_t)taps), split = (int)(b / (uint32_t)taps);
  const int r = tap / P.KW, s = tap - r * P.KW;
  const int ci0 = tci * 256, co0 = tco * 256;
  const int KT_all = (P.M + 63) >> 6;
  // a split takes every splits-th K-tile ("wgrad_il", default) or a contiguous range: interleaved, the splits of a tile
  // stream ONE window of the tensors together instead of `splits` streams far apart (bwd1x1.hpp measured the same)
  const bool il = P.interleave != 0 && P.splits > 1;
  const int kt_begin = il ? split : (int)((int64_t)KT_all * split / P.splits);
  const int kt_step = il ? P.splits : 1;
  const int KT = il ? (split < KT_all ? (KT_all - split + P.splits - 1) / P.splits : 0)
                    : (int)((int64_t)KT_all * (split + 1) / P.splits) - kt_begin;

  // DMA lanes: instruction jj of wave w fills pixel rows 4 * (jj*8 + w) .. +3 of a piece; lane l: pixel +(l >> 4), 16-byte
  // slot l & 15, fetching channel chunk slot ^ ((pixel & 3) << 2)
  const int px_lane = 4 * wave + (lane >> 4);                       // + 32 * jj
  const int chunk = (lane & 15) ^ (((lane >> 4) & 3) << 2);
  int chan_a[2], chan_b[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    chan_a[h] = ci0 + (chunk >> 3) * 128 + h * 64 + (chunk & 7) * 8;
    chan_b[h] = co0 + (chunk >> 2) * 64 + h * 32 + (chunk & 3) * 8;
  }
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.X, P.x_bytes), rsrcB = make_rsrc(P.DY, P.dy_bytes);
  const int hi0 = r - P.ph, wi0 = s - P.pw;

#define PP_CURSOR_T PPWCursor
#define PP_WCOMPUTE(c_)                                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                   \
      const int p_ = ((kt_begin + (c_).kt * kt_step) << 6) + 32 * jj + px_lane;                          \
      const bool ok_ = p_ < P.M;                                                                         \
      const int pp_ = ok_ ? p_ : 0;                                                                      \
      const int t_ = fdiv(pp_, P.fd_wo);                                                                 \
      const int wo_ = pp_ - t_ * P.Wo, n_ = fdiv(t_, P.fd_ho), ho_ = t_ - n_ * P.Ho;                     \
      const int hi_ = ho_ * P.sh + hi0, wi_ = wo_ * P.sw + wi0;                                          \
      const bool oka_ = ok_ && (unsigned)hi_ < (unsigned)P.H && (unsigned)wi_ < (unsigned)P.W;          \
      (c_).pa[jj] = oka_ ? ((n_ * P.H + hi_) * P.W + wi_) * P.x_pix_stride : -1;                         \
      (c_).pb[jj] = ok_ ? p_ * P.Cout : -1;                                                              \
    }                                                                                                    \
  }
#define PP_CURSOR_ZERO(c_) { (c_).kt = 0; PP_WCOMPUTE(c_); }
#define PP_NEXT(c_) { ++(c_).kt; PP_WCOMPUTE(c_); }
#define PP_ISSUE_A(h_, stage_, c_)                                                                       \
  {                                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                   \
      const int off_ = (c_).pa[jj] >= 0 ? (int)((uint32_t)((c_).pa[jj] + chan_a[h_]) * 2u) : (int)OOB;   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcA, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + (h_) * AH_BYTES + (jj * 8 + wave) * 1024), \
          16, off_, 0, 0, 0);                                                                            \
    }                                                                                                    \
  }
#define PP_ISSUE_B(h_, stage_, c_)                                                                       \
  {                                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                   \
      const int off_ = (c_).pb[jj] >= 0 ? (int)((uint32_t)((c_).pb[jj] + chan_b[h_]) * 2u) : (int)OOB;   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcB, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + 2 * AH_BYTES + (h_) * BH_BYTES + (jj * 8 + wave) * 1024), \
          16, off_, 0, 0, 0);                                                                            \
    }                                                                                                    \
  }

  // transposing fragment reads: a 32-channel x 16-pixel operand fragment = two ds_read_b64_tr_b16 (pixels +0..3, +4..7 of
  // the lane's 8); lane (g, j): pixel row 8 * (g >> 1) + (j >> 2), 16-byte chunk 2 * (g & 1) + ((j >> 1) & 1) of the
  // fragment's four, bytes (j & 1) * 8; the row's quad swizzle (pixel & 3 = (j >> 2) & 3) moves the fragment's quad
  const int g = lane >> 4, j16 = lane & 15, rs = (j16 >> 2) & 3;
  const int tr_row = (8 * (g >> 1) + (j16 >> 2)) * 256, tr_low = ((2 * (g & 1) + ((j16 >> 1) & 1)) << 4) + (j16 & 1) * 8;
  int a_tr[QM];
#pragma unroll
  for (int i = 0; i < QM; ++i) a_tr[i] = tr_row + ((((wm * 2 + i) ^ rs) * 4) << 4) + tr_low;
  const int b_tr = tr_row + (((wn ^ rs) * 4) << 4) + tr_low;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < TN; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  constexpr int AF = PH == 1 ? TM : QM;
  bf16x8 af[AF][4], b0[QN][4], b1[QN][4];

#define PP_READ_A(h_, stage_, ab_)                                                                       \
  _Pragma("unroll") for (int i = 0; i < QM; ++i)                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      af[(ab_) + i][ks] = lds_read_tr_pair(smem + (stage_) * STAGE + (h_) * AH_BYTES + a_tr[i] + ks * 4096,        \
                                           smem + (stage_) * STAGE + (h_) * AH_BYTES + a_tr[i] + ks * 4096 + 1024);
#define PP_READ_B(dst_, h_, stage_)                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
    dst_[0][ks] = lds_read_tr_pair(smem + (stage_) * STAGE + 2 * AH_BYTES + (h_) * BH_BYTES + b_tr + ks * 4096,    \
                                   smem + (stage_) * STAGE + 2 * AH_BYTES + (h_) * BH_BYTES + b_tr + ks * 4096 + 1024);
  // D[ci][co]: the X fragment is the MFMA's first operand
#define PP_QUAD(ha_, hb_, bsrc_, ab_)                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
    _Pragma("unroll") for (int i = 0; i < QM; ++i)                                                       \
      acc[(ha_) * QM + i][(hb_)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                              \
          af[(ab_) + i][ks], bsrc_[0][ks], acc[(ha_) * QM + i][(hb_)], 0, 0, 0);
  constexpr int W4 = 2 * NA + 2 * NB, W1 = NA;
#include "convpp_loop.inc"
#undef PP_READ_A
#undef PP_READ_B
#undef PP_QUAD
#undef PP_ISSUE_A
#undef PP_ISSUE_B
#undef PP_NEXT
#undef PP_CURSOR_T
#undef PP_CURSOR_ZERO
#undef PP_WCOMPUTE

  // partial tile -> slab `split` (or dW itself when the layer is not split).  D row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  // -> ci, column = lane & 31 -> co: stored straight from the accumulators that is 128 four-byte stores per lane, and those
  // stores -- not the MFMAs -- ended the workgroup (256 KB per workgroup, every workgroup of a round at the same moment).
  // Staged through a private LDS area per wave, half of its rows at a time ([64 rows][64 floats]), a lane stores 16
  // contiguous bytes and a wave instruction four whole 256-byte row segments.
  float* const out = P.OUT + (int64_t)split * P.slab_elems + (int64_t)tap * P.Cin * P.Cout;
  constexpr int SROW = 64;        // (256-byte rows: the 16-lane groups of ds_read_b128 land on distinct bank quarters without padding)
  float* const stg = reinterpret_cast<float*>(smem) + wave * 64 * SROW;
  static_assert(8 * 64 * SROW * 4 <= 2 * STAGE, "epilogue staging fits the K-tile stages");
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < QM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          stg[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * SROW + jj * 32 + (lane & 31)] = acc[h * QM + i][jj][e];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(stg + row * SROW + c4);
      const int ci = ci0 + wm * 128 + h * 64 + row, co = co0 + wn * 64 + c4;
      *reinterpret_cast<float4*>(out + (int64_t)ci * P.Cout + co) = v;
    }
  }
}

template <int PH>
__global__ __launch_bounds__(512) void k_wgrad_pp(WgradArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pp[];
  pp_wgrad_body<PH>(P, smem_pp, blockIdx.x, gridDim.x);
}

template <int WMD, int WND, int TMD, int TND, int PHD, bool CLSD = false>
__global__ __launch_bounds__(512) void k_bwd_pp(IgemmArgs PD, WgradArgs PW, uint32_t nd, uint32_t nw, uint32_t wgrad_first) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pp[];
  // longest jobs first: whichever body has the longer reduction per workgroup takes the low block indices
  const uint32_t b = blockIdx.x;
  const bool is_w = wgrad_first ? b < nw : b >= nd;
  if (is_w) {
    pp_wgrad_body<2>(PW, smem_pp, wgrad_first ? b : b - nd, nw);
  } else {
    pp_igemm_body<WMD, WND, TMD, TND, PHD, 1, CLSD>(PD, smem_pp, wgrad_first ? b - nw : b, nd);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Variant ids (rigl_tune_set("pp_fwd" / "pp_dgrad", id) forces one; 0 = never; -1 = the built-in rule)
enum { PP_NONE = 0, PP_256x256 = 1, PP_128x256 = 2, PP_256x128 = 3, PP_512x128 = 4 };

struct PPPlan { int variant; unsigned grid; int bm, bn; };



template <int WM, int WN, int TM, int TN, int PH, int MODE, bool CLS = false>
static bool pp_ready() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm_pp<WM, WN, TM, TN, PH, MODE, CLS>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                PPGeom<WM, WN, TM, TN, PH>::SMEM) == hipSuccess;
  return ready;
}
template <int WM, int WN, int TM, int TN, int PH, int MODE, bool CLS = false>
static bool pp_launch_one(dim3 grid, const IgemmArgs& a, hipStream_t st) {
  if (!pp_ready<WM, WN, TM, TN, PH, MODE, CLS>()) return false;
  RIGL_K_LAUNCH((k_igemm_pp<WM, WN, TM, TN, PH, MODE, CLS>), grid, dim3(512), (PPGeom<WM, WN, TM, TN, PH>::SMEM), st, a);
  return true;
}
// Parity-class tables of a strided dgrad on tiles of bm rows (the igemm plan's, on this body's tile height); returns
// the number of row tiles.
static int pp_fill_classes(IgemmArgs& a, int bm) {
  const int n_img = a.M / (a.RH * a.RW);
  int tiles = 0;
  for (int c = 0; c < 4; ++c) {
    const int ph = c / a.sw, pw = c % a.sw;
    int hc = 0, wc = 0;
    if (ph < a.sh && c < a.sh * a.sw) { hc = (a.RH - ph + a.sh - 1) / a.sh; wc = (a.RW - pw + a.sw - 1) / a.sw; }
    a.cls_hc[c] = hc > 0 ? hc : 1; a.cls_wc[c] = wc > 0 ? wc : 1;
    a.fd_chc[c] = make_fastdiv(a.cls_hc[c]); a.fd_cwc[c] = make_fastdiv(a.cls_wc[c]);
    a.cls_cnt[c] = hc > 0 && wc > 0 ? n_img * hc * wc : 0;
    a.cls_tile_begin[c] = tiles;
    tiles += (a.cls_cnt[c] + bm - 1) / bm;
  }
  a.cls_tile_begin[4] = tiles;
  a.cls_n = 0; a.cls_interleave = 1 << 30;
  for (int c = 0; c < 4; ++c) {
    const int tc = a.cls_tile_begin[c + 1] - a.cls_tile_begin[c];
    if (tc > 0) { a.cls_ids[a.cls_n++] = c; if (tc < a.cls_interleave) a.cls_interleave = tc; }
  }
  if (a.cls_n == 0) { a.cls_n = 1; a.cls_ids[0] = 0; a.cls_interleave = 0; }
  return tiles;
}
static inline bool pp_strided(const IgemmArgs& a) { return a.sh > 1 || a.sw > 1; }
// row tiles of a ping-pong GEMM (class-major for a strided dgrad)
template <int MODE>
static int pp_tiles_m(const IgemmArgs& a, int bm) {
  if (MODE == 1 && pp_strided(a)) { IgemmArgs t = a; return pp_fill_classes(t, bm); }
  return (a.M + bm - 1) / bm;
}

static inline void pp_dims(int variant, int& bm, int& bn) {
  switch (variant) {
    case PP_256x256: bm = 256; bn = 256; break;
    case PP_128x256: bm = 128; bn = 256; break;
    case PP_256x128: bm = 256; bn = 128; break;
    case PP_512x128: bm = 512; bn = 128; break;
    default: bm = bn = 0;
  }
}

// Is the ping-pong body legal for this GEMM at all?
template <int MODE>
static bool pp_legal(const IgemmArgs& a, int variant) {
  int bm, bn;
  pp_dims(variant, bm, bn);
  if (!bm) return false;
  if (a.Cred % 64 || a.a_pix_stride != a.Cred) return false;
  if (a.N % bn) return false;
  if (a.KH * a.KW > 32) return false;
  if (MODE == 1 && (a.sh > 2 || a.sw > 2)) return false;     // strided dgrad: parity classes, strides <= 2 like igemm_body
  if (a.BNX) return false;
  if (a.M < bm) return false;
  return true;
}

// Which ping-pong tile the shared backward launch gives this layer's dgrad (PP_NONE: the layer's backward stays on the
// igemm / tr bodies).  `a` = the dgrad GEMM (N = cin, Cred = cout, gathered tensor = dY).
static int pp_bwd_dgrad_variant(const IgemmArgs& a) {
  const int pp_bwd = RIGL_TUNE("pp_bwd", -1);
  if (pp_bwd == 0) return PP_NONE;
  const int64_t m_out = (int64_t)(a.M / (a.RH * a.RW)) * a.GH * a.GW;
  // the weight gradient that shares the launch has 256-channel tiles: cin (= N here) and cout (= Cred) multiples of 256
  if (a.N % 256 || a.Cred % 256 || m_out < 256 || a.KH * a.KW > 32 || a.BNX || a.sh > 2 || a.sw > 2) return PP_NONE;
  if (pp_bwd > 0) return pp_bwd == 2 ? PP_128x256 : PP_256x256;
  // built-in rule (tools/pp_sweep.py --passes bwd, batch 128): dgrad reductions of >= 16 K-tiles (a strided dgrad: its
  // longest parity class, ceil(KH / sh) x ceil(KW / sw) taps); 256x256 dgrad tiles where they give at least ~1/3 of the
  // CUs a tile (a strided dgrad counts the tiles of ONE class), the 128-row tiles below that
  const int kt_d = ((a.KH + a.sh - 1) / a.sh) * ((a.KW + a.sw - 1) / a.sw) * (a.Cred / 64);
  if (kt_d < RIGL_TUNE("pp_bwd_min_kt", 16)) return PP_NONE;     // (8-tile reductions measured neutral to slightly slower)
  const int64_t rows = a.M / (a.sh * a.sw);
  return ((rows + 255) / 256) * (a.N / 256) >= 90 ? PP_256x256 : PP_128x256;
}

template <int MODE>
static PPPlan plan_pp(const IgemmArgs& a) {
  PPPlan p = {PP_NONE, 0u, 0, 0};
  const int forced = MODE == 0 ? RIGL_TUNE("pp_fwd", -1) : RIGL_TUNE("pp_dgrad", -1);
  int v = PP_NONE;
  if (forced >= 0) v = forced;
  else if (MODE == 0) {
    // Built-in rule (measured per layer at batch 128 and 512, tools/pp_sweep.py, profiles/r3/pp_sweep_*.txt): reductions
    // of >= 8 K-tiles of 64; the 128x64-per-wave tiles (256x256 / 512x128) where they still give ~3/4 of the CUs a
    // workgroup (one workgroup per CU), else the 64x64-per-wave tiles with half the rows.
    const int kt = a.KH * a.KW * (a.Cred / 64);
    const int64_t fill = (int64_t)num_cus() * 3 / 4;
    if (kt >= 8) {
      if (a.N % 256 == 0) v = (int64_t)((a.M + 255) / 256) * (a.N / 256) >= fill ? PP_256x256 : PP_128x256;
      else if (a.N % 128 == 0) v = (int64_t)((a.M + 511) / 512) * (a.N / 128) >= fill ? PP_512x128 : PP_256x128;
    }
  }
  // dgrad: a layer's dX comes from ONE kernel whichever entry point computes it, so the stand-alone dgrad follows the
  // shared backward launch's choice ("pp_bwd", below): the ping-pong dgrad where that launch runs on the 8-wave bodies.
  else v = pp_bwd_dgrad_variant(a);
  if (v == PP_NONE || !pp_legal<MODE>(a, v)) return p;
  pp_dims(v, p.bm, p.bn);
  p.variant = v;
  p.grid = (unsigned)(pp_tiles_m<MODE>(a, p.bm) * (a.N / p.bn));
  return p;
}

// Forward layers whose tiles would occupy at most half of the CUs (the 7x7 layers at batch 128: 98 tiles) split the
// reduction in two ("pp_ksplit" = 0: never).  Measured at batch 128 (forward alone, us): 7x7x512 3x3 52.4 -> 43.5 (565 -> 681
// TFLOP/s), its stride-2 sibling 50.9 -> 42.9, 2048->512 1x1 (32 K-tiles) 26.9 -> 28.5 -- the 256 KB hand-off per tile
// (write-through stores, drained before the ticket; the partner's read) costs ~8 us, so only reductions of >= 48 K-tiles
// split ("pp_ksplit_min_kt"; round 6 measured 32: the 7x7 2048 -> 512 forwards join, conv_fwd -0.015 ms per step -- but the
// 14x14 3x3 layer at HALF the batch (98 tiles, 36 K-tiles) then splits while the full batch does not, and a layer's bits would
// depend on the batch size: tests/test_fullsize_properties_gpu.py; 48 stays).  Forward only: a layer's dX must have the same bits from the stand-alone dgrad and from the shared backward
// launch, which does not split (its weight-gradient workgroups already fill the chip).
static inline bool pp_ksplit_ok(const IgemmArgs& a, const PPPlan& p) {
  if (!p.variant || RIGL_TUNE("pp_ksplit", 1) == 0) return false;
  const int kt = a.KH * a.KW * (a.Cred / 64);
  // (7x7 output grids from 32 K-tiles: the 2048 -> 512 forwards, 98 tiles at batch 128 and 49 at 64 -- the same side of the
  // tile-count rule at both, unlike the 14x14 layer above; conv_fwd 1.955 -> 1.935 ms)
  const int min_kt = a.RH * a.RW <= 64 ? 32 : RIGL_TUNE("pp_ksplit_min_kt", 48);
  return 2 * (int64_t)p.grid <= (int64_t)num_cus() && kt >= min_kt;
}

template <int MODE>
static bool launch_pp(const PPPlan& p, const IgemmArgs& a0, hipStream_t st) {
  IgemmArgs a = a0;
  a.fd_rw = make_fastdiv(a.RW); a.fd_rh = make_fastdiv(a.RH);
  a.tiles_n = a.N / p.bn;
  const dim3 grid(MODE == 0 && a.ksplit == 2 ? 2 * p.grid : p.grid);
  // phases per K-tile: the 128x64 wave tiles run 2 (16 MFMAs between barriers), the 64x64 wave tiles 1 with three stages
  if constexpr (MODE == 1) {
    if (pp_strided(a)) {
      pp_fill_classes(a, p.bm);
      switch (p.variant) {
        case PP_256x256: return pp_launch_one<2, 4, 4, 2, 2, 1, true>(grid, a, st);
        case PP_128x256: return pp_launch_one<2, 4, 2, 2, 1, 1, true>(grid, a, st);
        case PP_256x128: return pp_launch_one<4, 2, 2, 2, 1, 1, true>(grid, a, st);
        case PP_512x128: return pp_launch_one<4, 2, 4, 2, 2, 1, true>(grid, a, st);
        default: return false;
      }
    }
  }
  switch (p.variant) {
    case PP_256x256: return pp_launch_one<2, 4, 4, 2, 2, MODE>(grid, a, st);
    case PP_128x256: return pp_launch_one<2, 4, 2, 2, 1, MODE>(grid, a, st);
    case PP_256x128: return pp_launch_one<4, 2, 2, 2, 1, MODE>(grid, a, st);
    case PP_512x128: return pp_launch_one<4, 2, 4, 2, 2, MODE>(grid, a, st);
    default: return false;
  }
}

// ---- backward on the ping-pong bodies -------------------------------------------------------------------------------------
// Knobs: "pp_wgrad" (stand-alone weight gradient: -1 rule = off, 1 wherever legal), "pp_bwd" (the shared launch: -1 rule,
// 0 never, 1 dgrad on the 128x64-per-wave tile (256x256), 2 dgrad on the 64x64-per-wave tile (128x256)), "pp_slab_mb".
struct PPBwdPlan {
  bool use;
  int tiles_ci, tiles_co, splits;
  int64_t slab;         // elements of one partial slab = the whole dW
  unsigned nd, nw;
  bool wgrad_first;
};

// 256-channel tiles of dW; the tiny- / small-Cin repack paths keep their own kernels
static inline bool pp_wgrad_legal(const RiglConvDesc* d) {
  const int64_t M = (int64_t)d->n * d->ho * d->wo;
  return d->cin % 256 == 0 && d->cout % 256 == 0 && M >= 256 && d->kh * d->kw <= 32;
}
// Upper bound of the split count of any ping-pong weight-gradient plan (the workspace is sized for it): at least 256
// pixels per split, at most two rounds of workgroups, at most "pp_slab_mb" (40) MB of slabs -- every split is one more
// fp32 copy of dW written and read back (16 MB: slower on every layer measured, 24 / 40 / 64 MB within noise of each other).
static inline int pp_wgrad_max_splits(const RiglConvDesc* d) {
  const int64_t M = (int64_t)d->n * d->ho * d->wo, kt_all = (M + 63) / 64;
  const int64_t base = (int64_t)d->kh * d->kw * (d->cin / 256) * (d->cout / 256);
  const int64_t dw_bytes = (int64_t)d->kh * d->kw * d->cin * d->cout * 4;
  int64_t s = kt_all / 4;
  const int64_t by_slots = (2 * (int64_t)num_cus() + base - 1) / base, by_bytes = ((int64_t)RIGL_TUNE("pp_slab_mb", 40) << 20) / dw_bytes;
  if (s > by_slots) s = by_slots;
  if (s > by_bytes) s = by_bytes;
  return (int)(s < 1 ? 1 : s);
}
static inline size_t pp_wgrad_workspace(const RiglConvDesc* d) {
  if (!pp_wgrad_legal(d)) return 0;
  const int sp = pp_wgrad_max_splits(d);
  return sp > 1 ? (size_t)sp * d->kh * d->kw * d->cin * d->cout * 4 : 0;
}

// nd = dgrad workgroups that share the launch (0: stand-alone weight gradient), kt_d = their length in 256x256x64 K-tiles
static PPBwdPlan plan_wgrad_pp(const RiglConvDesc* d, unsigned nd, int kt_d) {
  PPBwdPlan p = {};
  p.tiles_ci = d->cin / 256; p.tiles_co = d->cout / 256;
  p.slab = (int64_t)d->kh * d->kw * d->cin * d->cout;
  const int64_t M = (int64_t)d->n * d->ho * d->wo, kt_all = (M + 63) / 64;
  const int64_t base = (int64_t)d->kh * d->kw * p.tiles_ci * p.tiles_co, cus = num_cus();
  // whole rounds of one workgroup per CU, with room for at least `base` (and a quarter of a round of) weight-gradient
  // workgroups behind the dgrad tiles
  const int64_t room = base > cus / 4 ? base : cus / 4;
  const int64_t slots = ((int64_t)nd + room + cus - 1) / cus * cus;
  int64_t s = (slots - (int64_t)nd) / base;
  const int smax = pp_wgrad_max_splits(d);
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  p.splits = (int)s;
  p.nd = nd; p.nw = (unsigned)(base * s);
  p.wgrad_first = kt_all / s >= kt_d;          // longest jobs first
  p.use = true;
  return p;
}

static WgradArgs pp_wgrad_args(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const PPBwdPlan& p, float* dw,
                               void* workspace) {
  WgradArgs a = {};
  a.interleave = RIGL_TUNE("wgrad_il", 1);
  a.X = x; a.DY = dy; a.M = d->n * d->ho * d->wo; a.Cin = d->cin; a.Cout = d->cout; a.KH = d->kh; a.KW = d->kw;
  a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_top; a.pw = d->pad_left;
  a.tiles_ci = p.tiles_ci; a.tiles_co = p.tiles_co; a.splits = p.splits; a.slab_elems = p.slab;
  a.x_bytes = (uint32_t)((size_t)d->n * d->h * d->w * d->cin * 2);
  a.dy_bytes = (uint32_t)((size_t)a.M * d->cout * 2);
  a.x_pix_stride = d->cin;
  a.fd_wo = make_fastdiv(d->wo); a.fd_ho = make_fastdiv(d->ho);
  a.OUT = p.splits > 1 ? static_cast<float*>(workspace) : dw;
  return a;
}

static bool pp_wgrad_launch(const PPBwdPlan& p, const WgradArgs& aw, hipStream_t st) {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_pp<2>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                PPGeom<2, 4, 4, 2, 2>::SMEM) == hipSuccess;
  if (!ready) return false;
  RIGL_K_LAUNCH((k_wgrad_pp<2>), dim3(p.nw), dim3(512), (PPGeom<2, 4, 4, 2, 2>::SMEM), st, aw);
  return true;
}
template <int WMD, int WND, int TMD, int TND, int PHD, bool CLSD>
static bool pp_bwd_launch_one(const IgemmArgs& ad, const WgradArgs& aw, unsigned nd, unsigned nw, bool wgrad_first, hipStream_t st) {
  constexpr int SM_D = PPGeom<WMD, WND, TMD, TND, PHD>::SMEM, SM_W = PPGeom<2, 4, 4, 2, 2>::SMEM, SM = SM_D > SM_W ? SM_D : SM_W;
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_pp<WMD, WND, TMD, TND, PHD, CLSD>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, SM) == hipSuccess;
  if (!ready) return false;
  RIGL_K_LAUNCH((k_bwd_pp<WMD, WND, TMD, TND, PHD, CLSD>), dim3(nd + nw), dim3(512), SM, st, ad, aw, nd, nw, (uint32_t)(wgrad_first ? 1u : 0u));
  return true;
}
// dvar = the dgrad tile (256 output columns: the weight gradient's 256-channel tiles imply cin % 256 == 0), strided =
// parity-class dgrad
static bool pp_bwd_launch(int dvar, bool strided, const IgemmArgs& ad, const WgradArgs& aw, const PPBwdPlan& pw, hipStream_t st) {
  if (strided) {
    switch (dvar) {
      case PP_256x256: return pp_bwd_launch_one<2, 4, 4, 2, 2, true>(ad, aw, pw.nd, pw.nw, pw.wgrad_first, st);
      case PP_128x256: return pp_bwd_launch_one<2, 4, 2, 2, 1, true>(ad, aw, pw.nd, pw.nw, pw.wgrad_first, st);
      default: return false;
    }
  }
  switch (dvar) {
    case PP_256x256: return pp_bwd_launch_one<2, 4, 4, 2, 2, false>(ad, aw, pw.nd, pw.nw, pw.wgrad_first, st);
    case PP_128x256: return pp_bwd_launch_one<2, 4, 2, 2, 1, false>(ad, aw, pw.nd, pw.nw, pw.wgrad_first, st);
    default: return false;
  }
}
