// K1, forward / dgrad of a 3x3 stride-1 SAME convolution with the INPUT SLAB resident in LDS ("c3").
// Included by conv.hip inside namespace rigl::k1.
//
// The igemm kernel treats a 3x3 conv as nine 1x1 GEMMs: the activation operand crosses L2 -> LDS nine times and
// every K-tile pays the loader.  Here a workgroup owns a tile of whole image rows (or whole images) and, per block of
// 32 channels, fetches the pixels it needs ONCE into LDS; the nine taps then read that slab as shifted views while
// only the 8 KB weight tiles stream through a 3-deep DMA ring.
//
// Layout that makes the views free.  The slab is ZERO-PADDED and its MFMA rows enumerate PADDED ENTRIES:
//   pixel (n, h, w)  <->  padded entry  PG = (n (H + 1) + h) RP + w + 1,   RP = W + 1
// entry (row, 0) is a zero pixel (w = -1 of its row and w = W of the row above), one all-zero row follows every
// image.  Row r of the tile IS entry tile_P0 + r, so
//   * tap (dh, dw) is the constant entry offset dh RP + dw: one scalar add per K-tile, no bounds test, no select;
//   * a wave's 32 lanes read 32 CONSECUTIVE entries, 80 bytes apart (64 of data + 16 unused): 16 consecutive entries
//     fall into 16 distinct 16-byte bank slots (5 r mod 16) -- ds_read_b128 is conflict-free without an XOR swizzle
//     (the first version interleaved pads with pixels: 43 % of its LDS cycles were bank conflicts, profiles/r2);
//   * row block i of a wave is the immediate offset i * 2560 from the lane's base address.
//   Rows that are pads, gap rows or beyond the tile's RT RP entries are computed and discarded (the same 12.5 % of the
//   MFMA slots a 196-row tile spends on padding to 224).  `buffer_load ... lds` writes lane-linearly, so every fifth
//   lane lands on the unused tail of an entry and pad entries are out-of-range lanes (the hardware writes zeros).
//
// Register tile.  4 waves = (2 along N | 2 along M) x 2 along K: a wave owns ALL row blocks (7, or 4 of 8) x 2 column
// blocks and multiplies one of the two 16-wide K-steps of every 32-channel K-tile; the two partial sums are added
// through LDS at the end in a fixed order, identical in every variant (so the bits do not depend on the plan).
// Per K-step that is 9 (or 6) fragment reads for 14 (8) MFMAs -- 0.64 KB of LDS reads per MFMA where one column
// block per wave needs 1.14 -- with 224 + 72 VGPRs (one wave per SIMD, the 512-register budget).
// Main loop: the software pipeline across the barrier of conv196.hpp (fragments of tile kt + 1 are requested before
// the MFMAs of tile kt are issued; the slab of channel block cb + 1 is fetched during the nine K-tiles of block cb).
// The lane-level index arithmetic is restated and checked on the CPU in tools/emu/c3_emu.py.

#ifndef RIGL_C3_ABLATE
#define RIGL_C3_ABLATE 0      // development ablations: 1 no MFMA, 2 no fragment reads, 4 no DMA, 8 no per-tile barrier
#endif

struct C3Args {
  const uint16_t* A;   // activations: x (fwd) or dy (dgrad), NHWC
  const uint16_t* B;   // packed weights: OHWI shadow (fwd) / HWIO shadow (dgrad)
  void* C;
  const uint16_t* ADD;
  float* STATS;        // [tiles_m][2][N] or NULL
  int M, N, Cred, H, W, nimg;
  int RP, RT;          // padded row pitch (W + 1); padded rows per tile
  int per;             // tiles per image (tiles inside an image), or 0: a tile is RT / (H + 1) whole images
  int b_row_stride, b_tap_stride, ldc, tiles_n;
  uint32_t a_bytes, b_bytes;
  FastDiv fd_rp, fd_h1, fd_per;   // / RP, / (H + 1), / per
};

constexpr int C3_SLAB_BYTES = 24576, C3_PITCH = 80;

template <int BN, int RBT>
constexpr int c3_smem_bytes() {
  constexpr int WM = BN == 128 ? 1 : 2, RB = (RBT + WM - 1) / WM, BMC = RBT * 32;
  constexpr int RING = 2 * C3_SLAB_BYTES + 3 * BN * 64;
  constexpr int DUMP = 2 * RB * 2 * 16 * 64 * 4;                       // two waves hand their accumulators over
  constexpr int EPI = BMC * (BN + 8) * 2 + BMC * 4 + THREADS * 8;      // bf16 staging + row table + statistics scratch
  constexpr int a = RING > DUMP ? RING : DUMP;
  return a > EPI ? a : EPI;
}

template <int MODE /*0 fwd, 1 dgrad*/, int BN, int RBT>
__global__ __launch_bounds__(THREADS, 1) void k_c3(C3Args P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WN = BN / 64, WM = BN == 128 ? 1 : 2, RB = (RBT + WM - 1) / WM, LB = BN / 64, BMC = RBT * 32;
  constexpr int SLAB_B = C3_SLAB_BYTES, PITCH = C3_PITCH, B_STAGE = BN * 64, OFF_B = 2 * SLAB_B, LA = SLAB_B / 4096;
  constexpr int CS_LD = BN + 8, EPI = BMC * CS_LD * 2;
  static_assert(WN * WM == 2, "two wave columns/rows x two K halves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave & 1, wn = (wave >> 1) % WN, wm = (wave >> 1) / WN;
  const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = (int)(tile / (uint32_t)P.tiles_n);
  const int n0 = (int)(tile % (uint32_t)P.tiles_n) * BN;
  int prow0;                                  // padded row of the tile's row 0
  if (P.per > 0) { const int n = fdiv(tile_m, P.fd_per); prow0 = n * (P.H + 1) + (tile_m - n * P.per) * P.RT; }
  else prow0 = tile_m * P.RT;
  const int RP = P.RP;
  const int S0 = prow0 * RP - RP - 1;         // global padded entry held by slab entry 0 (may be negative: tile 0)
  const int l4 = lane >> 2, dchunk = (lane & 3) ^ ((lane >> 4) & 3), hi = lane >> 5, lr0 = lane & 31;
  const int CB = P.Cred >> 5, KT = CB * 9;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);

  // ---- per-lane DMA source offsets ----------------------------------------------------------------------------
  uint32_t vb[LB], va[LA];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int row = (j * 4 + wave) * 16 + l4, n = n0 + row;
    vb[j] = (row < BN && n < P.N) ? (uint32_t)(n * P.b_row_stride + dchunk * 8) * 2u : OOB;
  }
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int o = (j * 4 + wave) * 1024 + lane * 16;
    const int e = o / PITCH, col = (o - e * PITCH) >> 4;
    const int pg = S0 + e;
    const int pgc = pg < 0 ? 0 : pg;
    const int prow = fdiv(pgc, P.fd_rp), c = pgc - prow * RP;
    const int n_ = fdiv(prow, P.fd_h1), h_ = prow - n_ * (P.H + 1);
    const bool ok = col < 4 && pg >= 0 && c >= 1 && h_ < P.H && n_ < P.nimg;
    va[j] = ok ? (uint32_t)(((n_ * P.H + h_) * P.W + c - 1) * P.Cred + col * 8) * 2u : OOB;
  }
#define C3_ISSUE_B(st_, cb_, tap_)                                                                     \
  {                                                                                                    \
    unsigned char* base_ = smem + OFF_B + (st_) * B_STAGE;                                             \
    const uint32_t add_ = (uint32_t)((tap_) * P.b_tap_stride + (cb_) * 32) * 2u;                       \
    if (!(RIGL_C3_ABLATE & 4))                                                                         \
    _Pragma("unroll") for (int j = 0; j < LB; ++j)                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
          rsrcB, (__attribute__((address_space(3))) void*)(base_ + (j * 4 + wave) * 1024), 16,         \
          (int)(vb[j] + add_), 0, 0, 0);                                                               \
  }
#define C3_ISSUE_A(cb_)                                                                                \
  {                                                                                                    \
    unsigned char* base_ = smem + ((cb_) & 1) * SLAB_B;                                                \
    const uint32_t add_ = (uint32_t)((cb_) * 64);                                                      \
    if (!(RIGL_C3_ABLATE & 4))                                                                         \
    _Pragma("unroll") for (int j = 0; j < LA; ++j)                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
          rsrcA, (__attribute__((address_space(3))) void*)(base_ + (j * 4 + wave) * 1024), 16,         \
          (int)(va[j] + add_), 0, 0, 0);                                                               \
  }

  // ---- per-lane fragment addresses (this wave's K-step folded in) ---------------------------------------------
  const int kx = wk * 32;
  const int a_base = (RP + 1 + wm * RB * 32 + lr0) * PITCH + hi * 16 + kx;
  int b_rd[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int row = wn * 64 + jb * 32 + lr0;
    b_rd[jb] = OFF_B + ((row * 64 + (((hi ^ (row >> 2)) & 3) << 4)) ^ kx);
  }

  f32x16 acc[RB][2];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jb][e] = 0.f;

  struct Frag { bf16x8 b[2]; bf16x8 a[RB]; };
#define C3_READ(F_, st_, cb_, tap_)                                                                    \
  if (!(RIGL_C3_ABLATE & 2)) {                                                                         \
    const int r_ = (tap_) >= 6 ? 2 : ((tap_) >= 3 ? 1 : 0), s_ = (tap_) - 3 * r_;                      \
    const int dh_ = MODE == 0 ? r_ - 1 : 1 - r_, dw_ = MODE == 0 ? s_ - 1 : 1 - s_;                    \
    const unsigned char* ap_ = smem + a_base + ((cb_) & 1) * SLAB_B + (dh_ * RP + dw_) * PITCH;        \
    _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                     \
      F_.a[i] = *reinterpret_cast<const bf16x8*>(ap_ + i * 32 * PITCH);                                \
    _Pragma("unroll") for (int jb = 0; jb < 2; ++jb)                                                   \
      F_.b[jb] = *reinterpret_cast<const bf16x8*>(smem + b_rd[jb] + (st_) * B_STAGE);                  \
  }
#define C3_MFMA(F_)                                                                                    \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                     \
      _Pragma("unroll") for (int jb = 0; jb < 2; ++jb) {                                               \
        if (RIGL_C3_ABLATE & 1) asm volatile("" :: "v"(F_.a[i]), "v"(F_.b[jb]));                       \
        else acc[i][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F_.b[jb], F_.a[i], acc[i][jb], 0, 0, 0); \
      }                                                                                                \
  }
#define C3_NEXT3(x_) { x_ = (x_) == 2 ? 0 : (x_) + 1; }

  // ---- main loop (see conv196.hpp for the discipline; K-tile kt = (channel block, tap), stage kt % 3) ------------
  C3_ISSUE_A(0);
  C3_ISSUE_B(0, 0, 0); C3_ISSUE_B(1, 0, 1); C3_ISSUE_B(2, 0, 2);          // KT >= 9
  wait_vmcnt<2 * LB>();
  __builtin_amdgcn_s_barrier();
  int st = 0;
  int cbc = 0, tapc = 0, cb1 = 0, tap1 = 1, cb3 = 0, tap3 = 3;
  Frag F0 = {}, F1 = {};
  // between two MFMA batches: tile kt's fragments are in registers (kt + 1 < KT)
#define C3_MID()                                                                                       \
  {                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
    const bool slab_pending = (tapc == 1 || tapc == 2) && cbc + 1 < CB;                                \
    if (slab_pending) wait_vmcnt<LB + LA>();                                                           \
    else if (kt + 2 < KT) wait_vmcnt<LB>();                                                            \
    else wait_vmcnt<0>();                                                                              \
    if (!(RIGL_C3_ABLATE & 8)) __builtin_amdgcn_s_barrier();                                           \
    if (kt + 3 < KT) C3_ISSUE_B(st, cb3, tap3);                                                        \
    if (tapc == 0 && cbc + 1 < CB) C3_ISSUE_A(cbc + 1);                                                \
    C3_NEXT3(st);                                                                                      \
    if (++tapc == 9) { tapc = 0; ++cbc; }                                                              \
    if (++tap3 == 9) { tap3 = 0; ++cb3; }                                                              \
  }
#define C3_ADV1() { if (++tap1 == 9) { tap1 = 0; ++cb1; } }
  C3_READ(F0, 0, 0, 0);
  int kt = 0;
  // (scheduling fences: left alone the compiler moves half of the next tile's fragment reads into the middle of the MFMA
  //  batch and then waits for ALL of them before the batch's last MFMAs -- a few hundred cycles per K-tile)
#define C3_FENCE() __builtin_amdgcn_sched_barrier(0)
  for (; kt + 2 < KT; kt += 2) {
    C3_MID(); C3_READ(F1, st, cb1, tap1); C3_ADV1(); C3_FENCE(); C3_MFMA(F0); C3_FENCE();
    ++kt;
    C3_MID(); C3_READ(F0, st, cb1, tap1); C3_ADV1(); C3_FENCE(); C3_MFMA(F1); C3_FENCE();
    --kt;
  }
  // one or two tiles are left, F0 holds the first; both batches always run (on zeros when there is no second tile) so
  // that no accumulator lives on two control-flow paths
  if (kt + 1 < KT) { C3_MID(); C3_READ(F1, st, cb1, tap1); }
  else {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RB; ++i) F1.a[i] = z;
    F1.b[0] = z; F1.b[1] = z;
  }
  C3_MFMA(F0);
  C3_MFMA(F1);
  wait_vmcnt<0>();
  __syncthreads();
#undef C3_ISSUE_A
#undef C3_ISSUE_B
#undef C3_READ
#undef C3_MFMA
#undef C3_NEXT3
#undef C3_MID
#undef C3_ADV1
#undef C3_FENCE

  // ---- K split: the odd wave of each pair hands its partial sums over; even + odd, always in that order --------------
  {
    float* dump = reinterpret_cast<float*>(smem);
    const int pair = wave >> 1;
    if (wk == 1) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 16; ++e) dump[(((pair * RB + i) * 2 + jb) * 16 + e) * 64 + lane] = acc[i][jb][e];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][jb][e] += dump[(((pair * RB + i) * 2 + jb) * 16 + e) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);
  int* rowpix = reinterpret_cast<int*>(smem + EPI);          // output pixel of tile row r, or -1 (pad / gap / beyond the tile)
  if (tid < BMC) {
    const int pr = fdiv(tid, P.fd_rp), c = tid - pr * RP;
    const int prow = prow0 + pr;
    const int n_ = fdiv(prow, P.fd_h1), h_ = prow - n_ * (P.H + 1);
    const bool ok = pr < P.RT && c >= 1 && h_ < P.H && n_ < P.nimg;
    rowpix[tid] = ok ? (n_ * P.H + h_) * P.W + c - 1 : -1;
  }
  if (wk == 0) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (wm * RB + i >= RBT) continue;         // (wave-uniform) the eighth block of a 7-block tile has no staging rows
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (wm * RB + i) * 32 + lr0;
          const int col = wn * 64 + jb * 32 + 8 * q + 4 * hi;
          const f32x2 lo = {acc[i][jb][4 * q], acc[i][jb][4 * q + 1]}, hi2 = {acc[i][jb][4 * q + 2], acc[i][jb][4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          *reinterpret_cast<uint2*>(Cs + row * CS_LD + col) = pk;
        }
    }
  }
  __syncthreads();
  if (MODE == 0 && P.STATS) {
    // one partial row per tile over its valid rows; PARTS row slices per column, combined in ascending order
    constexpr int PARTS = THREADS / BN, RPS = BMC / PARTS;
    const int col = tid % BN, part = tid / BN;
    float sy = 0.f, sq = 0.f;
    for (int r2 = 0; r2 < RPS; ++r2) {
      const int row = part * RPS + r2;
      if (rowpix[row] < 0) continue;
      const float v = __uint_as_float((uint32_t)Cs[row * CS_LD + col] << 16);
      sy += v; sq = fmaf(v, v, sq);
    }
    float2* red = reinterpret_cast<float2*>(smem + EPI + BMC * 4);
    red[tid] = make_float2(sy, sq);
    __syncthreads();
    if (part == 0 && n0 + col < P.N) {
#pragma unroll
      for (int k = 1; k < PARTS; ++k) { sy += red[k * BN + col].x; sq += red[k * BN + col].y; }
      float* stp = P.STATS + (int64_t)tile_m * 2 * P.N + n0 + col;
      stp[0] = sy; stp[P.N] = sq;
    }
  }
  uint16_t* C = static_cast<uint16_t*>(P.C);
  constexpr int CH = BN / 8, ITERS = BMC * CH / THREADS;
  static_assert(BMC * CH % THREADS == 0, "whole output chunks per thread");
  uint4 addv[ITERS];
  if (P.ADD) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
      const int m = rowpix[row], n = n0 + ch * 8;
      addv[it] = (m >= 0 && n < P.N) ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)m * P.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
    const int m = rowpix[row], n = n0 + ch * 8;
    if (m >= 0 && n < P.N) {
      uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_LD + ch * 8);
      if (P.ADD) {
        const uint4 q = addv[it];
        v.x = add_bf16x2(v.x, q.x); v.y = add_bf16x2(v.y, q.y); v.z = add_bf16x2(v.z, q.z); v.w = add_bf16x2(v.w, q.w);
      }
      *reinterpret_cast<uint4*>(C + (int64_t)m * P.ldc + n) = v;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
template <int MODE, int BN, int RBT>
static bool c3_ready_one() {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_c3<MODE, BN, RBT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, c3_smem_bytes<BN, RBT>()) == hipSuccess;
  return ok;
}

// RIGL_C3: 0 = never (default, see profiles/r2/README.md), 1 = the rule below, 2 = wherever the shape is legal, dgrad too.
static int c3_mode() {
  static const int v = [] { const char* e = getenv("RIGL_C3"); return e ? atoi(e) : 0; }();
  return v;
}

struct C3Plan { bool use; int bn, rbt, rp, rt, per, tiles_m; unsigned grid; };

// rows x cols GEMM of a 3x3 stride-1 SAME convolution over nimg H x W maps with `cred` reduction channels per tap
static C3Plan plan_c3(int kh, int kw, int sh, int sw, int pt, int pl, int H, int W, int Ho, int Wo, int nimg, int N, int cred,
                      bool dgrad = false) {
  C3Plan p = {false, 128, 7, 0, 0, 0, 0, 0u};
  const int mode = c3_mode();
  if (mode <= 0) return p;
  if (mode == 1) {
    // Default rule, from the per-layer measurements (profiles/r2/README.md): forward passes of the 7x7 maps only
    // (256-row tiles of four images: -8...-12 % against the igemm kernel; level on 14x14, slower on 28x28 where two
    // igemm workgroups per CU hide each other's latencies and this kernel's single wave per SIMD cannot).  dgrad stays in
    // the launch it shares with the weight gradient (see plan_t196).
    static const int max_hw = [] { const char* e = getenv("RIGL_C3_MAX_HW"); return e ? atoi(e) : 7; }();
    if (dgrad || H > max_hw || W > max_hw) return p;
  }
  if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || pt != 1 || pl != 1 || Ho != H || Wo != W) return p;
  if ((cred & 31) || (N & 7) || N < 64 || nimg <= 0) return p;
  const int RP = W + 1, cap = C3_SLAB_BYTES / C3_PITCH;
  double best = 0.0;
  for (int rbt = 7; rbt <= 8; ++rbt) {
    const int bmc = rbt * 32;
    int rt, per, tiles;
    if ((H + 1) * RP <= bmc) {                 // whole images per tile (their gap rows included)
      int k = bmc / ((H + 1) * RP);
      while (k > 1 && nimg % k) --k;
      rt = k * (H + 1); per = 0; tiles = nimg / k;
    } else {                                   // whole image rows per tile, tiles never straddle images
      rt = 0;
      for (int d = 1; d <= H; ++d) if (H % d == 0 && d * RP <= bmc) rt = d;
      if (!rt) continue;
      per = H / rt; tiles = nimg * per;
    }
    if (rt * RP + 2 * RP + 2 > cap) continue;  // the slab: the tile's entries + one padded row and one entry either side
    const double eff = (double)nimg * H * W / ((double)tiles * bmc);
    if (eff > best + 1e-9) { best = eff; p.rbt = rbt; p.rt = rt; p.per = per; p.tiles_m = tiles; }
  }
  if (best < 0.6) return p;                    // more than 40 % of the MFMA rows would be padding: the igemm kernel is better
  p.rp = RP;
  p.bn = ((int64_t)p.tiles_m * ((N + 127) / 128) >= num_cus()) ? 128 : 64;
  const int64_t tiles = (int64_t)p.tiles_m * ((N + p.bn - 1) / p.bn);
  if (tiles > 0x7fffffff) return p;
  p.grid = (unsigned)tiles;
  p.use = true;
  return p;
}

template <int MODE>
static bool launch_c3(const C3Plan& pl, C3Args& a, hipStream_t st) {
  a.RP = pl.rp; a.RT = pl.rt; a.per = pl.per;
  a.tiles_n = (a.N + pl.bn - 1) / pl.bn;
  a.fd_rp = make_fastdiv(pl.rp); a.fd_h1 = make_fastdiv(a.H + 1); a.fd_per = make_fastdiv(pl.per > 0 ? pl.per : 1);
  const dim3 grid(pl.grid), blk(THREADS);
#define RIGL_C3_GO(BN_, RBT_)                                                                          \
  {                                                                                                    \
    if (!c3_ready_one<MODE, BN_, RBT_>()) return false;                                                \
    RIGL_K_LAUNCH((k_c3<MODE, BN_, RBT_>), grid, blk, (c3_smem_bytes<BN_, RBT_>()), st, a);            \
    return true;                                                                                       \
  }
  if (pl.bn == 128) { if (pl.rbt == 7) RIGL_C3_GO(128, 7) else RIGL_C3_GO(128, 8) }
  else { if (pl.rbt == 7) RIGL_C3_GO(64, 7) else RIGL_C3_GO(64, 8) }
#undef RIGL_C3_GO
  return false;
}
