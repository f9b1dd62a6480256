// K1, "tile196" variant of the forward / dgrad implicit GEMM for stride-1 convolutions (1x1, and 3x3
// with SAME padding).  Included by conv.hip inside namespace rigl::k1 (it uses that file's helpers).
//
// Why another tile shape.  ResNet-50's row counts are M = batch * 49 * 4^k (7x7, 14x14, 28x28, 56x56
// maps), so a tile of 196 rows (four 7x7 images, one 14x14 image, a quarter of a 28x28 one) divides
// every layer exactly, and at the benchmarked per-GPU batch of 128 the tile counts of groups 2-4 are exact
// multiples of the 256 CUs (14x14x256 3x3: 128 row tiles x 2 column tiles = 256; 7x7x512 3x3: 32 x 8 of
// 64 columns = 256; 14x14 256->1024: 1024) where the 128-row grid gives 392 or 1568 tiles -- one and a half
// or two-and-a-bit rounds, a quarter of the chip idle in the last one.  196 rows are computed as 7 blocks
// of 32 (the last 28 rows multiply zeros: 12.5 % of the MFMA issue slots, less than the idle round cost).
//
// Workgroup = 4 waves: all four along N (BN = 128: one 32-column block each, every wave owns all 7 row
// blocks -- balanced, a K-step is 8 fragment reads for 7 MFMAs) or 2 along N x 2 along M (BN = 64: row
// blocks 0-3 | 4-6 plus an all-zero eighth).  Every output element is accumulated over (channel block,
// tap, K-step) in the same order in both variants and whatever tile it falls into, so the result does
// not depend on the plan (batch size, column width) -- only on the operands.
//
// 3x3 mode keeps the INPUT SLAB of the tile resident in LDS: for a block of 32 channels the image rows the tile
// touches (plus one row above and below) are fetched ONCE and the nine taps read them as shifted views -- the
// activation operand crosses L2 -> LDS once per channel block instead of nine times, and only the 8 KB weight
// tiles stream through the 3-deep DMA ring.  The slab is stored ZERO-PADDED: an image row of W pixels takes W + 2
// entries (a zero pixel at either end) and an all-zero row separates consecutive images, so a tap is nothing but a
// constant offset (dh (W + 2) + dw entries) from the lane's centre entry -- no per-tap bounds test, no select.
// Entries are 80 bytes apart (64 of data + 16 unused): consecutive entries then fall into distinct 16-byte bank
// slots (5 r mod 16), i.e. ds_read_b128 is conflict-free without an XOR swizzle, whose recomputation per tap was
// what made the first version of this kernel issue-bound (measured: 58 us with, 29 us without the fragment-read
// path on the 14x14x256 layer; ~14 instructions per MFMA at one wave per SIMD).  `buffer_load ... lds` writes
// lane-linearly, so every fifth lane lands on an unused tail and the pad entries are out-of-range lanes (zeros).
// The slab is double buffered over channel blocks: block cb + 1 is fetched while the nine K-tiles of block cb are
// multiplied.  Rows 196..223 of the tile repeat row 195 (computed, never stored, excluded from the statistics).
//
// Everything else follows conv.hip's igemm: `buffer_load ... lds` with the XOR swizzle on the source
// side, counted vmcnt + one raw barrier per K-tile, transposed accumulators, bf16 output staged through
// LDS, optional batch-norm partial statistics (one partial row per 196-row tile) and optional addend.
// The index arithmetic is restated lane by lane in tools/emu/t196_emu.py and checked there against a
// plain convolution (there is no GPU where this is written).

// Development ablations (never defined in the product build; tools/build_alt.sh + RIGL_HIP_LIB): bit 0 drops the MFMAs
// (fragments stay live), bit 1 the fragment reads, bit 2 the DMA loads, bit 3 the per-tile barrier.
#ifndef RIGL_T196_ABLATE
#define RIGL_T196_ABLATE 0
#endif

struct T196Args {
  const uint16_t* A;   // activations: x (fwd) or dy (dgrad), NHWC, row space == gathered space
  const uint16_t* B;   // packed weights: OHWI shadow (fwd) / HWIO shadow (dgrad)
  void* C;
  const uint16_t* ADD;
  float* STATS;        // [M / 196][2][N] or NULL
  int M, N, Cred;      // rows (pixels), columns, reduction channels per tap (multiple of 32)
  int H, W, nimg;
  int b_row_stride, b_tap_stride, ldc, tiles_n;
  uint32_t a_bytes, b_bytes;
  FastDiv fd_w, fd_h, fd_w2, fd_h1;   // / W, / H, / (W + 2), / (H + 1)
};

constexpr int T196_BMV = 196, T196_BMC = 224, T196_RB = 7;
constexpr int T196_SLAB_BYTES = 24576, T196_PITCH = 80;      // 3x3 slab buffer: up to 307 padded entries of 80 bytes

template <bool K3, int BN>
constexpr int t196_smem_bytes() {
  constexpr int EPI = T196_BMC * (BN + 8) * 2 + THREADS * 8;          // bf16 staging + statistics scratch
  constexpr int RING = K3 ? (2 * T196_SLAB_BYTES + 3 * BN * 64) : 3 * (256 * 64 + BN * 64);
  return EPI > RING ? EPI : RING;
}

template <int MODE /*0 fwd, 1 dgrad*/, bool K3, int BN>
__device__ __forceinline__ void t196_body(const T196Args& P, unsigned char* smem, uint32_t bid, uint32_t nblk) {
  constexpr int BMV = T196_BMV, BMC = T196_BMC;
  constexpr int WN = BN / 32, WM = 4 / WN, LB = BN / 64;
  constexpr int RB = WM == 1 ? T196_RB : 4;   // row blocks per wave (BN = 64: blocks 4 wm .. 4 wm + 3; block 7 is all zeros)
  constexpr int SLAB_B = T196_SLAB_BYTES, PITCH = T196_PITCH, B_STAGE = BN * 64, OFF_B = 2 * SLAB_B;
  constexpr int A_ST = 256 * 64, STAGE = A_ST + BN * 64;
  constexpr int LA = K3 ? SLAB_B / 4096 : 4;   // slab (per channel block) / A-tile (per K-tile) DMA instructions per wave
  constexpr int CS_LD = BN + 8, EPI = BMC * CS_LD * 2;
  static_assert(BN == 64 || BN == 128, "tile196: 64 or 128 columns");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, blk0 = (wave / WN) * 4;
  const uint32_t tile = xcd_remap(bid, nblk);
  const int tile_m = (int)(tile / (uint32_t)P.tiles_n);
  const int n0 = (int)(tile % (uint32_t)P.tiles_n) * BN;
  const int m0 = tile_m * BMV;
  const int l4 = lane >> 2, dchunk = (lane & 3) ^ ((lane >> 4) & 3), hi = lane >> 5;
  const int W2 = P.W + 2;
  // 3x3: padded row (image rows + one gap row per image, counted over the whole batch) stored first in the slab
  const int gr0 = fdiv(m0, P.fd_w), pr_base = gr0 + fdiv(gr0, P.fd_h) - 1;
  const int CB = P.Cred >> 5;
  const int KT = K3 ? CB * 9 : CB;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);

  // ---- per-lane DMA source offsets (bytes; fixed along the K loop up to a wave-uniform addend) ----------
  uint32_t vb[LB], va[LA];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int row = (j * 4 + wave) * 16 + l4, n = n0 + row;
    vb[j] = (row < BN && n < P.N) ? (uint32_t)(n * P.b_row_stride + dchunk * 8) * 2u : OOB;
  }
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int row = (j * 4 + wave) * 16 + l4;
    if (K3) {
      // lane -> (padded entry, 16-byte column) of the 80-byte-pitch slab; pads, gap rows, rows outside the batch and the
      // unused fifth column are out-of-range lanes (the hardware writes zeros)
      const int o = (j * 4 + wave) * 1024 + lane * 16;
      const int e = o / PITCH, col = (o - e * PITCH) >> 4;
      const int prr = fdiv(e, P.fd_w2), w_ = e - prr * W2 - 1;
      const int pr = pr_base + prr;                                      // >= -1
      const int n_ = fdiv(pr + P.H + 1, P.fd_h1) - 1, h_ = pr - n_ * (P.H + 1);
      const bool ok = col < 4 && (unsigned)w_ < (unsigned)P.W && h_ < P.H && pr >= 0 && n_ < P.nimg;
      va[j] = ok ? (uint32_t)(((n_ * P.H + h_) * P.W + w_) * P.Cred + col * 8) * 2u : OOB;
    } else {
      const int m = m0 + row;
      va[j] = (row < BMV && m < P.M) ? (uint32_t)(m * P.Cred + dchunk * 8) * 2u : OOB;
    }
  }
  // (an out-of-range offset stays out of range under the small addends below: the hardware writes zeros)
#define T196_ISSUE_B(st_, cb_, tap_)                                                                   \
  {                                                                                                   \
    unsigned char* base_ = smem + (K3 ? OFF_B + (st_) * B_STAGE : (st_) * STAGE + A_ST);              \
    const uint32_t add_ = (uint32_t)((tap_) * P.b_tap_stride + (cb_) * 32) * 2u;                      \
    if (!(RIGL_T196_ABLATE & 4))                                                                      \
    _Pragma("unroll") for (int j = 0; j < LB; ++j)                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                       \
          rsrcB, (__attribute__((address_space(3))) void*)(base_ + (j * 4 + wave) * 1024), 16,        \
          (int)(vb[j] + add_), 0, 0, 0);                                                              \
  }
#define T196_ISSUE_A(base_off_, cb_)                                                                  \
  {                                                                                                   \
    unsigned char* base_ = smem + (base_off_);                                                        \
    const uint32_t add_ = (uint32_t)((cb_) * 64);                                                     \
    if (!(RIGL_T196_ABLATE & 4))                                                                      \
    _Pragma("unroll") for (int j = 0; j < LA; ++j)                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                       \
          rsrcA, (__attribute__((address_space(3))) void*)(base_ + (j * 4 + wave) * 1024), 16,        \
          (int)(va[j] + add_), 0, 0, 0);                                                              \
  }

  // ---- per-lane fragment state ------------------------------------------------------------------------
  const int lr0 = lane & 31;
  const int row_b = wn * 32 + lr0;
  const int b_rd = row_b * 64 + (((hi ^ (row_b >> 2)) & 3) << 4);
  int a_fix[RB];                            // fragment offset of row block i: inside a stage (1x1) / a slab buffer, centre tap (3x3)
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int lr = (blk0 + i) * 32 + lr0;   // (rows 224..255 of the 1x1 A stage are zero-filled by the DMA: block 7 multiplies zeros)
    if (K3) {
      const int m = m0 + (lr < BMV ? lr : BMV - 1);                      // rows beyond 196 repeat row 195
      const int gr = fdiv(m, P.fd_w), w_ = m - gr * P.W;
      const int pe = (gr + fdiv(gr, P.fd_h) - pr_base) * W2 + w_ + 1;     // padded entry of the output pixel itself
      a_fix[i] = pe * PITCH + hi * 16;
    } else {
      a_fix[i] = lr * 64 + (((hi ^ (lr >> 2)) & 3) << 4);
    }
  }

  f32x16 acc[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // Fragments of one K-step: the wave's 32 weight rows and all 7 row blocks of the activation operand.
  struct Frag { bf16x8 b; bf16x8 a[RB]; };
  // a_off[i]: byte offset of row block i's K-step-0 fragment.  K-step 1: 1x1 (XOR-swizzled stage) = the same offset with
  // bit 5 flipped (chunk (2 + hi) ^ sw == (hi ^ sw) ^ 2); 3x3 (linear 80-byte entries) = 32 bytes further.
#define T196_READ(F_, bbase_, kx_)                                                                    \
  if (!(RIGL_T196_ABLATE & 2)) {                                                                      \
    F_.b = *reinterpret_cast<const bf16x8*>(smem + (bbase_) + (b_rd ^ (kx_)));                        \
    _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                    \
      F_.a[i] = *reinterpret_cast<const bf16x8*>(smem + (K3 ? a_off[i] + (kx_) : (a_off[i] ^ (kx_)))); \
  }
#define T196_MFMA(F_)                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                  \
      if (RIGL_T196_ABLATE & 1) asm volatile("" :: "v"(F_.a[i]), "v"(F_.b));                          \
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F_.b, F_.a[i], acc[i], 0, 0, 0);          \
    }                                                                                                 \
  }
  // fragment offsets of K-tile (stage st_ | slab of block cb_, tap tap_)
#define T196_AOFF(st_, cb_, tap_)                                                                     \
  {                                                                                                   \
    if constexpr (!K3) {                                                                              \
      _Pragma("unroll") for (int i = 0; i < RB; ++i) a_off[i] = (st_) * STAGE + a_fix[i];             \
    } else {                                                                                          \
      const int r_ = (tap_) >= 6 ? 2 : ((tap_) >= 3 ? 1 : 0), s_ = (tap_) - 3 * r_;                   \
      const int dh_ = MODE == 0 ? r_ - 1 : 1 - r_, dw_ = MODE == 0 ? s_ - 1 : 1 - s_;                 \
      const int tapoff_ = ((cb_) & 1) * SLAB_B + (dh_ * W2 + dw_) * PITCH;     /* wave-uniform */      \
      _Pragma("unroll") for (int i = 0; i < RB; ++i) a_off[i] = a_fix[i] + tapoff_;                   \
    }                                                                                                 \
  }
#define T196_BBASE(st_) (K3 ? OFF_B + (st_) * B_STAGE : (st_) * STAGE + A_ST)
#define T196_NEXT3(x_) { x_ = (x_) == 2 ? 0 : (x_) + 1; }

  // ---- main loop: a software pipeline across the barrier ------------------------------------------------
  // Tile kt lives in ring stage kt % 3; tiles kt+1, kt+2 (and, once tile kt is in registers, kt+3) are in flight.
  // Per tile:   request the LAST K-step's fragments of tile kt | MFMAs on the fragments requested a batch ago |
  //             lgkmcnt(0) (tile kt is now entirely in registers: its stage is free) | counted vmcnt: MY loads of
  //             tile kt+1 landed | raw barrier (everyone's landed, everyone is done with stage kt % 3) |
  //             DMA of tile kt+3 into stage kt % 3 | request the first fragments of tile kt+1 | MFMAs of the last
  //             K-step of tile kt.
  // Every MFMA batch therefore has its fragments requested a whole batch (7 MFMAs = 224 cycles) earlier, and the
  // barrier sits between two batches whose operands are already in registers.  Loads return in order, so
  // `vmcnt(N)` with N = the DMA instructions issued after tile kt+1's.
  // 3x3: K-tiles are (channel block cb, tap); the slab of block cb + 1 is requested in the iteration of tap 0 of
  // block cb, AFTER that iteration's weight tile, so it may stay in flight over the waits of taps 1 and 2
  // (N = LB + LA there) and is forced to land by the wait of tap 3 -- five K-tiles before its first use.  Its
  // buffer was last read by block cb - 1, whose reads completed before the barrier preceding the request.
  constexpr int L = K3 ? LB : LA + LB;       // DMA instructions per wave per K-tile (the slab comes on top)
  int a_off[RB];
  if constexpr (K3) T196_ISSUE_A(0, 0);
  if constexpr (!K3) {
    T196_ISSUE_A(0 * STAGE, 0); T196_ISSUE_B(0, 0, 0);
    if (KT > 1) { T196_ISSUE_A(1 * STAGE, 1); T196_ISSUE_B(1, 1, 0); }
    if (KT > 2) { T196_ISSUE_A(2 * STAGE, 2); T196_ISSUE_B(2, 2, 0); }
  } else {
    T196_ISSUE_B(0, 0, 0); T196_ISSUE_B(1, 0, 1); T196_ISSUE_B(2, 0, 2);     // KT >= 9
  }
  if (KT >= 3) wait_vmcnt<2 * L>(); else if (KT == 2) wait_vmcnt<L>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  int st = 0;                                // stage of tile kt (= the stage tile kt + 3 is loaded into)
  int cb1 = 0, tap1 = K3 ? 1 : 0;            // (block, tap) of tile kt + 1 (3x3); 1x1: cb1 = kt + 1 below
  int cb3 = 0, tap3 = K3 ? 3 : 0;            // ... of tile kt + 3
  int cbc = 0, tapc = 0;                     // ... of tile kt
  if constexpr (!K3) { cb1 = 1; cb3 = 3; }
  Frag F0 = {}, F1 = {};
  T196_AOFF(0, 0, 0);
  // The part of an iteration between the two MFMA batches (tile kt is in registers; kt + 1 < KT):
#define T196_MID()                                                                                    \
  {                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    const bool slab_pending = K3 && (tapc == 1 || tapc == 2) && cbc + 1 < CB;                         \
    if (slab_pending) wait_vmcnt<L + LA>();                                                           \
    else if (kt + 2 < KT) wait_vmcnt<L>();                                                            \
    else wait_vmcnt<0>();                                                                             \
    if (!(RIGL_T196_ABLATE & 8)) __builtin_amdgcn_s_barrier();                                        \
    if (kt + 3 < KT) {                                                                                \
      if constexpr (!K3) T196_ISSUE_A(st * STAGE, cb3);                                               \
      T196_ISSUE_B(st, cb3, tap3);                                                                    \
    }                                                                                                 \
    if (K3 && tapc == 0 && cbc + 1 < CB) T196_ISSUE_A(((cbc + 1) & 1) * SLAB_B, cbc + 1);             \
    T196_NEXT3(st);                        /* now the stage of tile kt + 1 */                         \
    T196_AOFF(st, cb1, tap1);                                                                         \
    if constexpr (K3) {                                                                               \
      if (++tapc == 9) { tapc = 0; ++cbc; }                                                           \
      if (++tap1 == 9) { tap1 = 0; ++cb1; }                                                           \
      if (++tap3 == 9) { tap3 = 0; ++cb3; }                                                           \
    } else { ++cb1; ++cb3; }                                                                          \
  }
  // (the last tile is peeled off: with a conditional middle section inside the loop the compiler merges the two
  //  paths in front of the second MFMA batch and makes it wait for the NEXT tile's fragment reads)
  int kt = 0;
  T196_READ(F0, T196_BBASE(0), 0);          // KT >= 1 (the plan requires Cred >= 32)
  for (; kt + 1 < KT; ++kt) {
    T196_READ(F1, T196_BBASE(st), 32);
    T196_MFMA(F0);
    T196_MID();
    T196_READ(F0, T196_BBASE(st), 0);
    T196_MFMA(F1);
  }
  T196_READ(F1, T196_BBASE(st), 32);
  T196_MFMA(F0);
  T196_MFMA(F1);
#undef T196_MID
  wait_vmcnt<0>();
  __syncthreads();   // all tiles consumed before the epilogue reuses the LDS
#undef T196_ISSUE_A
#undef T196_ISSUE_B
#undef T196_READ
#undef T196_MFMA
#undef T196_AOFF
#undef T196_BBASE
#undef T196_NEXT3

  // ---- epilogue (transposed accumulators: lane = output row, register quad = 4 consecutive channels) -----
  uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    if (blk0 + i >= T196_RB) continue;      // (wave-uniform) the all-zero eighth block has no staging rows
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (blk0 + i) * 32 + lr0;
        const int col = wn * 32 + 8 * q + 4 * hi;
        const f32x2 lo = {acc[i][4 * q], acc[i][4 * q + 1]}, hi2 = {acc[i][4 * q + 2], acc[i][4 * q + 3]};
        uint2 pk;
        pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
        pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
        *reinterpret_cast<uint2*>(Cs + row * CS_LD + col) = pk;
      }
  }
  __syncthreads();
  if (MODE == 0 && P.STATS) {
    // one partial row per tile over its 196 valid rows; PARTS row slices per column, combined in ascending order (deterministic)
    constexpr int PARTS = THREADS / BN, RPS = BMC / PARTS;
    const int col = tid % BN, part = tid / BN;
    float sy = 0.f, sq = 0.f;
#pragma unroll 8
    for (int r2 = 0; r2 < RPS; ++r2) {
      if (part * RPS + r2 >= BMV) break;
      const float v = __uint_as_float((uint32_t)Cs[(part * RPS + r2) * CS_LD + col] << 16);
      sy += v; sq = fmaf(v, v, sq);
    }
    float2* red = reinterpret_cast<float2*>(smem + EPI);
    red[tid] = make_float2(sy, sq);
    __syncthreads();
    if (part == 0 && n0 + col < P.N) {
#pragma unroll
      for (int k = 1; k < PARTS; ++k) { sy += red[k * BN + col].x; sq += red[k * BN + col].y; }
      float* st = P.STATS + (int64_t)tile_m * 2 * P.N + n0 + col;
      st[0] = sy; st[P.N] = sq;
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
      const int m = m0 + row, n = n0 + ch * 8;
      addv[it] = (row < BMV && m < P.M && n < P.N) ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)m * P.ldc + n)
                                                   : make_uint4(0u, 0u, 0u, 0u);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
    const int m = m0 + row, n = n0 + ch * 8;
    if (row < BMV && m < P.M && n < P.N) {
      uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_LD + ch * 8);
      if (P.ADD) {
        const uint4 q = addv[it];
        v.x = add_bf16x2(v.x, q.x); v.y = add_bf16x2(v.y, q.y); v.z = add_bf16x2(v.z, q.z); v.w = add_bf16x2(v.w, q.w);
      }
      *reinterpret_cast<uint4*>(C + (int64_t)m * P.ldc + n) = v;
    }
  }
}

// 2 workgroups per CU (<= 256 VGPRs): the LDS footprint (53-72 KB) allows exactly that.
template <int MODE, bool K3, int BN>
__global__ __launch_bounds__(THREADS, 2) void k_t196(T196Args P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t196[];
  t196_body<MODE, K3, BN>(P, smem_t196, blockIdx.x, gridDim.x);
}

// ---- host side ---------------------------------------------------------------------------------------------
template <int MODE, bool K3, int BN>
static bool t196_ready_one() {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t196<MODE, K3, BN>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             t196_smem_bytes<K3, BN>()) == hipSuccess;
  return ok;
}

// RIGL_T196: 0 = never (default: inside the ResNet-50 step the kernel loses 0.2 ms of forward time although it wins 0-20 %
// per layer in isolation, profiles/r2/README.md), 1 = forward passes of the maps <= 28x28 with K >= 128, 2 = wherever legal, dgrad too.
static int t196_mode() {
  static const int v = [] { const char* e = getenv("RIGL_T196"); return e ? atoi(e) : 0; }();
  return v;
}

struct T196Plan { bool use, k3; int bn; unsigned grid; };

// `rows` x `cols` GEMM with `cred` reduction channels per tap of a kxk stride-1 convolution over HxW maps.
static T196Plan plan_t196(int kh, int kw, int sh, int sw, int pt, int pl, int H, int W, int Ho, int Wo,
                          int64_t M, int N, int cred, bool dgrad = false) {
  T196Plan p = {false, false, 128, 0u};
  const int mode = t196_mode();
  if (mode <= 0) return p;
  if (sh != 1 || sw != 1 || Ho != H || Wo != W) return p;
  const bool k1 = kh == 1 && kw == 1 && pt == 0 && pl == 0;
  const bool k3 = false;                  // (3x3 layers are conv3x3.hpp's; this file's 3x3 mode is kept for reference only)
  if (!k1) return p;
  if (M % T196_BMV || (cred & 31) || (N & 7) || N < 64) return p;
  const int64_t tiles_m = M / T196_BMV;
  if (k3) {
    // the padded slab of a tile (image rows it touches + one above and below, W + 2 entries each, a gap row per image
    // boundary inside) must fit its LDS buffer; tile starts repeat with period lcm(196, H W) / 196 <= 16 for the shapes of interest
    const int cap = T196_SLAB_BYTES / T196_PITCH;
    for (int64_t t = 0; t < tiles_m && t < 64; ++t) {
      const int64_t m0 = t * T196_BMV, m1 = m0 + T196_BMV - 1;
      const int64_t g0 = m0 / W, g1 = m1 / W;
      const int64_t rows = (g1 + g1 / H) - (g0 + g0 / H) + 3;
      if (rows * (W + 2) > cap) return p;
    }
  }
  const int cus = num_cus();
  // 128 columns when that still gives every CU a tile, else 64 (the 7x7 maps: 32 row tiles at batch 128)
  p.bn = (tiles_m * ((N + 127) / 128) >= cus) ? 128 : 64;
  p.k3 = k3;
  const int64_t tiles = tiles_m * ((N + p.bn - 1) / p.bn);
  if (tiles > 0x7fffffff) return p;
  p.grid = (unsigned)tiles;
  if (mode >= 2) { p.use = true; return p; }
  // Default: FORWARD only.  A dgrad through this kernel would have to leave the launch it shares with the layer's weight
  // gradient (its accumulation order differs from the igemm body's, and a layer's dX must not depend on the entry point);
  // measured in the ResNet-50 step that costs more than the kernel gains (profiles/r2/README.md).
  if (dgrad) return p;
  // default rule: the maps where the 128-row grid leaves CUs idle (<= 28x28 at batch 128: fewer than ~6 tiles
  // per CU) and the reduction is long enough to matter; the 56x56 layers are HBM-bound and stay on the old tiles.
  static const int max_hw = [] { const char* e = getenv("RIGL_T196_MAX_HW"); return e ? atoi(e) : 28; }();
  static const int min_k = [] { const char* e = getenv("RIGL_T196_MIN_K"); return e ? atoi(e) : 128; }();
  p.use = H <= max_hw && W <= max_hw && (int64_t)kh * kw * cred >= min_k;
  return p;
}

template <int MODE>
static bool launch_t196(const T196Plan& pl, T196Args& a, hipStream_t st) {
  a.tiles_n = (a.N + pl.bn - 1) / pl.bn;
  a.fd_w = make_fastdiv(a.W); a.fd_h = make_fastdiv(a.H);
  a.fd_w2 = make_fastdiv(a.W + 2); a.fd_h1 = make_fastdiv(a.H + 1);
  a.nimg = a.M / (a.H * a.W);
  const dim3 grid(pl.grid), blk(THREADS);
#define RIGL_T196_GO(K3_, BN_)                                                                        \
  {                                                                                                   \
    if (!t196_ready_one<MODE, K3_, BN_>()) return false;                                              \
    RIGL_K_LAUNCH((k_t196<MODE, K3_, BN_>), grid, blk, (t196_smem_bytes<K3_, BN_>()), st, a);         \
    return true;                                                                                      \
  }
  if (pl.bn == 128) RIGL_T196_GO(false, 128) else RIGL_T196_GO(false, 64)
#undef RIGL_T196_GO
  return false;
}
