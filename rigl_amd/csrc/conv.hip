// K1: masked convolution (fwd / dgrad / wgrad) as implicit GEMM on the gfx950
// matrix cores: bf16 operands, fp32 accumulation, v_mfma_f32_32x32x16_bf16.
//
//   fwd    Y [M=N*Ho*Wo][Cout]  = im2col(X)[M][kh*kw*Cin]  x  W_ohwi[Cout][kh*kw*Cin]^T
//   dgrad  dX[M=N*H*W ][Cin ]  = gather(dY)[M][kh*kw*Cout] x  W_hwio[(tap,ci)][Cout]^T
//   wgrad  dW[tap][Cin][Cout]  = sum_m X^T[Cin][m] x dY^T[Cout][m]^T      (dense, fp32)
//
// The im2col matrix is never materialised: each K-tile is one filter tap and
// one block of channels, gathered straight from the NHWC activation (zero
// filled outside the image) into an XOR-swizzled LDS tile.  The weight operand
// is the pre-packed bf16 shadow of mask*W (rigl_pack_weights), so the 1-bit
// mask costs no bandwidth here.  Workgroup = 256 threads = 4 waves (2x2), each
// wave owns TMxTN 32x32 MFMA tiles (block tile 64*TM x 64*TN), K-tile BK.
// Staging: a 3-deep LDS-DMA ring (`buffer_load ... lds`, counted vmcnt, one raw
// barrier per K-tile) for fwd / dgrad / wgrad; wgrad reads its pixel-major tiles
// through ds_read_b64_tr_b16.  A register-staged double buffer (BK 64/32/16)
// remains for reductions narrower than 32 channels.  Epilogues optionally leave
// batch-norm partial statistics (fwd) or add a second gradient (dgrad).
#include <stdlib.h>

#include "common.hpp"

// Development ablations (never defined in the product build): RIGL_ABLATE=1 drops the MFMAs of the
// igemm K loop (fragments stay live through one scalar add), =2 drops its DMA loads, =3 the whole K loop.
#if defined(RIGL_ABLATE) && RIGL_ABLATE == 1
#define RIGL_MFMA_OR_ABLATE(i, j) acc[i][j][0] += (float)af[i][0] + (float)bfr[j][0];
#else
// Operands swapped (weights as the MFMA "A", activations as "B"): the accumulator tile is the
// TRANSPOSE, D[n][m] -- a lane then holds 4 consecutive output CHANNELS of one pixel per register
// quad, i.e. 8 contiguous bytes of the NHWC output row, so the epilogue packs with
// v_cvt_pk_bf16_f32 and stages through LDS with 8-byte stores (16 per thread instead of 64
// 2-byte ones).
#define RIGL_MFMA_OR_ABLATE(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#endif

namespace rigl {
namespace k1 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int THREADS = 256;

__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xFFFF0000u) + __uint_as_float(b & 0xFFFF0000u);
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ void store16(uint16_t* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// Byte offset of 16-byte chunk `chunk` of row `row` in a [rows][BK] bf16 tile.
// The chunk index is XORed with a row-derived value so that the 16-lane groups
// of ds_read_b128 (rows r..r+3, r+12.., r+20..) hit 16 distinct 16-B slots.
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  constexpr int CPR = BK / 8;
  constexpr int SH = (CPR == 8) ? 1 : (CPR == 4 ? 2 : 3);
  return row * (BK * 2) + (((chunk ^ (row >> SH)) & (CPR - 1)) << 4);
}

// XCD-aware, bijective block remap: consecutive logical tiles land on the same
// XCD (hardware dispatches block b to XCD b % 8), so tiles that share operand
// panels share an L2.  Speed only -- results never depend on placement.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Buffer loads (SRD in SGPRs): lanes that must read zero get an offset beyond
// num_records and the hardware returns 0 -- no branch, no select, and the
// compiler keeps all loads of a tile in flight behind one counted vmcnt
// (conditional `if (ok) v = *p` loads were serialised by a vmcnt(0) each).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr uint32_t OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// LDS-DMA issued through inline assembly.  The builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) works, but the compiler's
// wait-count pass then treats every later LDS read of the wave as a possible reader of the DMA's destination and puts an
// s_waitcnt vmcnt in front of it that waits for the wave's YOUNGEST DMA -- a wave that issues tile t + 3 and then reads tile t
// out of LDS stalls for the full memory latency in every iteration, and a prefetch ring is one tile deep whatever its size
// (found in round 6 in the ISA of bwdslice.hpp / bwd1x1.hpp: "DMA x5 ... s_waitcnt vmcnt(0) ... ds_read").  Hidden in an asm
// statement the DMA is an opaque memory operation: the kernels' own counted s_waitcnt vmcnt + barrier (which they had anyway)
// are then the ONLY synchronisation.  The compiler's counts for ordinary loads stay safe: it does not see these operations,
// so a vmcnt(N) it computes allows fewer operations in flight than it thinks, never more (loads return in order).
// rsrc4: the buffer descriptor as four dwords (make_rsrc4: base, base_hi, bytes, 0x00020000), lds: wave-uniform destination.
// M0 is a reserved register that cannot be named as a clobber: a kernel issues ALL its LDS-DMA this way or all through the builtin
// (rowstream.hpp and convpp.hpp keep the builtin: no later LDS read of the issuing wave sits between issue and its wait),
// so the compiler never holds a value of its own in M0 across one of these statements.
__device__ __forceinline__ u32x4 make_rsrc4(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
  return r;
}
__device__ __forceinline__ void lds_dma16(u32x4 rsrc4, const void* lds, int byte_off) {
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(l), "v"(byte_off), "s"(rsrc4) : "memory");
}
// the same with the non-temporal hint (a stream that is read once: keeps it out of the way of the lines other workgroups share)
__device__ __forceinline__ void lds_dma16_nt(u32x4 rsrc4, const void* lds, int byte_off) {
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds"
               :: "s"(l), "v"(byte_off), "s"(rsrc4) : "memory");
}
__device__ __forceinline__ void lds_dma4(u32x4 rsrc4, const void* lds, int byte_off) {
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "s"(l), "v"(byte_off), "s"(rsrc4) : "memory");
}
// RIGL_DMA16(rsrcQ, lds, off): the kernel holds the descriptor twice, rsrcQ (builtin form, -DRIGL_DMA_BUILTIN for A/B runs) and rsrcQ4
#ifdef RIGL_DMA_BUILTIN
#define RIGL_DMA16(r_, lds_, off_) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (__attribute__((address_space(3))) void*)(lds_), 16, off_, 0, 0, 0)
#else
#define RIGL_DMA16(r_, lds_, off_) lds_dma16(r_##4, lds_, off_)
#endif
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// s_waitcnt vmcnt(N) with a literal N (an "n"-constraint operand that depends on a
// template parameter made the host pass drop the kernel's launch stub).
template <int N> __device__ __forceinline__ void wait_vmcnt();
#define RIGL_WAIT_VMCNT(N) template <> __device__ __forceinline__ void wait_vmcnt<N>() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); }
RIGL_WAIT_VMCNT(0) RIGL_WAIT_VMCNT(1) RIGL_WAIT_VMCNT(2) RIGL_WAIT_VMCNT(3) RIGL_WAIT_VMCNT(4) RIGL_WAIT_VMCNT(5)
RIGL_WAIT_VMCNT(6) RIGL_WAIT_VMCNT(7) RIGL_WAIT_VMCNT(8) RIGL_WAIT_VMCNT(9) RIGL_WAIT_VMCNT(10) RIGL_WAIT_VMCNT(11)
RIGL_WAIT_VMCNT(12) RIGL_WAIT_VMCNT(13) RIGL_WAIT_VMCNT(14) RIGL_WAIT_VMCNT(15) RIGL_WAIT_VMCNT(16) RIGL_WAIT_VMCNT(17)
RIGL_WAIT_VMCNT(18) RIGL_WAIT_VMCNT(19) RIGL_WAIT_VMCNT(20) RIGL_WAIT_VMCNT(21) RIGL_WAIT_VMCNT(22) RIGL_WAIT_VMCNT(23)
RIGL_WAIT_VMCNT(24) RIGL_WAIT_VMCNT(25) RIGL_WAIT_VMCNT(26) RIGL_WAIT_VMCNT(27) RIGL_WAIT_VMCNT(28) RIGL_WAIT_VMCNT(29)
RIGL_WAIT_VMCNT(30) RIGL_WAIT_VMCNT(31) RIGL_WAIT_VMCNT(32) RIGL_WAIT_VMCNT(33) RIGL_WAIT_VMCNT(34) RIGL_WAIT_VMCNT(35)
RIGL_WAIT_VMCNT(36) RIGL_WAIT_VMCNT(37) RIGL_WAIT_VMCNT(38) RIGL_WAIT_VMCNT(39) RIGL_WAIT_VMCNT(40) RIGL_WAIT_VMCNT(41)
RIGL_WAIT_VMCNT(42) RIGL_WAIT_VMCNT(43) RIGL_WAIT_VMCNT(44) RIGL_WAIT_VMCNT(45) RIGL_WAIT_VMCNT(46) RIGL_WAIT_VMCNT(47)
RIGL_WAIT_VMCNT(48) RIGL_WAIT_VMCNT(49) RIGL_WAIT_VMCNT(50) RIGL_WAIT_VMCNT(51) RIGL_WAIT_VMCNT(52) RIGL_WAIT_VMCNT(53)
RIGL_WAIT_VMCNT(54) RIGL_WAIT_VMCNT(55) RIGL_WAIT_VMCNT(56) RIGL_WAIT_VMCNT(57) RIGL_WAIT_VMCNT(58) RIGL_WAIT_VMCNT(59)
RIGL_WAIT_VMCNT(60) RIGL_WAIT_VMCNT(61) RIGL_WAIT_VMCNT(62) RIGL_WAIT_VMCNT(63)
#undef RIGL_WAIT_VMCNT

__device__ __forceinline__ uint32_t dword_of(const uint4& v, int d) {
  return d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));
}

// Division by a launch-time constant: q = umulhi(n, magic) >> shift, exact for 0 <= n < 2^31
// (magic = ceil(2^(31+l) / d), l = ceil(log2 d), shift = l - 1; d == 1 is flagged by magic == 0).
// The per-thread pixel decompositions in the kernel prologues cost ~40 VALU per runtime division.
struct FastDiv { uint32_t magic, shift; };
static inline FastDiv make_fastdiv(int d) {
  FastDiv f = {0u, 0u};
  if (d <= 1) return f;
  int l = 0;
  while ((1ll << l) < (long long)d) ++l;
  const unsigned long long p = 1ull << (31 + l);
  f.magic = (uint32_t)((p + (unsigned long long)d - 1) / (unsigned long long)d);
  f.shift = (uint32_t)(l - 1);
  return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
  return f.magic ? (int)(__umulhi((uint32_t)n, f.magic) >> f.shift) : n;
}

struct IgemmArgs {
  const uint16_t* A;   // gathered activation tensor (x for fwd, dy for dgrad), NHWC
  const uint16_t* B;   // packed weights
  void* C;             // output rows
  const uint16_t* ADD; // optional bf16 tensor added to the bf16 output rows (same layout as C), or NULL
  // ADD of a SUBSAMPLED view (add_sh > 0): the addend has one row per output pixel (hi, wi) with hi % add_sh == 0 and
  // wi % add_sw == 0, laid out [image][add_ho][add_wo][N]; every other pixel adds nothing.  (The gradient of a strided
  // 1x1 conv that read the same tensor: rigl_masked_conv2d_bwd_sub.)
  int add_sh, add_sw, add_ho, add_wo;
  float* STATS;        // optional per-row-tile column statistics [tiles_m][2][N]: sum y, sum y^2 of the bf16 outputs (fwd);
                       // with BNX (dgrad): sum dz, sum dz * xhat -- the batch-norm backward reductions of the produced tensor
  const uint16_t* BNX; // dgrad only, optional: input x of the batch norm whose OUTPUT gradient this kernel produces ([M][N] like C)
  const uint8_t* BNBITS;  // its 1-bit-per-element ReLU mask (relu(bn + residual)), or NULL: mask recomputed from BNX
  const float* BNP;    // its saved statistics [4][N]: mean, invstd, scale, shift
  int bn_relu;
  int M, N, Cred;      // GEMM rows, columns, reduction channels per tap
  int KH, KW;
  int RH, RW;          // spatial size of the row space (ho,wo | h,w)
  int GH, GW;          // spatial size of the gathered tensor
  int sh, sw, ph, pw;
  int b_row_stride, b_tap_stride;
  int a_pix_stride;    // elements between consecutive gathered pixels (== Cred except on the tiny-Cin path)
  int ldc;
  int tiles_n;
  uint32_t a_bytes, b_bytes;   // sizes of A / B in bytes (< 2^31) for the buffer descriptors
  // strided dgrad, class-major tiling: rows are enumerated per stride-parity
  // class (h % sh, w % sw); a tile never mixes classes, so it only visits the
  // filter taps that can reach its pixels (3x3/2: 1, 2, 2 or 4 taps, not 9).
  int cls_tile_begin[5];       // tile_m prefix per class (sh*sw <= 4 classes)
  int cls_cnt[4], cls_hc[4], cls_wc[4];
  int cls_n, cls_ids[4], cls_interleave;   // non-empty classes and the common tile count they interleave over
  // two-way K split of the ping-pong forward (layers whose tiles would leave most CUs idle): each half of the reduction
  // is a workgroup; the first to finish leaves its fp32 partial tile in KS_SLAB[tile][half], the second adds it and
  // runs the epilogue (a + b = b + a: the same bits whichever arrives last).  KS_CNT[tile] = arrivals, zeroed per call.
  float* KS_SLAB; uint32_t* KS_CNT; int ksplit;
  FastDiv fd_rw, fd_rh, fd_cwc[4], fd_chc[4];
};

// Row of the subsampled addend that output row m (a pixel of the RH x RW row space) takes, or false: none.
__device__ __forceinline__ bool addend_sub_row(const IgemmArgs& P, int m, int64_t& am) {
  const int t = fdiv(m, P.fd_rw);
  const int wi = m - t * P.RW, img = fdiv(t, P.fd_rh), hi = t - img * P.RH;
  const int qh = hi / P.add_sh, qw = wi / P.add_sw;
  if (qh * P.add_sh != hi || qw * P.add_sw != wi) return false;
  am = ((int64_t)img * P.add_ho + qh) * P.add_wo + qw;
  return true;
}

// STAGES: 2 = register-staged double buffer; 3 = LDS-DMA ring of that depth (a 4-deep ring and a 2-deep one under a
// 128-VGPR cap were measured in rounds 1-2: 4-7 % slower over the layer set / neutral, and are gone).
// LDS bytes of one igemm workgroup (the kernels own the array; igemm_body gets a pointer so that
// a fused launch can run it next to another body in the same allocation).
// WM = wave rows of the workgroup (2 x WM waves of TM x TN 32x32 tiles each): 2 -> 256 threads, BM = 64*TM;
// 4 -> 512 threads, BM = 128*TM (the 256x128 tile: 24 KB of operands per K-tile for twice the MFMA work).
template <int TM, int TN, int BK, int MODE, bool OUT_F32, bool CLS, int STAGES, int WM = 2>
constexpr int igemm_smem_bytes() {
  constexpr int NST = STAGES;
  constexpr int NT = 128 * WM;
  constexpr int BM = 32 * WM * TM, BN = 64 * TN;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPI = OUT_F32 ? 0 : BM * (BN + 8) * 2;
  constexpr int EPI_TAB = EPI + (CLS ? BM * 4 : 0);
  constexpr int EPI_ALL = EPI_TAB + ((MODE == 0 && !OUT_F32) ? NT * 8 : 0);
  constexpr int BASE = (NST * STAGE > EPI_ALL) ? NST * STAGE : EPI_ALL;
  // dgrad: + the batch-norm statistics of the tile's columns, outside the ring where the 64 KB of static LDS allow it
  constexpr int EPI_PRM = EPI_TAB + 4 * BN * 4;
  if (MODE == 1 && !OUT_F32) return (BASE + 4 * BN * 4 <= 65536 && WM == 2) ? BASE + 4 * BN * 4 : (BASE > EPI_PRM ? BASE : EPI_PRM);
  return BASE;
}

template <int TM, int TN, int BK, int MODE /*0 fwd, 1 dgrad*/, bool OUT_F32, bool CLS, int STAGES, int WM = 2>
__device__ __forceinline__ void igemm_body(const IgemmArgs& P, unsigned char* smem, uint32_t bid, uint32_t nblk) {
  constexpr int NST = STAGES;
  constexpr int THREADS = 128 * WM;        // (shadows the file-wide 256 inside this body)
  constexpr int BM = 32 * WM * TM, BN = 64 * TN, CPR = BK / 8, RPP = THREADS / CPR;
  constexpr int APASS = (BM + RPP - 1) / RPP, BPASS = (BN + RPP - 1) / RPP;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int CS_LD = BN + 8;
  constexpr int EPI = OUT_F32 ? 0 : BM * CS_LD * 2;
  constexpr int EPI_TAB = EPI + (CLS ? BM * 4 : 0);       // + per-row output pixel table
  constexpr int EPI_ALL = EPI_TAB + ((MODE == 0 && !OUT_F32) ? THREADS * 8 : 0);   // + column-statistics scratch
  constexpr int BASE = (NST * STAGE > EPI_ALL) ? NST * STAGE : EPI_ALL;
  // dgrad: [4][BN] floats of batch-norm statistics -- behind ring and epilogue (filled at the top of the kernel) where
  // the 64 KB static limit allows, else inside the epilogue area (filled in the epilogue)
  constexpr bool PRM_OUT = MODE == 1 && !OUT_F32 && WM == 2 && BASE + 4 * BN * 4 <= 65536;
  constexpr int PRM_OFF = PRM_OUT ? BASE : EPI_TAB;
  constexpr int SMEM = (MODE == 1 && !OUT_F32) ? (PRM_OUT ? BASE + 4 * BN * 4 : (BASE > EPI_TAB + 4 * BN * 4 ? BASE : EPI_TAB + 4 * BN * 4)) : BASE;
  static_assert(SMEM <= 65536 || WM == 4, "static LDS limit (the 512-thread variant uses dynamic LDS)");
  static_assert(STAGES == 2 || (BM % RPP == 0 && BN % RPP == 0), "LDS-DMA needs whole 1-KB wave rows");
  static_assert(SMEM == igemm_smem_bytes<TM, TN, BK, MODE, OUT_F32, CLS, STAGES, WM>(), "LDS size formula out of sync");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t tile = xcd_remap(bid, nblk);
  const int tile_m = (int)(tile / (uint32_t)P.tiles_n);
  const int n0 = (int)(tile % (uint32_t)P.tiles_n) * BN;
  int m0 = tile_m * BM;
  int c_ph = 0, c_pw = 0, c_cnt = P.M, c_hc = 1, c_wc = 1;
  FastDiv c_fwc = {0u, 0u}, c_fhc = {0u, 0u};
  if (CLS) {
    // Classes differ in work (3x3/2: 4, 2, 2, 1 taps; 1x1/2: 1, 0, 0, 0), and the
    // XCD remap hands each XCD a contiguous range of tile_m -- so interleave the
    // classes over tile_m (round-robin while every class still has tiles, the
    // few left-over tiles class by class) to keep the 8 XCDs evenly loaded.
    int c, local;
    const int il = P.cls_interleave * P.cls_n;           // tiles covered by the round-robin part
    if (tile_m < il) {
      c = P.cls_ids[tile_m % P.cls_n]; local = tile_m / P.cls_n;
    } else {
      int rest = tile_m - il;
      c = 0; local = 0;
      for (int k = 0; k < 4; ++k) {
        const int tc = P.cls_tile_begin[k + 1] - P.cls_tile_begin[k];
        const int extra = tc > P.cls_interleave ? tc - P.cls_interleave : (P.cls_cnt[k] > 0 ? 0 : 0);
        const int have = P.cls_cnt[k] > 0 ? extra : tc;  // empty classes own no tiles
        if (rest < have) { c = k; local = (P.cls_cnt[k] > 0 ? P.cls_interleave : 0) + rest; break; }
        rest -= have;
      }
    }
    m0 = local * BM;                                     // index inside the class
    c_ph = c / P.sw; c_pw = c % P.sw;
    c_cnt = P.cls_cnt[c]; c_hc = P.cls_hc[c]; c_wc = P.cls_wc[c];
    c_fwc = P.fd_cwc[c]; c_fhc = P.fd_chc[c];
  }
  const int lrow = tid / CPR, lchunk = tid % CPR;

  // ---- per-thread gather rows (fixed for the whole K loop) ------------------
  int a_pix[APASS], a_c0[APASS], a_c1[APASS], a_out[APASS];
  bool a_ok[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int row = p * RPP + lrow, m = m0 + row;
    a_ok[p] = row < BM && m < (CLS ? c_cnt : P.M);
    const int mm = a_ok[p] ? m : 0;
    int rw, rh, n;
    if (CLS) {
      const int t = fdiv(mm, c_fwc), w2 = mm - t * c_wc;
      n = fdiv(t, c_fhc);
      rw = w2 * P.sw + c_pw; rh = (t - n * c_hc) * P.sh + c_ph;
    } else {
      const int t = fdiv(mm, P.fd_rw);
      rw = mm - t * P.RW; n = fdiv(t, P.fd_rh); rh = t - n * P.RH;
    }
    a_pix[p] = n * P.GH * P.GW;
    a_out[p] = a_ok[p] ? (n * P.RH + rh) * P.RW + rw : -1;
    if (MODE == 0) { a_c0[p] = rh * P.sh - P.ph; a_c1[p] = rw * P.sw - P.pw; }
    else { a_c0[p] = rh + P.ph; a_c1[p] = rw + P.pw; }
  }
  int b_off[BPASS];
  bool b_ok[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) {
    const int row = p * RPP + lrow, nn = n0 + row;
    b_ok[p] = row < BN && nn < P.N;
    b_off[p] = (b_ok[p] ? nn : 0) * P.b_row_stride;
  }

  const int kc_tiles = (P.Cred + BK - 1) / BK;
  // taps visited: all of them, or (class mode) only r = r0, r0+sh, ... and s = s0, s0+sw, ...
  const int r0 = CLS ? (c_ph + P.ph) % P.sh : 0, s0 = CLS ? (c_pw + P.pw) % P.sw : 0;
  const int r_step = CLS ? P.sh : 1, s_step = CLS ? P.sw : 1;
  const int n_r = r0 < P.KH ? (P.KH - r0 + r_step - 1) / r_step : 0;
  const int n_s = s0 < P.KW ? (P.KW - s0 + s_step - 1) / s_step : 0;
#if defined(RIGL_ABLATE) && RIGL_ABLATE == 3
  const int KT = 0;                       // development ablation: prologue + epilogue only
#else
  const int KT = n_r * n_s * kc_tiles;
#endif
  uint4 ra[APASS], rb[BPASS];
  __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);
  const u32x4 rsrcA4 = make_rsrc4(P.A, P.a_bytes), rsrcB4 = make_rsrc4(P.B, P.b_bytes);
  (void)rsrcA4; (void)rsrcB4;
  // dgrad with batch-norm reductions in the epilogue: request the x tile (and its ReLU bits) NOW -- nothing depends on
  // it until the tile is stored, so it travels under the whole K loop instead of stalling the epilogue.  (Class-major
  // rows only know their pixels in the epilogue; those few strided layers load there.)
  constexpr int E_CH = BN / 8, E_ITERS = OUT_F32 ? 1 : BM * E_CH / THREADS;
  uint4 bnx[E_ITERS];
  uint32_t bnb[E_ITERS];
  const bool bnst = MODE == 1 && !OUT_F32 && P.BNX != nullptr;
  if (PRM_OUT && bnst) {
    // ... and the per-channel statistics of the tile's columns, into LDS of their own (not part of the ring)
    float* bprm_w = reinterpret_cast<float*>(smem + PRM_OFF);
    for (int i = tid; i < 4 * BN; i += THREADS) {
      const int k = i / BN, c = i % BN;
      bprm_w[i] = (n0 + c < P.N) ? P.BNP[k * P.N + n0 + c] : 0.f;
    }
  }
  if (MODE == 1 && !OUT_F32 && !CLS && bnst) {
#pragma unroll
    for (int it = 0; it < E_ITERS; ++it) {
      const int idx = it * THREADS + tid, row = idx / E_CH, ch = idx % E_CH;
      const int m = m0 + row, n = n0 + ch * 8;
      const bool ok = m < P.M && n < P.N;
      bnx[it] = ok ? *reinterpret_cast<const uint4*>(P.BNX + (int64_t)m * P.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
      bnb[it] = (ok && P.BNBITS) ? (uint32_t)P.BNBITS[((int64_t)m * P.ldc + n) >> 3] : 0u;
    }
  }

  // (macros, not lambdas: by-reference lambda captures of the staging arrays
  //  kept them in scratch memory instead of registers)
#define RIGL_LOAD_TILE(r_, s_, cb_)                                                                   \
  {                                                                                                   \
    const int cofs = (cb_) * BK + lchunk * 8;                                                         \
    const bool c_ok = cofs < P.Cred;                                                                  \
    _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                               \
      bool ok = a_ok[p] && c_ok;                                                                      \
      int gh, gw;                                                                                     \
      if (MODE == 0) {                                                                                \
        gh = a_c0[p] + (r_); gw = a_c1[p] + (s_);                                                     \
      } else {                                                                                        \
        int th = a_c0[p] - (r_), tw = a_c1[p] - (s_);                                                 \
        ok = ok && th >= 0 && tw >= 0;                                                                \
        if (P.sh == 1) gh = th; else { gh = th / P.sh; ok = ok && (th - gh * P.sh) == 0; }            \
        if (P.sw == 1) gw = tw; else { gw = tw / P.sw; ok = ok && (tw - gw * P.sw) == 0; }            \
      }                                                                                               \
      ok = ok && (unsigned)gh < (unsigned)P.GH && (unsigned)gw < (unsigned)P.GW;                      \
      const int off = (a_pix[p] + gh * P.GW + gw) * P.a_pix_stride + cofs;                            \
      ra[p] = buf_load16(rsrcA, ok ? (uint32_t)off * 2u : OOB);                                       \
    }                                                                                                 \
    const int tap = (r_) * P.KW + (s_);                                                               \
    _Pragma("unroll") for (int p = 0; p < BPASS; ++p) {                                               \
      const int off = b_off[p] + tap * P.b_tap_stride + cofs;                                         \
      rb[p] = buf_load16(rsrcB, (b_ok[p] && c_ok) ? (uint32_t)off * 2u : OOB);                        \
    }                                                                                                 \
  }
#define RIGL_STORE_TILE(buf_)                                                                         \
  {                                                                                                   \
    unsigned char* As_ = smem + (buf_) * STAGE;                                                       \
    unsigned char* Bs_ = As_ + A_BYTES;                                                               \
    _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                               \
      const int row = p * RPP + lrow;                                                                 \
      if (row < BM) *reinterpret_cast<uint4*>(As_ + lds_off<BK>(row, lchunk)) = ra[p];                \
    }                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < BPASS; ++p) {                                               \
      const int row = p * RPP + lrow;                                                                 \
      if (row < BN) *reinterpret_cast<uint4*>(Bs_ + lds_off<BK>(row, lchunk)) = rb[p];                \
    }                                                                                                 \
  }

  // LDS-DMA staging (STAGES >= 3): `buffer_load_dwordx4 ... lds` writes
  // M0-base + lane*16, i.e. one wave instruction fills 1 KB = 64/CPR whole tile
  // rows in LINEAR order, so the XOR swizzle goes on the SOURCE side: the lane
  // that fills slot `lchunk` of its row fetches channel-chunk lchunk ^ f(row).
  // Out-of-image / out-of-range lanes use an offset beyond num_records and the
  // hardware writes zeros (measured: tools/probes/lds_dma_probe.hip).  No
  // registers are held, so several K-tiles stay in flight behind a counted vmcnt.
  constexpr int SWZ_SH = (CPR == 8) ? 1 : (CPR == 4 ? 2 : 3);
  const int dchunk = lchunk ^ ((lrow >> SWZ_SH) & (CPR - 1));
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define RIGL_DMA_ISSUE(r_, s_, cb_, stage_)                                                           \
  {                                                                                                   \
    unsigned char* As_ = smem + (stage_) * STAGE;                                                     \
    unsigned char* Bs_ = As_ + A_BYTES;                                                               \
    const int cofs = (cb_) * BK + dchunk * 8;                                                         \
    const bool c_ok = cofs < P.Cred;                                                                  \
    _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                               \
      bool ok = a_ok[p] && c_ok;                                                                      \
      int gh, gw;                                                                                     \
      if (MODE == 0) {                                                                                \
        gh = a_c0[p] + (r_); gw = a_c1[p] + (s_);                                                     \
      } else {                                                                                        \
        int th = a_c0[p] - (r_), tw = a_c1[p] - (s_);                                                 \
        ok = ok && th >= 0 && tw >= 0;                                                                \
        if (P.sh == 1) gh = th; else { gh = th / P.sh; ok = ok && (th - gh * P.sh) == 0; }            \
        if (P.sw == 1) gw = tw; else { gw = tw / P.sw; ok = ok && (tw - gw * P.sw) == 0; }            \
      }                                                                                               \
      ok = ok && (unsigned)gh < (unsigned)P.GH && (unsigned)gw < (unsigned)P.GW;                      \
      const int off = (a_pix[p] + gh * P.GW + gw) * P.a_pix_stride + cofs;                            \
      RIGL_DMA16(rsrcA, As_ + (p * RPP + wave_u * (64 / CPR)) * (BK * 2), ok ? (int)((uint32_t)off * 2u) : (int)OOB);                                        \
    }                                                                                                 \
    const int tap = (r_) * P.KW + (s_);                                                               \
    _Pragma("unroll") for (int p = 0; p < BPASS; ++p) {                                               \
      const int off = b_off[p] + tap * P.b_tap_stride + cofs;                                         \
      const bool okb = b_ok[p] && c_ok;   /* (a parenthesised condition inline here made the host pass drop the stub) */ \
      RIGL_DMA16(rsrcB, Bs_ + (p * RPP + wave_u * (64 / CPR)) * (BK * 2), okb ? (int)((uint32_t)off * 2u) : (int)OOB);                                       \
    }                                                                                                 \
  }
  // Fast DMA issue: everything about a gathered row that does not change along the K loop is
  // hoisted -- the element offset of the row's first visited tap and a bitmask "tap t lies inside
  // the image" -- so a K-tile costs one add, one bit test and one select per load instead of the
  // bounds / divisibility / multiply chain above (the igemm kernels were issue-bound: 3 waves per
  // SIMD each ~38 % of their time issuing, profiles/r1/pmc_sq_k1.txt).  Visited taps are
  // (r0 + i*r_step, s0 + j*s_step); in all supported cases the gathered pixel moves linearly with
  // (i, j): fwd +(i*r_step, j*s_step); dgrad stride 1 and parity-class dgrad -(i, j).
  // Used for dgrad only: measured on the ResNet-50 set it takes 9 % off dgrad (strided layers 10-30 %) and
  // nothing off fwd, whose bounds math had no divisions (and whose short-K stem pays for the longer prologue).
  constexpr bool FAST = STAGES != 2 && MODE == 1;
  const bool fast_ok = FAST && n_r * n_s <= 64 && (CLS || (P.sh == 1 && P.sw == 1));
  uint32_t f_lo[APASS], f_hi[APASS];
  int f_base[APASS], fb_base[BPASS];
  if (FAST) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      f_lo[p] = f_hi[p] = 0u;
      int gh0, gw0;
      if (MODE == 0) { gh0 = a_c0[p] + r0; gw0 = a_c1[p] + s0; }
      else if (CLS) { gh0 = (a_c0[p] - r0) / P.sh; gw0 = (a_c1[p] - s0) / P.sw; }   // exact inside a parity class
      else { gh0 = a_c0[p] - r0; gw0 = a_c1[p] - s0; }
      f_base[p] = (a_pix[p] + gh0 * P.GW + gw0) * P.a_pix_stride + dchunk * 8;
      if (fast_ok && a_ok[p]) {
        int t = 0;
        for (int i = 0; i < n_r; ++i)
          for (int j = 0; j < n_s; ++j, ++t) {
            const int gh = MODE == 0 ? gh0 + i * r_step : gh0 - i, gw = MODE == 0 ? gw0 + j * s_step : gw0 - j;
            const bool in = (unsigned)gh < (unsigned)P.GH && (unsigned)gw < (unsigned)P.GW;
            if (in) { if (t < 32) f_lo[p] |= 1u << t; else f_hi[p] |= 1u << (t - 32); }
          }
      }
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) fb_base[p] = b_off[p] + dchunk * 8;
  }
  const int f_dr = (MODE == 0 ? r_step * P.GW : -P.GW) * P.a_pix_stride;   // offset step per visited filter row
  const int f_ds = (MODE == 0 ? s_step : -1) * P.a_pix_stride;            // ... per visited filter column
  // (ti_, ri_, si_): index of the visited tap and its (row, column) position among the visited taps
#define RIGL_DMA_ISSUE_FAST(ti_, ri_, si_, r_, s_, cb_, stage_)                                       \
  {                                                                                                   \
    unsigned char* As_ = smem + (stage_) * STAGE;                                                     \
    unsigned char* Bs_ = As_ + A_BYTES;                                                               \
    const int sdelta = (ri_) * f_dr + (si_) * f_ds + (cb_) * BK;                                      \
    const bool c_ok = (cb_) * BK + dchunk * 8 < P.Cred;                                               \
    _Pragma("unroll") for (int p = 0; p < APASS; ++p) {                                               \
      const uint32_t bits = (ti_) < 32 ? f_lo[p] >> (ti_) : f_hi[p] >> ((ti_) - 32);                  \
      const bool ok = (bits & 1u) != 0u && c_ok;                                                      \
      const int boff = ok ? (int)((uint32_t)(f_base[p] + sdelta) * 2u) : (int)OOB;                    \
      RIGL_DMA16(rsrcA, As_ + (p * RPP + wave_u * (64 / CPR)) * (BK * 2), boff);                                                                             \
    }                                                                                                 \
    const int bdelta = ((r_) * P.KW + (s_)) * P.b_tap_stride + (cb_) * BK;                            \
    _Pragma("unroll") for (int p = 0; p < BPASS; ++p) {                                               \
      const bool okb = b_ok[p] && c_ok;                                                               \
      const int boff = okb ? (int)((uint32_t)(fb_base[p] + bdelta) * 2u) : (int)OOB;                  \
      RIGL_DMA16(rsrcB, Bs_ + (p * RPP + wave_u * (64 / CPR)) * (BK * 2), boff);                                                                             \
    }                                                                                                 \
  }
#define RIGL_COMPUTE_TILE(stage_)                                                                     \
  {                                                                                                   \
    const unsigned char* As = smem + (stage_) * STAGE;                                                \
    const unsigned char* Bs = As + A_BYTES;                                                           \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                          \
      const int chunk = ks * 2 + (lane >> 5);                                                         \
      bf16x8 af[TM], bfr[TN];                                                                         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        af[i] = *reinterpret_cast<const bf16x8*>(As + lds_off<BK>(wm * 32 * TM + i * 32 + (lane & 31), chunk)); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + lds_off<BK>(wn * 32 * TN + j * 32 + (lane & 31), chunk)); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
          RIGL_MFMA_OR_ABLATE(i, j)                                                                   \
    }                                                                                                 \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- main loop ------------------------------------------------------------
  int r = r0, s = s0, cb = 0, ti = 0, ri = 0, si = 0;
#define RIGL_ADVANCE() { if (++cb == kc_tiles) { cb = 0; ++ti; ++si; s += s_step; if (s >= P.KW) { s = s0; si = 0; r += r_step; ++ri; } } }
#if defined(RIGL_ABLATE) && RIGL_ABLATE == 2
#define RIGL_DMA_ANY(stage_) { }
#else
#define RIGL_DMA_ANY(stage_) { if (fast_ok) RIGL_DMA_ISSUE_FAST(ti, ri, si, r, s, cb, stage_) else RIGL_DMA_ISSUE(r, s, cb, stage_) }
#endif
  if constexpr (STAGES == 2) {
    // register-staged double buffer: tile kt+1's global loads are in flight during tile kt's MFMAs
    if (KT > 0) {
      RIGL_LOAD_TILE(r, s, cb);
      RIGL_STORE_TILE(0);
      RIGL_ADVANCE();
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < KT;
      if (more) { RIGL_LOAD_TILE(r, s, cb); RIGL_ADVANCE(); }
      RIGL_COMPUTE_TILE(buf);
      if (more) RIGL_STORE_TILE(buf ^ 1);
      __syncthreads();
    }
  } else {
    // NST-deep LDS-DMA ring: tiles kt+1 .. kt+NST-2 stay in flight while tile kt is
    // multiplied.  Per iteration: counted wait for MY loads of tile kt -> barrier (everyone's
    // landed, and everyone is done reading the stage about to be refilled) -> issue tile
    // kt+NST-1 into the stage tile kt-1 used -> MFMAs on tile kt.  One barrier per K-tile,
    // never vmcnt(0) in steady state.
    constexpr int L = APASS + BPASS;   // DMA instructions per thread per K-tile
    for (int t = 0; t < NST - 1; ++t)
      if (t < KT) { RIGL_DMA_ANY(t); RIGL_ADVANCE(); }
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + NST - 1 <= KT) wait_vmcnt<L * (NST - 2)>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + NST - 1 < KT) { RIGL_DMA_ANY((kt + NST - 1) % NST); RIGL_ADVANCE(); }
      RIGL_COMPUTE_TILE(kt % NST);
    }
    wait_vmcnt<0>();
    __syncthreads();   // all tiles consumed before the epilogue reuses the LDS
  }
#undef RIGL_LOAD_TILE
#undef RIGL_STORE_TILE
#undef RIGL_DMA_ISSUE
#undef RIGL_DMA_ISSUE_FAST
#undef RIGL_DMA_ANY
#undef RIGL_COMPUTE_TILE
#undef RIGL_ADVANCE

  // ---- epilogue -------------------------------------------------------------
  // The accumulators hold the transposed tile (operands swapped): D layout of the 32x32 MFMA is
  // col = lane & 31 -> output row m, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) -> output column n.
  if (OUT_F32) {
    float* C = static_cast<float*>(P.C);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 32 * TM + i * 32 + (lane & 31);
          const int n = n0 + wn * 32 * TN + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (m < P.M && n < P.N) C[(int64_t)m * P.ldc + n] = acc[i][j][e];
        }
  } else {
    uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);
    int* rowpix = reinterpret_cast<int*>(smem + EPI);
    if (CLS && lchunk == 0) {
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int row = p * RPP + lrow;
        if (row < BM) rowpix[row] = a_out[p];
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wm * 32 * TM + i * 32 + (lane & 31);
          const int col = wn * 32 * TN + j * 32 + 8 * q + 4 * (lane >> 5);
          const f32x2 lo = {acc[i][j][4 * q], acc[i][j][4 * q + 1]}, hi = {acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
          *reinterpret_cast<uint2*>(Cs + row * CS_LD + col) = pk;
        }
    __syncthreads();
    if (MODE == 0 && P.STATS) {
      // Batch-norm statistics of this row tile, from the bf16-rounded outputs the BN will
      // read: PARTS row-slices per column, combined in a fixed order (deterministic).
      // Rows beyond M hold exact zeros (their gathers were zero-filled).
      // One partial row per 128 output rows whatever the tile height (the batch-norm side counts
      // ceil(M / 128) of them), each the same slice-by-slice sum: a 256-row tile writes two.
      constexpr int PARTS = THREADS / BN, RPS = BM / PARTS, UNITS = BM / 128, SPU = PARTS / UNITS;
      const int col = tid % BN, part = tid / BN;
      float sy = 0.f, sq = 0.f;
#pragma unroll 8
      for (int r2 = 0; r2 < RPS; ++r2) {
        const float v = __uint_as_float((uint32_t)Cs[(part * RPS + r2) * CS_LD + col] << 16);
        sy += v; sq = fmaf(v, v, sq);
      }
      float2* red = reinterpret_cast<float2*>(smem + EPI_TAB);
      red[tid] = make_float2(sy, sq);
      __syncthreads();
      const int unit = tile_m * UNITS + part / SPU;
      if (part % SPU == 0 && n0 + col < P.N && (int64_t)unit * 128 < P.M) {
#pragma unroll
        for (int k = 1; k < SPU; ++k) { sy += red[(part + k) * BN + col].x; sq += red[(part + k) * BN + col].y; }
        float* st = P.STATS + (int64_t)unit * 2 * P.N + n0 + col;
        st[0] = sy; st[P.N] = sq;
      }
    }
    uint16_t* C = static_cast<uint16_t*>(P.C);
    constexpr int CH = BN / 8, ITERS = BM * CH / THREADS;
    static_assert(BM * CH % THREADS == 0, "whole output chunks per thread");
    static_assert(THREADS % CH == 0, "a thread keeps its column chunk over all of its rows");
    static_assert(CH == E_CH && ITERS == E_ITERS, "prefetch geometry == epilogue geometry");
    float q0[8], q1[8];
    if (!bnst) {
      // fully unrolled, addend loads first: all of a thread's chunks are in flight together
      // instead of one load -> add -> store round trip per chunk
      uint4 addv[ITERS];
      if (P.ADD) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
          const int m = CLS ? rowpix[row] : m0 + row, n = n0 + ch * 8;
          bool aok = m >= 0 && m < P.M && n < P.N;
          int64_t am = m;
          if (P.add_sh && aok) aok = addend_sub_row(P, m, am);
          addv[it] = aok ? *reinterpret_cast<const uint4*>(P.ADD + am * P.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
        const int m = CLS ? rowpix[row] : m0 + row, n = n0 + ch * 8;
        if (m >= 0 && m < P.M && n < P.N) {
          uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_LD + ch * 8);
          if (P.ADD) {   // fused gradient accumulation: out = bf16(bf16(acc) + addend), as the separate add would give
            const uint4 q = addv[it];
            v.x = add_bf16x2(v.x, q.x); v.y = add_bf16x2(v.y, q.y); v.z = add_bf16x2(v.z, q.z); v.w = add_bf16x2(v.w, q.w);
          }
          store16(C + (int64_t)m * P.ldc + n, v);
        }
      }
    } else {
      // Batch-norm backward reductions fused into the dgrad epilogue (rigl_masked_conv2d_bwd_bn): the tile being stored
      // IS the gradient w.r.t. y = relu?(bn(x) [+ residual]); sum dz and sum dz * xhat (dz = relu-masked gradient) per
      // column are exactly what the batch norm's own reduction pass would read this tensor and x again for
      // (bn.hip k_reduce<1>: resnet_model.py:41-82 through autodiff).  A thread owns one 8-column chunk for all its
      // rows; the per-channel statistics sit in LDS behind the staging tile; the matching x tile was requested at the
      // top of the kernel (stride-1 layers) and has long arrived.
      const int chf = tid % CH;
      const float* bprm = reinterpret_cast<const float*>(smem + PRM_OFF);   // [4][BN]: mean, invstd, scale, shift (filled at the top)
      if (!PRM_OUT) {
        float* bw = reinterpret_cast<float*>(smem + PRM_OFF);
        for (int i = tid; i < 4 * BN; i += THREADS) {
          const int k = i / BN, c = i % BN;
          bw[i] = (n0 + c < P.N) ? P.BNP[k * P.N + n0 + c] : 0.f;
        }
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) q0[j] = q1[j] = 0.f;
      constexpr int BATCH = ITERS < 4 ? ITERS : 4;
      static_assert(ITERS % BATCH == 0, "whole batches");
#pragma unroll
      for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
        uint4 addv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int idx = (b0 + u) * THREADS + tid, row = idx / CH, ch = idx % CH;
          const int m = CLS ? rowpix[row] : m0 + row, n = n0 + ch * 8;
          const bool ok = m >= 0 && m < P.M && n < P.N;
          addv[u] = (P.ADD && ok) ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)m * P.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
          if constexpr (CLS) {      // class-major rows: the pixel table exists only now (batch by batch: register budget)
            bnx[b0 + u] = ok ? *reinterpret_cast<const uint4*>(P.BNX + (int64_t)m * P.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
            bnb[b0 + u] = (ok && P.BNBITS) ? (uint32_t)P.BNBITS[((int64_t)m * P.ldc + n) >> 3] : 0u;
          }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int it = b0 + u;
          const int idx = it * THREADS + tid, row = idx / CH, ch = idx % CH;
          const int m = CLS ? rowpix[row] : m0 + row, n = n0 + ch * 8;
          if (m >= 0 && m < P.M && n < P.N) {
            uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_LD + ch * 8);
            if (P.ADD) {
              const uint4 q = addv[u];
              v.x = add_bf16x2(v.x, q.x); v.y = add_bf16x2(v.y, q.y); v.z = add_bf16x2(v.z, q.z); v.w = add_bf16x2(v.w, q.w);
            }
            *reinterpret_cast<uint4*>(C + (int64_t)m * P.ldc + n) = v;
            const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, xw[4] = {bnx[it].x, bnx[it].y, bnx[it].z, bnx[it].w};
            const float* pc = bprm + chf * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float dv = __uint_as_float((j & 1) ? (vw[j >> 1] & 0xFFFF0000u) : (vw[j >> 1] << 16));
              const float xv = __uint_as_float((j & 1) ? (xw[j >> 1] & 0xFFFF0000u) : (xw[j >> 1] << 16));
              bool on = true;
              if (P.bn_relu) on = P.BNBITS ? ((bnb[it] >> j) & 1u) != 0u : (fmaf(xv, pc[2 * BN + j], pc[3 * BN + j]) > 0.f);
              const float dz = on ? dv : 0.f;
              q0[j] += dz;
              q1[j] = fmaf(dz, (xv - pc[j]) * pc[BN + j], q1[j]);
            }
          }
        }
      }
    }
    if (bnst) {
      // Combine the row groups of every column chunk in a fixed order (deterministic): the lanes of a wave that share a
      // chunk are CH apart (xor-shuffle tree), the four waves meet in the staging area every thread has finished reading.
#pragma unroll
      for (int off = CH; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { q0[j] += __shfl_xor(q0[j], off); q1[j] += __shfl_xor(q1[j], off); }
      }
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);           // [waves][CH][16]
      if (lane < CH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(wave * CH + lane) * 16 + j] = q0[j]; red[(wave * CH + lane) * 16 + 8 + j] = q1[j]; }
      }
      __syncthreads();
      if (tid < CH * 16) {
        const int c = tid >> 4, v = tid & 15;
        float sum = red[c * 16 + v];
#pragma unroll
        for (int wv = 1; wv < THREADS / 64; ++wv) sum += red[(wv * CH + c) * 16 + v];
        const int n = n0 + c * 8 + (v & 7);
        if (n < P.N) P.STATS[(int64_t)tile_m * 2 * P.N + (int64_t)(v >> 3) * P.N + n] = sum;
      }
    }
  }
}

template <int TM, int TN, int BK, int MODE, bool OUT_F32, bool CLS, int STAGES>
__global__ __launch_bounds__(THREADS) void k_igemm(IgemmArgs P) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[igemm_smem_bytes<TM, TN, BK, MODE, OUT_F32, CLS, STAGES>()];
  igemm_body<TM, TN, BK, MODE, OUT_F32, CLS, STAGES>(P, smem, blockIdx.x, gridDim.x);
}
// 256x128 tile, 8 waves (4 x 2), 3-deep DMA ring: 72 KB of dynamic LDS, 2 workgroups per CU.
template <int MODE, bool CLS>
__global__ __launch_bounds__(512) void k_igemm_big(IgemmArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
  igemm_body<2, 2, 32, MODE, false, CLS, 3, 4>(P, smem_big, blockIdx.x, gridDim.x);
}
// ------------------------------------------------------------------ wgrad
struct WgradArgs {
  const uint16_t* X;   // NHWC input activations
  const uint16_t* DY;  // NHWC output gradients
  float* OUT;          // [splits][KH*KW][Cin][Cout] partial slabs (or dw itself when splits == 1)
  int M;               // N*Ho*Wo pixels (reduction length)
  int Cin, Cout;
  int KH, KW;
  int H, W, Ho, Wo;
  int sh, sw, ph, pw;
  int tiles_ci, tiles_co;
  int splits;
  int64_t slab_elems;  // KH*KW*Cin*Cout
  uint32_t x_bytes, dy_bytes;
  int x_pix_stride;    // elements between consecutive pixels of X (== Cin except on the tiny-Cin path)
  int fold;            // > 0 (k_wgrad_tr only): the KH taps are folded into the channel axis, `fold` channels per
                       // tap -- channel c of the GEMM is tap c / fold, input channel c % fold (KW == 1, Cin = taps*fold)
  FastDiv fd_wo, fd_ho;   // (ping-pong body only) launch-time divisors of the output-pixel decomposition
  int interleave;         // splits take interleaved K-tiles (knob "wgrad_il"; k_wgrad_tr: the 1x1 layers only)
};

// wgrad, LDS-DMA + transpose-read.  Both operands are reduction(pixel)-major
// in memory ([pixel][channel]) while the MFMA wants 8 consecutive k per lane (round 1's
// first kernel transposed 8x8 blocks in registers, ~100 VALU per K-tile).  gfx950's
// ds_read_b64_tr_b16 does that transposition in the LDS read path: per 16-lane
// group, lane j supplies the address of 4 consecutive channels of pixel j/4 and
// receives channel j of the 16-channel block for pixels 0..3 (measured:
// tools/probes/tr_read_probe.hip).  The LDS image can therefore stay
// [pixel][channel] -- exactly what `buffer_load ... lds` writes (lane-linear) --
// so the tile goes HBM -> LDS without touching registers, 4 stages deep behind a
// counted vmcnt, and the K loop holds only DMA issue, 8-byte tr reads and MFMAs.
// Bank conflicts: a 32-lane pass reads 4 pixel rows x 64 B; 64-B quads are XORed
// with the pixel index (source-side, in the DMA lane -> channel mapping) so the
// four rows land in four different 16-bank groups.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 lds_read_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// 16-B slot XOR for pixel row p of a [pixel][C*8 channels] image (C = 8 or 16 slots per row)
template <int C>
__device__ __forceinline__ int trswz(int p) { return ((p / (16 / C)) & (C / 4 - 1)) << 2; }

template <int TM, int TN, int STAGES>
constexpr int wgrad_tr_smem_bytes() { return STAGES * (64 * TM + 64 * TN) * 32 * 2; }

template <int TM, int TN, int STAGES>
__device__ __forceinline__ void wgrad_tr_body(const WgradArgs& P, unsigned char* smem, uint32_t bid, uint32_t nblk) {
  constexpr int BK = 32;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int CA = BM / 8, CB = BN / 8;
  constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int APASS = BK * CA / THREADS, BPASS = BK * CB / THREADS;
  static_assert(STAGES * STAGE <= 65536, "static LDS limit");
  static_assert(STAGES * STAGE == wgrad_tr_smem_bytes<TM, TN, STAGES>(), "LDS size formula out of sync");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  uint32_t b = xcd_remap(bid, nblk);
  const int tco = b % P.tiles_co; b /= P.tiles_co;
  const int tci = b % P.tiles_ci; b /= P.tiles_ci;
  const int tap = b % (P.KH * P.KW);
  const int split = b / (P.KH * P.KW);
  const int r = tap / P.KW, s = tap % P.KW;
  const int ci0 = tci * BM, co0 = tco * BN;
  const int KT_all = (P.M + BK - 1) / BK;
  const bool direct = !P.fold && P.KH == 1 && P.KW == 1 && P.sh == 1 && P.sw == 1 && P.ph == 0 && P.pw == 0;
  // the 1x1 layers ("wgrad_il"): a split takes every splits-th K-tile, so that the splits of a tile stream one window
  // of X and dY together instead of `splits` streams far apart; other layers keep contiguous pixel ranges (their
  // lane-local pixel decomposition advances incrementally)
  const bool il = P.interleave != 0 && direct && P.splits > 1;
  const int kt_begin = il ? split : (int)((int64_t)KT_all * split / P.splits);
  const int KT = il ? (split < KT_all ? (KT_all - split + P.splits - 1) / P.splits : 0)
                    : (int)((int64_t)KT_all * (split + 1) / P.splits) - kt_begin;
  const int m_step = il ? BK * P.splits : BK;
  const bool fast_inc = (BK / P.Wo + 1) <= P.Ho;
  const int inc_w = BK % P.Wo, inc_h = BK / P.Wo;
  const int hi0 = r - P.ph, wi0 = s - P.pw;
  const __amdgpu_buffer_rsrc_t rsrcX = make_rsrc(P.X, P.x_bytes), rsrcY = make_rsrc(P.DY, P.dy_bytes);
  const u32x4 rsrcX4 = make_rsrc4(P.X, P.x_bytes), rsrcY4 = make_rsrc4(P.DY, P.dy_bytes);
  (void)rsrcX4; (void)rsrcY4; (void)rsrcX; (void)rsrcY;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // ---- DMA lanes: slot q*256+tid of a stage = (pixel row, 16-B slot) ----------
  int a_m[APASS], a_ch[APASS], a_r[APASS], a_n[APASS], a_ho[APASS], a_wo[APASS];
  bool a_cok[APASS];
#pragma unroll
  for (int q = 0; q < APASS; ++q) {
    const int slot = q * THREADS + tid, p = slot / CA, pc = slot % CA;
    a_ch[q] = ci0 + ((pc ^ trswz<CA>(p)) << 3);
    a_cok[q] = a_ch[q] < P.Cin;
    a_r[q] = 0;
    if (P.fold) { a_r[q] = a_ch[q] / P.fold; a_ch[q] -= a_r[q] * P.fold; }   // folded taps: this lane's own filter row
    a_m[q] = kt_begin * BK + p;
    const int t = a_m[q] / P.Wo;
    a_wo[q] = a_m[q] % P.Wo; a_ho[q] = t % P.Ho; a_n[q] = t / P.Ho;
  }
  int b_m[BPASS], b_ch[BPASS];
  bool b_cok[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int slot = q * THREADS + tid, p = slot / CB, pc = slot % CB;
    b_ch[q] = co0 + ((pc ^ trswz<CB>(p)) << 3);
    b_cok[q] = b_ch[q] < P.Cout;
    b_m[q] = kt_begin * BK + p;
  }
#define RIGL_WTR_ISSUE(stage_)                                                                        \
  {                                                                                                   \
    unsigned char* As_ = smem + (stage_) * STAGE;                                                     \
    unsigned char* Bs_ = As_ + A_BYTES;                                                               \
    _Pragma("unroll") for (int q = 0; q < APASS; ++q) {                                               \
      bool ok = a_cok[q] && a_m[q] < P.M;                                                             \
      int off;                                                                                        \
      if (direct) {                                                                                   \
        off = a_m[q] * P.x_pix_stride + a_ch[q];                                                      \
      } else {                                                                                        \
        const int hi = a_ho[q] * P.sh + hi0 + a_r[q], wi = a_wo[q] * P.sw + wi0;                      \
        ok = ok && (unsigned)hi < (unsigned)P.H && (unsigned)wi < (unsigned)P.W;                      \
        off = ((a_n[q] * P.H + hi) * P.W + wi) * P.x_pix_stride + a_ch[q];                            \
      }                                                                                               \
      const int boff = ok ? (int)((uint32_t)off * 2u) : (int)OOB;                                     \
      RIGL_DMA16(rsrcX, As_ + (q * THREADS + wave_u * 64) * 16, boff); \
      a_m[q] += m_step;                                                                                \
      if (!direct) {                                                                                  \
        if (fast_inc) {                                                                               \
          a_wo[q] += inc_w; if (a_wo[q] >= P.Wo) { a_wo[q] -= P.Wo; ++a_ho[q]; }                      \
          a_ho[q] += inc_h; if (a_ho[q] >= P.Ho) { a_ho[q] -= P.Ho; ++a_n[q]; }                       \
        } else {                                                                                      \
          const int t = a_m[q] / P.Wo;                                                                \
          a_wo[q] = a_m[q] % P.Wo; a_ho[q] = t % P.Ho; a_n[q] = t / P.Ho;                             \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < BPASS; ++q) {                                               \
      const bool okb = b_cok[q] && b_m[q] < P.M;                                                      \
      const int boff = okb ? (int)((uint32_t)(b_m[q] * P.Cout + b_ch[q]) * 2u) : (int)OOB;            \
      RIGL_DMA16(rsrcY, Bs_ + (q * THREADS + wave_u * 64) * 16, boff); \
      b_m[q] += m_step;                                                                                \
    }                                                                                                 \
  }

  // ---- tr-read lanes ----------------------------------------------------------
  const int g = lane >> 4, j = lane & 15;
  const int prow = 8 * (g >> 1) + (j >> 2);               // pixel inside a 16-pixel k-step (t = 0)
  int a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int chunk = (wm * 32 * TM + i * 32) / 8 + 2 * (g & 1) + ((j >> 1) & 1);
    a_rd[i] = prow * (CA * 16) + ((chunk ^ trswz<CA>(prow)) << 4) + (j & 1) * 8;
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int chunk = (wn * 32 * TN + i * 32) / 8 + 2 * (g & 1) + ((j >> 1) & 1);
    b_rd[i] = prow * (CB * 16) + ((chunk ^ trswz<CB>(prow)) << 4) + (j & 1) * 8;
  }
#define RIGL_WTR_COMPUTE(stage_)                                                                      \
  {                                                                                                   \
    const unsigned char* As = smem + (stage_) * STAGE;                                                \
    const unsigned char* Bs = As + A_BYTES;                                                           \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                          \
      bf16x8 af[TM], bfr[TN];                                                                         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        af[i] = lds_read_tr_pair(As + a_rd[i] + (ks * 16) * (CA * 16), As + a_rd[i] + (ks * 16 + 4) * (CA * 16)); \
      _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                  \
        bfr[i] = lds_read_tr_pair(Bs + b_rd[i] + (ks * 16) * (CB * 16), Bs + b_rd[i] + (ks * 16 + 4) * (CB * 16)); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        _Pragma("unroll") for (int jj = 0; jj < TN; ++jj)                                             \
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jj], acc[i][jj], 0, 0, 0);  \
    }                                                                                                 \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < TN; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  constexpr int L = APASS + BPASS;
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < KT) RIGL_WTR_ISSUE(t);
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + STAGES - 1 <= KT) wait_vmcnt<L * (STAGES - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < KT) RIGL_WTR_ISSUE((kt + STAGES - 1) % STAGES);
    RIGL_WTR_COMPUTE(kt % STAGES);
  }
#undef RIGL_WTR_ISSUE
#undef RIGL_WTR_COMPUTE
  float* out = P.OUT + (int64_t)split * P.slab_elems + (int64_t)tap * P.Cin * P.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < TN; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int co = co0 + wn * 32 * TN + jj * 32 + (lane & 31);
        if (ci < P.Cin && co < P.Cout) out[(int64_t)ci * P.Cout + co] = acc[i][jj][e];
      }
}

template <int TM, int TN, int STAGES>
__global__ __launch_bounds__(THREADS) void k_wgrad_tr(WgradArgs P) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[wgrad_tr_smem_bytes<TM, TN, STAGES>()];
  wgrad_tr_body<TM, TN, STAGES>(P, smem, blockIdx.x, gridDim.x);
}

// dw[i] = sum_s slab[s][i] in a FIXED order (deterministic => identical masks
// run to run).  A workgroup owns 64 consecutive outputs; its 256 threads are 16
// float4 columns x 16 split-groups, so 16 independent 256-B reads are in flight
// per step instead of one thread walking all splits serially; group partials are
// combined through LDS in ascending group order.  n_out <= slab_elems lets the
// small-Cin (im2col) path drop its zero padding rows.
struct ReduceArgs {
  const float* slabs;
  float* dw;
  int64_t n_out, slab_elems;
  int splits;
};

// One 64-output group per workgroup.  (Rounds 2-3 also ran this as a third workgroup segment of the NEXT layer's
// backward launch -- bit-identical, +0.5 ms per step, removed.)
__global__ __launch_bounds__(THREADS) void k_wgrad_reduce(const float* __restrict__ slabs, float* __restrict__ dw,
                                                           int64_t n_out, int64_t slab_elems, int splits) {
  __shared__ float4 part[16][16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * 64 + col * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i0 + 3 < slab_elems && (slab_elems & 3) == 0) {
#pragma unroll 4
    for (int s2 = grp; s2 < splits; s2 += 16) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)s2 * slab_elems + i0);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  } else {
    for (int s2 = grp; s2 < splits; s2 += 16) {
      const float* p = slabs + (int64_t)s2 * slab_elems;
      if (i0 + 0 < slab_elems) acc.x += p[i0 + 0];
      if (i0 + 1 < slab_elems) acc.y += p[i0 + 1];
      if (i0 + 2 < slab_elems) acc.z += p[i0 + 2];
      if (i0 + 3 < slab_elems) acc.w += p[i0 + 3];
    }
  }
  part[grp][col] = acc;
  __syncthreads();
  if (grp == 0) {
    float4 r = part[0][col];
#pragma unroll
    for (int g2 = 1; g2 < 16; ++g2) {
      const float4 v = part[g2][col];
      r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    if (i0 + 0 < n_out) dw[i0 + 0] = r.x;
    if (i0 + 1 < n_out) dw[i0 + 1] = r.y;
    if (i0 + 2 < n_out) dw[i0 + 2] = r.z;
    if (i0 + 3 < n_out) dw[i0 + 3] = r.w;
  }
}

// (Round 6: a one-thread-per-float4-column form of the kernel above -- all slabs of a column walked by one thread, sixteen loads in
// flight, 1 KB of a slab per wave instruction, the same additions in the same order (dW bit-equal on seven layer shapes) --
// measured SLOWER in the step: conv_bwd 3.53 / 3.57 -> 3.68 ms; the 4096 four-wave workgroups of two loads per thread hide
// their latency better than 1024 single-wave workgroups of thirty-two.  Removed.)
// Few splits (the large layers: 3-7 slabs of up to 2.4 M outputs): the 16 split-groups above would leave most of
// the workgroup idle, so G = 4 or 8 groups x 256 / G float4 columns.  With G >= splits every group holds at most
// one slab and the combine is the plain sum in split order -- the same bits as the 16-group kernel.
template <int G>
__global__ __launch_bounds__(THREADS) void k_wgrad_reduce_few(const float* __restrict__ slabs, float* __restrict__ dw,
                                                               int64_t n_out, int64_t slab_elems, int splits) {
  constexpr int CW = THREADS / G;
  __shared__ float4 part[G][CW];
  const int col = threadIdx.x % CW, grp = threadIdx.x / CW;
  const int64_t i0 = ((int64_t)blockIdx.x * CW + col) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (grp < splits) {
    const float* p = slabs + (int64_t)grp * slab_elems;
    if (i0 + 3 < slab_elems && (slab_elems & 3) == 0) {
      const float4 v = *reinterpret_cast<const float4*>(p + i0);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    } else {
      if (i0 + 0 < slab_elems) acc.x += p[i0 + 0];
      if (i0 + 1 < slab_elems) acc.y += p[i0 + 1];
      if (i0 + 2 < slab_elems) acc.z += p[i0 + 2];
      if (i0 + 3 < slab_elems) acc.w += p[i0 + 3];
    }
  }
  part[grp][col] = acc;
  __syncthreads();
  if (grp == 0) {
    float4 r = part[0][col];
#pragma unroll
    for (int g2 = 1; g2 < G; ++g2) {
      const float4 v = part[g2][col];
      r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
    }
    if (i0 + 0 < n_out) dw[i0 + 0] = r.x;
    if (i0 + 1 < n_out) dw[i0 + 1] = r.y;
    if (i0 + 2 < n_out) dw[i0 + 2] = r.z;
    if (i0 + 3 < n_out) dw[i0 + 3] = r.w;
  }
}

static void launch_wgrad_reduce(const ReduceArgs& ra, hipStream_t st) {
  if (ra.splits <= 4)
    RIGL_K_LAUNCH(k_wgrad_reduce_few<4>, dim3((unsigned)ceil_div64(ra.n_out, 256)), dim3(THREADS), 0, st, ra.slabs, ra.dw, ra.n_out, ra.slab_elems, ra.splits);
  else if (ra.splits <= 8)
    RIGL_K_LAUNCH(k_wgrad_reduce_few<8>, dim3((unsigned)ceil_div64(ra.n_out, 128)), dim3(THREADS), 0, st, ra.slabs, ra.dw, ra.n_out, ra.slab_elems, ra.splits);
  else
    RIGL_K_LAUNCH(k_wgrad_reduce, dim3((unsigned)ceil_div64(ra.n_out, 64)), dim3(THREADS), 0, st, ra.slabs, ra.dw, ra.n_out, ra.slab_elems, ra.splits);
}

// Whole backward of a conv in ONE launch: the first `nw` workgroups run the weight-gradient GEMM, the
// next `nd` the dgrad implicit GEMM.  The two are independent (both read dY) and on their own each
// leaves much of the chip idle (196-784 tiles for 768 workgroup slots), so sharing a launch lets the
// dispatcher fill the machine with whichever still has tiles -- the overlap a second stream gives,
// without a second stream.  Same bodies, one LDS allocation (the larger of the two).
template <int TND, bool CLSD, int TMW, int TNW, int STW>
__global__ __launch_bounds__(THREADS) void k_bwd_fused(IgemmArgs PD, WgradArgs PW, uint32_t nd, uint32_t nw) {
  constexpr int SD = igemm_smem_bytes<2, TND, 32, 1, false, CLSD, 3>(), SW = wgrad_tr_smem_bytes<TMW, TNW, STW>();
  __shared__ __attribute__((aligned(16))) unsigned char smem[SD > SW ? SD : SW];
  // longest jobs first: a weight-gradient workgroup walks 1/splits of all pixels and runs several times as long as
  // a dgrad tile, so it must not be what is left for the tail (dgrad first: 13.85 ms per step, wgrad first: 13.74)
  if (blockIdx.x < nw) wgrad_tr_body<TMW, TNW, STW>(PW, smem, blockIdx.x, nw);
  else igemm_body<2, TND, 32, 1, false, CLSD, 3>(PD, smem, blockIdx.x - nw, nd);
}

// ------------------------------------------------------------------ small-Cin (stem) path
// Explicit im2col for layers whose Cin is not a multiple of 8 (the 7x7x3 stem):
// col[m][kk] = x[n, ho*sh-ph+r, wo*sw-pw+s, c] with kk = (r*KW+s)*Cin + c, zero
// padded to Kp columns.  The conv then runs as a 1x1 conv over `col`.
struct Im2colArgs {
  const uint16_t* X; uint16_t* COL;
  int M, Cin, KH, KW, H, W, Ho, Wo, sh, sw, ph, pw, K, Kp;
};
__global__ __launch_bounds__(THREADS) void k_im2col(Im2colArgs P) {
  const int CH = P.Kp / 8;
  const int64_t total = (int64_t)P.M * CH;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += stride) {
    const int m = (int)(idx / CH), ch = (int)(idx % CH);
    const int wo = m % P.Wo, t = m / P.Wo, ho = t % P.Ho, n = t / P.Ho;
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = ch * 8 + e;
      uint16_t x = 0;
      if (kk < P.K) {
        const int c = kk % P.Cin, tap = kk / P.Cin, s = tap % P.KW, r = tap / P.KW;
        const int hi = ho * P.sh - P.ph + r, wi = wo * P.sw - P.pw + s;
        if ((unsigned)hi < (unsigned)P.H && (unsigned)wi < (unsigned)P.W)
          x = P.X[((int64_t)(n * P.H + hi) * P.W + wi) * P.Cin + c];
      }
      v[e] = x;
    }
    *reinterpret_cast<uint4*>(P.COL + (int64_t)m * P.Kp + ch * 8) = *reinterpret_cast<const uint4*>(v);
  }
}
// wp[co][kk] = w_ohwi[co][kk] for kk < K else 0  (row length Kp, 16-byte aligned rows)
__global__ __launch_bounds__(THREADS) void k_pad_rows(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                       int rows, int K, int Kp) {
  const int64_t total = (int64_t)rows * Kp;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    const int rr = (int)(i / Kp), kk = (int)(i % Kp);
    dst[i] = kk < K ? src[(int64_t)rr * K + kk] : (uint16_t)0;
  }
}

// ------------------------------------------------------------------ tiny-Cin (stem) path
// Cin <= 4 (the 7x7x3 ImageNet stem, the 3x3x3 CIFAR stem).  Instead of an
// explicit im2col (513 MB at batch 128) the image is copied once into a
// zero-bordered, 4-channel NHWC buffer; one filter ROW then is KW*4 contiguous
// bf16 in memory, so the conv runs on the ordinary kernels as a KHx1 conv with
// "channel" count Cred = roundup(KW*4, 8) and pixel stride 4.  The surplus
// elements of a row window (4th channel, pixels beyond KW) meet zero weights.
struct PadArgs {
  const uint16_t* X; uint16_t* XP;
  int N, H, W, Cin, Hp, Wp, pt, pl;
};
__global__ __launch_bounds__(THREADS) void k_pad_input4(PadArgs P) {
  const int64_t total = (int64_t)P.N * P.Hp * P.Wp;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += stride) {
    const int xw = (int)(idx % P.Wp);
    int64_t t = idx / P.Wp;
    const int yh = (int)(t % P.Hp), n = (int)(t / P.Hp);
    const int hi = yh - P.pt, wi = xw - P.pl;
    uint16_t v[4] = {0, 0, 0, 0};
    if ((unsigned)hi < (unsigned)P.H && (unsigned)wi < (unsigned)P.W) {
      const uint16_t* src = P.X + ((int64_t)(n * P.H + hi) * P.W + wi) * P.Cin;
      for (int c = 0; c < P.Cin; ++c) v[c] = src[c];
    }
    *reinterpret_cast<uint2*>(P.XP + idx * 4) = *reinterpret_cast<const uint2*>(v);
  }
}
// wp[co][r][j], j = s*4 + c  <-  w_ohwi[co][(r*KW + s)*Cin + c]   (0 outside)
__global__ __launch_bounds__(THREADS) void k_stem_weights(const uint16_t* __restrict__ w, uint16_t* __restrict__ wp,
                                                           int cout, int KH, int KW, int Cin, int Cred) {
  const int total = cout * KH * Cred;
  for (int i = blockIdx.x * THREADS + threadIdx.x; i < total; i += gridDim.x * THREADS) {
    const int j = i % Cred, r = (i / Cred) % KH, co = i / (Cred * KH);
    const int s2 = j >> 2, c = j & 3;
    wp[i] = (s2 < KW && c < Cin) ? w[(int64_t)co * KH * KW * Cin + (r * KW + s2) * Cin + c] : (uint16_t)0;
  }
}
// dw[(r*KW+s)*Cin + c][co]  <-  t[r][s*4 + c][co]
// (shift: the column the first filter tap sits in -- 1 for the stem.hpp kernels, whose windows start one pixel early)
__global__ __launch_bounds__(THREADS) void k_stem_unpack(const float* __restrict__ t, float* __restrict__ dw, int KH,
                                                          int KW, int Cin, int Cred, int cout, int shift) {
  const int total = KH * KW * Cin * cout;
  for (int i = blockIdx.x * THREADS + threadIdx.x; i < total; i += gridDim.x * THREADS) {
    const int co = i % cout;
    int k = i / cout;
    const int c = k % Cin; k /= Cin;
    const int s2 = k % KW, r = k / KW;
    dw[i] = t[((int64_t)r * Cred + (s2 + shift) * 4 + c) * cout + co];
  }
}
struct TinyGeom { int cred, hp, wp; };
static inline bool tiny_cin(const RiglConvDesc* d) { return d->cin <= 4; }
static TinyGeom tiny_geom(const RiglConvDesc* d) {
  TinyGeom g;
  g.cred = (d->kw * 4 + 7) / 8 * 8;
  const int need_h = (d->ho - 1) * d->stride_h + d->kh, need_w = (d->wo - 1) * d->stride_w + g.cred / 4;
  g.hp = need_h > d->h + d->pad_top ? need_h : d->h + d->pad_top;
  g.wp = need_w > d->w + d->pad_left ? need_w : d->w + d->pad_left;
  return g;
}

// MFMA issue-rate probe ("re-measured on the box before use"): every wave of a 256-thread workgroup issues
// `iters` x 8 independent v_mfma_f32_32x32x16_bf16 back to back on register operands -- no memory traffic -- so the
// rate it reaches is the clock-and-power-limited dense bf16 peak of THIS box (bench.py prints it beside roofline.peak).
__global__ __launch_bounds__(THREADS) void k_mfma_probe(float* __restrict__ sink, int iters, float seed) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + 0.001f * (float)((threadIdx.x + i) & 63)); b[i] = (__bf16)(1.0f - 0.002f * (float)((threadIdx.x * 3 + i) & 31)); }
  f32x16 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[k][e];
  if (s == 12345.678f) sink[blockIdx.x * THREADS + threadIdx.x] = s;   // keeps the chain live, (almost) never stores
}

// ------------------------------------------------------------------ dispatch
template <int MODE, bool F32, bool CLS>
static void launch_igemm_t(const IgemmArgs& a, dim3 grid, bool wide_n, bool dma, hipStream_t st) {
  dim3 blk(THREADS);
  if (dma) {   // LDS-DMA ring, BK = 32: 3 stages = 48 KB of LDS -> 3 workgroups per CU (the 136-VGPR limit too)
    if (wide_n) RIGL_K_LAUNCH((k_igemm<2, 2, 32, MODE, F32, CLS, 3>), grid, blk, 0, st, a);
    else RIGL_K_LAUNCH((k_igemm<2, 1, 32, MODE, F32, CLS, 3>), grid, blk, 0, st, a);
    return;
  }
  // reductions narrower than 32 channels (8, 16, 24): the register-staged double buffer, BK = 16
  if (wide_n) RIGL_K_LAUNCH((k_igemm<2, 2, 16, MODE, F32, CLS, 2>), grid, blk, 0, st, a);
  else RIGL_K_LAUNCH((k_igemm<2, 1, 16, MODE, F32, CLS, 2>), grid, blk, 0, st, a);
}

struct IgemmPlan { bool wide_n, dma, cls, big; unsigned grid; };

static int num_cus();
// The 512-thread kernel needs 72 KB of dynamic LDS: opt in once; if the runtime refuses, the plan never picks it.
template <int MODE>
static bool big_tile_ready() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm_big<MODE, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                igemm_smem_bytes<2, 2, 32, MODE, false, false, 3, 4>()) == hipSuccess;
  return ready;
}

// Fills the launch-time parts of `a` (fast divisors, tile counts, parity-class tables) and picks the variant.
template <int MODE>
static IgemmPlan plan_igemm(IgemmArgs& a) {
  IgemmPlan pl;
  a.fd_rw = make_fastdiv(a.RW); a.fd_rh = make_fastdiv(a.RH);
  // 128x64 tiles when the 128x128 grid has fewer tiles than CUs (7x7 layers at batch 128: 196): twice the
  // workgroups, ~5 % faster; at 392 tiles the narrower tile's lower arithmetic intensity already loses.
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
  pl.wide_n = a.N > 64 && tiles128 >= 256;
  const int BM = 128;
  const int BN = pl.wide_n ? 128 : 64;
  a.tiles_n = (a.N + BN - 1) / BN;
  pl.dma = a.Cred >= 32;       // (narrower reductions: the register-staged body, BK = 16)
  pl.cls = false;
  pl.big = false;
  if (MODE == 1 && (a.sh > 1 || a.sw > 1) && a.sh <= 2 && a.sw <= 2) {
    // class-major rows: class c = (h % sh) * sw + (w % sw)
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
      tiles += (a.cls_cnt[c] + BM - 1) / BM;
    }
    a.cls_tile_begin[4] = tiles;
    a.cls_n = 0; a.cls_interleave = 1 << 30;
    for (int c = 0; c < 4; ++c) {
      const int tc = a.cls_tile_begin[c + 1] - a.cls_tile_begin[c];
      if (tc > 0) { a.cls_ids[a.cls_n++] = c; if (tc < a.cls_interleave) a.cls_interleave = tc; }
    }
    if (a.cls_n == 0) { a.cls_n = 1; a.cls_ids[0] = 0; a.cls_interleave = 0; }
    pl.cls = true;
    pl.grid = (unsigned)(tiles * a.tiles_n);
    return pl;
  }
  // 256x128 tiles (8 waves) where they turn two rounds of workgroups into one: the 128x128 grid does not fit the
  // 3 workgroups per CU of its 48 KB ring, the 256x128 grid fits the 2 per CU of its 72 KB -- at batch 128 the
  // 784-tile layers (28x28x128 3x3 and 512->128, 14x14 1024->512, 7x7 ->2048): -11...-20 % each.  On other
  // shapes the larger tile is neutral (14x14x256 3x3: 196 workgroups for 256 CUs) or loses (short reductions).
  // RIGL_CONV_BIG=0 never, =2 wherever it is legal (testing).
  static const int big_mode = [] { const char* e = getenv("RIGL_CONV_BIG"); return e ? atoi(e) : 1; }();
  const int64_t tiles256 = (int64_t)((a.M + 255) / 256) * ((a.N + 127) / 128);
  pl.big = MODE == 0 && big_mode > 0 && big_tile_ready<MODE>() && pl.wide_n && pl.dma && a.M >= 256 &&
           (big_mode == 2 || (tiles128 > 3 * (int64_t)num_cus() && tiles256 <= 2 * (int64_t)num_cus()));
  const int tiles_m = (a.M + (pl.big ? 256 : BM) - 1) / (pl.big ? 256 : BM);
  pl.grid = (unsigned)(tiles_m * a.tiles_n);
  return pl;
}

template <int MODE, bool F32>
static void launch_igemm(const IgemmArgs& a0, hipStream_t st) {
  IgemmArgs a = a0;
  const IgemmPlan pl = plan_igemm<MODE>(a);
  if (!F32 && pl.big) {
    RIGL_K_LAUNCH((k_igemm_big<MODE, false>), dim3(pl.grid), dim3(512), (igemm_smem_bytes<2, 2, 32, MODE, false, false, 3, 4>()), st, a);
    return;
  }
  if (pl.cls) launch_igemm_t<MODE, F32, true>(a, dim3(pl.grid), pl.wide_n, pl.dma, st);
  else launch_igemm_t<MODE, F32, false>(a, dim3(pl.grid), pl.wide_n, pl.dma, st);
}

static int check_desc(const RiglConvDesc* d, const char* who) {
  if (!d) return fail(RIGL_EINVAL, "%s: NULL descriptor", who);
  if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->ho <= 0 || d->wo <= 0 || d->cout <= 0 || d->kh <= 0 ||
      d->kw <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->pad_top < 0 || d->pad_left < 0)
    return fail(RIGL_EINVAL, "%s: non-positive dimension in descriptor", who);
  // every output pixel's window must start inside the padded image
  if ((d->ho - 1) * d->stride_h - d->pad_top >= d->h || (d->wo - 1) * d->stride_w - d->pad_left >= d->w)
    return fail(RIGL_EINVAL, "%s: output size inconsistent with input/stride/pad", who);
  const int64_t lim = (int64_t(1) << 30) - 1;   // bf16 tensors stay below 2^31 bytes (buffer descriptors)
  if ((int64_t)d->n * d->h * d->w * d->cin > lim || (int64_t)d->n * d->ho * d->wo * d->cout > lim ||
      (int64_t)d->kh * d->kw * d->cin * d->cout > lim)
    return fail(RIGL_EUNSUPPORTED, "%s: tensor exceeds 2^30 elements", who);
  return RIGL_OK;
}

static inline bool small_cin(const RiglConvDesc* d) { return (d->cin % 8) != 0 && d->cin > 4; }
static inline int kpad(const RiglConvDesc* d) { return (d->kh * d->kw * d->cin + 31) / 32 * 32; }

#include "convpp.hpp"
#include "bwd1x1.hpp"
#include "stem.hpp"
#include "c3x3.hpp"
#include "rowstream.hpp"
#include "bwdslice.hpp"

// Scratch of the K-split ping-pong forward (convpp.hpp: pp_ksplit_ok): two fp32 partial tiles + a counter per tile.
static size_t pp_ksplit_workspace(const RiglConvDesc* d) {
  if (d->cin <= 4 || (d->cin % 8) || (d->cout % 8)) return 0;
  IgemmArgs a = {};
  a.M = d->n * d->ho * d->wo; a.N = d->cout; a.Cred = d->cin; a.a_pix_stride = d->cin; a.KH = d->kh; a.KW = d->kw;
  a.RH = d->ho; a.RW = d->wo; a.GH = d->h; a.GW = d->w; a.sh = d->stride_h; a.sw = d->stride_w;
  const PPPlan p = plan_pp<0>(a);
  if (!pp_ksplit_ok(a, p)) return 0;
  return (size_t)p.grid * 2 * p.bm * p.bn * 4 + align_up((size_t)p.grid * 4, 256);
}

struct WgradPlan { int tm, tn, tiles_ci, tiles_co, splits; int64_t slab; };
// DMA ring depth of the tr kernel: 3 stages for the 128x128 tile (48 KB -> 3 workgroups/CU),
// 4 for the smaller tiles (measured per layer).
static int wgrad_stages(int tm, int tn) { return (tm == 2 && tn == 2) ? 3 : 4; }
static int num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
// Split-K plan.  Partial slabs cost 8 B of HBM traffic per output element per split, so
// the split count is the largest that still fits ONE co-resident round of workgroups
// (CUs x workgroups/CU at this tile's LDS footprint: 64/48/32 KB -> 2/3/4); one more
// workgroup than that starts a second, nearly empty round (measured: 513 workgroups of
// the 128x128 tile take 1.4x the time of 504).
static WgradPlan plan_wgrad(int M, int cin, int cout, int taps, bool fused = false) {
  WgradPlan p;
  p.tm = cin > 64 ? 2 : 1;
  p.tn = cout > 64 ? 2 : 1;
  p.tiles_ci = (cin + 64 * p.tm - 1) / (64 * p.tm);
  p.tiles_co = (cout + 64 * p.tn - 1) / (64 * p.tn);
  const int64_t base = (int64_t)p.tiles_ci * p.tiles_co * taps;
  const int kt = (M + 63) / 64;
  const int target = 0;
  const int lds = wgrad_stages(p.tm, p.tn) * 32 * 64 * (p.tm + p.tn) * 2;   // bytes of the tr kernel's DMA ring
  const int lds_cu = 160 * 1024;
  int occ = lds_cu / lds;
  if (occ > 4) occ = 4;
  int64_t slots = target > 0 ? target : (int64_t)num_cus() * occ;
  // Sharing the launch with the layer's dgrad tiles (k_bwd_fused, weight-gradient workgroups first): take 60 % of
  // the workgroup slots and leave the rest to the dgrad tiles from the start instead of making them wait for a
  // full round of long-running weight-gradient workgroups (ResNet-50 step: 100 % 13.87 ms, 83 % 13.99*, 67 % 13.70,
  // 58 % 13.62, 50 % 13.80, 42 % 14.09; * on a slower box whose 100 % was 14.25).
  static const int fused_pct = [] { const char* e = getenv("RIGL_WGRAD_FUSED_PCT"); return e ? atoi(e) : 60; }();
  // (the 64-wide weight-gradient tiles, planned on 4 workgroups per CU, do best at 75 %: 13.39 vs 13.44 ms; 45 % 13.57, 88 % 13.61)
  static const int fused_pct_small = [] { const char* e = getenv("RIGL_WGRAD_FUSED_PCT_SMALL"); return e ? atoi(e) : 75; }();
  if (fused && target <= 0) slots = slots * ((p.tm == 2 && p.tn == 2) ? fused_pct : fused_pct_small) / 100;
  int64_t s = slots / base;
  if (s > kt / 4) s = kt / 4;                // at least 256 pixels per split
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  p.splits = (int)s;
  p.slab = (int64_t)taps * cin * cout;
  return p;
}

// Tiny-Cin (stem) wgrad: the KH filter rows are folded into the channel axis (KH*cred = 224 "channels", one tap) so
// the 128-channel tile is full and dY is read by 2 channel tiles instead of KH = 7 taps.
static WgradPlan tiny_wgrad_plan(int M, int cred, int cout, int kh) { return plan_wgrad(M, kh * cred, cout, 1); }

}  // namespace k1
}  // namespace rigl

extern "C" {

size_t rigl_conv2d_workspace_bytes(const RiglConvDesc* d, int32_t which) {
  using namespace rigl;
  using namespace rigl::k1;
  if (!d) return 0;
  const int64_t M = (int64_t)d->n * d->ho * d->wo;
  if (tiny_cin(d)) {
    const TinyGeom tg = tiny_geom(d);
    const size_t xp = align_up((size_t)d->n * tg.hp * tg.wp * 4 * 2, 256);
    if (which == 0) return xp + align_up((size_t)d->cout * d->kh * tg.cred * 2, 256);
    if (which == 2) {
      WgradPlan p = tiny_wgrad_plan((int)M, tg.cred, d->cout, d->kh);
      const size_t generic = xp + align_up((size_t)p.slab * 4, 256) + align_up((size_t)p.splits * p.slab * 4, 256);
      const size_t direct = stem_wgrad_workspace(d);       // stem.hpp: no padded copy, one slab per workgroup
      return generic > direct ? generic : direct;
    }
    return 0;
  }
  if (small_cin(d)) {
    const int Kp = kpad(d);
    size_t col = align_up((size_t)M * Kp * 2, 256);
    if (which == 0) return col + align_up((size_t)d->cout * Kp * 2, 256);
    if (which == 2) {
      WgradPlan p = plan_wgrad((int)M, Kp, d->cout, 1);
      return col + align_up((size_t)p.splits * p.slab * 4, 256);
    }
    return 0;
  }
  if (which == 0) return pp_ksplit_workspace(d);     // the K-split partial tiles of the few-tile forwards (else 0)
  if (which == 2) {
    WgradPlan p = plan_wgrad((int)M, d->cin, d->cout, d->kh * d->kw);
    size_t need = p.splits > 1 ? align_up((size_t)p.splits * p.slab * 4, 256) : 0;
    const size_t npp = align_up(pp_wgrad_workspace(d), 256);   // the ping-pong weight gradients have their own split plans
    if (npp > need) need = npp;
    const size_t n11 = align_up(bwd1x1_workspace(d), 256);     // ... and the single-pass 1x1 backward one slab per workgroup
    if (n11 > need) need = n11;
    const size_t n33 = align_up(c3x3_wgrad_workspace(d), 256);  // ... as do the slab-resident 3x3 kernels (c3x3.hpp)
    if (n33 > need) need = n33;
    const size_t nbs = align_up(bs_workspace(d), 256);          // ... and the channel-sliced single-pass backward (bwdslice.hpp)
    if (nbs > need) need = nbs;
    return need;
  }
  return 0;
}

int32_t rigl_conv2d_stats_parts(const RiglConvDesc* d) {
  if (!d || d->cout <= 0 || (d->cout % 8)) return 0;
  const int64_t M = (int64_t)d->n * d->ho * d->wo;
  using namespace rigl::k1;
  if (c3x3_use(d)) return c3x3_stats_parts(d);     // c3x3.hpp: one partial per (persistent) workgroup
  { RsPlan rp; if (rs_use<0>(d, &rp)) return rp.gprime; }   // rowstream.hpp: one partial per workgroup of a column slice
  return (int32_t)((M + 127) / 128);     // one partial per 128-row output tile
}

int rigl_masked_conv2d_fwd_stats(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* w_ohwi, rigl_bf16* y,
                                 float* stats, size_t stats_floats, void* workspace, size_t workspace_bytes,
                                 rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::k1;
  int rc = check_desc(d, "rigl_masked_conv2d_fwd");
  if (rc) return rc;
  if (!x || !w_ohwi || !y) return fail(RIGL_EINVAL, "rigl_masked_conv2d_fwd: NULL tensor");
  if (d->cout % 8) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_fwd: cout %% 8 != 0 (use the reference kernel)");
  if (stats && stats_floats < (size_t)rigl_conv2d_stats_parts(d) * 2 * d->cout)
    return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_fwd_stats: stats buffer %zu floats < %zu", stats_floats,
                (size_t)rigl_conv2d_stats_parts(d) * 2 * d->cout);
  hipStream_t st = as_stream(stream);
  prof_set_tag(d);
  ProfFamily prof(PROF_CONV_FWD);
  IgemmArgs a = {};
  a.C = y; a.M = d->n * d->ho * d->wo; a.N = d->cout; a.ldc = d->cout; a.STATS = stats;
  const size_t need = rigl_conv2d_workspace_bytes(d, 0);
  // (the ordinary layers' only workspace is the optional K-split scratch: without it they run unsplit)
  if ((tiny_cin(d) || small_cin(d)) && need && (!workspace || workspace_bytes < need))
    return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_fwd: workspace %zu < %zu", workspace_bytes, need);
  if (tiny_cin(d)) {
    const TinyGeom tg = tiny_geom(d);
    const size_t xp_bytes = (size_t)d->n * tg.hp * tg.wp * 4 * 2;
    uint16_t* xp = static_cast<uint16_t*>(workspace);
    uint16_t* wp = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + align_up(xp_bytes, 256));
    if (stem_direct_legal(d)) {
      // the ImageNet stem: the image patch of a 16 x 16 output tile resident in LDS, no padded copy (stem.hpp)
      RIGL_K_LAUNCH(k_stem_weights_shift, dim3(56), dim3(THREADS), 0, st, w_ohwi, wp, d->cout);
      if (launch_stem_fwd(d, x, wp, y, stats, st)) {
        RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd");
        return RIGL_OK;
      }
    }
    PadArgs pa = {x, xp, d->n, d->h, d->w, d->cin, tg.hp, tg.wp, d->pad_top, d->pad_left};
    RIGL_K_LAUNCH(k_pad_input4, dim3(2048), dim3(THREADS), 0, st, pa);
    RIGL_K_LAUNCH(k_stem_weights, dim3(64), dim3(THREADS), 0, st, w_ohwi, wp, d->cout, d->kh, d->kw, d->cin, tg.cred);
    a.A = xp; a.B = wp; a.Cred = tg.cred; a.a_pix_stride = 4; a.KH = d->kh; a.KW = 1; a.RH = d->ho; a.RW = d->wo;
    a.GH = tg.hp; a.GW = tg.wp; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = a.pw = 0;
    a.b_row_stride = d->kh * tg.cred; a.b_tap_stride = tg.cred;
    a.a_bytes = (uint32_t)xp_bytes; a.b_bytes = (uint32_t)((size_t)d->cout * d->kh * tg.cred * 2);
  } else if (small_cin(d)) {
    const int K = d->kh * d->kw * d->cin, Kp = kpad(d);
    uint16_t* col = static_cast<uint16_t*>(workspace);
    uint16_t* wp = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + align_up((size_t)a.M * Kp * 2, 256));
    Im2colArgs ia = {x, col, a.M, d->cin, d->kh, d->kw, d->h, d->w, d->ho, d->wo, d->stride_h, d->stride_w, d->pad_top, d->pad_left, K, Kp};
    RIGL_K_LAUNCH(k_im2col, dim3(2048), dim3(THREADS), 0, st, ia);
    RIGL_K_LAUNCH(k_pad_rows, dim3(64), dim3(THREADS), 0, st, w_ohwi, wp, d->cout, K, Kp);
    // row space = one long row of M pixels in a single "image"
    a.A = col; a.B = wp; a.Cred = Kp; a.a_pix_stride = Kp; a.KH = a.KW = 1; a.RH = 1; a.RW = a.M; a.GH = 1; a.GW = a.M;
    a.sh = a.sw = 1; a.ph = a.pw = 0; a.b_row_stride = Kp; a.b_tap_stride = 0;
    a.a_bytes = (uint32_t)((size_t)a.M * Kp * 2); a.b_bytes = (uint32_t)((size_t)d->cout * Kp * 2);
  } else {
    a.A = x; a.B = w_ohwi; a.Cred = d->cin; a.a_pix_stride = d->cin; a.KH = d->kh; a.KW = d->kw; a.RH = d->ho; a.RW = d->wo;
    a.GH = d->h; a.GW = d->w; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_top; a.pw = d->pad_left;
    a.b_row_stride = d->kh * d->kw * d->cin; a.b_tap_stride = d->cin;
    a.a_bytes = (uint32_t)((size_t)d->n * d->h * d->w * d->cin * 2);
    a.b_bytes = (uint32_t)((size_t)d->kh * d->kw * d->cin * d->cout * 2);
    {
      RsPlan rp;
      if (rs_use<0>(d, &rp)) {                 // 1x1 / stride 1: rows streamed through registers against an LDS-stationary filter slice
        launch_rs<0>(d, rp, x, w_ohwi, nullptr, y, stats, st);
        RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd");
        return RIGL_OK;
      }
    }
    if (c3x3_use(d)) {                         // 3x3, 64 -> 64: input patch resident in LDS, filter in registers (c3x3.hpp)
      launch_c3x3<false>(d, x, w_ohwi, nullptr, y, stats, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd");
      return RIGL_OK;
    }
    const PPPlan pp = plan_pp<0>(a);           // long reductions: the 8-wave ping-pong body
    if (pp.variant && pp_ksplit_ok(a, pp) && workspace && workspace_bytes >= need && need) {
      a.ksplit = 2;
      a.KS_SLAB = static_cast<float*>(workspace);
      a.KS_CNT = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + (size_t)pp.grid * 2 * pp.bm * pp.bn * 4);
      RIGL_HIP(hipMemsetAsync(a.KS_CNT, 0, (size_t)pp.grid * 4, st));
    }
    if (pp.variant && launch_pp<0>(pp, a, st)) {
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd");
      return RIGL_OK;
    }
  }
  launch_igemm<0, false>(a, st);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd");
  return RIGL_OK;
}

// y = conv(relu(bn(x_pre)), w) with the apply pass of the batch norm done on the operand load (rowstream.hpp, BNL kernels):
// scale_shift = [2][cin] fp32 (rows 2-3 of the batch norm's `saved`), a_out = the activated tensor, written as a side output.
int32_t rigl_conv2d_fwd_takes_bn_input(const RiglConvDesc* d) {
  using namespace rigl;
  using namespace rigl::k1;
  if (!d || check_desc(d, "rigl_conv2d_fwd_takes_bn_input")) return 0;
  if ((d->cin % 8) || (d->cout % 8)) return 0;
  return rs_bnl_use(d) ? 1 : 0;
}

int rigl_masked_conv2d_fwd_bnrelu(const RiglConvDesc* d, const rigl_bf16* x_pre, const float* scale_shift, rigl_bf16* a_out,
                                  const rigl_bf16* w_ohwi, rigl_bf16* y, float* stats_partial, size_t partial_floats,
                                  void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::k1;
  (void)workspace; (void)workspace_bytes;
  int rc = check_desc(d, "rigl_masked_conv2d_fwd_bnrelu");
  if (rc) return rc;
  if (!x_pre || !scale_shift || !a_out || !w_ohwi || !y) return fail(RIGL_EINVAL, "rigl_masked_conv2d_fwd_bnrelu: NULL pointer");
  RsPlan rp;
  if ((d->cin % 8) || (d->cout % 8) || !rs_bnl_use(d, &rp))
    return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_fwd_bnrelu: this layer's forward does not take the transform (rigl_conv2d_fwd_takes_bn_input)");
  if (stats_partial && partial_floats < (size_t)rp.gprime * 2 * d->cout)
    return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_fwd_bnrelu: partial buffer %zu floats < %zu", partial_floats, (size_t)rp.gprime * 2 * d->cout);
  prof_set_tag(d);
  ProfFamily prof(PROF_CONV_FWD);
  launch_rs_bnl(d, rp, x_pre, scale_shift, a_out, w_ohwi, y, stats_partial, as_stream(stream));
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd_bnrelu");
  return RIGL_OK;
}

int rigl_masked_conv2d_fwd(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* w_ohwi, rigl_bf16* y,
                           void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  return rigl_masked_conv2d_fwd_stats(d, x, w_ohwi, y, nullptr, 0, workspace, workspace_bytes, stream);
}

static rigl::k1::IgemmArgs dgrad_args(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                      const rigl_bf16* addend, rigl_bf16* dx) {
  rigl::k1::IgemmArgs a = {};
  a.A = dy; a.B = w_hwio; a.C = dx; a.ADD = addend;
  a.M = d->n * d->h * d->w; a.N = d->cin; a.Cred = d->cout; a.ldc = d->cin;
  a.KH = d->kh; a.KW = d->kw; a.RH = d->h; a.RW = d->w; a.GH = d->ho; a.GW = d->wo;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_top; a.pw = d->pad_left;
  a.b_row_stride = d->cout; a.b_tap_stride = d->cin * d->cout; a.a_pix_stride = d->cout;
  a.a_bytes = (uint32_t)((size_t)d->n * d->ho * d->wo * d->cout * 2);
  a.b_bytes = (uint32_t)((size_t)d->kh * d->kw * d->cin * d->cout * 2);
  return a;
}

// Which dgrad kernel a layer gets decides whether its epilogue can carry the batch-norm backward reductions: only
// the igemm body does (the default for every cin % 8 == cout % 8 == 0 layer); returns its number of row tiles, else 0.
int32_t rigl_conv2d_dgrad_stats_parts(const RiglConvDesc* d) {
  using namespace rigl;
  using namespace rigl::k1;
  if (!d || check_desc(d, "rigl_conv2d_dgrad_stats_parts")) return 0;
  if ((d->cin % 8) || (d->cout % 8)) return 0;
  if (bwd1x1_kind(d)) return 0;               // the single-pass 1x1 backward has no reduction epilogue
  if (bs_use(d)) return 0;                    // nor has its channel-sliced form
  if (c3x3_use(d)) return 0;                  // nor has the slab-resident 3x3 dgrad
  if (rs_use<1>(d)) return 0;                 // nor the row-streaming dgrad
  IgemmArgs a = dgrad_args(d, nullptr, nullptr, nullptr, nullptr);
  if (plan_pp<1>(a).variant) return 0;        // the ping-pong dgrad has no reduction epilogue
  const IgemmPlan pl = plan_igemm<1>(a);
  return (int32_t)(pl.grid / (unsigned)a.tiles_n);
}

static int attach_bn(rigl::k1::IgemmArgs& a, const RiglConvDesc* d, const RiglBnReduceFuse* bn) {
  using namespace rigl;
  if (!bn) return RIGL_OK;
  if (!bn->x || !bn->params || !bn->partial) return fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd_bn: NULL pointer in the batch-norm block");
  const int32_t parts = rigl_conv2d_dgrad_stats_parts(d);
  if (parts <= 0) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_bwd_bn: this layer's dgrad kernel has no reduction epilogue");
  if (bn->partial_floats < (size_t)parts * 2 * d->cin)
    return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd_bn: partial buffer %zu floats < %zu", bn->partial_floats, (size_t)parts * 2 * d->cin);
  a.BNX = bn->x; a.BNBITS = bn->relu_bits; a.BNP = bn->params; a.bn_relu = bn->relu; a.STATS = bn->partial;
  return RIGL_OK;
}

// The addend of a dgrad may be the gradient of a SUBSAMPLED view of the tensor (sh, sw > 1: one row per pixel with
// h % sh == 0 and w % sw == 0); only the igemm body's epilogue takes that form, so such a call stays on it.
struct AddendSub { int sh, sw; };
static inline bool addend_is_sub(const rigl_bf16* addend, AddendSub s) { return addend && (s.sh > 1 || s.sw > 1); }
static void set_addend_sub(rigl::k1::IgemmArgs& a, const RiglConvDesc* d, AddendSub s) {
  a.add_sh = s.sh; a.add_sw = s.sw;
  a.add_ho = (d->h + s.sh - 1) / s.sh; a.add_wo = (d->w + s.sw - 1) / s.sw;
}

static int dgrad_impl(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio, const rigl_bf16* addend,
                      rigl_bf16* dx, const RiglBnReduceFuse* bn, rigl_stream_t stream, AddendSub sub = {1, 1});

int rigl_masked_conv2d_dgrad_acc(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                 const rigl_bf16* addend, rigl_bf16* dx, void* workspace, size_t workspace_bytes,
                                 rigl_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  return dgrad_impl(d, dy, w_hwio, addend, dx, nullptr, stream);
}

static int dgrad_impl(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio, const rigl_bf16* addend,
                      rigl_bf16* dx, const RiglBnReduceFuse* bn, rigl_stream_t stream, AddendSub sub) {
  using namespace rigl;
  using namespace rigl::k1;
  int rc = check_desc(d, "rigl_masked_conv2d_dgrad");
  if (rc) return rc;
  if (!dy || !w_hwio || !dx) return fail(RIGL_EINVAL, "rigl_masked_conv2d_dgrad: NULL tensor");
  if ((d->cin % 8) || (d->cout % 8)) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_dgrad: cin/cout %% 8 != 0 (use the reference kernel)");
  hipStream_t st = as_stream(stream);
  prof_set_tag(d);
  ProfFamily prof(PROF_CONV_DGRAD);
  if (addend_is_sub(addend, sub)) {
    if (bn) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_dgrad: a subsampled addend and the batch-norm reductions do not combine");
    IgemmArgs a = dgrad_args(d, dy, w_hwio, addend, dx);
    set_addend_sub(a, d, sub);
    launch_igemm<1, false>(a, st);
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
    return RIGL_OK;
  }
  // (the single-pass kernels first, in the order of bwd_impl: a layer's dX has the same bits from both entry points)
  if (!bn && bwd1x1_kind(d)) {
    // the big-M 1x1 layers: dX from the single-pass backward kernel (without its weight-gradient half), so that it has
    // the bits rigl_masked_conv2d_bwd gives it
    if (launch_bwd1x1(d, nullptr, dy, w_hwio, addend, dx, nullptr, st)) {
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
      return RIGL_OK;
    }
  }
  {
    BsPlan bp;
    if (!bn && bs_use(d, &bp)) {                // the many-input-channel 1x1 layers: the channel-sliced single-pass kernel, dX half only
      launch_bs(d, bp, nullptr, dy, w_hwio, addend, dx, nullptr, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
      return RIGL_OK;
    }
  }
  {
    RsPlan rp;
    if (!bn && rs_use<1>(d, &rp)) {             // dX[M][cin] = dY[M][cout] x W[cin][cout]^T, rows streamed (rowstream.hpp)
      launch_rs<1>(d, rp, dy, w_hwio, addend, dx, nullptr, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
      return RIGL_OK;
    }
  }
  if (!bn && c3x3_use(d)) {
    launch_c3x3<true>(d, dy, w_hwio, addend, dx, nullptr, st);
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
    return RIGL_OK;
  }
  IgemmArgs a = dgrad_args(d, dy, w_hwio, addend, dx);
  rc = attach_bn(a, d, bn);
  if (rc) return rc;
  const PPPlan pp = plan_pp<1>(a);
  if (pp.variant && launch_pp<1>(pp, a, st)) {
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
    return RIGL_OK;
  }
  launch_igemm<1, false>(a, st);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad");
  return RIGL_OK;
}

int rigl_masked_conv2d_dgrad(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio, rigl_bf16* dx,
                             void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  return rigl_masked_conv2d_dgrad_acc(d, dy, w_hwio, nullptr, dx, workspace, workspace_bytes, stream);
}

int rigl_masked_conv2d_wgrad(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, float* dw,
                             void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::k1;
  int rc = check_desc(d, "rigl_masked_conv2d_wgrad");
  if (rc) return rc;
  if (!x || !dy || !dw) return fail(RIGL_EINVAL, "rigl_masked_conv2d_wgrad: NULL tensor");
  if (d->cout % 8) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_wgrad: cout %% 8 != 0 (use the reference kernel)");
  hipStream_t st = as_stream(stream);
  const size_t need = rigl_conv2d_workspace_bytes(d, 2);
  if (need && (!workspace || workspace_bytes < need)) return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_wgrad: workspace %zu < %zu", workspace_bytes, need);
  prof_set_tag(d);
  ProfFamily prof(PROF_CONV_WGRAD);
  if (RIGL_TUNE("pp_wgrad", -1) > 0 && !tiny_cin(d) && !small_cin(d) && pp_wgrad_legal(d)) {
    const PPBwdPlan pw = plan_wgrad_pp(d, 0u, 0);
    const WgradArgs aw = pp_wgrad_args(d, x, dy, pw, dw, workspace);
    if (pp_wgrad_launch(pw, aw, st)) {
      if (pw.splits > 1) {
        ReduceArgs ra = {static_cast<const float*>(workspace), dw, pw.slab, pw.slab, pw.splits};
        launch_wgrad_reduce(ra, st);
      }
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad");
      return RIGL_OK;
    }
  }
  if (c3x3_use(d)) {
    // 3x3, 64 -> 64: X patch and dY tile resident in LDS, nine taps per wave (c3x3.hpp), one slab per workgroup
    float* slabs = static_cast<float*>(workspace);
    launch_c3x3_wgrad(d, x, dy, slabs, st);
    ReduceArgs ra = {slabs, dw, (int64_t)9 * 64 * 64, (int64_t)9 * 64 * 64, c3x3_wgrad_geom(d).grid};
    launch_wgrad_reduce(ra, st);
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad");
    return RIGL_OK;
  }
  if (tiny_cin(d) && stem_direct_legal(d) && RIGL_TUNE("stem_wgrad", 1) != 0) {
    // the ImageNet stem: patch and dY tile resident in LDS, both operands by transposing reads (stem.hpp)
    float* tmp = static_cast<float*>(workspace);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)7 * 32 * 64 * 4, 256));
    if (launch_stem_wgrad(d, x, dy, slabs, st)) {
      ReduceArgs ra = {slabs, tmp, (int64_t)7 * 32 * 64, (int64_t)7 * 32 * 64, stem_wgrad_grid(d)};
      launch_wgrad_reduce(ra, st);
      RIGL_K_LAUNCH(k_stem_unpack, dim3(64), dim3(THREADS), 0, st, tmp, dw, 7, 7, 3, 32, 64, 1);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad");
      return RIGL_OK;
    }
  }
  WgradArgs a = {};
  a.interleave = RIGL_TUNE("wgrad_il", 1);
  a.DY = dy; a.M = d->n * d->ho * d->wo; a.Cout = d->cout;
  a.dy_bytes = (uint32_t)((size_t)a.M * d->cout * 2);
  int64_t n_out;
  char* ws = static_cast<char*>(workspace);
  float* tiny_tmp = nullptr;          // [kh][cred][cout] reduced result of the tiny-Cin path
  TinyGeom tg = {0, 0, 0};
  if (tiny_cin(d)) {
    tg = tiny_geom(d);
    const size_t xp_bytes = (size_t)d->n * tg.hp * tg.wp * 4 * 2;
    uint16_t* xp = reinterpret_cast<uint16_t*>(ws);
    ws += align_up(xp_bytes, 256);
    PadArgs pa = {x, xp, d->n, d->h, d->w, d->cin, tg.hp, tg.wp, d->pad_top, d->pad_left};
    RIGL_K_LAUNCH(k_pad_input4, dim3(2048), dim3(THREADS), 0, st, pa);
    a.X = xp; a.Cin = tg.cred; a.x_pix_stride = 4; a.KH = d->kh; a.KW = 1; a.H = tg.hp; a.W = tg.wp;
    a.Ho = d->ho; a.Wo = d->wo; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = a.pw = 0;
    a.x_bytes = (uint32_t)xp_bytes;
    a.fold = tg.cred; a.Cin = d->kh * tg.cred; a.KH = 1;   // [kh][cred] is one channel axis
    n_out = (int64_t)d->kh * tg.cred * d->cout;
    tiny_tmp = reinterpret_cast<float*>(ws);
    ws += align_up((size_t)n_out * 4, 256);
  } else if (small_cin(d)) {
    const int K = d->kh * d->kw * d->cin, Kp = kpad(d);
    uint16_t* col = reinterpret_cast<uint16_t*>(ws);
    ws += align_up((size_t)a.M * Kp * 2, 256);
    Im2colArgs ia = {x, col, a.M, d->cin, d->kh, d->kw, d->h, d->w, d->ho, d->wo, d->stride_h, d->stride_w, d->pad_top, d->pad_left, K, Kp};
    RIGL_K_LAUNCH(k_im2col, dim3(2048), dim3(THREADS), 0, st, ia);
    a.X = col; a.Cin = Kp; a.x_pix_stride = Kp; a.KH = a.KW = 1; a.H = 1; a.W = a.M; a.Ho = 1; a.Wo = a.M;
    a.sh = a.sw = 1; a.ph = a.pw = 0;
    a.x_bytes = (uint32_t)((size_t)a.M * Kp * 2);
    n_out = (int64_t)K * d->cout;
  } else {
    a.X = x; a.Cin = d->cin; a.x_pix_stride = d->cin; a.KH = d->kh; a.KW = d->kw; a.H = d->h; a.W = d->w;
    a.Ho = d->ho; a.Wo = d->wo; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_top; a.pw = d->pad_left;
    a.x_bytes = (uint32_t)((size_t)d->n * d->h * d->w * d->cin * 2);
    n_out = (int64_t)d->kh * d->kw * d->cin * d->cout;
  }
  WgradPlan p = plan_wgrad(a.M, a.Cin, a.Cout, a.KH * a.KW);
  a.tiles_ci = p.tiles_ci; a.tiles_co = p.tiles_co; a.splits = p.splits; a.slab_elems = p.slab;
  const bool two_pass = p.splits > 1 || small_cin(d) || tiny_cin(d);
  a.OUT = two_pass ? reinterpret_cast<float*>(ws) : dw;
  dim3 grid((unsigned)((int64_t)p.tiles_ci * p.tiles_co * a.KH * a.KW * p.splits)), blk(THREADS);
  if (wgrad_stages(p.tm, p.tn) == 3) RIGL_K_LAUNCH((k_wgrad_tr<2, 2, 3>), grid, blk, 0, st, a);
  else if (p.tm == 2) RIGL_K_LAUNCH((k_wgrad_tr<2, 1, 4>), grid, blk, 0, st, a);
  else if (p.tn == 2) RIGL_K_LAUNCH((k_wgrad_tr<1, 2, 4>), grid, blk, 0, st, a);
  else RIGL_K_LAUNCH((k_wgrad_tr<1, 1, 4>), grid, blk, 0, st, a);
  if (two_pass) {
    ReduceArgs ra = {reinterpret_cast<const float*>(ws), tiny_tmp ? tiny_tmp : dw, n_out, p.slab, p.splits};
    launch_wgrad_reduce(ra, st);
  }
  if (tiny_tmp)
    RIGL_K_LAUNCH(k_stem_unpack, dim3(64), blk, 0, st, tiny_tmp, dw, d->kh, d->kw, d->cin, tg.cred, d->cout, 0);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad");
  return RIGL_OK;
}

static int bwd_impl(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                    const rigl_bf16* addend, float* dw, rigl_bf16* dx, void* workspace, size_t workspace_bytes,
                    const RiglBnReduceFuse* bn, rigl_stream_t stream, AddendSub sub = {1, 1},
                    const RiglConvDesc* dg = nullptr, const uint8_t* addend_bits = nullptr);

// rigl_masked_conv2d_bwd with the batch-norm backward reductions of the tensor dX is the gradient of riding in the dgrad
// epilogue.
int rigl_masked_conv2d_bwd_bn(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                              const rigl_bf16* addend, float* dw, rigl_bf16* dx, void* workspace, size_t workspace_bytes,
                              const RiglBnReduceFuse* bn, rigl_stream_t stream) {
  if (bn && !dx) return rigl::fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd_bn: the reductions ride on dX, which was not requested");
  return bwd_impl(d, x, dy, w_hwio, addend, dw, dx, workspace, workspace_bytes, bn, stream);
}

static int bwd_impl(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                    const rigl_bf16* addend, float* dw, rigl_bf16* dx, void* workspace, size_t workspace_bytes,
                    const RiglBnReduceFuse* bn, rigl_stream_t stream, AddendSub sub, const RiglConvDesc* dg,
                    const uint8_t* addend_bits) {
  // dg (rigl_masked_conv2d_bwd_grid): the dgrad half runs on THIS descriptor -- the stride-1 twin of a strided 1x1 conv
  // on its own output grid -- while the weight gradient reads x through d; the two halves of the shared launch take
  // their geometry from separate argument blocks anyway.
  using namespace rigl;
  using namespace rigl::k1;
  static const bool fuse = [] { const char* e = getenv("RIGL_BWD_FUSED"); return e ? atoi(e) != 0 : true; }();
  const RiglConvDesc* dd = dg ? dg : d;
  int rc = check_desc(d, "rigl_masked_conv2d_bwd");
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  prof_set_tag(d);
  const bool subadd = addend_is_sub(addend, sub);          // (only the igemm body's epilogue: the special paths step aside)
  if (subadd && bn) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_bwd: a subsampled addend and the batch-norm reductions do not combine");
  if (subadd && !dx) return fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd: an addend without dX");
  const bool both = dx && x && dy && w_hwio && dw;          // both gradients asked for, every operand there
  const bool whole = both && !subadd && !dg;                // ... and nothing that keeps the layer off its special kernels
  const size_t need = rigl_conv2d_workspace_bytes(d, 2);
  // A layer's dX must come from the same kernel whichever entry point computes it (rigl_masked_conv2d_dgrad or this
  // one): the ping-pong body accumulates in another order than the igemm body of the shared launch, so layers whose
  // dgrad has a ping-pong plan run wgrad and dgrad as two launches.
  bool dgrad_pp = false;
  if (dx && (d->cin % 8) == 0 && (d->cout % 8) == 0 && !bn && !subadd && !dg) {
    const IgemmArgs ap = dgrad_args(d, dy, w_hwio, addend, dx);
    dgrad_pp = plan_pp<1>(ap).variant != 0;
  }
  // The big-M 1x1 layers: dX and dW in ONE pass over dY (bwd1x1.hpp), one slab per workgroup, then the reduce
  if (whole && !bn && bwd1x1_kind(d) && bwd1x1_ready(d)) {
    if (need && (!workspace || workspace_bytes < need))
      return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);
    ProfFamily prof(PROF_CONV_BWD);
    if (launch_bwd1x1(d, x, dy, w_hwio, addend, dx, static_cast<float*>(workspace), st)) {
      const int64_t n_out = (int64_t)d->cin * d->cout;
      ReduceArgs ra = {static_cast<const float*>(workspace), dw, n_out, n_out, bwd1x1_splits()};
      launch_wgrad_reduce(ra, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
      return RIGL_OK;
    }
  }
  // The many-input-channel 1x1 layers: the same single pass per 128-channel slice (bwdslice.hpp), one slab per row group
  {
    BsPlan bp;
    // (a subsampled addend -- the first block of a group -- rides in the same DMA ring: rows without an addend row read zeros)
    if (both && !dg && !bn && bs_use(d, &bp)) {
      if (need && (!workspace || workspace_bytes < need))
        return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);
      ProfFamily prof(PROF_CONV_BWD);
      launch_bs(d, bp, x, dy, w_hwio, addend, dx, static_cast<float*>(workspace), st, sub.sh, sub.sw, addend_bits);
      const int64_t n_out = (int64_t)d->cin * d->cout;
      ReduceArgs ra = {static_cast<const float*>(workspace), dw, n_out, n_out, bp.G};
      launch_wgrad_reduce(ra, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
      return RIGL_OK;
    }
  }
  if (addend_bits) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_bwd_masked: this layer's kernels take no masked addend (ask rigl_conv2d_bwd_takes_masked_addend)");
  // Layers whose dgrad streams rows (rowstream.hpp): the weight gradient with its stand-alone plan, then the dgrad -- two
  // launches (+ reduce) instead of the shared one
  {
    RsPlan rp;
    if (whole && !bn && rs_use<1>(d, &rp)) {
      ProfFamily prof(PROF_CONV_BWD);
      rc = rigl_masked_conv2d_wgrad(d, x, dy, dw, workspace, workspace_bytes, stream);
      if (rc) return rc;
      prof_current_kind() = PROF_CONV_BWD;
      launch_rs<1>(d, rp, dy, w_hwio, addend, dx, nullptr, st);
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
      return RIGL_OK;
    }
  }
  // 3x3, 64 -> 64 (c3x3.hpp): weight gradient (+ reduce) and dgrad are launches of their own, each on resident patches
  if (whole && !bn && c3x3_use(d)) {
    if (need && (!workspace || workspace_bytes < need))
      return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);
    ProfFamily prof(PROF_CONV_BWD);
    float* slabs = static_cast<float*>(workspace);
    launch_c3x3_wgrad(d, x, dy, slabs, st);
    ReduceArgs ra = {slabs, dw, (int64_t)9 * 64 * 64, (int64_t)9 * 64 * 64, c3x3_wgrad_geom(d).grid};
    launch_wgrad_reduce(ra, st);
    launch_c3x3<true>(d, dy, w_hwio, addend, dx, nullptr, st);
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
    return RIGL_OK;
  }
  // The shared launch on the 8-wave ping-pong bodies ("pp_bwd"): layers whose weight gradient has 256-channel tiles and
  // whose dgrad is a long reduction.
  if (RIGL_TUNE("pp_bwd", -1) != 0 && both && !subadd && !bn && !tiny_cin(d) && !small_cin(d) && pp_wgrad_legal(d)) {
    IgemmArgs ad = dgrad_args(dd, dy, w_hwio, addend, dx);
    const int dvar = RIGL_TUNE("pp_dgrad", -1) >= 0 ? PP_NONE : pp_bwd_dgrad_variant(ad);   // (a forced stand-alone dgrad tile wins)
    if (dvar != PP_NONE && pp_legal<1>(ad, dvar)) {
      int bm, bn2;
      pp_dims(dvar, bm, bn2);
      const bool strided = pp_strided(ad);
      const int tiles_m = strided ? pp_fill_classes(ad, bm) : (ad.M + bm - 1) / bm;
      const unsigned nd = (unsigned)(tiles_m * (ad.N / bn2));
      // dgrad workgroup length in 256x256-tile K-tile units (a parity class visits about taps / (sh * sw) of the taps)
      const int kt_d = (int)((int64_t)dd->kh * dd->kw * (dd->cout / 64) * bm * bn2 / (256 * 256) / (dd->stride_h * dd->stride_w));
      const PPBwdPlan pw = plan_wgrad_pp(d, nd, kt_d);
      if (need && (!workspace || workspace_bytes < need))
        return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);
      ProfFamily prof(PROF_CONV_BWD);
      const WgradArgs aw = pp_wgrad_args(d, x, dy, pw, dw, workspace);
      ad.fd_rw = make_fastdiv(ad.RW); ad.fd_rh = make_fastdiv(ad.RH);
      ad.tiles_n = ad.N / bn2;
      if (pp_bwd_launch(dvar, strided, ad, aw, pw, st)) {
        if (pw.splits > 1) {
          ReduceArgs ra = {static_cast<const float*>(workspace), dw, pw.slab, pw.slab, pw.splits};
          launch_wgrad_reduce(ra, st);
        }
        RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
        return RIGL_OK;
      }
    }
  }
  if (fuse && !dgrad_pp && both && !tiny_cin(d) && !small_cin(d) && (d->cin % 8) == 0 && (d->cout % 8) == 0) {
    IgemmArgs ad = dgrad_args(dd, dy, w_hwio, addend, dx);
    if (subadd) set_addend_sub(ad, dd, sub);
    rc = attach_bn(ad, d, bn);
    if (rc) return rc;
    const IgemmPlan pd = plan_igemm<1>(ad);
    WgradArgs aw = {};
    aw.interleave = RIGL_TUNE("wgrad_il", 1);
    aw.DY = dy; aw.M = d->n * d->ho * d->wo; aw.Cout = d->cout;
    aw.dy_bytes = (uint32_t)((size_t)aw.M * d->cout * 2);
    aw.X = x; aw.Cin = d->cin; aw.x_pix_stride = d->cin; aw.KH = d->kh; aw.KW = d->kw; aw.H = d->h; aw.W = d->w;
    aw.Ho = d->ho; aw.Wo = d->wo; aw.sh = d->stride_h; aw.sw = d->stride_w; aw.ph = d->pad_top; aw.pw = d->pad_left;
    aw.x_bytes = (uint32_t)((size_t)d->n * d->h * d->w * d->cin * 2);
    const WgradPlan p = plan_wgrad(aw.M, aw.Cin, aw.Cout, aw.KH * aw.KW, true);
    // (launched alone back to back, a large short-reduction dgrad -- the 56x56 / 28x28 1x1 "reduce" convs -- is 9-28 %
    // slower when it shares the launch, the other layers 3-10 % faster; inside the training step sharing always won)
    if (pd.dma) {
      if (need && (!workspace || workspace_bytes < need))
        return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_bwd: workspace %zu < %zu", workspace_bytes, need);
      ProfFamily prof(PROF_CONV_BWD);
      aw.tiles_ci = p.tiles_ci; aw.tiles_co = p.tiles_co; aw.splits = p.splits; aw.slab_elems = p.slab;
      const bool two_pass = p.splits > 1;
      aw.OUT = two_pass ? static_cast<float*>(workspace) : dw;
      const unsigned nd = pd.grid, nw = (unsigned)((int64_t)p.tiles_ci * p.tiles_co * aw.KH * aw.KW * p.splits);
      const dim3 grid(nd + nw), blk(THREADS);
#define RIGL_FUSED(TND, CLSD)                                                                                          \
      {                                                                                                                \
        if (p.tm == 2 && p.tn == 2) RIGL_K_LAUNCH((k_bwd_fused<TND, CLSD, 2, 2, 3>), grid, blk, 0, st, ad, aw, nd, nw); \
        else if (p.tm == 2) RIGL_K_LAUNCH((k_bwd_fused<TND, CLSD, 2, 1, 4>), grid, blk, 0, st, ad, aw, nd, nw);          \
        else if (p.tn == 2) RIGL_K_LAUNCH((k_bwd_fused<TND, CLSD, 1, 2, 4>), grid, blk, 0, st, ad, aw, nd, nw);          \
        else RIGL_K_LAUNCH((k_bwd_fused<TND, CLSD, 1, 1, 4>), grid, blk, 0, st, ad, aw, nd, nw);                         \
      }
      if (pd.wide_n) { if (pd.cls) RIGL_FUSED(2, true) else RIGL_FUSED(2, false) }
      else { if (pd.cls) RIGL_FUSED(1, true) else RIGL_FUSED(1, false) }
#undef RIGL_FUSED
      if (two_pass) {
        const int64_t n_out = (int64_t)d->kh * d->kw * d->cin * d->cout;
        ReduceArgs ra = {static_cast<const float*>(workspace), dw, n_out, p.slab, p.splits};
        launch_wgrad_reduce(ra, st);
      }
      RIGL_CHECK_LAUNCH("rigl_masked_conv2d_bwd");
      return RIGL_OK;
    }
  }
  rc = rigl_masked_conv2d_wgrad(d, x, dy, dw, workspace, workspace_bytes, stream);
  if (rc || !dx) return rc;
  return dgrad_impl(dd, dy, w_hwio, addend, dx, bn, stream, sub);
}

// Whole backward of one masked conv in one call: dW (dense) and, when dx is given, dX (+ addend).
// Ordinary layers run both GEMMs in ONE launch (k_bwd_fused / k_bwd_pp: the longer-running workgroups first) followed by
// the split-K reduce; the tiny-/small-Cin paths (extra repack kernels, no dX for the stem) fall back to the two
// separate launches.
int rigl_masked_conv2d_bwd(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                           const rigl_bf16* addend, float* dw, rigl_bf16* dx, void* workspace, size_t workspace_bytes,
                           rigl_stream_t stream) {
  return bwd_impl(d, x, dy, w_hwio, addend, dw, dx, workspace, workspace_bytes, nullptr, stream);
}

// rigl_masked_conv2d_bwd whose addend arrives UNMASKED together with a 1-bit-per-element mask (bit set = the addend counts):
// the gradient of relu(bn3 + shortcut) w.r.t. the shortcut is the block output's gradient where the ReLU was on
// (resnet_model.py:497-501), so the batch norm's backward need not write that masked copy -- its consumer, the dgrad
// epilogue of the block's first conv, masks on the fly.  Only the layers of rigl_conv2d_bwd_takes_masked_addend.
int32_t rigl_conv2d_bwd_takes_masked_addend(const RiglConvDesc* d) {
  using namespace rigl;
  using namespace rigl::k1;
  if (!d || check_desc(d, "rigl_conv2d_bwd_takes_masked_addend")) return 0;
  return bs_use(d) ? 1 : 0;
}
int rigl_masked_conv2d_bwd_masked(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                  const rigl_bf16* addend, const uint8_t* addend_bits, float* dw, rigl_bf16* dx,
                                  void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  if (!addend || !addend_bits || !dx) return rigl::fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd_masked: addend, addend_bits and dx are required");
  return bwd_impl(d, x, dy, w_hwio, addend, dw, dx, workspace, workspace_bytes, nullptr, stream, AddendSub{1, 1}, nullptr, addend_bits);
}

// rigl_masked_conv2d_bwd whose addend is the gradient of a SUBSAMPLED view of the conv's input: [n][ceil(h / sub_h)][ceil(w /
// sub_w)][cin], added at the pixels with h % sub_h == 0 and w % sub_w == 0 (the first block of a ResNet group: the block
// input feeds conv1 and a strided 1x1 projection, whose input gradient exists only at every sub-th pixel and is handed over
// compact instead of as a full-size tensor that is three quarters zeros).
// Backward of a STRIDED 1x1 conv without padding whose dX is wanted on the conv's own grid only (the pixels it read):
// dx_grid[n][ho][wo][cin] = the dgrad of the stride-1 1x1 conv over that grid (its consumer: rigl_masked_conv2d_bwd_sub of
// the other reader of the tensor), dW from x as it lies -- one shared launch, the dgrad half on the compact descriptor.
int rigl_masked_conv2d_bwd_grid(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                float* dw, rigl_bf16* dx_grid, void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::k1;
  int rc = check_desc(d, "rigl_masked_conv2d_bwd_grid");
  if (rc) return rc;
  if (d->kh != 1 || d->kw != 1 || d->pad_top || d->pad_left || d->ho != (d->h + d->stride_h - 1) / d->stride_h ||
      d->wo != (d->w + d->stride_w - 1) / d->stride_w)
    return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_bwd_grid: a 1x1 conv without padding whose output grid is every stride-th pixel");
  if (!x || !dy || !w_hwio || !dw || !dx_grid) return fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd_grid: NULL tensor");
  if ((d->cin % 8) || (d->cout % 8)) return fail(RIGL_EUNSUPPORTED, "rigl_masked_conv2d_bwd_grid: cin/cout %% 8 != 0");
  RiglConvDesc g = *d;
  g.h = d->ho; g.w = d->wo; g.stride_h = g.stride_w = 1;
  return bwd_impl(d, x, dy, w_hwio, nullptr, dw, dx_grid, workspace, workspace_bytes, nullptr, stream, AddendSub{1, 1}, &g);
}

int rigl_masked_conv2d_bwd_sub(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const rigl_bf16* w_hwio,
                               const rigl_bf16* addend, int32_t sub_h, int32_t sub_w, float* dw, rigl_bf16* dx,
                               void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  if (sub_h < 1 || sub_w < 1) return rigl::fail(RIGL_EINVAL, "rigl_masked_conv2d_bwd_sub: subsampling factors must be >= 1");
  return bwd_impl(d, x, dy, w_hwio, addend, dw, dx, workspace, workspace_bytes, nullptr, stream, AddendSub{sub_h, sub_w});
}


// Dense bf16 MFMA probe: enqueues blocks x 256 threads x iters x 8 MFMAs (2*32*32*16 FLOP each per wave); the caller
// times it with events and divides.  `sink` needs blocks*256 floats (never written in practice).
int rigl_probe_mfma_bf16(int32_t blocks, int32_t iters, float* sink, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::k1;
  if (blocks <= 0 || iters <= 0 || !sink) return fail(RIGL_EINVAL, "rigl_probe_mfma_bf16: bad argument");
  hipLaunchKernelGGL(k_mfma_probe, dim3((unsigned)blocks), dim3(THREADS), 0, as_stream(stream), sink, (int)iters, 0.5f);
  RIGL_CHECK_LAUNCH("rigl_probe_mfma_bf16");
  return RIGL_OK;
}

}  // extern "C"
