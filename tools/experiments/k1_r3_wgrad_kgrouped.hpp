// ARCHIVED EXPERIMENT (round 3) -- not compiled into librigl_hip.so.
// K-grouped weight gradient for the ping-pong skeleton (convpp.hpp / convpp_loop.inc): 128 x 128 channel tiles, the 8 waves
// = 2 x 2 wave tiles of 64 x 64 x TWO pixel groups whose partial tiles meet in LDS, so a workgroup writes a quarter of the
// split-K slab bytes of the 256 x 256 body.  Bit-exact against the fp64 reference on all 23 ResNet-50 shapes at batch 128
// (tests/k1_check.py with RIGL_PP_WK=1 at the time), deterministic, and SLOWER than the 256 x 256 body where both apply:
//   stand-alone weight gradient, batch 128:  14x14x256 3x3  62.2 vs 57.0 us,  7x7x512 3x3  85.5 vs 55.4 us
//   shared backward launch:                   14x14x256 3x3 105.7 vs 75.7 us,  7x7x512 3x3  94.2 vs 70.9 us
//   ResNet-50 step with it as the default weight-gradient body: 12.60 ms vs 12.35 ms
// (gpurun_out/r3f, r3g: tools/pp_sweep.py --passes bwd wgrad).  The slab bytes were not what ended those workgroups: 12
// MFMAs per phase and 24 eight-byte transposing reads per wave per K-tile lose more MFMA rate than the slabs cost.
// To revive: paste between pp_wgrad_body and pp_reduce_body, give k_bwd_pp a template switch for the body, and let the
// host plan tiles of 128 channels / K-tiles of 96 pixels (git history: commit "convpp: K-grouped weight gradient").
// ---- weight gradient, K-grouped: 128 x 128 channel tiles ------------------------------------------------------------------
// Every split of a weight gradient costs one more fp32 copy of dW written and read back, and at batch 128 the 256x256
// tiles above need 17-39 splits to fill the chip (40 MB of slabs per layer: the slab stores, not the MFMAs, end those
// workgroups).  Here the 8 waves are 2 x 2 wave tiles of 64x64 channels x TWO pixel groups: each group multiplies its own
// 48 pixels of every 96-pixel K-tile, the two 64x64 partials of a wave pair meet in LDS in a fixed order at the end, and the
// workgroup writes ONE 128x128 tile -- a quarter of the splits (and of the slab bytes) for the same number of workgroups,
// and layers with 128 channels (28x28 3x3) get a ping-pong weight gradient at all.  Wave group = pixel group, so the
// ping-pong partner of a wave is the wave that accumulates the other half of the same output tile.  Skeleton: PH = 1 (the
// whole K-tile per phase: 12 MFMAs per wave), three 48 KB stages; stage = [96 pixels][128 channels] of X, then of dY.
struct PPKCursor { int kt; int pa[3], pb[3]; };

__device__ __forceinline__ void pp_wgrad_k_body(const WgradArgs& P, unsigned char* const smem, uint32_t bid, uint32_t nblk) {
  constexpr int PH = 1, QM = 1, KS = 3, PXT = 96, PXG = 48;
  constexpr int PART = PXT * 256, STAGE = 2 * PART;       // 24 KB of X + 24 KB of dY
  static_assert(3 * STAGE <= 160 * 1024, "three stages");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, gk = grp, gm = (wave >> 1) & 1, gn = wave & 1;
  uint32_t b = xcd_remap(bid, nblk);
  const int tco = (int)(b % (uint32_t)P.tiles_co); b /= (uint32_t)P.tiles_co;
  const int tci = (int)(b % (uint32_t)P.tiles_ci); b /= (uint32_t)P.tiles_ci;
  const int taps = P.KH * P.KW;
  const int tap = (int)(b % (uint32_t)taps), split = (int)(b / (uint32_t)taps);
  const int r = tap / P.KW, s = tap - r * P.KW;
  const int ci0 = tci * 128, co0 = tco * 128;
  const int KT_all = (P.M + PXT - 1) / PXT;
  const int kt_begin = (int)((int64_t)KT_all * split / P.splits);
  const int KT = (int)((int64_t)KT_all * (split + 1) / P.splits) - kt_begin;

  // DMA lanes: instruction jj (0..2) of wave w fills pixel rows 4 * (jj*8 + w) .. +3 of a part; lane l: pixel +(l >> 4),
  // slot l & 15 fetching channel chunk slot ^ ((pixel & 3) << 2)
  const int px_lane = 4 * wave + (lane >> 4);                       // + 32 * jj
  const int chunk = (lane & 15) ^ (((lane >> 4) & 3) << 2);
  const int chan_a = ci0 + chunk * 8, chan_b = co0 + chunk * 8;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.X, P.x_bytes), rsrcB = make_rsrc(P.DY, P.dy_bytes);
  const int hi0 = r - P.ph, wi0 = s - P.pw;

#define PP_CURSOR_T PPKCursor
#define PP_WCOMPUTE(c_)                                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int jj = 0; jj < 3; ++jj) {                                                   \
      const int p_ = (kt_begin + (c_).kt) * PXT + 32 * jj + px_lane;                                     \
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
  // the skeleton's four pieces: "A0" = all of X's part, "B0" = all of dY's, "A1" / "B1" empty
#define PP_ISSUE_A(h_, stage_, c_)                                                                       \
  if ((h_) == 0) {                                                                                       \
    _Pragma("unroll") for (int jj = 0; jj < 3; ++jj) {                                                   \
      const int off_ = (c_).pa[jj] >= 0 ? (int)((uint32_t)((c_).pa[jj] + chan_a) * 2u) : (int)OOB;       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcA, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + (jj * 8 + wave) * 1024), 16, off_, 0, 0, 0); \
    }                                                                                                    \
  }
#define PP_ISSUE_B(h_, stage_, c_)                                                                       \
  if ((h_) == 0) {                                                                                       \
    _Pragma("unroll") for (int jj = 0; jj < 3; ++jj) {                                                   \
      const int off_ = (c_).pb[jj] >= 0 ? (int)((uint32_t)((c_).pb[jj] + chan_b) * 2u) : (int)OOB;       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
          rsrcB, (__attribute__((address_space(3))) void*)(smem + (stage_) * STAGE + PART + (jj * 8 + wave) * 1024), 16, off_, 0, 0, 0); \
    }                                                                                                    \
  }

  const int g = lane >> 4, j16 = lane & 15, rs = (j16 >> 2) & 3;
  const int tr_row = (gk * PXG + 8 * (g >> 1) + (j16 >> 2)) * 256, tr_low = ((2 * (g & 1) + ((j16 >> 1) & 1)) << 4) + (j16 & 1) * 8;
  int a_tr[2], b_tr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_tr[i] = tr_row + ((((gm * 2 + i) ^ rs) * 4) << 4) + tr_low;
    b_tr[i] = PART + tr_row + ((((gn * 2 + i) ^ rs) * 4) << 4) + tr_low;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  bf16x8 af[2][KS], b0[1][KS], b1[1][KS];

#define PP_READ_A(h_, stage_, ab_)                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                      \
    af[ab_][ks] = lds_read_tr_pair(smem + (stage_) * STAGE + a_tr[h_] + ks * 4096, smem + (stage_) * STAGE + a_tr[h_] + ks * 4096 + 1024);
#define PP_READ_B(dst_, h_, stage_)                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                      \
    dst_[0][ks] = lds_read_tr_pair(smem + (stage_) * STAGE + b_tr[h_] + ks * 4096, smem + (stage_) * STAGE + b_tr[h_] + ks * 4096 + 1024);
#define PP_QUAD(ha_, hb_, bsrc_, ab_)                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                      \
    acc[ha_][hb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ab_][ks], bsrc_[0][ks], acc[ha_][hb_], 0, 0, 0);
  constexpr int W4 = 6, W2 = 3, W1 = 3;                   // six DMA instructions per thread per K-tile
  (void)W2; (void)W1;
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

  // the two pixel groups' partials of every 64x64 wave tile meet in LDS: group 1 parks its accumulators ([value][lane]:
  // conflict-free), group 0 adds them to its own -- always (group 0) + (group 1) -- and stores the tile
  float* const xch = reinterpret_cast<float*>(smem) + (gm * 2 + gn) * 64 * 64;
  if (gk == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[((i * 2 + jj) * 16 + e) * 64 + lane] = acc[i][jj][e];
  }
  __syncthreads();
  if (gk == 0) {
    float* const out = P.OUT + (int64_t)split * P.slab_elems + (int64_t)tap * P.Cin * P.Cout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + gm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int co = co0 + gn * 64 + jj * 32 + (lane & 31);
          out[(int64_t)ci * P.Cout + co] = acc[i][jj][e] + xch[((i * 2 + jj) * 16 + e) * 64 + lane];
        }
  }
}

__global__ __launch_bounds__(512) void k_wgrad_ppk(WgradArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pp[];
  pp_wgrad_k_body(P, smem_pp, blockIdx.x, gridDim.x);
}
constexpr int PPK_SMEM = 3 * 2 * 96 * 256;

// One launch per layer backward: the split-K weight-gradient workgroups and the dgrad tiles of a layer share one grid -- two independent GEMMs that each leave CUs idle at batch 128 share the chip (the k_bwd_fused idea
// on the 8-wave bodies).  dX is bit-identical to k_igemm_pp<..., 1> launched alone.
