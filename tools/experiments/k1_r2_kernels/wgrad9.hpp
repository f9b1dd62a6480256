// K1, weight gradient of a 3x3 stride-1 SAME convolution with ALL NINE TAPS in one workgroup ("wgrad9").
// Included by conv.hip inside namespace rigl::k1.
//
//   dW[r][s][ci][co] = sum_p  X[p + (r-1) W + (s-1)][ci] * dY[p][co]        (p = flat NHW pixel, inside the image)
//
// The per-tap kernel (k_wgrad_tr) gives every tap its own workgroups: X crosses L2 -> LDS nine times, and a
// 128x128x32-pixel K-tile moves 16 KB per 1 MFLOP -- the loader, not the matrix core, sets its pace.  Here a
// workgroup owns a 64 (ci) x 64 (co) tile of ALL nine taps over a range of pixels: per 32-pixel K-tile it fetches
// 4 KB of dY and 4 KB of X ONCE and issues 9 taps x 2 K-steps x 4 waves = 72 MFMAs on them (290 FLOP per loaded
// byte instead of 64).  X lives in a circular LDS ring of 256 pixel rows ([pixel][64 channels], exactly what
// `buffer_load ... lds` writes); a tap reads it through ds_read_b64_tr_b16 at the pixel offset of the tap, so the
// nine operands are nine views of the same bytes.  Pixels a tap must not see (image border / the neighbouring
// image) are redirected per lane to a zero row by a 9-bit mask derived from the pixel's (h, w).
//
// Decomposition: (Cin/64) x (Cout/64) tiles x `splits` pixel ranges, splits chosen so that the grid is one
// workgroup per CU (every 3x3 of ResNet-50 at batch 128: 256 workgroups x 49 K-tiles -- perfectly balanced).
// 4 waves (2 x 2 over ci x co), 9 accumulator tiles of 32x32 per wave (144 VGPRs); one wave per SIMD, so the loop
// is a software pipeline across the barrier like tile196's: every MFMA batch (9 MFMAs) has its fragments
// requested a batch earlier.  Partial slabs [split][9][Cin][Cout] fp32 + the deterministic fixed-order reduce
// of conv.hip complete dW (HWIO order).
// The lane-level index arithmetic is restated and checked on the CPU in tools/emu/w9_emu.py.

struct W9Args {
  const uint16_t* X;    // [M][Cin] bf16 (NHWC, stride 1, SAME: M = N*H*W pixels for both operands)
  const uint16_t* DY;   // [M][Cout]
  float* OUT;           // [splits][9][Cin][Cout]
  int M, Cin, Cout, H, W;
  int tiles_ci, tiles_co, splits, kt_per_split;   // K-tiles (32 pixels) per split
  int hb;               // halo in 32-pixel chunks: ceil((W + 1) / 32)
  int64_t slab_elems;
  uint32_t x_bytes, dy_bytes;
  FastDiv fd_w, fd_h;
};

constexpr int W9_XROWS = 256, W9_XRING_B = W9_XROWS * 128, W9_DY_ST = 32 * 128;
constexpr int W9_OFF_DY = W9_XRING_B, W9_OFF_ZERO = W9_OFF_DY + 3 * W9_DY_ST, W9_SMEM = W9_OFF_ZERO + 64;

__global__ __launch_bounds__(THREADS) void k_wgrad9(W9Args P) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[W9_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  uint32_t b = xcd_remap(blockIdx.x, gridDim.x);
  const int tco = (int)(b % (uint32_t)P.tiles_co); b /= (uint32_t)P.tiles_co;
  const int tci = (int)(b % (uint32_t)P.tiles_ci);
  const int split = (int)(b / (uint32_t)P.tiles_ci);
  const int ci0 = tci * 64, co0 = tco * 64;
  const int p_begin = split * P.kt_per_split * 32;
  int p_end = p_begin + P.kt_per_split * 32;
  if (p_end > P.M) p_end = P.M;
  const int KT = p_end > p_begin ? (p_end - p_begin + 31) >> 5 : 0;
  const int HB = P.hb;
  const int xb0 = p_begin - 32 * HB;          // X pixel held by ring row 0 of chunk 0
  const __amdgpu_buffer_rsrc_t rsrcX = make_rsrc(P.X, P.x_bytes), rsrcY = make_rsrc(P.DY, P.dy_bytes);

  // ---- DMA lanes: one wave instruction = 8 pixel rows x 128 B; lane -> (row l8, 16-B slot) ----------------------
  const int l8 = lane >> 3, slot = lane & 7;
  const int dsw = ((l8 >> 1) & 1) << 2;       // trswz of the row (rows 8-aligned per instruction: only l8 matters)
  const int row_w = wave * 8 + l8;            // row inside a 32-row chunk / tile
  // (element offsets; pixel validity is checked per chunk)
  const int x_lane = row_w * P.Cin + ci0 + ((slot ^ dsw) << 3);
  const int y_lane = row_w * P.Cout + co0 + ((slot ^ dsw) << 3);
#define W9_ISSUE_X(c_)                                                                                 \
  {                                                                                                    \
    const int pix_ = xb0 + (c_) * 32 + row_w;                                                          \
    const uint32_t off_ = ((unsigned)pix_ < (unsigned)P.M)                                             \
                              ? (uint32_t)((xb0 + (c_) * 32) * P.Cin + x_lane) * 2u : OOB;             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
        rsrcX, (__attribute__((address_space(3))) void*)(smem + ((((c_) * 32 + wave * 8) & (W9_XROWS - 1)) * 128)), 16, \
        (int)off_, 0, 0, 0);                                                                           \
  }
#define W9_ISSUE_Y(t_, st_)                                                                            \
  {                                                                                                    \
    const int pix_ = p_begin + (t_) * 32 + row_w;                                                      \
    const uint32_t off_ = (pix_ < p_end) ? (uint32_t)((p_begin + (t_) * 32) * P.Cout + y_lane) * 2u : OOB; \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                          \
        rsrcY, (__attribute__((address_space(3))) void*)(smem + W9_OFF_DY + (st_) * W9_DY_ST + wave * 1024), 16, \
        (int)off_, 0, 0, 0);                                                                           \
  }

  // ---- tr-read lanes -------------------------------------------------------------------------------------------
  const int g = lane >> 4, j = lane & 15;
  const int prow = 8 * (g >> 1) + (j >> 2);                    // pixel this lane supplies inside a 16-pixel K-step
  const int sub = (j & 1) * 8;
  const int chunkA = wm * 4 + 2 * (g & 1) + ((j >> 1) & 1);    // 16-B slot of the lane's 4 channels (ci side)
  const int chunkB = wn * 4 + 2 * (g & 1) + ((j >> 1) & 1);
  const int b_rd = prow * 128 + ((chunkB ^ (((prow >> 1) & 1) << 2)) << 4) + sub;
  if (tid < 4) *reinterpret_cast<uint4*>(smem + W9_OFF_ZERO + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  struct Frag { bf16x8 b; bf16x8 a[9]; };
  // 9-bit validity of the taps for output pixel p (flat): bit r*3+s set iff (h + r - 1, w + s - 1) lies inside the image
#define W9_MASK(p_, out_)                                                                              \
  {                                                                                                    \
    const int t_ = fdiv((p_), P.fd_w), w_ = (p_) - t_ * P.W, n_ = fdiv(t_, P.fd_h), h_ = t_ - n_ * P.H; \
    uint32_t m_ = 0x1FFu;                                                                              \
    if (h_ == 0) m_ &= ~0x007u;                                                                        \
    if (h_ == P.H - 1) m_ &= ~0x1C0u;                                                                  \
    if (w_ == 0) m_ &= ~0x049u;                                                                        \
    if (w_ == P.W - 1) m_ &= ~0x124u;                                                                  \
    out_ = m_;                                                                                         \
  }
  // fragments of K-step ks_ of tile (t_ in stage st_): pixels pl_ = ks_*16 + prow (+4)
#define W9_READ(F_, t_, st_, ks_, mlo_, mhi_)                                                          \
  {                                                                                                    \
    const unsigned char* Bs_ = smem + W9_OFF_DY + (st_) * W9_DY_ST + (ks_) * 2048;                     \
    F_.b = lds_read_tr_pair(Bs_ + b_rd, Bs_ + b_rd + 512);                                             \
    const int q0_ = (t_) * 32 + 32 * HB + (ks_) * 16 + prow;      /* ring row of the lane's pixel, before the tap shift */ \
    _Pragma("unroll") for (int tp = 0; tp < 9; ++tp) {                                                 \
      const int sh_ = (tp / 3 - 1) * P.W + (tp % 3 - 1);                                               \
      const int r0_ = (q0_ + sh_) & (W9_XROWS - 1), r1_ = (q0_ + 4 + sh_) & (W9_XROWS - 1);            \
      const int a0_ = r0_ * 128 + ((chunkA ^ (((r0_ >> 1) & 1) << 2)) << 4) + sub;                     \
      const int a1_ = r1_ * 128 + ((chunkA ^ (((r1_ >> 1) & 1) << 2)) << 4) + sub;                     \
      const unsigned char* p0_ = smem + ((((mlo_) >> tp) & 1u) ? a0_ : W9_OFF_ZERO + sub);             \
      const unsigned char* p1_ = smem + ((((mhi_) >> tp) & 1u) ? a1_ : W9_OFF_ZERO + sub);             \
      F_.a[tp] = lds_read_tr_pair(p0_, p1_);                                                           \
    }                                                                                                  \
  }
#define W9_MFMA(F_)                                                                                    \
  {                                                                                                    \
    _Pragma("unroll") for (int tp = 0; tp < 9; ++tp)                                                   \
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F_.a[tp], F_.b, acc[tp], 0, 0, 0);             \
  }

  if (KT > 0) {
    // prologue: the 2*HB halo chunks of X, then the pairs (dY tile j, X chunk j + 2 HB) for j = 0, 1, 2
    for (int c = 0; c < 2 * HB; ++c) W9_ISSUE_X(c);
    W9_ISSUE_Y(0, 0); W9_ISSUE_X(2 * HB);
    if (KT > 1) { W9_ISSUE_Y(1, 1); W9_ISSUE_X(2 * HB + 1); }
    if (KT > 2) { W9_ISSUE_Y(2, 2); W9_ISSUE_X(2 * HB + 2); }
    if (KT >= 3) wait_vmcnt<4>(); else if (KT == 2) wait_vmcnt<2>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    uint32_t m00, m01, m10, m11;             // masks of the lane's four pixels of a tile: K-step 0 / 1, row prow / prow + 4
    W9_MASK(p_begin + prow, m00); W9_MASK(p_begin + prow + 4, m01);
    W9_MASK(p_begin + 16 + prow, m10); W9_MASK(p_begin + 20 + prow, m11);
    Frag F0, F1;
    int st = 0, kt = 0;
    W9_READ(F0, 0, 0, 0, m00, m01);
    for (; kt + 1 < KT; ++kt) {
      W9_READ(F1, kt, st, 1, m10, m11);
      W9_MFMA(F0);
      // ---- between the batches: tile kt is in registers ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (kt + 2 < KT) wait_vmcnt<2>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + 3 < KT) { W9_ISSUE_Y(kt + 3, st); W9_ISSUE_X(kt + 3 + 2 * HB); }
      st = st == 2 ? 0 : st + 1;
      const int pn = p_begin + (kt + 1) * 32 + prow;
      W9_MASK(pn, m00); W9_MASK(pn + 4, m01); W9_MASK(pn + 16, m10); W9_MASK(pn + 20, m11);
      W9_READ(F0, kt + 1, st, 0, m00, m01);
      W9_MFMA(F1);
    }
    W9_READ(F1, kt, st, 1, m10, m11);
    W9_MFMA(F0);
    W9_MFMA(F1);
  }
#undef W9_ISSUE_X
#undef W9_ISSUE_Y
#undef W9_MASK
#undef W9_READ
#undef W9_MFMA

  // ---- partial slab: D[ci][co], lane = co column, register e = ci row (e&3) + 8 (e>>2) + 4 (lane>>5) -----------
  float* out = P.OUT + (int64_t)split * P.slab_elems;
  const int co = co0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = ci0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      out[((int64_t)tp * P.Cin + ci) * P.Cout + co] = acc[tp][e];
    }
}

// RIGL_W9: 0 = never (default), 1 = wherever the shape is legal.  Experimental: this first version recomputes swizzled,
// predicated addresses for every tap (~200 VALU per 9 MFMAs at one wave per SIMD) and measures 77 us on every ResNet-50
// 3x3 against 58-87 us for the per-tap kernel (profiles/r2/README.md); the padded-entry layout of conv3x3.hpp is the fix.
static int w9_mode() {
  static const int v = [] { const char* e = getenv("RIGL_W9"); return e ? atoi(e) : 0; }();
  return v;
}
struct W9Plan { bool use; int tiles_ci, tiles_co, splits, kt_per_split, hb; int64_t slab; };
static W9Plan plan_w9(const RiglConvDesc* d) {
  W9Plan p = {false, 0, 0, 1, 0, 1, 0};
  if (w9_mode() <= 0) return p;
  if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_top != 1 || d->pad_left != 1 ||
      d->ho != d->h || d->wo != d->w)
    return p;
  if ((d->cin & 63) || (d->cout & 63) || d->w + 1 > 64) return p;
  const int64_t M = (int64_t)d->n * d->h * d->w;
  if (M > 0x3fffffff) return p;
  p.tiles_ci = d->cin / 64; p.tiles_co = d->cout / 64;
  const int64_t tiles = (int64_t)p.tiles_ci * p.tiles_co;
  const int64_t kt_all = (M + 31) / 32;
  // one workgroup per CU: the accumulators of a workgroup are 147 KB of fp32, every extra split writes and re-reads
  // that much, and one wave per SIMD already keeps the matrix core of its CU busy
  static const int target = [] { const char* e = getenv("RIGL_W9_WGS"); return e ? atoi(e) : 0; }();
  int64_t splits = (target > 0 ? target : num_cus()) / tiles;
  if (splits < 1) splits = 1;
  if (splits > kt_all) splits = kt_all;
  p.kt_per_split = (int)((kt_all + splits - 1) / splits);
  p.splits = (int)((kt_all + p.kt_per_split - 1) / p.kt_per_split);
  p.hb = (d->w + 1 + 31) / 32;
  p.slab = (int64_t)9 * d->cin * d->cout;
  p.use = true;
  return p;
}
