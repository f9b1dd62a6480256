// K1, the ImageNet stem (7x7 stride-2 conv of a 3-channel image; included inside namespace rigl::k1 of conv.hip).
//
// Through the generic bodies the stem is a 7x1 conv over a zero-bordered 4-channel copy of the image ("tiny-Cin path"
// above): every output pixel's seven 64-byte filter-row windows are fetched from L2 into LDS, 448 B per pixel x 1.6 M
// pixels = 0.72 GB of L2 -> LDS traffic for a layer whose HBM traffic is 0.24 GB (129 us at batch 128, plus 31 us for
// the padded copy).  A pixel's windows overlap its neighbours' 12-fold, so here the image patch of a 16 x 16 output
// tile (37 x 40 input pixels) is brought into LDS ONCE, straight from the 3-channel tensor (24-byte groups of 4 pixels
// -> four 8-byte 4-channel pixels), and every MFMA operand is a 16-byte LDS read from it:
//   D[co][pixel] += W[co][k] * X[k][pixel],   k = (kh, kw', c) with kw' = kw + 1 in 0..7 and c in 0..3  (K = 7 x 32 = 224),
// v_mfma_f32_32x32x16_bf16 with the filter as the A operand (register-resident for the whole kernel: 2 x 14 fragments)
// and 32 pixels (two tile rows of 16) as B: lane (pixel, k-group) reads 16 contiguous bytes = two neighbouring input
// pixels x 4 channels at LDS column 2 * pixel_column + 4 * half + 2 * k-group.  The patch starts one pixel left of the
// window (column 2 * ow0 - 4, a multiple of 4 pixels, so the 24-byte global groups are 8-byte aligned and whole groups
// are inside or outside the image); kw' = 0 carries zero weights.  Row pitch 48 pixels = 384 B: two tile rows are
// 768 B = 3 bank rows apart, which makes the four 16-lane groups of ds_read_b128 conflict-free.
// Workgroups are persistent (2 per CU) and walk tiles in image order; the next tile's patch is in registers while the
// current one is multiplied.  The output tile leaves through LDS as whole 128-byte pixel rows, with the batch-norm
// partial sums of the bf16-rounded outputs (two 128-pixel partials per tile, fixed order) like every K1 forward epilogue.
// Reference: resnet_model.py:456-501 (conv2d_fixed_padding 7x7/2 + batch norm), pruning_layers.py:139-157.
#pragma once

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_stem;
struct StemArgs {
  const uint16_t* X;     // [N][H][W][3] bf16
  const uint16_t* WP;    // [64][7][32] bf16: k_stem_weights_shift's packed filter (kw' * 4 + c)
  uint16_t* Y;           // [N][Ho][Wo][64] bf16
  float* STATS;          // [tiles * 2][2][64] or NULL
  int N, H, W, Ho, Wo, pt;
  int tiles_h, tiles_w, tiles;
};

// wp[co][r][kw' * 4 + c]  <-  w_ohwi[co][(r * 7 + kw' - 1) * 3 + c]   (0 for kw' = 0 and c = 3)
__global__ __launch_bounds__(THREADS) void k_stem_weights_shift(const uint16_t* __restrict__ w, uint16_t* __restrict__ wp, int cout) {
  const int total = cout * 7 * 32;
  for (int i = blockIdx.x * THREADS + threadIdx.x; i < total; i += gridDim.x * THREADS) {
    const int j = i & 31, r = (i >> 5) % 7, co = i / (7 * 32);
    const int kwp = j >> 2, c = j & 3;
    wp[i] = (kwp >= 1 && c < 3) ? w[(int64_t)co * 147 + (r * 7 + kwp - 1) * 3 + c] : (uint16_t)0;
  }
}

constexpr int STEM_PW = 48, STEM_PROWS = 37, STEM_PATCH = STEM_PROWS * STEM_PW * 8;   // 14 208 B
constexpr int STEM_OROW = 144, STEM_OUT = 256 * STEM_OROW;                            // 36 864 B
constexpr int STEM_RED = 4 * 2 * 2 * 64 * 4;                                          // [wave][half][q][64] floats
constexpr int STEM_SMEM = 2 * STEM_PATCH + STEM_OUT + STEM_RED;

__global__ __launch_bounds__(THREADS, 2) void k_stem_fwd(StemArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_stem[];
  unsigned char* const patch0 = smem_stem;
  unsigned char* const ostage = smem_stem + 2 * STEM_PATCH;
  float* const red = reinterpret_cast<float*>(smem_stem + 2 * STEM_PATCH + STEM_OUT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- the filter, register-resident: A fragment (cf, s): lane (co = cf * 32 + (l & 31), k-group l >> 5) holds
  //      k = 16 * (s & 1) + 8 * (l >> 5) .. + 7 of filter row s >> 1
  bf16x8 wfr[2][14];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int s = 0; s < 14; ++s)
      wfr[cf][s] = *reinterpret_cast<const bf16x8*>(P.WP + ((cf * 32 + (lane & 31)) * 7 + (s >> 1)) * 32 + 16 * (s & 1) + 8 * (lane >> 5));

  const __amdgpu_buffer_rsrc_t rsrcXs = make_rsrc(P.X, (uint32_t)((size_t)P.N * P.H * P.W * 6));
  // ---- patch tasks of this thread: (row, group of 4 pixels), 37 x 10 = 370 of them over 256 threads
  constexpr int NTASK = STEM_PROWS * 10;
  const int t1 = tid + THREADS;
  const int pr0 = tid / 10, pg0 = tid % 10, pr1 = t1 / 10, pg1 = t1 % 10;
  const bool has1 = t1 < NTASK;
  uint2 ld[2][3];
  // loads the 24 bytes of task (pr, pg) of tile `t` (zeros outside the image)
#define STEM_LOAD(slot_, pr_, pg_, n_, ih0_, iw0_)                                                        \
  {                                                                                                       \
    const int ih_ = (ih0_) + (pr_), iw_ = (iw0_) + 4 * (pg_);                                             \
    const bool ok_ = (unsigned)ih_ < (unsigned)P.H && iw_ >= 0 && iw_ + 3 < P.W;                          \
    /* buffer loads, out of range where the pixels do not exist: `ok ? src[i] : 0` had become six dword loads under branches */ \
    const uint32_t off_ = ok_ ? (uint32_t)((((n_) * P.H + ih_) * P.W + iw_) * 6) : OOB;                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                    \
      const u32x2_stem v_ = __builtin_amdgcn_raw_buffer_load_b64(rsrcXs, (int)(off_ + 8u * i_), 0, 0);     \
      ld[slot_][i_] = make_uint2(v_.x, v_.y);                                                             \
    }                                                                                                     \
  }
  // 24 bytes = 12 bf16 = 4 pixels x 3 channels  ->  4 x (3 channels + 0) as two 16-byte LDS writes
#define STEM_STORE(slot_, buf_, pr_, pg_)                                                                 \
  {                                                                                                       \
    const uint32_t d0 = ld[slot_][0].x, d1 = ld[slot_][0].y, d2 = ld[slot_][1].x, d3 = ld[slot_][1].y,    \
                   d4 = ld[slot_][2].x, d5 = ld[slot_][2].y;                                              \
    uint4 a_, b_;                                                                                         \
    a_.x = d0; a_.y = d1 & 0xFFFFu; a_.z = (d1 >> 16) | (d2 << 16); a_.w = d2 >> 16;                      \
    b_.x = d3; b_.y = d4 & 0xFFFFu; b_.z = (d4 >> 16) | (d5 << 16); b_.w = d5 >> 16;                      \
    unsigned char* dst_ = (buf_) + ((pr_) * STEM_PW + 4 * (pg_)) * 8;                                     \
    *reinterpret_cast<uint4*>(dst_) = a_; *reinterpret_cast<uint4*>(dst_ + 16) = b_;                      \
  }
#define STEM_TILE_ORIGIN(t_, n_, oh0_, ow0_)                                                              \
  {                                                                                                       \
    const int per_ = P.tiles_h * P.tiles_w;                                                               \
    n_ = (t_) / per_;                                                                                     \
    const int r_ = (t_) - n_ * per_;                                                                      \
    oh0_ = (r_ / P.tiles_w) * 16; ow0_ = (r_ % P.tiles_w) * 16;                                           \
  }

  int t = blockIdx.x;
  if (t >= P.tiles) return;
  {
    int n, oh0, ow0;
    STEM_TILE_ORIGIN(t, n, oh0, ow0);
    STEM_LOAD(0, pr0, pg0, n, 2 * oh0 - P.pt, 2 * ow0 - 4);
    if (has1) STEM_LOAD(1, pr1, pg1, n, 2 * oh0 - P.pt, 2 * ow0 - 4);
  }
  // B fragment of k-step s, pixel fragment pf (tile rows 2 pf, 2 pf + 1): lane (pixel l & 31, k-group l >> 5)
  const int b_lane = ((2 * ((lane & 31) >> 4)) * STEM_PW + 2 * (lane & 15) + 2 * (lane >> 5)) * 8;
  int it = 0;
  for (; t < P.tiles; t += gridDim.x, ++it) {
    unsigned char* const patch = patch0 + (it & 1) * STEM_PATCH;
    int n, oh0, ow0;
    STEM_TILE_ORIGIN(t, n, oh0, ow0);
    STEM_STORE(0, patch, pr0, pg0);
    if (has1) STEM_STORE(1, patch, pr1, pg1);
    const int tn = t + gridDim.x;
    if (tn < P.tiles) {                      // the next tile's patch: in flight under this tile's MFMAs
      int n2, oh2, ow2;
      STEM_TILE_ORIGIN(tn, n2, oh2, ow2);
      STEM_LOAD(0, pr0, pg0, n2, 2 * oh2 - P.pt, 2 * ow2 - 4);
      if (has1) STEM_LOAD(1, pr1, pg1, n2, 2 * oh2 - P.pt, 2 * ow2 - 4);
    }
    __syncthreads();                         // patch complete; the previous tile's output stage and sums are consumed

    // ---- wave w: pixel fragments 2 w, 2 w + 1 (tile rows 4 w .. 4 w + 3) x both channel fragments
    f32x16 acc[2][2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cf][pf][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      bf16x8 xf[2];
#pragma unroll
      for (int pf = 0; pf < 2; ++pf) {
        const int row0 = 2 * (2 * (2 * wave + pf)) + (s >> 1);       // patch row of tile row 2 (2 w + pf), filter row s >> 1
        xf[pf] = *reinterpret_cast<const bf16x8*>(patch + b_lane + (row0 * STEM_PW + 4 * (s & 1)) * 8);
      }
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc[cf][pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[cf][s], xf[pf], acc[cf][pf], 0, 0, 0);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keeps the fragment reads of at most four k-steps in registers
    }
    // ---- accumulators -> output stage [256 pixels][64 channels] bf16 (144-byte rows): lane = pixel (l & 31) of its
    //      fragment, element e = channel cf * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int pf = 0; pf < 2; ++pf) {
        const int px = (2 * wave + pf) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x2 lo = {acc[cf][pf][4 * g], acc[cf][pf][4 * g + 1]}, hi = {acc[cf][pf][4 * g + 2], acc[cf][pf][4 * g + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
          *reinterpret_cast<uint2*>(ostage + px * STEM_OROW + (cf * 32 + 8 * g + 4 * (lane >> 5)) * 2) = pk;
        }
      }
    __syncthreads();
    // ---- out in whole pixel rows: thread (row = pass * 32 + tid / 8, 16-byte chunk tid % 8); sums of the rounded values
    const int chunk = tid & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {            // the two 128-pixel halves = the two partials of this tile
      float s0[8], s1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
#pragma unroll
      for (int p4 = 0; p4 < 4; ++p4) {
        const int row = (h * 4 + p4) * 32 + (tid >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(ostage + row * STEM_OROW + chunk * 16);
        const int64_t m = ((int64_t)n * P.Ho + oh0 + (row >> 4)) * P.Wo + ow0 + (row & 15);
        store16(P.Y + m * 64 + chunk * 8, v);
        if (P.STATS) {
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a = __uint_as_float(w4[q] << 16), b = __uint_as_float(w4[q] & 0xFFFF0000u);
            s0[2 * q] += a; s1[2 * q] = fmaf(a, a, s1[2 * q]);
            s0[2 * q + 1] += b; s1[2 * q + 1] = fmaf(b, b, s1[2 * q + 1]);
          }
        }
      }
      if (P.STATS) {
        // lanes l, l ^ 8, l ^ 16, l ^ 32 share a chunk: fixed-order butterfly, then the four waves through LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int off = 8; off < 64; off <<= 1) {
            s0[j] += __shfl_xor(s0[j], off);
            s1[j] += __shfl_xor(s1[j], off);
          }
        }
        if (lane < 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            red[((wave * 2 + h) * 2 + 0) * 64 + chunk * 8 + j] = s0[j];
            red[((wave * 2 + h) * 2 + 1) * 64 + chunk * 8 + j] = s1[j];
          }
        }
      }
    }
    if (P.STATS) {
      __syncthreads();
      {
        const int h = tid >> 7, q = (tid >> 6) & 1, c = tid & 63;     // 256 threads = 2 halves x 2 quantities x 64 channels
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[((w * 2 + h) * 2 + q) * 64 + c];
        P.STATS[(((int64_t)t * 2 + h) * 2 + q) * 64 + c] = sum;
      }
    }
  }
#undef STEM_LOAD
#undef STEM_STORE
#undef STEM_TILE_ORIGIN
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
// dW[kh][kw'][c][co] = sum over pixels X[2 oh + kh - pt][2 ow + kw' - 4][c] * dY[oh][ow][co]: through the generic tr body
// the padded image is gathered 448 B per output pixel (0.93 GB of L2 -> LDS traffic, 191 us + 31 us for the copy).  Here
// a workgroup walks 8 x 16 output-pixel tiles: the 21 x 40 input patch (as in the forward kernel) and the 128 x 64 dY
// tile (LDS-DMA, rows XOR-swizzled like bwd1x1's 128-byte rows) are resident, the reduction index of a
// v_mfma_f32_32x32x16_bf16 is 16 output pixels of one tile row, and BOTH operands are transposing reads
// (ds_read_b64_tr_b16): for the image the "row" of reduction index ow is the 64 contiguous bytes starting at patch
// pixel 2 ow -- rows 16 bytes apart, overlapping, which a read does not mind -- and the 32 values in it are exactly
// (kw', c) of one filter row.  Wave w accumulates filter rows w and w + 4 (x both halves of the 64 output channels) in
// registers over all its tiles and writes one [7][32][64] fp32 slab per workgroup; the usual fixed-order reduce sums
// the slabs and k_stem_unpack drops the kw' = 0 and c = 3 rows.
struct StemWgradArgs {
  const uint16_t* X;     // [N][H][W][3] bf16
  const uint16_t* DY;    // [N][Ho][Wo][64] bf16
  float* SLAB;           // [gridDim][7][32][64] fp32
  int N, H, W, Ho, Wo, pt;
  int tiles_h, tiles_w, tiles;
  uint32_t dy_bytes;
};
constexpr int STEMW_PROWS = 21, STEMW_PATCH = STEMW_PROWS * STEM_PW * 8;     // 8 064 B
constexpr int STEMW_DY = 128 * 128;                                          // 16 384 B
constexpr int STEMW_SMEM = 2 * (STEMW_PATCH + STEMW_DY);

__global__ __launch_bounds__(THREADS, 2) void k_stem_wgrad(StemWgradArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_stem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rsrcY = make_rsrc(P.DY, P.dy_bytes);
  const u32x4 rsrcY4 = make_rsrc4(P.DY, P.dy_bytes);
  const __amdgpu_buffer_rsrc_t rsrcXs = make_rsrc(P.X, (uint32_t)((size_t)P.N * P.H * P.W * 6));
  (void)rsrcY; (void)rsrcY4;

  f32x16 acc[2][2];                              // [filter row slot: wave, wave + 4][channel half]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][cf][e] = 0.f;

  const int pr0 = tid / 10, pg0 = tid % 10;      // patch task (row, group of 4 pixels): 21 x 10 = 210 <= 256
  const bool has0 = tid < STEMW_PROWS * 10;
  uint2 ld[3];
#define STEMW_TILE_ORIGIN(t_, n_, oh0_, ow0_)                                                             \
  {                                                                                                       \
    const int per_ = P.tiles_h * P.tiles_w;                                                               \
    n_ = (t_) / per_;                                                                                     \
    const int r_ = (t_) - n_ * per_;                                                                      \
    oh0_ = (r_ / P.tiles_w) * 8; ow0_ = (r_ % P.tiles_w) * 16;                                            \
  }
#define STEMW_LOAD(n_, ih0_, iw0_)                                                                        \
  {                                                                                                       \
    const int ih_ = (ih0_) + pr0, iw_ = (iw0_) + 4 * pg0;                                                 \
    const bool ok_ = has0 && (unsigned)ih_ < (unsigned)P.H && iw_ >= 0 && iw_ + 3 < P.W;                  \
    const uint32_t off_ = ok_ ? (uint32_t)((((n_) * P.H + ih_) * P.W + iw_) * 6) : OOB;   /* (buffer loads: k_stem_fwd) */ \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                    \
      const u32x2_stem v_ = __builtin_amdgcn_raw_buffer_load_b64(rsrcXs, (int)(off_ + 8u * i_), 0, 0);     \
      ld[i_] = make_uint2(v_.x, v_.y);                                                                    \
    }                                                                                                     \
  }
  // the dY tile: 16 wave-instructions of 8 pixels x 128 B, four per wave; lane l: pixel + (l >> 3), 16-byte slot l & 7,
  // fetching the logical chunk slot ^ swizzle(pixel)
#define STEMW_DMA(buf_, n_, oh0_, ow0_)                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                       \
      const int i_ = q * 4 + wave, px_ = i_ * 8 + (lane >> 3);                                            \
      const int64_t m_ = ((int64_t)(n_) * P.Ho + (oh0_) + (px_ >> 4)) * P.Wo + (ow0_) + (px_ & 15);       \
      const int off_ = (int)((uint32_t)m_ * 128u + (uint32_t)(((lane & 7) ^ dual_swz<128>(px_)) << 4));   \
      RIGL_DMA16(rsrcY, (buf_) + i_ * 1024, off_);   /* (asm form: the loop reads LDS behind its DMA issue, conv.hip lds_dma16) */ \
    }                                                                                                     \
  }

  int t = blockIdx.x;
  if (t < P.tiles) {
    int n, oh0, ow0;
    STEMW_TILE_ORIGIN(t, n, oh0, ow0);
    STEMW_DMA(smem_stem + STEMW_PATCH, n, oh0, ow0);
    STEMW_LOAD(n, 2 * oh0 - P.pt, 2 * ow0 - 4);
  }
  // transposing-read lanes: 16-lane group g = l >> 4 -> reduction half g >> 1 (pixels 8 (g >> 1) ..), value block g & 1
  // (16 of the fragment's 32 rows); inside a group lane j: pixel j >> 2 (+ 4 for the second read), 8-byte piece j & 3
  const int grp = lane >> 4, j16 = lane & 15;
  const int a_lane = (2 * (8 * (grp >> 1) + (j16 >> 2))) * 8 + 32 * (grp & 1) + 8 * (j16 & 3);
  const int t_row = 8 * (grp >> 1) + (j16 >> 2);
  const int t_low = 2 * (grp & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
#define STEMW_Y_OFF(chunk_, plus4_) \
  ((t_row + (plus4_)) * 128 + ((((chunk_) + t_low) ^ dual_swz<128>(t_row + (plus4_))) << 4) + t_half)
  int it = 0;
  for (; t < P.tiles; t += gridDim.x, ++it) {
    unsigned char* const patch = smem_stem + (it & 1) * (STEMW_PATCH + STEMW_DY);
    unsigned char* const dyt = patch + STEMW_PATCH;
    wait_vmcnt<0>();                           // this tile's dY (DMA) and patch (registers) have arrived
    if (has0) {
      const uint32_t d0 = ld[0].x, d1 = ld[0].y, d2 = ld[1].x, d3 = ld[1].y, d4 = ld[2].x, d5 = ld[2].y;
      uint4 a_, b_;
      a_.x = d0; a_.y = d1 & 0xFFFFu; a_.z = (d1 >> 16) | (d2 << 16); a_.w = d2 >> 16;
      b_.x = d3; b_.y = d4 & 0xFFFFu; b_.z = (d4 >> 16) | (d5 << 16); b_.w = d5 >> 16;
      unsigned char* dst_ = patch + (pr0 * STEM_PW + 4 * pg0) * 8;
      *reinterpret_cast<uint4*>(dst_) = a_; *reinterpret_cast<uint4*>(dst_ + 16) = b_;
    }
    __syncthreads();                           // tile resident; every wave is done with the other buffer pair
    const int tn = t + gridDim.x;
    if (tn < P.tiles) {
      int n2, oh2, ow2;
      STEMW_TILE_ORIGIN(tn, n2, oh2, ow2);
      STEMW_DMA(smem_stem + ((it + 1) & 1) * (STEMW_PATCH + STEMW_DY) + STEMW_PATCH, n2, oh2, ow2);
      STEMW_LOAD(n2, 2 * oh2 - P.pt, 2 * ow2 - 4);
    }
#pragma unroll
    for (int R = 0; R < 8; ++R) {              // one tile row = 16 pixels of reduction
      const unsigned char* yrow = dyt + R * 16 * 128;
      bf16x8 yf[2];
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) yf[cf] = lds_read_tr_pair(yrow + STEMW_Y_OFF(cf * 4, 0), yrow + STEMW_Y_OFF(cf * 4, 4));
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int kh = wave + 4 * a;
        if (kh < 7) {
          const unsigned char* xrow = patch + ((2 * R + kh) * STEM_PW) * 8 + a_lane;
          const bf16x8 xf = lds_read_tr_pair(xrow, xrow + 4 * 16);
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) acc[a][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf[cf], acc[a][cf], 0, 0, 0);
        }
      }
    }
  }
#undef STEMW_TILE_ORIGIN
#undef STEMW_LOAD
#undef STEMW_DMA
#undef STEMW_Y_OFF
  // the workgroup's partial: D row (e & 3) + 8 (e >> 2) + 4 (l >> 5) = (kw', c) of filter row kh, column l & 31 = channel
  float* const out = P.SLAB + (int64_t)blockIdx.x * (7 * 32 * 64);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int kh = wave + 4 * a;
    if (kh < 7) {
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          out[(kh * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 64 + cf * 32 + (lane & 31)] = acc[a][cf][e];
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Legal: 7x7, stride 2, 3 input channels, 64 output channels, 3 pixels of left padding (the reference's fixed_padding),
// output a whole number of 16 x 16 tiles, input width a multiple of 4.  "stem_direct" = 0 turns the kernel off.
static inline bool stem_direct_legal(const RiglConvDesc* d) {
  return d->kh == 7 && d->kw == 7 && d->stride_h == 2 && d->stride_w == 2 && d->cin == 3 && d->cout == 64 && d->pad_left == 3 &&
         d->pad_top >= 0 && d->pad_top <= 3 && (d->ho % 16) == 0 && (d->wo % 16) == 0 && (d->w % 4) == 0 &&
         RIGL_TUNE("stem_direct", 1) != 0;
}
static bool launch_stem_fwd(const RiglConvDesc* d, const rigl_bf16* x, const uint16_t* wp, rigl_bf16* y, float* stats, hipStream_t st) {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                STEM_SMEM) == hipSuccess;
  if (!ready) return false;
  StemArgs a = {};
  a.X = x; a.WP = wp; a.Y = y; a.STATS = stats;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pt = d->pad_top;
  a.tiles_h = d->ho / 16; a.tiles_w = d->wo / 16; a.tiles = d->n * a.tiles_h * a.tiles_w;
  const int grid = a.tiles < 2 * num_cus() ? a.tiles : 2 * num_cus();
  RIGL_K_LAUNCH(k_stem_fwd, dim3((unsigned)grid), dim3(THREADS), STEM_SMEM, st, a);
  return true;
}

// The weight gradient: one slab per workgroup + the reduced [7][32][64] image (k_stem_unpack finishes with shift 1).
static inline int stem_wgrad_grid(const RiglConvDesc* d) {
  const int tiles = d->n * (d->ho / 8) * (d->wo / 16);
  return tiles < 2 * num_cus() ? tiles : 2 * num_cus();
}
static inline size_t stem_wgrad_workspace(const RiglConvDesc* d) {
  return stem_direct_legal(d) ? align_up((size_t)7 * 32 * 64 * 4, 256) + (size_t)stem_wgrad_grid(d) * 7 * 32 * 64 * 4 : 0;
}
static bool launch_stem_wgrad(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, float* slabs, hipStream_t st) {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                STEMW_SMEM) == hipSuccess;
  if (!ready) return false;
  StemWgradArgs a = {};
  a.X = x; a.DY = dy; a.SLAB = slabs;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo; a.pt = d->pad_top;
  a.tiles_h = d->ho / 8; a.tiles_w = d->wo / 16; a.tiles = d->n * a.tiles_h * a.tiles_w;
  a.dy_bytes = (uint32_t)((size_t)d->n * d->ho * d->wo * 64 * 2);
  RIGL_K_LAUNCH(k_stem_wgrad, dim3((unsigned)stem_wgrad_grid(d)), dim3(THREADS), STEMW_SMEM, st, a);
  return true;
}
