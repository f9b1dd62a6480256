// K1, the 3x3 stride-1 convs with 64 input and 64 output channels (ResNet-50 group 1: 56x56x64, three layers; included
// inside namespace rigl::k1 of conv.hip).  Reference: layers.masked_conv2d of the bottleneck's middle conv
// (rigl/imagenet_resnet/resnet_model.py:471-482 through pruning_layers.py:139-157) and its autodiff
// (sparse_optimizers_base.py:478-485 for the dense dW).
//
// On the generic bodies these layers sat at 4.7-5.1x their roofline bound: with N = 64 columns the igemm
// tile is 128x64, its K loop 18 tiles of 32, and every input pixel is fetched NINE times from L2 -- 462 MB of L2 -> LDS
// traffic per layer at ~8 TB/s = the 57 us the forward took.  Here the input is fetched ONCE:
//   forward / dgrad (k_c3x3): PERSISTENT workgroups, one per CU (512 threads), walk tiles of TH output rows x the full
//     width of one image.  A tile's input patch ((TH + 2) rows x (W + 2) pixels x 64 channels, zero border by out-of-range
//     DMA lanes) goes HBM -> LDS by LDS-DMA and is indexed LINEARLY, q = row * (W + 2) + column, so that filter tap
//     (dr, dc) of output pixel q is patch pixel q + dr * (W + 2) + dc -- a constant LDS offset; an M-tile of the implicit
//     GEMM is 32 consecutive q (the two halo columns per row are computed and dropped: 3.4 % at W = 56).  Two patch
//     buffers: the pieces of the NEXT tile's patch are issued between the M-tiles of this one.  The 3x3x64 filter of a
//     wave's 32 output channels is REGISTER-RESIDENT (36 MFMA A-fragments = 144 VGPRs, copied once per workgroup through
//     LDS), so the K loop holds only ds_read_b128 of activation fragments (16-byte chunks XORed with (q >> 1) & 7:
//     conflict-free at every tap shift, any 16 lanes of a read group differ in q mod 16) and v_mfma_f32_32x32x16_bf16,
//     the four fragments of tap t + 1 requested above the MFMAs of tap t.  8 waves = 2 channel halves x 4 M-tile phases.
//     dgrad is the same kernel on dY with the filter read flipped and transposed from the HWIO shadow
//     (dx[p][ci] = sum dy[p + 1 - tap][co] w[tap][ci][co]).
//   weight gradient (k_c3x3_wgrad): persistent, one 8-wave workgroup per CU, tiles of TH = 4 rows in two LDS stages (X
//     patch + dY tile, same linear indexing, dY's two surplus columns zero); both operands come out of
//     ds_read_b64_tr_b16 (the reduction index is the pixel), the two waves of a SIMD keep the nine taps (5 + 4) of one
//     32 x 32 block of dW in accumulator registers and read ONE dY fragment per k-step.  One [9][64][64] fp32 slab per
//     workgroup, launch_wgrad_reduce finishes.
// Epilogues as in the other bodies: bf16 outputs staged per wave through LDS and stored as 16 bytes per lane, batch-norm
// statistics of the bf16 outputs (one partial per WORKGROUP here: rigl_conv2d_stats_parts), the dgrad addend.
// Measured at batch 128 (56x56, gpurun r4e-r4l; kernels alone, us): forward 60 -> 39 (what moved it: the filter through
// LDS instead of 288 uncoalesced fragment loads per workgroup, 10 000 -> 7 400 cycles of prologue; reads ahead of the
// MFMAs, 51 -> 43; statistics once per workgroup instead of an xor tree per tile, -4 000 cycles per tile; per tile now
// ~11 000 cycles of M-tiles + ~2 700 of waiting for the piece-issuing waves, against 8 064 cycles of MFMA on the busiest
// SIMD), dgrad 72 -> 42, weight gradient 92 -> 47 + 7 (reduce), one-call backward 133 -> 98; in the step K1 -0.19 ms.
#pragma once

struct C3Args {
  const uint16_t* X;     // gathered activations [N][H][W][64] (x for the forward, dy for dgrad)
  const uint16_t* WT;    // forward: OHWI [64][9][64]; dgrad: HWIO [9][64][64]
  uint16_t* Y;           // [N][H][W][64]
  const uint16_t* ADD;   // dgrad: optional addend (bf16, like Y)
  float* STATS;          // forward: optional [tiles][2][64]
  int N, H, W, PW, TH, tiles_h, tiles;
  int alloc_px;          // patch pixels DMA-filled (the real patch, then zeros up to the last row any M-tile reads)
  uint32_t x_bytes;
  FastDiv fd_pw, fd_th;
  unsigned long long* TRACE;   // development (-DRIGL_C3_TRACE): [grid][64] s_memtime stamps of wave 0, else unused
};
#ifdef RIGL_C3_TRACE
#define C3_STAMP(i_) { if (tid == 0 && P.TRACE && (i_) < 64) P.TRACE[blockIdx.x * 64 + (i_)] = __builtin_amdgcn_s_memtime(); }
#else
#define C3_STAMP(i_) { }
#endif

constexpr int C3_STG_ROWB = 80;                      // staged output row of a wave: 32 channels x 2 B + 16 (bank spread)
constexpr int C3_STG_WAVE = 32 * C3_STG_ROWB;
constexpr int C3_THREADS = 512;                      // 8 waves = 2 channel halves x 4 M-tile phases, one workgroup per CU
constexpr int C3_RED_BYTES = 8 * 2 * 32 * 4;         // statistics scratch [8 waves][2][32]

// The whole reduction (9 taps x 64 channels) of one M-tile: 36 ds_read_b128 + 36 MFMAs, no control flow.  The four
// fragments of tap t + 1 are requested before the MFMAs of tap t (the compiler's own schedule read one fragment, waited,
// multiplied: with the filter taking 144 of the 256 registers it never ran reads ahead).  (Per-tap fragment offsets kept
// in nine more registers instead of the ~4 VALU per read below spilled 22-37 registers and cost 18 us: measured, dropped.)
__device__ __forceinline__ void c3_ktile(const unsigned char* const patch, const bf16x8 (&wf)[9][4], const int qa, const int PW,
                                         const int hi, f32x16& acc) {
  bf16x8 xf[2][4];
#define C3_READ_TAP(tap_, buf_)                                                                          \
  {                                                                                                      \
    const int pa_ = qa + ((tap_) / 3) * PW + ((tap_) % 3);                                               \
    const unsigned char* const ra_ = patch + pa_ * 128;                                                  \
    const int sw_ = (pa_ >> 1) & 7;                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      xf[buf_][ks] = *reinterpret_cast<const bf16x8*>(ra_ + (((2 * ks + hi) ^ sw_) << 4));               \
  }
  C3_READ_TAP(0, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap < 8) C3_READ_TAP(tap + 1, (tap + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);               // (the reads stay ABOVE this tap's MFMAs)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap][ks], xf[tap & 1][ks], acc, 0, 0, 0);
  }
#undef C3_READ_TAP
}

// Persistent workgroups (one per CU) walk tiles L = blockIdx.x, + gridDim.x, ...: the patch of tile L + grid is in flight
// (LDS-DMA into the other buffer) while tile L is multiplied, so the chip's HBM stream and its MFMA work overlap instead
// of alternating (the first version -- one tile per workgroup, two workgroups per CU -- ran every CU's load phase and
// every CU's compute phase at the same moments: 55 us per layer, no faster than the generic body).
template <bool DGRAD>
__global__ __launch_bounds__(C3_THREADS, 2) void k_c3x3(C3Args P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_c3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int coh = wave & 1, mph = wave >> 1;           // channel half, M-tile phase (M-tiles mph, mph + 4, ...)
  const int PW = P.PW;
  const int patch_bytes = P.alloc_px * 128;
  unsigned char* const stg = smem_c3 + 2 * patch_bytes + wave * C3_STG_WAVE;
  const __amdgpu_buffer_rsrc_t rsrcX = make_rsrc(P.X, P.x_bytes);
  const u32x4 rsrcX4 = make_rsrc4(P.X, P.x_bytes);
  (void)rsrcX4; (void)rsrcX;
  const int npx = (P.TH + 2) * PW;
  const int n_inst = P.alloc_px >> 3;

  // The patch of a tile: 1-KB LDS-DMA pieces; lane (i, l) of piece i fills pixel i * 8 + (l >> 3), 16-byte slot l & 7 with
  // source chunk slot ^ ((pixel >> 1) & 7); pixels outside the image (and the zero tail) use an out-of-range offset ->
  // zeros.  A wave takes every nw-th piece; its pixel's (row, column) advances incrementally (no division per piece), so
  // the pieces can be issued a few at a time BETWEEN M-tiles (all pieces up front, by every wave at once, left the MFMA
  // pipes idle for ~2 300 cycles per tile).
  struct Issue { int i, pr, pc, n, h0; unsigned char* dst; };
  Issue iq;
  iq.i = 1 << 30; iq.pr = iq.pc = iq.n = iq.h0 = 0; iq.dst = smem_c3;
#define C3_ISSUE_BEGIN(L_, buf_, wi_)                                                                    \
  {                                                                                                      \
    const uint32_t tl_ = xcd_remap((uint32_t)(L_), (uint32_t)P.tiles);                                   \
    iq.n = fdiv((int)tl_, P.fd_th); iq.h0 = ((int)tl_ - iq.n * P.tiles_h) * P.TH;                        \
    iq.dst = smem_c3 + (buf_) * patch_bytes;                                                             \
    iq.i = (wi_);                                                                                        \
    const int px_ = iq.i * 8 + (lane >> 3);                                                              \
    iq.pr = fdiv(px_, P.fd_pw); iq.pc = px_ - iq.pr * PW;                                                \
  }
  // up to cnt_ pieces of this wave's share, stepping nw_ pieces (nw_ * 8 pixels: at most two row wraps for W >= 22)
#define C3_ISSUE_SOME(cnt_, nw_)                                                                         \
  {                                                                                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < (cnt_); ++k_) {                                              \
      if (iq.i < n_inst) {                                                                               \
        const int px = iq.i * 8 + (lane >> 3), slot = lane & 7;                                          \
        const int chunk = slot ^ ((px >> 1) & 7);                                                        \
        const int h = iq.h0 - 1 + iq.pr, w = iq.pc - 1;                                                  \
        const bool ok = px < npx && (unsigned)h < (unsigned)P.H && (unsigned)w < (unsigned)P.W;         \
        const int off = ((iq.n * P.H + h) * P.W + w) * 64 + chunk * 8;                                   \
        RIGL_DMA16(rsrcX, iq.dst + iq.i * 1024, ok ? (int)((uint32_t)off * 2u) : (int)OOB);    \
        iq.i += (nw_);                                                                                   \
        iq.pc += 8 * (nw_);                                                                              \
        if (iq.pc >= PW) { iq.pc -= PW; ++iq.pr; }                                                       \
        if (iq.pc >= PW) { iq.pc -= PW; ++iq.pr; }                                                       \
        if (iq.pc >= PW) { iq.pc -= PW; ++iq.pr; }                                                       \
      }                                                                                                  \
    }                                                                                                    \
  }
#define C3_ISSUE_REST(nw_) { while (iq.i < n_inst) C3_ISSUE_SOME(1, nw_) }
  int L = blockIdx.x;
  C3_STAMP(0);
  if (L < P.tiles) { C3_ISSUE_BEGIN(L, 0, wave); C3_ISSUE_REST(8); }
  // ---- the filter: 73 728 contiguous bytes (OHWI for the forward, HWIO for dgrad) copied ONCE per workgroup into the LDS
  // behind patch buffer 0 by 72 coalesced LDS-DMA instructions, then every wave reads the 36 MFMA A-fragments of its 32
  // output channels into registers (fetched straight from global memory, a fragment load touched 32 rows x 32 bytes: the
  // eight waves' 288 loads took 10 000 cycles of texture-address time per workgroup).  Rows of the LDS image = the
  // fragment rows (forward: output channel, 72 chunks of 16 bytes; dgrad: (tap, channel), 8 chunks); the low three bits
  // of a row's chunk index are XORed with (row >> 1) & 7 on the DMA's source side -- conflict-free ds_read_b128.
  {
    const __amdgpu_buffer_rsrc_t rsrcW = make_rsrc(P.WT, 9 * 64 * 64 * 2);
    const u32x4 rsrcW4 = make_rsrc4(P.WT, 9 * 64 * 64 * 2);
    (void)rsrcW; (void)rsrcW4;
    unsigned char* const wl = smem_c3 + patch_bytes;
    constexpr int CPR = DGRAD ? 8 : 72;                // chunks per row
    for (int i = wave; i < 72; i += 8) {
      const int p = i * 64 + lane, row = p / CPR, c = p - row * CPR;
      const int src = row * CPR + (c ^ ((row >> 1) & 7));
      RIGL_DMA16(rsrcW, wl + i * 1024, src * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 wf[9][4];
  {
    const unsigned char* const wl = smem_c3 + patch_bytes;
    const int oc = coh * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int row = DGRAD ? (8 - tap) * 64 + oc : oc;
        const int c = (DGRAD ? 0 : tap * 8) + 2 * ks + (lane >> 5);
        wf[tap][ks] = *reinterpret_cast<const bf16x8*>(wl + (row * (DGRAD ? 8 : 72) + (c ^ ((row >> 1) & 7))) * 16);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the loop's first barrier then frees this LDS for patch buffer 1)
  const int hi = lane >> 5;
  const int s_ch = lane & 3;                           // the 16-byte chunk of a staged row this lane stores
  float sy[8], sq[8];                                  // statistics of this lane's staged chunk over ALL tiles of the workgroup
#pragma unroll
  for (int c = 0; c < 8; ++c) sy[c] = sq[c] = 0.f;
  C3_STAMP(1);

  for (int it = 0; L < P.tiles; ++it, L += gridDim.x) {
    // patch L has landed in every wave's share, and every wave is done with the other buffer (tile L - grid)
    C3_STAMP(2 + it * 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C3_STAMP(3 + it * 5);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    C3_STAMP(4 + it * 5);
    // the next patch is requested by the six waves of M-tile phases 1..3 (three M-tiles of a 13-M-tile patch each; the two
    // phase-0 waves have four), four pieces before each of their M-tiles, the rest behind the last
    const bool issuer = wave >= 2 && L + (int)gridDim.x < P.tiles;
    if (issuer) C3_ISSUE_BEGIN(L + (int)gridDim.x, (it + 1) & 1, wave - 2);
    C3_STAMP(5 + it * 5);
    const unsigned char* const patch = smem_c3 + (it & 1) * patch_bytes;
    const uint32_t tile = xcd_remap((uint32_t)L, (uint32_t)P.tiles);
    const int n = fdiv((int)tile, P.fd_th), h0 = ((int)tile - n * P.tiles_h) * P.TH;
    const int rows = P.H - h0 < P.TH ? P.H - h0 : P.TH;
    const int MT = (rows * PW + 31) >> 5;

    for (int mt = mph; mt < MT; mt += 4) {
      if (issuer) C3_ISSUE_SOME(4, 6);
      f32x16 acc0;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
      c3_ktile(patch, wf, mt * 32 + (lane & 31), PW, hi, acc0);
      // ---- epilogue of the M-tile: D row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) -> channel, column = lane & 31 ->
      // pixel; staged [32 pixels][32 channels] bf16, stored as 16 bytes per lane
      {
        const f32x16& a = acc0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 lo = {a[4 * q], a[4 * q + 1]}, hi2 = {a[4 * q + 2], a[4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          *reinterpret_cast<uint2*>(stg + (lane & 31) * C3_STG_ROWB + (8 * q + 4 * hi) * 2) = pk;
        }
        const int qbase = mt * 32;
        int64_t ooff[2];
        bool okv[2];
        uint4 addv[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int row = i2 * 16 + (lane >> 2);
          const int q = qbase + row;
          const int r = fdiv(q, P.fd_pw), c = q - r * PW;
          okv[i2] = c < P.W && r < rows;
          ooff[i2] = ((int64_t)(n * P.H + h0 + r) * P.W + c) * 64 + coh * 32 + s_ch * 8;
          if (DGRAD) addv[i2] = (P.ADD && okv[i2]) ? *reinterpret_cast<const uint4*>(P.ADD + ooff[i2]) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int row = i2 * 16 + (lane >> 2);
          uint4 v = *reinterpret_cast<const uint4*>(stg + row * C3_STG_ROWB + s_ch * 16);
          if (okv[i2]) {
            if (!DGRAD && P.STATS) {
              const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float f = __uint_as_float((c & 1) ? (vw[c >> 1] & 0xFFFF0000u) : (vw[c >> 1] << 16));
                sy[c] += f; sq[c] = fmaf(f, f, sq[c]);
              }
            }
            if (DGRAD && P.ADD) {
              const uint4 q4 = addv[i2];
              v.x = add_bf16x2(v.x, q4.x); v.y = add_bf16x2(v.y, q4.y); v.z = add_bf16x2(v.z, q4.z); v.w = add_bf16x2(v.w, q4.w);
            }
            store16(P.Y + ooff[i2], v);
          }
        }
      }
    }
    if (issuer) C3_ISSUE_REST(6);
    C3_STAMP(6 + it * 5);
  }
  if (!DGRAD && P.STATS) {
    // Column sums of the bf16 outputs of ALL tiles this workgroup produced: ONE partial row per workgroup
    // (rigl_conv2d_stats_parts = the grid).  Every lane leaves its 16 sums in LDS; thread (k, channel) then adds the 4
    // M-tile phases x 16 lanes that own the channel's chunk in a fixed order -- deterministic, no shuffles (the xor tree
    // per tile cost 4 000 cycles of ds_bpermute per tile).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave is done with the patch buffers: they become scratch
    float* const sc = reinterpret_cast<float*>(smem_c3);      // [8 waves][64 lanes][16]
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[(wave * 64 + lane) * 16 + c] = sy[c]; sc[(wave * 64 + lane) * 16 + 8 + c] = sq[c]; }
    __syncthreads();
    if (tid < 128) {
      const int k = tid >> 6, co = tid & 63, ch2 = co >> 5, chunk = (co & 31) >> 3, j = co & 7;
      float s2 = 0.f;
      for (int m = 0; m < 4; ++m)
        for (int l = 0; l < 16; ++l) s2 += sc[(((m * 2 + ch2) * 64) + l * 4 + chunk) * 16 + k * 8 + j];
      P.STATS[((int64_t)blockIdx.x * 2 + k) * 64 + co] = s2;
    }
  }
  C3_STAMP(63);
#undef C3_ISSUE_BEGIN
#undef C3_ISSUE_SOME
#undef C3_ISSUE_REST
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------
struct C3WArgs {
  const uint16_t* X;     // [N][H][W][64]
  const uint16_t* DY;    // [N][H][W][64]
  float* SLAB;           // [grid][9][64][64]
  int N, H, W, PW, TH, tiles_h, tiles;
  int kp;                // reduction pixels per tile, TH * PW rounded up to 16
  int x_px;              // X patch pixels DMA-filled (kp + 2 * PW + 2 rounded up to 8; zeros behind the real patch)
  uint32_t x_bytes, dy_bytes;
  FastDiv fd_pw, fd_th;
};

// 8 waves: wave = tg * 4 + (cit * 2 + cot): the 32 x 32 block (cit, cot) of dW for the taps of group tg (0: taps 0..4, 1:
// taps 5..8) -- the two waves of a SIMD (w, w + 4) carry the nine taps of one block between them.  Persistent, one
// workgroup per CU; two LDS stages (X patch + dY tile each): the next tile's pieces are issued one per k-step while this
// tile is multiplied (the first version -- 4 waves, one stage, two workgroups per CU -- loaded, then multiplied).
template <int TG>
__device__ __forceinline__ void c3w_tile(const unsigned char* const xp, const unsigned char* const yp, const int KS,
                                         const int (&a_off)[5], const int y_off, f32x16 (&acc)[5]) {
  constexpr int NT = TG ? 4 : 5;
  // a k-step is 16 pixels = 2048 bytes further; the second half of a fragment's 8 pixels 4 pixels = 512 bytes further
  // (neither changes bit 1 of the pixel index, so the swizzled offsets a_off / y_off hold for every k-step)
  bf16x8 bfr[2], af[2][NT];
#define C3W_READ(kk_, buf_)                                                                              \
  {                                                                                                      \
    const unsigned char* const xk_ = xp + (kk_) * 2048;                                                  \
    const unsigned char* const yk_ = yp + (kk_) * 2048;                                                  \
    bfr[buf_] = lds_read_tr_pair(yk_ + y_off, yk_ + y_off + 512);                                        \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) af[buf_][t] = lds_read_tr_pair(xk_ + a_off[t], xk_ + a_off[t] + 512); \
  }
  C3W_READ(0, 0);
  for (int kk = 0; kk < KS; kk += 2) {
    if (kk + 1 < KS) C3W_READ(kk + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][t], bfr[0], acc[t], 0, 0, 0);
    if (kk + 1 < KS) {
      if (kk + 2 < KS) C3W_READ(kk + 2, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][t], bfr[1], acc[t], 0, 0, 0);
    }
  }
#undef C3W_READ
}

__global__ __launch_bounds__(C3_THREADS, 2) void k_c3x3_wgrad(C3WArgs P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_c3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2, cit = (wave >> 1) & 1, cot = wave & 1;
  const int PW = P.PW;
  const int stage_bytes = (P.x_px + P.kp) * 128;
  const __amdgpu_buffer_rsrc_t rsrcX = make_rsrc(P.X, P.x_bytes), rsrcY = make_rsrc(P.DY, P.dy_bytes);
  const u32x4 rsrcX4 = make_rsrc4(P.X, P.x_bytes), rsrcY4 = make_rsrc4(P.DY, P.dy_bytes);
  (void)rsrcX4; (void)rsrcY4; (void)rsrcX; (void)rsrcY;
  // transposing fragment reads (the k_wgrad_tr recipe): lane (g, j) of a 16-pixel k-step: pixel 8 * (g >> 1) + (j >> 2)
  // (+ 4 for the second read), 16-byte chunk 2 * (g & 1) + ((j >> 1) & 1) of the fragment's four, bytes (j & 1) * 8; the
  // row's 64-byte quads are XORed with bit 1 of the pixel index on the DMA's source side
  const int g = lane >> 4, j16 = lane & 15;
  const int t_pix = 8 * (g >> 1) + (j16 >> 2), t_chunk = 2 * (g & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
#define C3_TR_OFF(pix_, tile_) ((pix_) * 128 + ((((tile_) * 4 + t_chunk) ^ ((((pix_) >> 1) & 1) << 2)) << 4) + t_half)
  int a_off[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int tap = (tg ? 5 : 0) + (t < (tg ? 4 : 5) ? t : 0);
    a_off[t] = C3_TR_OFF(t_pix + (tap / 3) * PW + (tap % 3), cit);
  }
  const int y_off = C3_TR_OFF(t_pix, cot);
#undef C3_TR_OFF
  const int n_xi = P.x_px >> 3, n_pieces = n_xi + (P.kp >> 3);
  const int npx = (P.TH + 2) * PW;

  // pieces of a stage: 0 .. n_xi - 1 the X patch, then the dY tile; a wave takes every 8th
  int q_i = 1 << 30, q_n = 0, q_h0 = 0, q_rows = 0;
  unsigned char* q_dst = smem_c3;
#define C3W_BEGIN(t_, buf_)                                                                              \
  {                                                                                                      \
    q_n = fdiv((t_), P.fd_th); q_h0 = ((t_) - q_n * P.tiles_h) * P.TH;                                   \
    q_rows = P.H - q_h0 < P.TH ? P.H - q_h0 : P.TH;                                                      \
    q_dst = smem_c3 + (buf_) * stage_bytes; q_i = wave;                                                  \
  }
#define C3W_ISSUE_ONE()                                                                                  \
  {                                                                                                      \
    if (q_i < n_pieces) {                                                                                \
      const bool isx_ = q_i < n_xi;                                                                      \
      const int px = (isx_ ? q_i : q_i - n_xi) * 8 + (lane >> 3), slot = lane & 7;                       \
      const int chunk = slot ^ (((px >> 1) & 1) << 2);                                                   \
      const int pr = fdiv(px, P.fd_pw), pc = px - pr * PW;                                               \
      const int h = isx_ ? q_h0 - 1 + pr : q_h0 + pr, w = isx_ ? pc - 1 : pc;                            \
      const bool ok = isx_ ? (px < npx && (unsigned)h < (unsigned)P.H && (unsigned)w < (unsigned)P.W)    \
                           : (pr < q_rows && pc < P.W);                                                  \
      const int off = ((q_n * P.H + h) * P.W + w) * 64 + chunk * 8;                                      \
      const int boff = ok ? (int)((uint32_t)off * 2u) : (int)OOB;                                        \
      if (isx_) RIGL_DMA16(rsrcX, q_dst + q_i * 1024, boff); \
      else RIGL_DMA16(rsrcY, q_dst + q_i * 1024, boff); \
      q_i += 8;                                                                                          \
    }                                                                                                    \
  }
  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  int tile = blockIdx.x;
  if (tile < P.tiles) { C3W_BEGIN(tile, 0); while (q_i < n_pieces) C3W_ISSUE_ONE(); }
  const int KS = P.kp >> 4;
  for (int it = 0; tile < P.tiles; ++it, tile += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // stage `it & 1` landed everywhere; every wave is done with the other
    asm volatile("" ::: "memory");
    const bool more = tile + (int)gridDim.x < P.tiles;
    if (more) { C3W_BEGIN(tile + (int)gridDim.x, (it + 1) & 1); while (q_i < n_pieces) C3W_ISSUE_ONE(); }
    const unsigned char* const xp = smem_c3 + (it & 1) * stage_bytes;
    const unsigned char* const yp = xp + P.x_px * 128;
    if (tg == 0) c3w_tile<0>(xp, yp, KS, a_off, y_off, acc);
    else c3w_tile<1>(xp, yp, KS, a_off, y_off, acc);
  }
#undef C3W_BEGIN
#undef C3W_ISSUE_ONE
  // the workgroup's partial dW -> its slab: D row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) = ci, column lane & 31 = co
  float* const out = P.SLAB + (int64_t)blockIdx.x * (9 * 64 * 64);
  const int t0 = tg ? 5 : 0, nt = tg ? 4 : 5;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if (t < nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = cit * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), co = cot * 32 + (lane & 31);
        out[((t0 + t) * 64 + ci) * 64 + co] = acc[t][e];
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Legal: 3x3, stride 1, one pixel of padding on every side (ho == h, wo == w), 64 -> 64 channels, W + 2 <= 64.  "c3x3" = 0
// turns the kernels off (the layer then runs on the igemm / tr bodies in every entry point).
struct C3Geom { int th, tiles_h, mt_max, alloc_px, smem; };

static inline bool c3x3_legal(const RiglConvDesc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_top == 1 && d->pad_left == 1 &&
         d->cin == 64 && d->cout == 64 && d->ho == d->h && d->wo == d->w && d->w >= 24 && d->w <= 62 &&
         (int64_t)d->n * d->h * d->w >= 4096 && RIGL_TUNE("c3x3", 1) != 0;
}
// forward / dgrad tile height: the largest whose patch + zero tail fits ONE OF TWO patch buffers of a CU's LDS beside the
// eight staging areas and the statistics scratch
constexpr int C3_PATCH_BUDGET_PX = 544;
static_assert(2 * C3_PATCH_BUDGET_PX * 128 + 8 * C3_STG_WAVE + C3_RED_BYTES <= 160 * 1024, "LDS per CU");
static C3Geom c3x3_geom(const RiglConvDesc* d) {
  C3Geom g = {0, 0, 0, 0, 0};
  const int pw = d->w + 2;
  for (int th = d->h; th >= 1; --th) {
    const int mt = (th * pw + 31) / 32;
    const int alloc = (mt * 32 + 2 * pw + 2 + 7) / 8 * 8;
    if (alloc <= C3_PATCH_BUDGET_PX) { g.th = th; g.mt_max = mt; g.alloc_px = alloc; break; }
  }
  if (!g.th) return g;
  // (prefer a height that divides H when it costs at most one row)
  if (d->h % g.th && g.th > 1 && d->h % (g.th - 1) == 0) {
    g.th -= 1; g.mt_max = (g.th * pw + 31) / 32; g.alloc_px = (g.mt_max * 32 + 2 * pw + 2 + 7) / 8 * 8;
  }
  g.tiles_h = (d->h + g.th - 1) / g.th;
  g.smem = 2 * g.alloc_px * 128 + 8 * C3_STG_WAVE + C3_RED_BYTES;
  if (g.smem < g.alloc_px * 128 + 9 * 64 * 64 * 2) g.smem = g.alloc_px * 128 + 9 * 64 * 64 * 2;   // (the filter's prologue image)
  if (g.smem < 8 * 64 * 16 * 4) g.smem = 8 * 64 * 16 * 4;                                          // (the statistics scratch)
  return g;
}
template <bool DGRAD>
static bool c3x3_ready() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_c3x3<DGRAD>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  return ready;
}
static bool c3x3_wgrad_ready() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_c3x3_wgrad),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  return ready;
}
// Does this layer run on the kernels of this file?  ONE decision for every entry point (a layer's dX has the same bits from
// rigl_masked_conv2d_dgrad and rigl_masked_conv2d_bwd, and rigl_conv2d_stats_parts must describe the forward that runs).
static bool c3x3_use(const RiglConvDesc* d) {
  return c3x3_legal(d) && c3x3_geom(d).th > 0 && c3x3_ready<false>() && c3x3_ready<true>() && c3x3_wgrad_ready();
}
static inline int c3x3_grid(const RiglConvDesc* d) { const int t = d->n * c3x3_geom(d).tiles_h; return t < num_cus() ? t : num_cus(); }
static inline int32_t c3x3_stats_parts(const RiglConvDesc* d) { return c3x3_grid(d); }     // one partial per (persistent) workgroup

template <bool DGRAD>
static void launch_c3x3(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* wt, const rigl_bf16* addend, rigl_bf16* y,
                        float* stats, hipStream_t st) {
  const C3Geom g = c3x3_geom(d);
  C3Args a = {};
  a.X = x; a.WT = wt; a.Y = y; a.ADD = addend; a.STATS = stats;
  a.N = d->n; a.H = d->h; a.W = d->w; a.PW = d->w + 2; a.TH = g.th; a.tiles_h = g.tiles_h; a.tiles = d->n * g.tiles_h;
  a.alloc_px = g.alloc_px;
  a.x_bytes = (uint32_t)((size_t)d->n * d->h * d->w * 64 * 2);
  a.fd_pw = make_fastdiv(a.PW); a.fd_th = make_fastdiv(a.tiles_h);
#ifdef RIGL_C3_TRACE
  { const char* e = getenv("RIGL_C3_TRACE_PTR"); a.TRACE = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  const int grid = c3x3_grid(d);                                       // persistent: one workgroup per CU
  RIGL_K_LAUNCH(k_c3x3<DGRAD>, dim3((unsigned)grid), dim3(C3_THREADS), (unsigned)g.smem, st, a);
}

// weight gradient: TH = the largest height whose X patch + dY tile fit half a CU's LDS; two workgroups per CU, persistent
struct C3WGeom { int th, tiles_h, kp, x_px, smem, grid; };
static C3WGeom c3x3_wgrad_geom(const RiglConvDesc* d) {
  C3WGeom g = {0, 0, 0, 0, 0, 0};
  const int pw = d->w + 2;
  const int budget_px = (160 * 1024 / 2) / 128;        // one of two stages
  for (int th = d->h; th >= 1; --th) {
    const int kp = (th * pw + 15) / 16 * 16;
    const int x_px = (kp + 2 * pw + 2 + 7) / 8 * 8;
    if (x_px + kp <= budget_px) { g.th = th; g.kp = kp; g.x_px = x_px; break; }
  }
  if (!g.th) return g;
  g.tiles_h = (d->h + g.th - 1) / g.th;
  g.smem = 2 * (g.x_px + g.kp) * 128;
  const int tiles = d->n * g.tiles_h;
  g.grid = tiles < num_cus() ? tiles : num_cus();      // persistent: one workgroup per CU
  return g;
}
static inline size_t c3x3_wgrad_workspace(const RiglConvDesc* d) {
  return c3x3_legal(d) ? (size_t)c3x3_wgrad_geom(d).grid * 9 * 64 * 64 * 4 : 0;
}
static void launch_c3x3_wgrad(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, float* slabs, hipStream_t st) {
  const C3WGeom g = c3x3_wgrad_geom(d);
  C3WArgs a = {};
  a.X = x; a.DY = dy; a.SLAB = slabs;
  a.N = d->n; a.H = d->h; a.W = d->w; a.PW = d->w + 2; a.TH = g.th; a.tiles_h = g.tiles_h; a.tiles = d->n * g.tiles_h;
  a.kp = g.kp; a.x_px = g.x_px;
  a.x_bytes = a.dy_bytes = (uint32_t)((size_t)d->n * d->h * d->w * 64 * 2);
  a.fd_pw = make_fastdiv(a.PW); a.fd_th = make_fastdiv(a.tiles_h);
  RIGL_K_LAUNCH(k_c3x3_wgrad, dim3((unsigned)g.grid), dim3(C3_THREADS), (unsigned)g.smem, st, a);
}
