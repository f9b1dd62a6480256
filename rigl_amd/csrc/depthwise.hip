// K1d: dense depthwise convolution (depth multiplier 1) fwd / dgrad / wgrad for
// NHWC bf16 activations -- MobileNet-v1's unmasked depthwise 3x3 layers
// (contrib_layers.separable_conv2d(num_outputs=None),
// rigl/imagenet_resnet/mobilenetv1_model.py:81-92; SURVEY F7: the depthwise
// convs are NOT masked, only the 1x1 pointwise convs are).
//
// HBM-bound, no MFMA (9 MACs per output element): one thread owns 8 channels
// (16 B) of one pixel, neighbouring taps are served by L1/L2.  Weights are the
// fp32 HWIO tensor [kh][kw][C][1] = flat [kh*kw][C], read directly (no shadow).
// wgrad reduces over pixels with per-block partial sums combined in a fixed
// order (deterministic), like the BN statistics.
#include <stdlib.h>

#include "common.hpp"

namespace rigl {
namespace kdw {

constexpr int THREADS = 256;
constexpr int MAX_TAPS = 25;   // up to 5x5

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t pack2(float a, float b) { return f2bf(a) | (f2bf(b) << 16); }
__device__ __forceinline__ void unpack8(const uint4& v, float f[8]) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float o[8]) {
  uint4 r;
  r.x = pack2(o[0], o[1]); r.y = pack2(o[2], o[3]); r.z = pack2(o[4], o[5]); r.w = pack2(o[6], o[7]);
  return r;
}

// y[n,ho,wo,c] = sum_{r,s} x[n, ho*sh-pt+r, wo*sw-pl+s, c] * w[r,s,c]
__global__ __launch_bounds__(THREADS) void k_fwd(RiglConvDesc d, const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                 uint16_t* __restrict__ y) {
  const int cg = d.cin / 8;
  const int64_t total = (int64_t)d.n * d.ho * d.wo * cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    const int c0 = (int)(i % cg) * 8;
    int64_t p = i / cg;
    const int wo = (int)(p % d.wo); p /= d.wo;
    const int ho = (int)(p % d.ho);
    const int n = (int)(p / d.ho);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < d.kh; ++r) {
      const int hi = ho * d.stride_h - d.pad_top + r;
      if ((unsigned)hi >= (unsigned)d.h) continue;
      for (int s = 0; s < d.kw; ++s) {
        const int wi = wo * d.stride_w - d.pad_left + s;
        if ((unsigned)wi >= (unsigned)d.w) continue;
        float xv[8];
        unpack8(*reinterpret_cast<const uint4*>(x + ((int64_t)(n * d.h + hi) * d.w + wi) * d.cin + c0), xv);
        const float* wp = w + (int64_t)(r * d.kw + s) * d.cin + c0;
        const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
        acc[0] = fmaf(xv[0], w0.x, acc[0]); acc[1] = fmaf(xv[1], w0.y, acc[1]);
        acc[2] = fmaf(xv[2], w0.z, acc[2]); acc[3] = fmaf(xv[3], w0.w, acc[3]);
        acc[4] = fmaf(xv[4], w1.x, acc[4]); acc[5] = fmaf(xv[5], w1.y, acc[5]);
        acc[6] = fmaf(xv[6], w1.z, acc[6]); acc[7] = fmaf(xv[7], w1.w, acc[7]);
      }
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8(acc);
  }
}

// dx[n,h,w,c] = sum_{r,s : (h+pt-r) % sh == 0, ...} dy[n,(h+pt-r)/sh,(w+pl-s)/sw,c] * w[r,s,c]
__global__ __launch_bounds__(THREADS) void k_dgrad(RiglConvDesc d, const uint16_t* __restrict__ dy, const float* __restrict__ w,
                                                   uint16_t* __restrict__ dx) {
  const int cg = d.cin / 8;
  const int64_t total = (int64_t)d.n * d.h * d.w * cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    const int c0 = (int)(i % cg) * 8;
    int64_t p = i / cg;
    const int wi = (int)(p % d.w); p /= d.w;
    const int hi = (int)(p % d.h);
    const int n = (int)(p / d.h);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < d.kh; ++r) {
      const int th = hi + d.pad_top - r;
      if (th < 0 || th % d.stride_h) continue;
      const int ho = th / d.stride_h;
      if (ho >= d.ho) continue;
      for (int s = 0; s < d.kw; ++s) {
        const int tw = wi + d.pad_left - s;
        if (tw < 0 || tw % d.stride_w) continue;
        const int wo = tw / d.stride_w;
        if (wo >= d.wo) continue;
        float gv[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + ((int64_t)(n * d.ho + ho) * d.wo + wo) * d.cin + c0), gv);
        const float* wp = w + (int64_t)(r * d.kw + s) * d.cin + c0;
        const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
        acc[0] = fmaf(gv[0], w0.x, acc[0]); acc[1] = fmaf(gv[1], w0.y, acc[1]);
        acc[2] = fmaf(gv[2], w0.z, acc[2]); acc[3] = fmaf(gv[3], w0.w, acc[3]);
        acc[4] = fmaf(gv[4], w1.x, acc[4]); acc[5] = fmaf(gv[5], w1.y, acc[5]);
        acc[6] = fmaf(gv[6], w1.z, acc[6]); acc[7] = fmaf(gv[7], w1.w, acc[7]);
      }
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8(acc);
  }
}

// partial[part][tap][c] = sum over this part's output pixels of x * dy
struct WGeom { int tpr, rpb, parts, cg; int64_t M, rows_per_part; };

__global__ __launch_bounds__(THREADS) void k_wgrad_partial(RiglConvDesc d, WGeom G, const uint16_t* __restrict__ x,
                                                           const uint16_t* __restrict__ dy, float* __restrict__ partial) {
  __shared__ float red[THREADS][9];
  const int tx = threadIdx.x % G.tpr, ty = threadIdx.x / G.tpr;
  const int cgi = blockIdx.y * G.tpr + tx;
  const bool c_ok = cgi < G.cg;
  const int taps = d.kh * d.kw;
  const int64_t r0 = (int64_t)blockIdx.x * G.rows_per_part;
  int64_t r1 = r0 + G.rows_per_part;
  if (r1 > G.M) r1 = G.M;
  // one tap at a time keeps the register footprint at 8 accumulators
  for (int tap = 0; tap < taps; ++tap) {
    const int r = tap / d.kw, s = tap % d.kw;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c_ok) {
      for (int64_t m = r0 + ty; m < r1; m += G.rpb) {
        const int wo = (int)(m % d.wo);
        const int64_t t = m / d.wo;
        const int ho = (int)(t % d.ho), n = (int)(t / d.ho);
        const int hi = ho * d.stride_h - d.pad_top + r, wi = wo * d.stride_w - d.pad_left + s;
        if ((unsigned)hi >= (unsigned)d.h || (unsigned)wi >= (unsigned)d.w) continue;
        float xv[8], gv[8];
        unpack8(*reinterpret_cast<const uint4*>(x + ((int64_t)(n * d.h + hi) * d.w + wi) * d.cin + cgi * 8), xv);
        unpack8(*reinterpret_cast<const uint4*>(dy + m * d.cin + cgi * 8), gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], gv[j], acc[j]);
      }
    }
    __syncthreads();
    // (own sums stay in registers, the partner's are read with one wait per tree round: bn.hip k_reduce; same additions)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int st = G.rpb >> 1; st > 0; st >>= 1) {
      if (ty < st) {
        float oth[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) oth[j] = red[threadIdx.x + st * G.tpr][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += oth[j];
        if (st > 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
        }
      }
      if (st > 1) __syncthreads();
    }
    if (ty == 0 && c_ok) {
      float* p = partial + ((int64_t)blockIdx.x * taps + tap) * d.cin + cgi * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = acc[j];
    }
  }
}

__global__ __launch_bounds__(THREADS) void k_wgrad_final(const float* __restrict__ partial, float* __restrict__ dw, int n_out,
                                                         int parts) {
  const int i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n_out) return;
  double a = 0.0;
  for (int p = 0; p < parts; ++p) a += (double)partial[(int64_t)p * n_out + i];
  dw[i] = (float)a;
}


// ---------------------------------------------------------------------------------------------------------
// 3x3 specialisations (stride 1 / 2): what MobileNet-v1 runs.  The generic kernels above spend their time in
// 64-bit index divisions, runtime tap loops and (wgrad) nine passes over the data with <= 128 workgroups:
// 408 GB/s of algorithmic traffic, 9.4 ms of a 13.5 ms MobileNet step (profiles/r2).  Here a thread owns 8
// channels (16 B) of a strip of TW = 4 consecutive produced pixels: the 3 x ((TW - 1) S + 3) gathered pixels it
// needs are fetched with buffer loads (out-of-image taps = out-of-range offsets, the hardware returns zeros:
// no branches), unpacked once, and every gathered pixel feeds up to three outputs from registers.  Index math is
// 32-bit with launch-time magic-number division.  Accumulation order per output is tap-major (r, s) as in the
// generic kernels, so forward / dgrad results are bit-identical to them.
constexpr int TW = 4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr uint32_t OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
struct FastDiv { uint32_t magic, shift; };     // q = umulhi(n, magic) >> shift, exact for 0 <= n < 2^31 (magic == 0: d == 1)
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

// XCD-aware block remap (hardware hands block b to XCD b % 8): consecutive logical blocks -- neighbouring image rows,
// which share their halo rows -- land on one XCD's L2 instead of eight.  Bijective; speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

struct G3 {
  int N, GH, GW;        // gathered tensor (x for fwd / wgrad, dy for dgrad)
  int PH, PW;           // produced tensor rows / columns (y | dx); for wgrad: the dy tensor
  int C, cg, SW;        // channels, 8-channel groups, strips per produced row
  int pt, pl;           // gathered row / column of tap (0, 0) for produced pixel (0, 0) is (-pt, -pl)
  int total;            // work items: N * PH * SW * cg (fwd, dgrad) | N * PH * SW (wgrad)
  uint32_t g_bytes, p_bytes;
  FastDiv fd_cg, fd_sw, fd_ph;
};

// produced[n, p, q, c] = sum_{r, s} gathered[n, p S - pt + r, q S - pl + s, c] * w[tap(r, s), c]
// FLIP (the stride-1 dgrad): tap(r, s) = (2 - r, 2 - s).
// One work item (strip of TW produced pixels x 8 channels).  STATS: also add the bf16-rounded outputs and their squares to
// q0 / q1 (the batch-norm statistics of the tensor as the batch norm will read it).
template <int S, bool FLIP, bool STATS>
__device__ __forceinline__ void fwd3_item(const G3& g, int i, const uint16_t* __restrict__ in, const float* __restrict__ w,
                                          uint16_t* __restrict__ out, float q0[8], float q1[8]) {
  constexpr int NC = (TW - 1) * S + 3;
  const int t = fdiv(i, g.fd_cg), cgi = i - t * g.cg;
  const int t2 = fdiv(t, g.fd_sw), strip = t - t2 * g.SW;
  const int n = fdiv(t2, g.fd_ph), po = t2 - n * g.PH;
  const int c0 = cgi * 8, q0s = strip * TW;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, g.g_bytes);
  const int gw0 = q0s * S - g.pl;
  float acc[TW][8];
#pragma unroll
  for (int j = 0; j < TW; ++j)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int gh = po * S - g.pt + r;
    const bool rok = (unsigned)gh < (unsigned)g.GH;
    const int rowbase = ((n * g.GH + gh) * g.GW) * g.C + c0;     // element index of pixel (gh, 0), channel c0
    uint4 px[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int gw = gw0 + j;
      const bool ok = rok && (unsigned)gw < (unsigned)g.GW;
      px[j] = buf_load16(rs, ok ? (uint32_t)(rowbase + gw * g.C) * 2u : OOB);
    }
    float wv[3][8];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = FLIP ? (2 - r) * 3 + (2 - s) : r * 3 + s;
      const float* wp = w + tap * g.C + c0;
      const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
      wv[s][0] = w0.x; wv[s][1] = w0.y; wv[s][2] = w0.z; wv[s][3] = w0.w;
      wv[s][4] = w1.x; wv[s][5] = w1.y; wv[s][6] = w1.z; wv[s][7] = w1.w;
    }
    // pixel-major: a gathered pixel is unpacked once and feeds every (output j, tap s) with j S + s == k; for a fixed
    // output the taps still arrive in the order s = 0, 1, 2 (k ascending), i.e. tap-major like the generic kernel
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float xv[8];
      unpack8(px[k], xv);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (k - s < 0 || (k - s) % S != 0 || (k - s) / S >= TW) continue;
        const int j = (k - s) / S;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = fmaf(xv[c], wv[s][c], acc[j][c]);
      }
    }
  }
  const int obase = ((n * g.PH + po) * g.PW) * g.C + c0;
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    if (q0s + j < g.PW) {
      const uint4 pk = pack8(acc[j]);
      *reinterpret_cast<uint4*>(out + obase + (q0s + j) * g.C) = pk;
      if (STATS) {
        float v[8];
        unpack8(pk, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) { q0[c] += v[c]; q1[c] = fmaf(v[c], v[c], q1[c]); }
      }
    }
  }
}

template <int S, bool FLIP>
__global__ __launch_bounds__(THREADS) void k_fwd3(G3 g, const uint16_t* __restrict__ in, const float* __restrict__ w,
                                                  uint16_t* __restrict__ out) {
  const int i = (int)xcd_remap(blockIdx.x, gridDim.x) * THREADS + threadIdx.x;
  if (i >= g.total) return;
  fwd3_item<S, FLIP, false>(g, i, in, w, out, nullptr, nullptr);
}

// The forward with the batch-norm statistics of its output in the epilogue (MobileNet-v1 follows every depthwise conv with
// a batch norm, mobilenetv1_model.py:188-198, whose first pass would re-read the tensor): a workgroup walks
// stats_reps x 256 consecutive items -- the items of a thread are a multiple of 256 apart, so they share one channel
// group ((channels / 8) divides 256) -- and leaves one partial row [2][C]: sum y, sum y^2 of the bf16 outputs, combined
// over the threads of each channel group in a fixed order (deterministic).  Consumed by rigl_bn_fwd_stats.
inline int stats_reps() {      // items per thread of the statistics forward (RIGL_DW_STATS_REPS, default 2: a wave waits for its stores between items)
  static const int v = [] { const char* e = getenv("RIGL_DW_STATS_REPS"); const int r = e ? atoi(e) : 2; return r < 1 ? 1 : r > 64 ? 64 : r; }();
  return v;
}
template <int S>
__global__ __launch_bounds__(THREADS) void k_fwd3_stats(G3 g, int stats_reps, const uint16_t* __restrict__ in,
                                                        const float* __restrict__ w, uint16_t* __restrict__ out,
                                                        float* __restrict__ stats) {
  __shared__ float red[THREADS / 64][64][17];
  const int b = (int)xcd_remap(blockIdx.x, gridDim.x);
  float q[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) q[c] = 0.f;
#pragma unroll 1
  for (int rep = 0; rep < stats_reps; ++rep) {
    const int i = (b * stats_reps + rep) * THREADS + threadIdx.x;
    if (i < g.total) fwd3_item<S, false, true>(g, i, in, w, out, q, q + 8);
  }
  // lanes cg apart hold the same channel group: butterfly over them first, then one LDS row per (wave, channel group)
  const int span = g.cg < 64 ? g.cg : 64;
  for (int off = 32; off >= span; off >>= 1)
#pragma unroll
    for (int c = 0; c < 16; ++c) q[c] += __shfl_xor(q[c], off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < span)
#pragma unroll
    for (int c = 0; c < 16; ++c) red[wave][lane][c] = q[c];
  __syncthreads();
  const int wps = g.cg > 64 ? g.cg >> 6 : 1;        // waves one pass over the channel groups takes
  for (int p = threadIdx.x; p < g.cg * 16; p += THREADS) {
    const int cgq = p >> 4, v = p & 15;
    float sum = 0.f;
    for (int r = cgq >> 6; r < THREADS / 64; r += wps) sum += red[r][cgq & 63][v];
    stats[((int64_t)b * 2 + (v >> 3)) * g.C + cgq * 8 + (v & 7)] = sum;
  }
}


// Stride-2 dgrad: dx[n, h, w, c] = sum over the taps whose parity matches, dy[n, (h + pt - r) / 2, (w + pl - s) / 2, c] * w[r, s, c].
// A strip of 4 dx pixels (w0 % 4 == 0) touches 3 dy columns; which (pixel, tap) pairs meet which column depends only
// on the parity of pl (template), rows r = {0, 2} or {1} on the parity of h + pt.
template <int PLODD>
__global__ __launch_bounds__(THREADS) void k_dgrad3s2(G3 g, const uint16_t* __restrict__ dy, const float* __restrict__ w,
                                                      uint16_t* __restrict__ dx) {
  const int i = (int)xcd_remap(blockIdx.x, gridDim.x) * THREADS + threadIdx.x;
  if (i >= g.total) return;
  const int t = fdiv(i, g.fd_cg), cgi = i - t * g.cg;
  const int t2 = fdiv(t, g.fd_sw), strip = t - t2 * g.SW;
  const int n = fdiv(t2, g.fd_ph), h = t2 - n * g.PH;
  const int c0 = cgi * 8, w0 = strip * TW;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(dy, g.g_bytes);
  const int e0 = w0 + g.pl - 2 + PLODD;             // first even tw the strip can use (may be negative)
  const int wo_b = e0 >> 1;                          // arithmetic shift: e0 is even
  float acc[TW][8];
#pragma unroll
  for (int j = 0; j < TW; ++j)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
  const int par = (h + g.pt) & 1;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if ((r & 1) != par) continue;                    // th = h + pt - r must be even
    const int th = h + g.pt - r;
    const int ho = th >> 1;
    const bool rok = th >= 0 && ho < g.GH;
    const int rowbase = ((n * g.GH + ho) * g.GW) * g.C + c0;
    float gv[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int wo = wo_b + k;
      const bool ok = rok && (unsigned)wo < (unsigned)g.GW;
      unpack8(buf_load16(rs, ok ? (uint32_t)(rowbase + wo * g.C) * 2u : OOB), gv[k]);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* wp = w + (r * 3 + s) * g.C + c0;
      const float4 wa = *reinterpret_cast<const float4*>(wp), wb = *reinterpret_cast<const float4*>(wp + 4);
      const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        if (((j - s - PLODD) & 1) != 0) continue;    // tw = w0 + j + pl - s must be even (compile-time)
        const int k = (j - s + 2 - PLODD) / 2;       // dy column slot: 0..2
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = fmaf(gv[k][c], wv[c], acc[j][c]);
      }
    }
  }
  const int obase = ((n * g.PH + h) * g.PW) * g.C + c0;
#pragma unroll
  for (int j = 0; j < TW; ++j)
    if (w0 + j < g.PW) *reinterpret_cast<uint4*>(dx + obase + (w0 + j) * g.C) = pack8(acc[j]);
}

// Weight gradient, all nine taps at once: 72 fp32 accumulators per thread; x and dy are each read ONCE
// (the generic kernel walks the data once per tap).  partial[part][tap][c], combined by k_wgrad_final2.
struct WG3 { int tpr, rpb, parts, items_per_part; };
template <int S>
__global__ __launch_bounds__(THREADS, 2) void k_wgrad3(G3 g, WG3 G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                    float* __restrict__ partial) {
  constexpr int NC = (TW - 1) * S + 3;
  __shared__ float red[THREADS][9];
  const int tx = threadIdx.x % G.tpr, ty = threadIdx.x / G.tpr;
  const int cgi = blockIdx.y * G.tpr + tx;
  const bool c_ok = cgi < g.cg;
  const int c0 = cgi * 8;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, g.g_bytes), rd = make_rsrc(dy, g.p_bytes);
  float acc[9][8];
#pragma unroll
  for (int tpi = 0; tpi < 9; ++tpi)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[tpi][c] = 0.f;
  const int part = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = part * G.items_per_part;
  int m1 = m0 + G.items_per_part;
  if (m1 > g.total) m1 = g.total;
  if (c_ok) {
    for (int m = m0 + ty; m < m1; m += G.rpb) {
      const int t2 = fdiv(m, g.fd_sw), strip = m - t2 * g.SW;
      const int n = fdiv(t2, g.fd_ph), po = t2 - n * g.PH;
      const int q0 = strip * TW;
      float dv[TW][8];
      const int dbase = ((n * g.PH + po) * g.PW) * g.C + c0;
#pragma unroll
      for (int j = 0; j < TW; ++j)
        unpack8(buf_load16(rd, (q0 + j < g.PW) ? (uint32_t)(dbase + (q0 + j) * g.C) * 2u : OOB), dv[j]);
      const int gw0 = q0 * S - g.pl;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int gh = po * S - g.pt + r;
        const bool rok = (unsigned)gh < (unsigned)g.GH;
        const int rowbase = ((n * g.GH + gh) * g.GW) * g.C + c0;
        uint4 px[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int gw = gw0 + k;
          const bool ok = rok && (unsigned)gw < (unsigned)g.GW;
          px[k] = buf_load16(rx, ok ? (uint32_t)(rowbase + gw * g.C) * 2u : OOB);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          float xv[8];
          unpack8(px[k], xv);
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            if (k - s < 0 || (k - s) % S != 0 || (k - s) / S >= TW) continue;
            const int j = (k - s) / S;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[r * 3 + s][c] = fmaf(xv[c], dv[j][c], acc[r * 3 + s][c]);
          }
        }
      }
    }
  }
  // combine the item-lanes of each channel group, 8 channels of all 9 taps at a time... one channel per round
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    __syncthreads();
    float own[9];
#pragma unroll
    for (int tpi = 0; tpi < 9; ++tpi) { own[tpi] = acc[tpi][c]; red[threadIdx.x][tpi] = own[tpi]; }
    __syncthreads();
    for (int st = G.rpb >> 1; st > 0; st >>= 1) {
      if (ty < st) {
        float oth[9];
#pragma unroll
        for (int tpi = 0; tpi < 9; ++tpi) oth[tpi] = red[threadIdx.x + st * G.tpr][tpi];
#pragma unroll
        for (int tpi = 0; tpi < 9; ++tpi) own[tpi] += oth[tpi];
        if (st > 1) {
#pragma unroll
          for (int tpi = 0; tpi < 9; ++tpi) red[threadIdx.x][tpi] = own[tpi];
        }
      }
      if (st > 1) __syncthreads();
    }
    if (ty == 0 && c_ok) {
#pragma unroll
      for (int tpi = 0; tpi < 9; ++tpi) partial[((int64_t)part * 9 + tpi) * g.C + c0 + c] = own[tpi];
    }
  }
}

// dw[i] = sum over parts of partial[p][i], fixed order: 16 part-lanes per output (lane-strided, then lanes ascending), in double.
__global__ __launch_bounds__(THREADS) void k_wgrad_final2(const float* __restrict__ partial, float* __restrict__ dw, int n_out,
                                                          int parts) {
  __shared__ double acc[16][17];
  const int ol = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ol;
  double a = 0.0;
  if (i < n_out)
    for (int p = pl; p < parts; p += 16) a += (double)partial[(int64_t)p * n_out + i];
  acc[pl][ol] = a;
  __syncthreads();
  if (pl == 0 && i < n_out) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k][ol];
    dw[i] = (float)s;
  }
}

static bool use3(const RiglConvDesc* d) {
  static const bool on = [] { const char* e = getenv("RIGL_DW3"); return e ? atoi(e) != 0 : true; }();
  if (!on || d->kh != 3 || d->kw != 3 || d->stride_h != d->stride_w || (d->stride_h != 1 && d->stride_h != 2)) return false;
  const int64_t xb = (int64_t)d->n * d->h * d->w * d->cin * 2, yb = (int64_t)d->n * d->ho * d->wo * d->cin * 2;
  const int64_t items = (int64_t)d->n * (d->h > d->ho ? d->h : d->ho) * (((d->w > d->wo ? d->w : d->wo) + TW - 1) / TW) * (d->cin / 8);
  return xb < (int64_t(1) << 31) && yb < (int64_t(1) << 31) && items < (int64_t(1) << 31);
}

// gathered = [N][gh][gw][C], produced = [N][ph][pw][C]
static G3 make_g3(const RiglConvDesc* d, int gh, int gw, int ph, int pw, int pt, int pl, bool with_cg) {
  G3 g;
  g.N = d->n; g.GH = gh; g.GW = gw; g.PH = ph; g.PW = pw; g.C = d->cin; g.cg = d->cin / 8;
  g.SW = (pw + TW - 1) / TW;
  g.pt = pt; g.pl = pl;
  g.total = d->n * ph * g.SW * (with_cg ? g.cg : 1);
  g.g_bytes = (uint32_t)((int64_t)d->n * gh * gw * d->cin * 2);
  g.p_bytes = (uint32_t)((int64_t)d->n * ph * pw * d->cin * 2);
  g.fd_cg = make_fastdiv(g.cg); g.fd_sw = make_fastdiv(g.SW); g.fd_ph = make_fastdiv(ph);
  return g;
}

static WG3 make_wg3(const G3& g) {
  static const int max_parts = [] { const char* e = getenv("RIGL_DW_PARTS"); return e ? atoi(e) : 512; }();
  WG3 G;
  int tpr = 1;
  while (tpr < g.cg && tpr < THREADS) tpr <<= 1;
  G.tpr = tpr; G.rpb = THREADS / tpr;
  int64_t parts = ((int64_t)g.total + (int64_t)G.rpb * 4 - 1) / ((int64_t)G.rpb * 4);     // >= 4 strips per item-lane
  const int64_t cap_bytes = (int64_t(8) << 20) / ((int64_t)9 * g.C * 4);                   // <= 8 MB of partial sums
  if (parts > cap_bytes) parts = cap_bytes;
  if (parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  int64_t ipp = ((int64_t)g.total + parts - 1) / parts;
  ipp = (ipp + G.rpb - 1) / G.rpb * G.rpb;
  G.items_per_part = (int)ipp;
  G.parts = (int)(((int64_t)g.total + ipp - 1) / ipp);
  return G;
}

static unsigned stream_grid(int64_t total) {
  int64_t b = (total + THREADS * 2 - 1) / (THREADS * 2);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static WGeom make_wgeom(const RiglConvDesc* d) {
  WGeom g;
  g.M = (int64_t)d->n * d->ho * d->wo;
  g.cg = d->cin / 8;
  int tpr = 1;
  while (tpr < g.cg && tpr < THREADS) tpr <<= 1;
  g.tpr = tpr; g.rpb = THREADS / tpr;
  int64_t parts = (g.M + (int64_t)g.rpb * 32 - 1) / ((int64_t)g.rpb * 32);
  if (parts > 128) parts = 128;
  if (parts < 1) parts = 1;
  int64_t rpp = (g.M + parts - 1) / parts;
  rpp = (rpp + g.rpb - 1) / g.rpb * g.rpb;
  g.rows_per_part = rpp;
  g.parts = (int)((g.M + rpp - 1) / rpp);
  return g;
}

static int check(const RiglConvDesc* d, const char* who) {
  if (!d) return fail(RIGL_EINVAL, "%s: NULL descriptor", who);
  if (d->cin != d->cout) return fail(RIGL_EINVAL, "%s: depth multiplier 1 only (cin == cout)", who);
  if (d->cin % 8) return fail(RIGL_EUNSUPPORTED, "%s: channels %% 8 != 0", who);
  if (d->kh * d->kw > MAX_TAPS) return fail(RIGL_EUNSUPPORTED, "%s: kernel too large", who);
  if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->ho <= 0 || d->wo <= 0 || d->stride_h <= 0 || d->stride_w <= 0)
    return fail(RIGL_EINVAL, "%s: non-positive dimension", who);
  return RIGL_OK;
}

}  // namespace kdw
}  // namespace rigl

extern "C" {

size_t rigl_depthwise_conv2d_workspace_bytes(const RiglConvDesc* d) {
  if (!d || d->cin % 8) return 0;
  if (rigl::kdw::use3(d)) {
    const rigl::kdw::G3 g3 = rigl::kdw::make_g3(d, d->h, d->w, d->ho, d->wo, d->pad_top, d->pad_left, false);
    return rigl::align_up((size_t)rigl::kdw::make_wg3(g3).parts * 9 * d->cin * 4, 256);
  }
  rigl::kdw::WGeom g = rigl::kdw::make_wgeom(d);
  return rigl::align_up((size_t)g.parts * d->kh * d->kw * d->cin * 4, 256);
}

// Partial rows the forward leaves with rigl_depthwise_conv2d_fwd_stats (0: this shape has no statistics epilogue).
int32_t rigl_depthwise_conv2d_stats_parts(const RiglConvDesc* d) {
  using namespace rigl;
  if (!d || kdw::check(d, "rigl_depthwise_conv2d_stats_parts") || !kdw::use3(d)) return 0;
  const int cg = d->cin / 8;
  if (cg > kdw::THREADS || kdw::THREADS % cg) return 0;
  const kdw::G3 g = kdw::make_g3(d, d->h, d->w, d->ho, d->wo, d->pad_top, d->pad_left, true);
  const int64_t per = (int64_t)kdw::THREADS * kdw::stats_reps();
  return (int32_t)((g.total + per - 1) / per);
}

int rigl_depthwise_conv2d_fwd_stats(const RiglConvDesc* d, const rigl_bf16* x, const float* w, rigl_bf16* y, float* stats,
                                    size_t stats_floats, rigl_stream_t stream) {
  using namespace rigl;
  if (!stats) return rigl_depthwise_conv2d_fwd(d, x, w, y, stream);
  int rc = kdw::check(d, "rigl_depthwise_conv2d_fwd_stats");
  if (rc) return rc;
  if (!x || !w || !y) return fail(RIGL_EINVAL, "rigl_depthwise_conv2d_fwd_stats: NULL tensor");
  const int32_t parts = rigl_depthwise_conv2d_stats_parts(d);
  if (parts <= 0) return fail(RIGL_EUNSUPPORTED, "rigl_depthwise_conv2d_fwd_stats: no statistics epilogue for this shape");
  if (stats_floats < (size_t)parts * 2 * d->cin)
    return fail(RIGL_EWORKSPACE, "rigl_depthwise_conv2d_fwd_stats: stats buffer %zu floats < %zu", stats_floats, (size_t)parts * 2 * d->cin);
  ProfScope prof(PROF_DEPTHWISE, as_stream(stream));
  const kdw::G3 g = kdw::make_g3(d, d->h, d->w, d->ho, d->wo, d->pad_top, d->pad_left, true);
  if (d->stride_h == 1) hipLaunchKernelGGL(kdw::k_fwd3_stats<1>, dim3((unsigned)parts), dim3(kdw::THREADS), 0, as_stream(stream), g, kdw::stats_reps(), x, w, y, stats);
  else hipLaunchKernelGGL(kdw::k_fwd3_stats<2>, dim3((unsigned)parts), dim3(kdw::THREADS), 0, as_stream(stream), g, kdw::stats_reps(), x, w, y, stats);
  RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_fwd_stats");
  return RIGL_OK;
}

int rigl_depthwise_conv2d_fwd(const RiglConvDesc* d, const rigl_bf16* x, const float* w, rigl_bf16* y,
                              rigl_stream_t stream) {
  using namespace rigl;
  int rc = kdw::check(d, "rigl_depthwise_conv2d_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return fail(RIGL_EINVAL, "rigl_depthwise_conv2d_fwd: NULL tensor");
  ProfScope prof(PROF_DEPTHWISE, as_stream(stream));
  if (kdw::use3(d)) {
    const kdw::G3 g = kdw::make_g3(d, d->h, d->w, d->ho, d->wo, d->pad_top, d->pad_left, true);
    const dim3 grid((unsigned)((g.total + kdw::THREADS - 1) / kdw::THREADS));
    if (d->stride_h == 1) hipLaunchKernelGGL((kdw::k_fwd3<1, false>), grid, dim3(kdw::THREADS), 0, as_stream(stream), g, x, w, y);
    else hipLaunchKernelGGL((kdw::k_fwd3<2, false>), grid, dim3(kdw::THREADS), 0, as_stream(stream), g, x, w, y);
    RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_fwd");
    return RIGL_OK;
  }
  hipLaunchKernelGGL(kdw::k_fwd, dim3(kdw::stream_grid((int64_t)d->n * d->ho * d->wo * d->cin / 8)), dim3(kdw::THREADS), 0,
                     as_stream(stream), *d, x, w, y);
  RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_fwd");
  return RIGL_OK;
}

int rigl_depthwise_conv2d_dgrad(const RiglConvDesc* d, const rigl_bf16* dy, const float* w, rigl_bf16* dx,
                                rigl_stream_t stream) {
  using namespace rigl;
  int rc = kdw::check(d, "rigl_depthwise_conv2d_dgrad");
  if (rc) return rc;
  if (!dy || !w || !dx) return fail(RIGL_EINVAL, "rigl_depthwise_conv2d_dgrad: NULL tensor");
  ProfScope prof(PROF_DEPTHWISE, as_stream(stream));
  if (kdw::use3(d)) {
    if (d->stride_h == 1) {       // a correlation with the flipped filter: the forward body, gathering dy
      const kdw::G3 g = kdw::make_g3(d, d->ho, d->wo, d->h, d->w, 2 - d->pad_top, 2 - d->pad_left, true);
      const dim3 grid((unsigned)((g.total + kdw::THREADS - 1) / kdw::THREADS));
      hipLaunchKernelGGL((kdw::k_fwd3<1, true>), grid, dim3(kdw::THREADS), 0, as_stream(stream), g, dy, w, dx);
    } else {
      const kdw::G3 g = kdw::make_g3(d, d->ho, d->wo, d->h, d->w, d->pad_top, d->pad_left, true);
      const dim3 grid((unsigned)((g.total + kdw::THREADS - 1) / kdw::THREADS));
      if (d->pad_left & 1) hipLaunchKernelGGL(kdw::k_dgrad3s2<1>, grid, dim3(kdw::THREADS), 0, as_stream(stream), g, dy, w, dx);
      else hipLaunchKernelGGL(kdw::k_dgrad3s2<0>, grid, dim3(kdw::THREADS), 0, as_stream(stream), g, dy, w, dx);
    }
    RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_dgrad");
    return RIGL_OK;
  }
  hipLaunchKernelGGL(kdw::k_dgrad, dim3(kdw::stream_grid((int64_t)d->n * d->h * d->w * d->cin / 8)), dim3(kdw::THREADS), 0,
                     as_stream(stream), *d, dy, w, dx);
  RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_dgrad");
  return RIGL_OK;
}

int rigl_depthwise_conv2d_wgrad(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, float* dw,
                                void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  int rc = kdw::check(d, "rigl_depthwise_conv2d_wgrad");
  if (rc) return rc;
  if (!x || !dy || !dw) return fail(RIGL_EINVAL, "rigl_depthwise_conv2d_wgrad: NULL tensor");
  const size_t need = rigl_depthwise_conv2d_workspace_bytes(d);
  if (!workspace || workspace_bytes < need) return fail(RIGL_EWORKSPACE, "rigl_depthwise_conv2d_wgrad: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_DEPTHWISE, st);
  float* partial = static_cast<float*>(workspace);
  if (kdw::use3(d)) {
    const kdw::G3 g3 = kdw::make_g3(d, d->h, d->w, d->ho, d->wo, d->pad_top, d->pad_left, false);
    const kdw::WG3 wg = kdw::make_wg3(g3);
    const dim3 grid3((unsigned)wg.parts, (unsigned)((g3.cg + wg.tpr - 1) / wg.tpr));
    if (d->stride_h == 1) hipLaunchKernelGGL(kdw::k_wgrad3<1>, grid3, dim3(kdw::THREADS), 0, st, g3, wg, x, dy, partial);
    else hipLaunchKernelGGL(kdw::k_wgrad3<2>, grid3, dim3(kdw::THREADS), 0, st, g3, wg, x, dy, partial);
    const int n_out3 = 9 * d->cin;
    hipLaunchKernelGGL(kdw::k_wgrad_final2, dim3((unsigned)((n_out3 + 15) / 16)), dim3(kdw::THREADS), 0, st, partial, dw, n_out3, wg.parts);
    RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_wgrad");
    return RIGL_OK;
  }
  kdw::WGeom g = kdw::make_wgeom(d);
  dim3 grid((unsigned)g.parts, (unsigned)((g.cg + g.tpr - 1) / g.tpr));
  hipLaunchKernelGGL(kdw::k_wgrad_partial, grid, dim3(kdw::THREADS), 0, st, *d, g, x, dy, partial);
  const int n_out = d->kh * d->kw * d->cin;
  hipLaunchKernelGGL(kdw::k_wgrad_final, dim3((unsigned)((n_out + kdw::THREADS - 1) / kdw::THREADS)), dim3(kdw::THREADS), 0, st,
                     partial, dw, n_out, g.parts);
  RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_wgrad");
  return RIGL_OK;
}

}  // extern "C"
