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
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int st = G.rpb >> 1; st > 0; st >>= 1) {
      if (ty < st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] += red[threadIdx.x + st * G.tpr][j];
      }
      __syncthreads();
    }
    if (ty == 0 && c_ok) {
      float* p = partial + ((int64_t)blockIdx.x * taps + tap) * d.cin + cgi * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = red[threadIdx.x][j];
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
  rigl::kdw::WGeom g = rigl::kdw::make_wgeom(d);
  return rigl::align_up((size_t)g.parts * d->kh * d->kw * d->cin * 4, 256);
}

int rigl_depthwise_conv2d_fwd(const RiglConvDesc* d, const rigl_bf16* x, const float* w, rigl_bf16* y,
                              rigl_stream_t stream) {
  using namespace rigl;
  int rc = kdw::check(d, "rigl_depthwise_conv2d_fwd");
  if (rc) return rc;
  if (!x || !w || !y) return fail(RIGL_EINVAL, "rigl_depthwise_conv2d_fwd: NULL tensor");
  ProfScope prof(PROF_DEPTHWISE, as_stream(stream));
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
  kdw::WGeom g = kdw::make_wgeom(d);
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_DEPTHWISE, st);
  float* partial = static_cast<float*>(workspace);
  dim3 grid((unsigned)g.parts, (unsigned)((g.cg + g.tpr - 1) / g.tpr));
  hipLaunchKernelGGL(kdw::k_wgrad_partial, grid, dim3(kdw::THREADS), 0, st, *d, g, x, dy, partial);
  const int n_out = d->kh * d->kw * d->cin;
  hipLaunchKernelGGL(kdw::k_wgrad_final, dim3((unsigned)((n_out + kdw::THREADS - 1) / kdw::THREADS)), dim3(kdw::THREADS), 0, st,
                     partial, dw, n_out, g.parts);
  RIGL_CHECK_LAUNCH("rigl_depthwise_conv2d_wgrad");
  return RIGL_OK;
}

}  // extern "C"
