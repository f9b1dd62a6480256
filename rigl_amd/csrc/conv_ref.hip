// Direct (non-MFMA) convolution kernels with the same operand layouts as the
// MFMA path.  They serve shapes the implicit-GEMM path does not take (channel
// counts that are not multiples of 8: the MNIST MLP's 300/100/10 units, a
// 10-class logits layer) and act as an independent on-device cross-check in the
// tests.  One thread per output element, fp32 accumulation in (r, s, c) order.
#include "common.hpp"

namespace rigl {
namespace kref {

constexpr int THREADS = 256;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(THREADS) void k_fwd(RiglConvDesc d, const uint16_t* __restrict__ x,
                                                 const uint16_t* __restrict__ w_ohwi, uint16_t* __restrict__ y) {
  const int64_t total = (int64_t)d.n * d.ho * d.wo * d.cout;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  const int K = d.kh * d.kw * d.cin;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += stride) {
    const int co = (int)(idx % d.cout);
    int64_t m = idx / d.cout;
    const int wo = (int)(m % d.wo); m /= d.wo;
    const int ho = (int)(m % d.ho);
    const int n = (int)(m / d.ho);
    float acc = 0.f;
    for (int r = 0; r < d.kh; ++r) {
      const int hi = ho * d.stride_h - d.pad_top + r;
      if ((unsigned)hi >= (unsigned)d.h) continue;
      for (int s = 0; s < d.kw; ++s) {
        const int wi = wo * d.stride_w - d.pad_left + s;
        if ((unsigned)wi >= (unsigned)d.w) continue;
        const uint16_t* xp = x + ((int64_t)(n * d.h + hi) * d.w + wi) * d.cin;
        const uint16_t* wp = w_ohwi + (int64_t)co * K + (r * d.kw + s) * d.cin;
        for (int c = 0; c < d.cin; ++c) acc = fmaf(bf2f(xp[c]), bf2f(wp[c]), acc);
      }
    }
    y[idx] = f2bf(acc);
  }
}

__global__ __launch_bounds__(THREADS) void k_dgrad(RiglConvDesc d, const uint16_t* __restrict__ dy,
                                                   const uint16_t* __restrict__ w_hwio, uint16_t* __restrict__ dx) {
  const int64_t total = (int64_t)d.n * d.h * d.w * d.cin;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += stride) {
    const int ci = (int)(idx % d.cin);
    int64_t m = idx / d.cin;
    const int wi = (int)(m % d.w); m /= d.w;
    const int hi = (int)(m % d.h);
    const int n = (int)(m / d.h);
    float acc = 0.f;
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
        const uint16_t* gp = dy + ((int64_t)(n * d.ho + ho) * d.wo + wo) * d.cout;
        const uint16_t* wp = w_hwio + ((int64_t)(r * d.kw + s) * d.cin + ci) * d.cout;
        for (int c = 0; c < d.cout; ++c) acc = fmaf(bf2f(gp[c]), bf2f(wp[c]), acc);
      }
    }
    dx[idx] = f2bf(acc);
  }
}

__global__ __launch_bounds__(THREADS) void k_wgrad(RiglConvDesc d, const uint16_t* __restrict__ x,
                                                   const uint16_t* __restrict__ dy, float* __restrict__ dw) {
  const int64_t total = (int64_t)d.kh * d.kw * d.cin * d.cout;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += stride) {
    const int co = (int)(idx % d.cout);
    int64_t t = idx / d.cout;
    const int ci = (int)(t % d.cin); t /= d.cin;
    const int s = (int)(t % d.kw);
    const int r = (int)(t / d.kw);
    float acc = 0.f;
    for (int n = 0; n < d.n; ++n)
      for (int ho = 0; ho < d.ho; ++ho) {
        const int hi = ho * d.stride_h - d.pad_top + r;
        if ((unsigned)hi >= (unsigned)d.h) continue;
        for (int wo = 0; wo < d.wo; ++wo) {
          const int wi = wo * d.stride_w - d.pad_left + s;
          if ((unsigned)wi >= (unsigned)d.w) continue;
          acc = fmaf(bf2f(x[((int64_t)(n * d.h + hi) * d.w + wi) * d.cin + ci]),
                     bf2f(dy[((int64_t)(n * d.ho + ho) * d.wo + wo) * d.cout + co]), acc);
        }
      }
    dw[idx] = acc;
  }
}

static unsigned grid_for(int64_t total) {
  int64_t b = (total + THREADS - 1) / THREADS;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace kref
}  // namespace rigl

extern "C" {

int rigl_conv2d_fwd_ref(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* w_ohwi, rigl_bf16* y,
                        rigl_stream_t stream) {
  using namespace rigl;
  if (!d || !x || !w_ohwi || !y) return fail(RIGL_EINVAL, "rigl_conv2d_fwd_ref: NULL argument");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_FWD, st);
  hipLaunchKernelGGL(kref::k_fwd, dim3(kref::grid_for((int64_t)d->n * d->ho * d->wo * d->cout)), dim3(kref::THREADS), 0,
                     st, *d, x, w_ohwi, y);
  RIGL_CHECK_LAUNCH("rigl_conv2d_fwd_ref");
  return RIGL_OK;
}

int rigl_conv2d_dgrad_ref(const RiglConvDesc* d, const rigl_bf16* dy, const rigl_bf16* w_hwio, rigl_bf16* dx,
                          rigl_stream_t stream) {
  using namespace rigl;
  if (!d || !dy || !w_hwio || !dx) return fail(RIGL_EINVAL, "rigl_conv2d_dgrad_ref: NULL argument");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_DGRAD, st);
  hipLaunchKernelGGL(kref::k_dgrad, dim3(kref::grid_for((int64_t)d->n * d->h * d->w * d->cin)), dim3(kref::THREADS), 0,
                     st, *d, dy, w_hwio, dx);
  RIGL_CHECK_LAUNCH("rigl_conv2d_dgrad_ref");
  return RIGL_OK;
}

int rigl_conv2d_wgrad_ref(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, float* dw,
                          rigl_stream_t stream) {
  using namespace rigl;
  if (!d || !x || !dy || !dw) return fail(RIGL_EINVAL, "rigl_conv2d_wgrad_ref: NULL argument");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_WGRAD, st);
  hipLaunchKernelGGL(kref::k_wgrad, dim3(kref::grid_for((int64_t)d->kh * d->kw * d->cin * d->cout)), dim3(kref::THREADS),
                     0, st, *d, x, dy, dw);
  RIGL_CHECK_LAUNCH("rigl_conv2d_wgrad_ref");
  return RIGL_OK;
}

}  // extern "C"
