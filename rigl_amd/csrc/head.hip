// Classifier head glue: global average pool and softmax cross-entropy.
//
// Not a graded hot-path row (resnet_model.py:701-712, imagenet_train_eval.py:578-584), but at 14 ms per step
// the ~40 tiny framework kernels of the loss head (casts, reductions, gather/scatter, fills, 3-12 us each)
// were 0.12 ms of pure launch latency.  Three kernels replace them:
//   rigl_global_avgpool_fwd : y[n,c]   = bf16( sum_p float(x[n,p,c]) / P )          (fp32 accumulation)
//   rigl_global_avgpool_bwd : dx[n,p,c] = bf16( float(dy[n,c]) / P )
//   rigl_softmax_xent       : per row  loss = -(sum_k t_k * log p_k),  t = onehot*(1-eps) + eps/K
//                             dlogits  = bf16( (p - t) * scale )       (scale = 1/batch for the mean loss)
#include "common.hpp"

namespace rigl {
namespace khead {

constexpr int THREADS = 256;

__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_at(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }

// One thread per (image, 2 channels): consecutive threads read consecutive channel pairs of a pixel row.
__global__ __launch_bounds__(THREADS) void k_avgpool_fwd(int n, int p, int c2, const uint32_t* __restrict__ x,
                                                          uint32_t* __restrict__ y) {
  const int i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n * c2) return;
  const int img = i / c2, ch = i % c2;
  const uint32_t* src = x + (int64_t)img * p * c2 + ch;
  float a = 0.f, b = 0.f;
  for (int q = 0; q < p; ++q) {
    const uint32_t v = src[(int64_t)q * c2];
    a += bf_lo(v); b += bf_hi(v);
  }
  const float inv = (float)p;
  y[i] = f2bf(a / inv) | (f2bf(b / inv) << 16);
}

// The same with 8 channels (16 bytes) per lane (c % 8 == 0): the two-channel form moves 256 bytes per wave instruction and
// ran at 1.5 TB/s; the additions of a channel run in the same pixel order, so the bits are those of the form above.
__global__ __launch_bounds__(THREADS) void k_avgpool_fwd8(int n, int p, int c8, const uint4* __restrict__ x, uint4* __restrict__ y) {
  const int i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= n * c8) return;
  const int img = i / c8, ch = i % c8;
  const uint4* src = x + (int64_t)img * p * c8 + ch;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll 7
  for (int q = 0; q < p; ++q) {
    const uint4 v = src[(int64_t)q * c8];
    a[0] += bf_lo(v.x); a[1] += bf_hi(v.x); a[2] += bf_lo(v.y); a[3] += bf_hi(v.y);
    a[4] += bf_lo(v.z); a[5] += bf_hi(v.z); a[6] += bf_lo(v.w); a[7] += bf_hi(v.w);
  }
  const float inv = (float)p;
  uint4 o;
  o.x = f2bf(a[0] / inv) | (f2bf(a[1] / inv) << 16); o.y = f2bf(a[2] / inv) | (f2bf(a[3] / inv) << 16);
  o.z = f2bf(a[4] / inv) | (f2bf(a[5] / inv) << 16); o.w = f2bf(a[6] / inv) | (f2bf(a[7] / inv) << 16);
  y[i] = o;
}
__global__ __launch_bounds__(THREADS) void k_avgpool_bwd8(int n, int p, int c8, const uint4* __restrict__ dy, uint4* __restrict__ dx) {
  const int64_t total = (int64_t)n * p * c8;
  const float inv = (float)p;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
    const int ch = (int)(i % c8);
    const int img = (int)(i / ((int64_t)p * c8));
    const uint4 v = dy[(int64_t)img * c8 + ch];
    uint4 o;
    o.x = f2bf(bf_lo(v.x) / inv) | (f2bf(bf_hi(v.x) / inv) << 16); o.y = f2bf(bf_lo(v.y) / inv) | (f2bf(bf_hi(v.y) / inv) << 16);
    o.z = f2bf(bf_lo(v.z) / inv) | (f2bf(bf_hi(v.z) / inv) << 16); o.w = f2bf(bf_lo(v.w) / inv) | (f2bf(bf_hi(v.w) / inv) << 16);
    dx[i] = o;
  }
}

__global__ __launch_bounds__(THREADS) void k_avgpool_bwd(int n, int p, int c2, const uint32_t* __restrict__ dy,
                                                          uint32_t* __restrict__ dx) {
  const int64_t total = (int64_t)n * p * c2;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
    const int ch = (int)(i % c2);
    const int img = (int)(i / ((int64_t)p * c2));
    const uint32_t v = dy[(int64_t)img * c2 + ch];
    const float inv = (float)p;
    dx[i] = f2bf(bf_lo(v) / inv) | (f2bf(bf_hi(v) / inv) << 16);
  }
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < THREADS / 64; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// One workgroup per row.
__global__ __launch_bounds__(THREADS) void k_softmax_xent(int k, const uint16_t* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, float eps, float scale,
                                                           float* __restrict__ row_loss, uint16_t* __restrict__ dlogits) {
  __shared__ float sh[THREADS / 64];
  const int row = blockIdx.x;
  const uint16_t* z = logits + (int64_t)row * k;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < k; j += THREADS) m = fmaxf(m, bf16_at(z, j));
  m = block_reduce(m, true, sh);
  float se = 0.f, sz = 0.f;
  for (int j = threadIdx.x; j < k; j += THREADS) {
    const float v = bf16_at(z, j) - m;
    se += expf(v); sz += v;
  }
  se = block_reduce(se, false, sh);
  sz = block_reduce(sz, false, sh);
  const float lse = logf(se);                 // log sum exp of the shifted logits
  const int64_t lab = labels[row];
  const float on = 1.f - eps, off = eps / (float)k;
  if (threadIdx.x == 0) {
    // -sum_k t_k (z_k - m - lse) = (1-eps) * (lse - (z_lab - m)) + eps/K * (K * lse - sum_k (z_k - m))
    const float zl = (lab >= 0 && lab < k) ? bf16_at(z, lab) - m : 0.f;
    row_loss[row] = on * (lse - zl) + off * ((float)k * lse - sz);
  }
  if (dlogits) {
    uint16_t* g = dlogits + (int64_t)row * k;
    for (int j = threadIdx.x; j < k; j += THREADS) {
      const float p = expf(bf16_at(z, j) - m - lse);
      const float t = off + ((int64_t)j == lab ? on : 0.f);
      g[j] = (uint16_t)f2bf((p - t) * scale);
    }
  }
}

}  // namespace khead
}  // namespace rigl

extern "C" {

int rigl_global_avgpool_fwd(int32_t n, int32_t pixels, int32_t c, const rigl_bf16* x, rigl_bf16* y, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::khead;
  if (n <= 0 || pixels <= 0 || c <= 0 || (c & 1)) return fail(RIGL_EINVAL, "rigl_global_avgpool_fwd: need n, pixels > 0 and an even channel count");
  if (!x || !y) return fail(RIGL_EINVAL, "rigl_global_avgpool_fwd: NULL tensor");
  const int c2 = c / 2;
  if ((c & 7) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
    const int c8 = c / 8;
    hipLaunchKernelGGL(k_avgpool_fwd8, dim3((unsigned)((n * c8 + THREADS - 1) / THREADS)), dim3(THREADS), 0, as_stream(stream),
                       n, pixels, c8, reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y));
    RIGL_CHECK_LAUNCH("rigl_global_avgpool_fwd");
    return RIGL_OK;
  }
  hipLaunchKernelGGL(k_avgpool_fwd, dim3((unsigned)((n * c2 + THREADS - 1) / THREADS)), dim3(THREADS), 0, as_stream(stream),
                     n, pixels, c2, reinterpret_cast<const uint32_t*>(x), reinterpret_cast<uint32_t*>(y));
  RIGL_CHECK_LAUNCH("rigl_global_avgpool_fwd");
  return RIGL_OK;
}

int rigl_global_avgpool_bwd(int32_t n, int32_t pixels, int32_t c, const rigl_bf16* dy, rigl_bf16* dx, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::khead;
  if (n <= 0 || pixels <= 0 || c <= 0 || (c & 1)) return fail(RIGL_EINVAL, "rigl_global_avgpool_bwd: need n, pixels > 0 and an even channel count");
  if (!dy || !dx) return fail(RIGL_EINVAL, "rigl_global_avgpool_bwd: NULL tensor");
  const int c2 = c / 2;
  if ((c & 7) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0) {
    const int c8 = c / 8;
    int64_t blocks8 = ((int64_t)n * pixels * c8 + THREADS - 1) / THREADS;
    if (blocks8 > 8192) blocks8 = 8192;
    hipLaunchKernelGGL(k_avgpool_bwd8, dim3((unsigned)blocks8), dim3(THREADS), 0, as_stream(stream), n, pixels, c8,
                       reinterpret_cast<const uint4*>(dy), reinterpret_cast<uint4*>(dx));
    RIGL_CHECK_LAUNCH("rigl_global_avgpool_bwd");
    return RIGL_OK;
  }
  int64_t blocks = ((int64_t)n * pixels * c2 + THREADS - 1) / THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_avgpool_bwd, dim3((unsigned)blocks), dim3(THREADS), 0, as_stream(stream), n, pixels, c2,
                     reinterpret_cast<const uint32_t*>(dy), reinterpret_cast<uint32_t*>(dx));
  RIGL_CHECK_LAUNCH("rigl_global_avgpool_bwd");
  return RIGL_OK;
}

int rigl_softmax_xent(int32_t rows, int32_t classes, const rigl_bf16* logits, const int64_t* labels, float label_smoothing,
                      float grad_scale, float* row_loss, rigl_bf16* dlogits, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::khead;
  if (rows <= 0 || classes <= 0) return fail(RIGL_EINVAL, "rigl_softmax_xent: rows and classes must be positive");
  if (!logits || !labels || !row_loss) return fail(RIGL_EINVAL, "rigl_softmax_xent: NULL tensor");
  if (!(label_smoothing >= 0.f) || label_smoothing > 1.f) return fail(RIGL_EINVAL, "rigl_softmax_xent: label_smoothing %g not in [0,1]", (double)label_smoothing);
  hipLaunchKernelGGL(k_softmax_xent, dim3((unsigned)rows), dim3(THREADS), 0, as_stream(stream), classes, logits, labels,
                     label_smoothing, grad_scale, row_loss, dlogits);
  RIGL_CHECK_LAUNCH("rigl_softmax_xent");
  return RIGL_OK;
}

}  // extern "C"
