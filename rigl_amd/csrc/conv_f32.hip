// K1 in fp32 arithmetic: the validation twin of the bf16 kernels.
//
// The reference trains in float32 unless told otherwise (--precision, imagenet_train_eval.py:56-59, 553-554) and the
// float tolerance of the path is stated against fp32 ("loss / gradients within 1e-5").  The bf16-operand kernels can
// only show that per kernel (relative to sum |a||b|); these kernels compute the same three products -- y = conv(x,
// mask * W), dX, dense dW -- from fp32 activations and the fp32 master weights with the mask applied on the fly, on
// v_mfma_f32_32x32x2_f32 (fp32 multiply, fp32 accumulate), so that a whole network can be trained for a few steps and
// held to 1e-5 against a float64 evaluation of the same model (tests/test_k1_fp32_gpu.py).
//
// NOT tuned and not on the measured path: one wave owns one 32 x 32 output tile, operands come straight from global
// memory into the MFMA operand registers (lane l supplies row / column l & 31 at reduction index l >> 5; along a
// contiguous reduction axis a lane loads four consecutive values and the four MFMAs pair index j of the two
// half-waves' quads -- any pairing is a valid order of the same sum).  The weight gradient splits its reduction (the
// N*Ho*Wo positions) into slabs that are summed in a fixed order, so results are deterministic.
// Any shape is taken (bounds are checked per element; the quad loads need the reduction axis to be a multiple of 8).
#include "common.hpp"

namespace rigl {
namespace kf32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;  // four independent waves

__device__ __forceinline__ float mask_at(const uint32_t* __restrict__ bits, int64_t i) {
  return bits ? (float)((bits[i >> 5] >> (i & 31)) & 1u) : 1.f;
}

#define F32_MFMA(a, b, acc) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)

// accumulator element v of lane l is output row 8 * (v / 4) + 4 * (l / 32) + v % 4, column l % 32
__device__ __forceinline__ int acc_row(int v, int half) { return 8 * (v >> 2) + 4 * half + (v & 3); }

// y[m][co] = sum_{r,s,c} x[pix(m, r, s)][c] * (mask * W)[r][s][c][co]
template <bool VEC>
__global__ __launch_bounds__(THREADS) void k_fwd(RiglConvDesc d, const float* __restrict__ x, const float* __restrict__ w,
                                                 const uint32_t* __restrict__ bits, float* __restrict__ y,
                                                 int64_t mtiles, int ntiles) {
  const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (tile >= mtiles * ntiles) return;
  const int nt = (int)(tile % ntiles);
  const int64_t mt = tile / ntiles;
  const int64_t M = (int64_t)d.n * d.ho * d.wo;
  const int64_t row = mt * 32 + li;
  const bool rv = row < M;
  const int wo = (int)(row % d.wo);
  const int ho = (int)((row / d.wo) % d.ho);
  const int n = (int)(row / ((int64_t)d.wo * d.ho));
  const int col = nt * 32 + li;
  const bool cv = col < d.cout;
  // four accumulators: independent MFMA chains, and a quarter of the sequential additions per partial sum
  f32x16 acc, acc1, acc2, acc3;
  for (int v = 0; v < 16; ++v) acc[v] = acc1[v] = acc2[v] = acc3[v] = 0.f;
  for (int r = 0; r < d.kh; ++r) {
    const int hi = ho * d.stride_h - d.pad_top + r;
    for (int s = 0; s < d.kw; ++s) {
      const int wi = wo * d.stride_w - d.pad_left + s;
      const bool inb = rv && (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w;
      const float* xp = x + ((int64_t)(n * d.h + hi) * d.w + wi) * d.cin;
      const int64_t wb = (int64_t)(r * d.kw + s) * d.cin * d.cout + col;
      if (VEC) {
        for (int c0 = 0; c0 < d.cin; c0 += 8) {
          const int c = c0 + 4 * half;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (inb) a = *(const f32x4*)(xp + c);
          float b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t idx = wb + (int64_t)(c + j) * d.cout;
            b[j] = cv ? w[idx] * mask_at(bits, idx) : 0.f;
          }
          F32_MFMA(a[0], b[0], acc);
          F32_MFMA(a[1], b[1], acc1);
          F32_MFMA(a[2], b[2], acc2);
          F32_MFMA(a[3], b[3], acc3);
        }
      } else {
        for (int c0 = 0; c0 < d.cin; c0 += 2) {
          const int c = c0 + half;
          const bool ok = c < d.cin;
          const int64_t idx = wb + (int64_t)c * d.cout;
          const float a = (inb && ok) ? xp[c] : 0.f;
          const float b = (cv && ok) ? w[idx] * mask_at(bits, idx) : 0.f;
          F32_MFMA(a, b, acc);
        }
      }
    }
  }
  if (!cv) return;
  acc = (acc + acc1) + (acc2 + acc3);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int64_t orow = mt * 32 + acc_row(v, half);
    if (orow < M) y[orow * d.cout + col] = acc[v];
  }
}

// dx[p][ci] = sum_{r,s,co} dy[q(p, r, s)][co] * (mask * W)[r][s][ci][co]  (+ addend[p][ci])
template <bool VEC>
__global__ __launch_bounds__(THREADS) void k_dgrad(RiglConvDesc d, const float* __restrict__ dy, const float* __restrict__ w,
                                                   const uint32_t* __restrict__ bits, const float* __restrict__ addend,
                                                   float* __restrict__ dx, int64_t mtiles, int ntiles) {
  const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (tile >= mtiles * ntiles) return;
  const int nt = (int)(tile % ntiles);
  const int64_t mt = tile / ntiles;
  const int64_t M = (int64_t)d.n * d.h * d.w;
  const int64_t row = mt * 32 + li;
  const bool rv = row < M;
  const int wi = (int)(row % d.w);
  const int hi = (int)((row / d.w) % d.h);
  const int n = (int)(row / ((int64_t)d.w * d.h));
  const int ci = nt * 32 + li;
  const bool cv = ci < d.cin;
  f32x16 acc, acc1, acc2, acc3;
  for (int v = 0; v < 16; ++v) acc[v] = acc1[v] = acc2[v] = acc3[v] = 0.f;
  for (int r = 0; r < d.kh; ++r) {
    const int th = hi + d.pad_top - r;
    const int ho = th / d.stride_h;
    const bool hv = th >= 0 && th % d.stride_h == 0 && ho < d.ho;
    for (int s = 0; s < d.kw; ++s) {
      const int tw = wi + d.pad_left - s;
      const int wo = tw / d.stride_w;
      const bool inb = rv && hv && tw >= 0 && tw % d.stride_w == 0 && wo < d.wo;
      const float* gp = dy + ((int64_t)(n * d.ho + ho) * d.wo + wo) * d.cout;
      const int64_t wb = ((int64_t)(r * d.kw + s) * d.cin + ci) * d.cout;
      if (VEC) {
        for (int c0 = 0; c0 < d.cout; c0 += 8) {
          const int c = c0 + 4 * half;
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (inb) a = *(const f32x4*)(gp + c);
          if (cv) {
            b = *(const f32x4*)(w + wb + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] *= mask_at(bits, wb + c + j);
          }
          F32_MFMA(a[0], b[0], acc);
          F32_MFMA(a[1], b[1], acc1);
          F32_MFMA(a[2], b[2], acc2);
          F32_MFMA(a[3], b[3], acc3);
        }
      } else {
        for (int c0 = 0; c0 < d.cout; c0 += 2) {
          const int c = c0 + half;
          const bool ok = c < d.cout;
          const float a = (inb && ok) ? gp[c] : 0.f;
          const float b = (cv && ok) ? w[wb + c] * mask_at(bits, wb + c) : 0.f;
          F32_MFMA(a, b, acc);
        }
      }
    }
  }
  if (!cv) return;
  acc = (acc + acc1) + (acc2 + acc3);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int64_t orow = mt * 32 + acc_row(v, half);
    if (orow < M) {
      const int64_t o = orow * d.cin + ci;
      dx[o] = addend ? acc[v] + addend[o] : acc[v];
    }
  }
}

// slab[split][r][s][ci][co] = sum over the split's positions m of x[pix(m, r, s)][ci] * dy[m][co]
__global__ __launch_bounds__(THREADS) void k_wgrad(RiglConvDesc d, const float* __restrict__ x, const float* __restrict__ dy,
                                                   float* __restrict__ out, int itiles, int jtiles, int nsplit,
                                                   int64_t chunk) {
  const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
  int64_t tile = (int64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  const int taps = d.kh * d.kw;
  if (tile >= (int64_t)nsplit * taps * itiles * jtiles) return;
  const int jt = (int)(tile % jtiles); tile /= jtiles;
  const int it = (int)(tile % itiles); tile /= itiles;
  const int tap = (int)(tile % taps);
  const int split = (int)(tile / taps);
  const int r = tap / d.kw, s = tap % d.kw;
  const int64_t M = (int64_t)d.n * d.ho * d.wo;
  const int64_t mb = split * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  const int ci = it * 32 + li, co = jt * 32 + li;
  const bool iv = ci < d.cin, jv = co < d.cout;
  f32x16 acc, acc1, acc2, acc3;
  for (int v = 0; v < 16; ++v) acc[v] = acc1[v] = acc2[v] = acc3[v] = 0.f;
  for (int64_t m0 = mb; m0 < me; m0 += 8) {
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + 2 * j + half;
      const int wo = (int)(m % d.wo);
      const int ho = (int)((m / d.wo) % d.ho);
      const int n = (int)(m / ((int64_t)d.wo * d.ho));
      const int hi = ho * d.stride_h - d.pad_top + r;
      const int wi = wo * d.stride_w - d.pad_left + s;
      const bool mv = m < me;
      const bool inb = mv && iv && (unsigned)hi < (unsigned)d.h && (unsigned)wi < (unsigned)d.w;
      a[j] = inb ? x[((int64_t)(n * d.h + hi) * d.w + wi) * d.cin + ci] : 0.f;
      b[j] = (mv && jv) ? dy[m * d.cout + co] : 0.f;
    }
    F32_MFMA(a[0], b[0], acc);
    F32_MFMA(a[1], b[1], acc1);
    F32_MFMA(a[2], b[2], acc2);
    F32_MFMA(a[3], b[3], acc3);
  }
  if (!jv) return;
  acc = (acc + acc1) + (acc2 + acc3);
  float* slab = out + (int64_t)split * taps * d.cin * d.cout;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int i = it * 32 + acc_row(v, half);
    if (i < d.cin) slab[((int64_t)tap * d.cin + i) * d.cout + co] = acc[v];
  }
}

// dw[i] = slab[0][i] + slab[1][i] + ... in that order (double accumulator: the order is then immaterial to fp32)
__global__ __launch_bounds__(THREADS) void k_sum_slabs(const float* __restrict__ slabs, float* __restrict__ dw, int64_t size,
                                                       int nsplit) {
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < size; i += stride) {
    double t = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) t += (double)slabs[(int64_t)sp * size + i];
    dw[i] = (float)t;
  }
}

static bool desc_ok(const RiglConvDesc* d) {
  return d && d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0 && d->kh > 0 &&
         d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_top >= 0 && d->pad_left >= 0;
}

static unsigned blocks_for(int64_t waves) { return (unsigned)((waves + THREADS / 64 - 1) / (THREADS / 64)); }

// the reduction of the weight gradient is cut so that the launch has a few thousand waves, in slabs of at least 256
// positions (a multiple of 8: the loop's step)
static void wgrad_split(const RiglConvDesc* d, int* nsplit, int64_t* chunk) {
  const int64_t M = (int64_t)d->n * d->ho * d->wo;
  const int64_t tiles = (int64_t)d->kh * d->kw * ((d->cin + 31) / 32) * ((d->cout + 31) / 32);
  int64_t want = (4096 + tiles - 1) / tiles;
  const int64_t most = (M + 255) / 256;
  if (want > most) want = most;
  if (want < 1) want = 1;
  int64_t c = (M + want - 1) / want;
  c = (c + 7) / 8 * 8;
  *chunk = c;
  *nsplit = (int)((M + c - 1) / c);
}

}  // namespace kf32
}  // namespace rigl

extern "C" {

int rigl_masked_conv2d_fwd_f32(const RiglConvDesc* d, const float* x, const float* w_hwio, const uint32_t* mask_bits,
                               float* y, rigl_stream_t stream) {
  using namespace rigl;
  if (!kf32::desc_ok(d) || !x || !w_hwio || !y) return fail(RIGL_EINVAL, "rigl_masked_conv2d_fwd_f32: bad argument");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_FWD, st);
  const int64_t mtiles = ((int64_t)d->n * d->ho * d->wo + 31) / 32;
  const int ntiles = (d->cout + 31) / 32;
  const dim3 grid(kf32::blocks_for(mtiles * ntiles)), block(kf32::THREADS);
  if (d->cin % 8 == 0)
    hipLaunchKernelGGL(kf32::k_fwd<true>, grid, block, 0, st, *d, x, w_hwio, mask_bits, y, mtiles, ntiles);
  else
    hipLaunchKernelGGL(kf32::k_fwd<false>, grid, block, 0, st, *d, x, w_hwio, mask_bits, y, mtiles, ntiles);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_fwd_f32");
  return RIGL_OK;
}

int rigl_masked_conv2d_dgrad_f32(const RiglConvDesc* d, const float* dy, const float* w_hwio, const uint32_t* mask_bits,
                                 const float* addend, float* dx, rigl_stream_t stream) {
  using namespace rigl;
  if (!kf32::desc_ok(d) || !dy || !w_hwio || !dx) return fail(RIGL_EINVAL, "rigl_masked_conv2d_dgrad_f32: bad argument");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_DGRAD, st);
  const int64_t mtiles = ((int64_t)d->n * d->h * d->w + 31) / 32;
  const int ntiles = (d->cin + 31) / 32;
  const dim3 grid(kf32::blocks_for(mtiles * ntiles)), block(kf32::THREADS);
  if (d->cout % 8 == 0)
    hipLaunchKernelGGL(kf32::k_dgrad<true>, grid, block, 0, st, *d, dy, w_hwio, mask_bits, addend, dx, mtiles, ntiles);
  else
    hipLaunchKernelGGL(kf32::k_dgrad<false>, grid, block, 0, st, *d, dy, w_hwio, mask_bits, addend, dx, mtiles, ntiles);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_dgrad_f32");
  return RIGL_OK;
}

size_t rigl_conv2d_wgrad_f32_workspace_bytes(const RiglConvDesc* d) {
  using namespace rigl;
  if (!kf32::desc_ok(d)) return 0;
  int nsplit;
  int64_t chunk;
  kf32::wgrad_split(d, &nsplit, &chunk);
  return nsplit > 1 ? (size_t)nsplit * d->kh * d->kw * d->cin * d->cout * sizeof(float) : 0;
}

int rigl_masked_conv2d_wgrad_f32(const RiglConvDesc* d, const float* x, const float* dy, float* dw, void* workspace,
                                 size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  if (!kf32::desc_ok(d) || !x || !dy || !dw) return fail(RIGL_EINVAL, "rigl_masked_conv2d_wgrad_f32: bad argument");
  int nsplit;
  int64_t chunk;
  kf32::wgrad_split(d, &nsplit, &chunk);
  const int64_t size = (int64_t)d->kh * d->kw * d->cin * d->cout;
  if (nsplit > 1 && (!workspace || workspace_bytes < (size_t)nsplit * size * sizeof(float)))
    return fail(RIGL_EWORKSPACE, "rigl_masked_conv2d_wgrad_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  ProfScope prof(PROF_CONV_WGRAD, st);
  const int itiles = (d->cin + 31) / 32, jtiles = (d->cout + 31) / 32;
  const int64_t waves = (int64_t)nsplit * d->kh * d->kw * itiles * jtiles;
  float* out = nsplit > 1 ? (float*)workspace : dw;
  hipLaunchKernelGGL(kf32::k_wgrad, dim3(kf32::blocks_for(waves)), dim3(kf32::THREADS), 0, st, *d, x, dy, out, itiles,
                     jtiles, nsplit, chunk);
  RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad_f32");
  if (nsplit > 1) {
    int64_t blocks = (size + kf32::THREADS - 1) / kf32::THREADS;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kf32::k_sum_slabs, dim3((unsigned)blocks), dim3(kf32::THREADS), 0, st, (const float*)workspace, dw,
                       size, nsplit);
    RIGL_CHECK_LAUNCH("rigl_masked_conv2d_wgrad_f32 (slab sum)");
  }
  return RIGL_OK;
}

}  // extern "C"
