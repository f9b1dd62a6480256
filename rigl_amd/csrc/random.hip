// Stateless random tensors with TensorFlow's bit layout, for RigL's drop noise
// and SET's random grow scores:
//   stateless_random_normal (shape, seed=[s0, s1]) * stddev + mean
//   stateless_random_uniform(shape, seed=[s0, s1]) * (max - min) + min
// (rigl/sparse_optimizers_base.py:402-418, 260-274, 523-534).  TensorFlow 1.15
// (tensorflow/core/kernels/stateless_random_ops.cc, lib/random/philox_random.h,
// random_distributions.h): the int32 seed pair is scrambled by one Philox call
// into (key, counter); element i is lane i % 4 of Philox-4x32-10(counter + i/4);
// floats take 23 mantissa bits ([1,2) - 1); normals are Box-Muller on
// consecutive lane pairs.  One thread per group of 4 elements; HBM-bound write.
// sinf / cosf / logf are ocml's (<= 2 ulp from glibc's, which TF's CPU kernel calls).
#include "common.hpp"

namespace rigl {
namespace krand {

constexpr int THREADS = 256;
constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;

struct Philox { uint32_t c[4]; };

__host__ __device__ inline Philox philox4x32_10(Philox ctr, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    if (r) { k0 += W0; k1 += W1; }
    const uint64_t p0 = (uint64_t)M0 * ctr.c[0], p1 = (uint64_t)M1 * ctr.c[2];
    Philox n;
    n.c[0] = (uint32_t)(p1 >> 32) ^ ctr.c[1] ^ k0;
    n.c[1] = (uint32_t)p1;
    n.c[2] = (uint32_t)(p0 >> 32) ^ ctr.c[3] ^ k1;
    n.c[3] = (uint32_t)p0;
    ctr = n;
  }
  return ctr;
}

__device__ __forceinline__ float u32_to_float(uint32_t x) {
  return __uint_as_float((127u << 23) | (x & 0x7FFFFFu)) - 1.0f;
}

struct Args {
  float* out;
  int64_t n;
  uint32_t key0, key1;
  uint32_t c0, c1, c2, c3;   // 128-bit base counter
  int dist;                  // 0 uniform [0,1), 1 standard normal
  float scale, shift;
};

__global__ __launch_bounds__(THREADS) void k_fill(Args A) {
  const int64_t groups = (A.n + 3) / 4;
  for (int64_t g = (int64_t)blockIdx.x * THREADS + threadIdx.x; g < groups; g += (int64_t)gridDim.x * THREADS) {
    // counter + g as a 128-bit add
    const uint64_t lo0 = ((uint64_t)A.c1 << 32) | A.c0, hi0 = ((uint64_t)A.c3 << 32) | A.c2;
    const uint64_t lo = lo0 + (uint64_t)g, hi = hi0 + (lo < lo0 ? 1u : 0u);
    Philox c;
    c.c[0] = (uint32_t)lo; c.c[1] = (uint32_t)(lo >> 32); c.c[2] = (uint32_t)hi; c.c[3] = (uint32_t)(hi >> 32);
    const Philox s = philox4x32_10(c, A.key0, A.key1);
    float f[4];
    if (A.dist == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = u32_to_float(s.c[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        float u1 = u32_to_float(s.c[i]);
        if (u1 < 1.0e-7f) u1 = 1.0e-7f;
        const float v1 = (2.0f * 3.14159265358979323846f) * u32_to_float(s.c[i + 1]);
        const float u2 = sqrtf(-2.0f * logf(u1));
        f[i] = sinf(v1) * u2;
        f[i + 1] = cosf(v1) * u2;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = g * 4 + i;
      if (e < A.n) A.out[e] = f[i] * A.scale + A.shift;      // two roundings, like the TF graph's Mul + Add
    }
  }
}

// All tensors of a mask update in ONE launch: 54 separate fills of a ResNet-50 update were 54 launches of
// ~4 us each next to a 0.35 ms update.  Block b belongs to the item whose block range contains it.
constexpr int BATCH = 64;
struct BatchArgs { Args a[BATCH]; uint32_t blk_begin[BATCH + 1]; int count; };
__global__ __launch_bounds__(THREADS) void k_fill_batched(BatchArgs B) {
  int lo = 0, hi = B.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (B.blk_begin[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const Args& A = B.a[lo];
  const uint32_t nblk = B.blk_begin[lo + 1] - B.blk_begin[lo], bid = blockIdx.x - B.blk_begin[lo];
  const int64_t groups = (A.n + 3) / 4;
  for (int64_t g = (int64_t)bid * THREADS + threadIdx.x; g < groups; g += (int64_t)nblk * THREADS) {
    const uint64_t lo0 = ((uint64_t)A.c1 << 32) | A.c0, hi0 = ((uint64_t)A.c3 << 32) | A.c2;
    const uint64_t l = lo0 + (uint64_t)g, h = hi0 + (l < lo0 ? 1u : 0u);
    Philox c;
    c.c[0] = (uint32_t)l; c.c[1] = (uint32_t)(l >> 32); c.c[2] = (uint32_t)h; c.c[3] = (uint32_t)(h >> 32);
    const Philox s = philox4x32_10(c, A.key0, A.key1);
    float f[4];
    if (A.dist == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = u32_to_float(s.c[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        float u1 = u32_to_float(s.c[i]);
        if (u1 < 1.0e-7f) u1 = 1.0e-7f;
        const float v1 = (2.0f * 3.14159265358979323846f) * u32_to_float(s.c[i + 1]);
        const float u2 = sqrtf(-2.0f * logf(u1));
        f[i] = sinf(v1) * u2;
        f[i + 1] = cosf(v1) * u2;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = g * 4 + i;
      if (e < A.n) A.out[e] = f[i] * A.scale + A.shift;
    }
  }
}

static Args make_args(float* out, int64_t n, int32_t seed0, int32_t seed1, int32_t dist, float scale, float shift) {
  // GenerateKey: int32 -> uint64 sign-extends; scramble with one Philox call under a fixed key
  const uint64_t s0 = (uint64_t)(int64_t)seed0, s1 = (uint64_t)(int64_t)seed1;
  Philox c;
  c.c[0] = (uint32_t)s0; c.c[1] = (uint32_t)(s0 >> 32); c.c[2] = (uint32_t)s1; c.c[3] = (uint32_t)(s1 >> 32);
  const Philox mix = philox4x32_10(c, 0x3ec8f720u, 0x02461e29u);
  Args a;
  a.out = out; a.n = n; a.key0 = mix.c[0]; a.key1 = mix.c[1];
  a.c0 = 0; a.c1 = 0; a.c2 = mix.c[2]; a.c3 = mix.c[3];
  a.dist = dist; a.scale = scale; a.shift = shift;
  return a;
}

}  // namespace krand
}  // namespace rigl

extern "C" {

int rigl_stateless_random(float* out, int64_t n, int32_t seed0, int32_t seed1, int32_t dist, float scale, float shift,
                          rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::krand;
  if (n < 0 || (n > 0 && !out)) return fail(RIGL_EINVAL, "rigl_stateless_random: bad arguments");
  if (dist != 0 && dist != 1) return fail(RIGL_EINVAL, "rigl_stateless_random: dist must be 0 (uniform) or 1 (normal)");
  if (n == 0) return RIGL_OK;
  const Args a = make_args(out, n, seed0, seed1, dist, scale, shift);
  int64_t blocks = ((n + 3) / 4 + THREADS - 1) / THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(THREADS), 0, as_stream(stream), a);
  RIGL_CHECK_LAUNCH("rigl_stateless_random");
  return RIGL_OK;
}

int rigl_stateless_random_batched(const RiglRandomItem* items, int32_t n_items, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::krand;
  if (n_items < 0 || (n_items > 0 && !items)) return fail(RIGL_EINVAL, "rigl_stateless_random_batched: bad arguments");
  for (int i = 0; i < n_items; ++i) {
    if (items[i].n < 0 || (items[i].n > 0 && !items[i].out)) return fail(RIGL_EINVAL, "rigl_stateless_random_batched: item %d: bad out / n", i);
    if (items[i].dist != 0 && items[i].dist != 1) return fail(RIGL_EINVAL, "rigl_stateless_random_batched: item %d: dist must be 0 or 1", i);
  }
  for (int b = 0; b < n_items; b += BATCH) {
    BatchArgs B;
    B.count = 0;
    uint32_t blk = 0;
    for (int i = b; i < n_items && B.count < BATCH; ++i) {
      if (items[i].n == 0) continue;
      int64_t blocks = ((items[i].n + 3) / 4 + THREADS * 4 - 1) / (THREADS * 4);     // ~4 groups per thread
      if (blocks > 1024) blocks = 1024;
      if (blocks < 1) blocks = 1;
      B.a[B.count] = make_args(items[i].out, items[i].n, items[i].seed0, items[i].seed1, items[i].dist, items[i].scale, items[i].shift);
      B.blk_begin[B.count] = blk;
      blk += (uint32_t)blocks;
      ++B.count;
    }
    if (B.count == 0) continue;
    for (int i = B.count; i <= BATCH; ++i) B.blk_begin[i] = blk;
    hipLaunchKernelGGL(k_fill_batched, dim3(blk), dim3(THREADS), 0, as_stream(stream), B);
  }
  RIGL_CHECK_LAUNCH("rigl_stateless_random_batched");
  return RIGL_OK;
}

}  // extern "C"
