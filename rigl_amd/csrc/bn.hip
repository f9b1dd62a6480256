// Fused batch-norm (+ residual add) (+ ReLU), forward and backward, for NHWC
// bf16 activations with fp32 parameters / statistics.
//
// Not one of the graded hot-path rows but inside the images/s
// number: between two masked convs the reference runs
// tf.layers.batch_normalization(fused=True) + tf.nn.relu (+ the residual add)
// (rigl/imagenet_resnet/resnet_model.py:41-82, 456-501).  Doing that with
// stock ops costs ~13 full passes over the activation per layer; here it is
//   forward : statistics (1 read) + apply (1 read [+1 residual], 1 write)
//   backward: reductions (reads dy, x [,y]) + apply (reads dy, x [,y], writes dx [,dres])
// All kernels are HBM-bound streams: 16 B per lane, channel-group-major thread
// layout so every row access is one contiguous C*2-byte segment, per-block
// partial sums combined in a FIXED order (deterministic statistics).
#include "common.hpp"

namespace rigl {
namespace kbn {

constexpr int THREADS = 256;
constexpr int MAX_PARTS = 512;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// v_cvt_pk_bf16_f32: hardware round-to-nearest-even pack of two floats
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float f[8]) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// Streaming accesses of the element-wise passes ("bn_nt": 1 = stores, 2 = loads and stores marked non-temporal)
__device__ __forceinline__ void st16(uint16_t* p, const uint4& v, int nt) {
  if (nt) { const u32x4_t t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p)); }
  else *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ uint4 ld16(const uint16_t* p, int nt) {
  if (nt > 1) { const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}

struct Geom {
  int nt;             // non-temporal mode of the apply passes
  int il;             // reductions: parts interleave groups of rows
  int64_t M;
  int C, cg;          // channels, 8-channel groups
  int tpr, rpb;       // threads per row (pow2 >= min(cg,256)), rows per pass
  int parts;          // row partitions (= gridDim.x of the reduction kernels)
  int64_t rows_per_part;
  int fixed;          // apply passes: the grid stride is a multiple of the channel groups, so a thread sees ONE channel group in
                      // every iteration and keeps its per-channel parameters in registers (no LDS copy of all C channels per
                      // workgroup: at C = 2048 that copy was 56 KB -- two workgroups per CU -- and 3.5x the bytes the workgroup streams)
};

__device__ __forceinline__ void ld8f(const float* __restrict__ p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// Reduction over rows of up to two per-channel quantities.
//   MODE 0 (fwd stats): q0 = sum x, q1 = sum x^2
//   MODE 1 (bwd)      : q0 = sum dz, q1 = sum dz * xhat,  dz = relu-masked dy
// MSK: where the ReLU mask comes from -- 0 recomputed from x (scale, shift), 1 the saved output y,
// 2 the 1-bit-per-element mask the forward left (1/16 of y's bytes).
template <int MODE, bool RELU, int MSK>
__global__ __launch_bounds__(THREADS) void k_reduce(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ y,
                                                    const uint8_t* __restrict__ mbits,
                                                    const uint16_t* __restrict__ dy, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, float* __restrict__ partial) {
  __shared__ float red[THREADS][17];
  const int tx = threadIdx.x % G.tpr, ty = threadIdx.x / G.tpr;
  const int cgi = blockIdx.y * G.tpr + tx;
  const bool c_ok = cgi < G.cg;
  const int64_t r0 = (int64_t)blockIdx.x * G.rows_per_part;
  int64_t r1 = r0 + G.rows_per_part;
  if (r1 > G.M) r1 = G.M;
  float q0[8], q1[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { q0[j] = q1[j] = 0.f; mu[j] = is[j] = sc[j] = sh[j] = 0.f; }
  if (MODE == 1 && c_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[cgi * 8 + j]; is[j] = invstd[cgi * 8 + j];
      if (RELU && MSK == 0) { sc[j] = scale[cgi * 8 + j]; sh[j] = shift[cgi * 8 + j]; }
    }
  }
  if (c_ok) {
    // a part = every parts-th group of rpb rows ("bn_il", default: the grid streams one window of the tensors) or a
    // contiguous range of rows
    const int64_t rb = G.il ? (int64_t)blockIdx.x * G.rpb + ty : r0 + ty, re = G.il ? G.M : r1;
    const int64_t rstep = G.il ? (int64_t)G.parts * G.rpb : (int64_t)G.rpb;
#pragma unroll 2
    for (int64_t r = rb; r < re; r += rstep) {
      const int64_t off = r * G.C + (int64_t)cgi * 8;
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { q0[j] += xv[j]; q1[j] = fmaf(xv[j], xv[j], q1[j]); }
      } else {
        float dv[8], yv[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + off), dv);
        uint32_t mb = 0u;
        if (RELU && MSK == 1) unpack8(*reinterpret_cast<const uint4*>(y + off), yv);
        if (RELU && MSK == 2) mb = mbits[r * G.cg + cgi];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bool on = true;
          if (RELU) on = MSK == 1 ? (yv[j] > 0.f) : (MSK == 2 ? ((mb >> j) & 1u) != 0u : (fmaf(xv[j], sc[j], sh[j]) > 0.f));
          const float dz = on ? dv[j] : 0.f;
          q0[j] += dz;
          q1[j] = fmaf(dz, (xv[j] - mu[j]) * is[j], q1[j]);
        }
      }
    }
  }
  // combine the `rpb` row-lanes of each channel group (fixed tree order).  A thread keeps its own sums in registers and reads
  // its partner's sixteen with ONE wait per round (written as `red[t][j] += red[t + s * tpr][j]` the compiler serialised
  // read -> wait -> write sixteen times a round: ~2 us of tail in a 9-15 us launch); same additions in the same order.
  float own[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { own[j] = q0[j]; own[8 + j] = q1[j]; red[threadIdx.x][j] = q0[j]; red[threadIdx.x][8 + j] = q1[j]; }
  __syncthreads();
  for (int s = G.rpb >> 1; s > 0; s >>= 1) {
    if (ty < s) {
      float oth[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) oth[j] = red[threadIdx.x + s * G.tpr][j];
#pragma unroll
      for (int j = 0; j < 16; ++j) own[j] += oth[j];
      if (s > 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[threadIdx.x][j] = own[j];
      }
    }
    if (s > 1) __syncthreads();
  }
  if (ty == 0 && c_ok) {
    float* p = partial + ((int64_t)blockIdx.x * 2) * G.C + cgi * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = own[j]; p[G.C + j] = own[8 + j]; }
  }
}

// Sums the per-part partials of CL channels with 256/CL part-lanes each (fixed
// order: lane-strided, then lanes ascending) -- a 256-thread block per CL
// channels instead of one thread walking all parts serially.  CL = 16 for the
// <= 512 partials of k_reduce, 4 (64 part-lanes) for the per-row-tile partials a
// conv epilogue leaves (up to M/128 of them).
template <int CL>
__device__ __forceinline__ void sum_partials(const Geom& G, const float* __restrict__ partial, double& s0, double& s1,
                                             int& c, bool& leader) {
  constexpr int PL = THREADS / CL;
  __shared__ double acc[2][PL][CL + 1];
  const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
  c = blockIdx.x * CL + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < G.C) {
    // FLIGHT rows of both quantities are requested before the first is added: the partials were written by other XCDs
    // (they come from the fabric, not this L2), so the kernel's time is round trips, not bytes.  Rows past the end read
    // as zero; the additions run in the same ascending order whatever FLIGHT is.
#ifndef RIGL_BN_FIN_FLIGHT
#define RIGL_BN_FIN_FLIGHT 16
#endif
    constexpr int FLIGHT = RIGL_BN_FIN_FLIGHT;
    for (int p0 = pl; p0 < G.parts; p0 += PL * FLIGHT) {
      float v0[FLIGHT], v1[FLIGHT];
#pragma unroll
      for (int j = 0; j < FLIGHT; ++j) {
        const int p = p0 + j * PL;
        const int pc = p < G.parts ? p : G.parts - 1;       // unconditional loads (a branch per load would serialise them)
        const float l0 = partial[((int64_t)pc * 2) * G.C + c], l1 = partial[((int64_t)pc * 2 + 1) * G.C + c];
        v0[j] = p < G.parts ? l0 : 0.f;
        v1[j] = p < G.parts ? l1 : 0.f;
      }
#pragma unroll
      for (int j = 0; j < FLIGHT; ++j) { a0 += (double)v0[j]; a1 += (double)v1[j]; }
    }
  }
  acc[0][pl][cl] = a0; acc[1][pl][cl] = a1;
  __syncthreads();
  leader = pl == 0 && c < G.C;
  s0 = s1 = 0.0;
  if constexpr (CL == 1) {
    // 256 part-lanes of one channel: 16 threads sum 16 lanes each, the leader sums those (a fixed two-level order)
    __shared__ double acc2[2][16];
    if (threadIdx.x < 16) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { t0 += acc[0][threadIdx.x * 16 + k][0]; t1 += acc[1][threadIdx.x * 16 + k][0]; }
      acc2[0][threadIdx.x] = t0; acc2[1][threadIdx.x] = t1;
    }
    __syncthreads();
    if (leader) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { s0 += acc2[0][k]; s1 += acc2[1][k]; }
    }
  } else if (leader) {
#pragma unroll 8
    for (int k = 0; k < PL; ++k) { s0 += acc[0][k][cl]; s1 += acc[1][k][cl]; }
  }
}

// Forward finalize: mean / invstd, running statistics, fused scale & shift.
template <int CL>
__global__ __launch_bounds__(THREADS) void k_fwd_finalize(Geom G, const float* __restrict__ partial,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float momentum, float eps, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ scale,
                                                           float* __restrict__ shift) {
  double s0, s1; int c; bool leader;
  sum_partials<CL>(G, partial, s0, s1, c, leader);
  if (!leader) return;
  const double m = (double)G.M;
  const double mean = s0 / m;
  double var = s1 / m - mean * mean;           // biased variance normalises (TF fused BN)
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) {
    const double unbiased = G.M > 1 ? var * m / (m - 1.0) : var;   // moving variance uses Bessel's correction
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// y = relu?(x * scale + shift (+ residual))
template <bool RELU, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void k_fwd_apply(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        uint16_t* __restrict__ y, uint8_t* __restrict__ mbits) {
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [2][C] (unused when G.fixed)
  const bool fixed = G.fixed != 0;
  float psc[8], psh[8];
  if (fixed) {
    const int c0 = (int)(((int64_t)blockIdx.x * THREADS + threadIdx.x) % G.cg) * 8;
    ld8f(scale + c0, psc); ld8f(shift + c0, psh);
  } else {
    for (int i = threadIdx.x; i < G.C; i += THREADS) { prm[i] = scale[i]; prm[G.C + i] = shift[i]; }
    __syncthreads();
  }
  const int64_t total = G.M * G.cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    if (!fixed) {
      const int cgi = (int)(i % G.cg);
#pragma unroll
      for (int j = 0; j < 8; ++j) { psc[j] = prm[cgi * 8 + j]; psh[j] = prm[G.C + cgi * 8 + j]; }
    }
    float xv[8], rv[8];
    unpack8(ld16(x + i * 8, G.nt), xv);
    if (HAS_RES) unpack8(ld16(res + i * 8, G.nt), rv);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = fmaf(xv[j], psc[j], psh[j]);
      if (HAS_RES) v += rv[j];
      o[j] = RELU ? fmaxf(v, 0.f) : v;
    }
    uint4 out;
    out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
    st16(y + i * 8, out, G.nt);
    if (RELU && mbits) {          // 1 bit per element: was the activation positive
      uint32_t mb = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) mb |= (o[j] > 0.f ? 1u : 0u) << j;
      mbits[i] = (uint8_t)mb;
    }
  }
}

// Backward finalize: dgamma, dbeta and the per-channel coefficients of dx.  CL = 16 for the <= 512 partials of
// k_reduce, 4 for the per-row-tile partials a dgrad epilogue leaves (rigl_masked_conv2d_bwd_bn: up to M / 128 of them).
template <int CL>
__global__ __launch_bounds__(THREADS) void k_bwd_finalize(Geom G, const float* __restrict__ partial,
                                                           const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef /*[3][C]: a, b, c*/) {
  double s0, s1; int c; bool leader;
  sum_partials<CL>(G, partial, s0, s1, c, leader);
  if (!leader) return;
  dbeta[c] = (float)s0;
  dgamma[c] = (float)s1;
  coef[c] = gamma[c] * invstd[c];
  coef[G.C + c] = (float)(s0 / (double)G.M);
  coef[2 * G.C + c] = (float)(s1 / (double)G.M);
}

// dx = a * (dz - b - xhat * c);  dres = dz
template <bool RELU, int MSK, bool HAS_DRES>
__global__ __launch_bounds__(THREADS) void k_bwd_apply(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ y,
                                                        const uint8_t* __restrict__ mbits,
                                                        const uint16_t* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ coef,
                                                        uint16_t* __restrict__ dx, uint16_t* __restrict__ dres) {
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [7][C]: mean, invstd, a, b, c, scale, shift (unused when G.fixed)
  const bool fixed = G.fixed != 0;
  float pm[8], pis[8], pa[8], pb[8], pc[8], psc[8], psh[8];
  if (fixed) {
    const int c0 = (int)(((int64_t)blockIdx.x * THREADS + threadIdx.x) % G.cg) * 8;
    ld8f(mean + c0, pm); ld8f(invstd + c0, pis);
    ld8f(coef + c0, pa); ld8f(coef + G.C + c0, pb); ld8f(coef + 2 * G.C + c0, pc);
    if (RELU && MSK == 0) { ld8f(scale + c0, psc); ld8f(shift + c0, psh); }
  } else {
    for (int i = threadIdx.x; i < G.C; i += THREADS) {
      prm[i] = mean[i]; prm[G.C + i] = invstd[i];
      prm[2 * G.C + i] = coef[i]; prm[3 * G.C + i] = coef[G.C + i]; prm[4 * G.C + i] = coef[2 * G.C + i];
      if (RELU && MSK == 0) { prm[5 * G.C + i] = scale[i]; prm[6 * G.C + i] = shift[i]; }
    }
    __syncthreads();
  }
  const int64_t total = G.M * G.cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    if (!fixed) {
      const int c0 = (int)(i % G.cg) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pm[j] = prm[c0 + j]; pis[j] = prm[G.C + c0 + j]; pa[j] = prm[2 * G.C + c0 + j]; pb[j] = prm[3 * G.C + c0 + j];
        pc[j] = prm[4 * G.C + c0 + j];
        if (RELU && MSK == 0) { psc[j] = prm[5 * G.C + c0 + j]; psh[j] = prm[6 * G.C + c0 + j]; }
      }
    }
    float xv[8], dv[8], yv[8];
    unpack8(ld16(x + i * 8, G.nt), xv);
    unpack8(ld16(dy + i * 8, G.nt), dv);
    uint32_t mb = 0u;
    if (RELU && MSK == 1) unpack8(*reinterpret_cast<const uint4*>(y + i * 8), yv);
    if (RELU && MSK == 2) mb = mbits[i];
    float o[8], z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bool on = true;
      if (RELU) on = MSK == 1 ? (yv[j] > 0.f) : (MSK == 2 ? ((mb >> j) & 1u) != 0u : (fmaf(xv[j], psc[j], psh[j]) > 0.f));
      const float dz = on ? dv[j] : 0.f;
      const float xh = (xv[j] - pm[j]) * pis[j];
      o[j] = pa[j] * (dz - pb[j] - xh * pc[j]);
      z[j] = dz;
    }
    uint4 out;
    out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
    st16(dx + i * 8, out, G.nt);
    if (HAS_DRES) {
      out.x = pack2(z[0], z[1]); out.y = pack2(z[2], z[3]); out.z = pack2(z[4], z[5]); out.w = pack2(z[6], z[7]);
      st16(dres + i * 8, out, G.nt);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Two batch norms meeting in one add: out = relu(bn(x) + bn2(x2)) -- the first block of every ResNet group, whose
// shortcut is a projection conv + batch norm (resnet_model.py:456-501).  As separate calls the shortcut is written by
// its batch norm and read back by the main one, and in the backward the masked gradient dz is written by the main
// batch norm for the shortcut's to read twice.  Here neither tensor exists: the forward applies both affine maps in
// one pass (the shortcut rounded to bf16 in the register, where it would have been stored), the backward takes both
// batch norms' reductions in one pass over dy and writes both input gradients in another.  Same arithmetic in the same
// order as the separate calls: bit-identical results (tests/test_bn_gpu.py).
// ---------------------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(THREADS) void k_fwd_apply_pair(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                             uint16_t* __restrict__ y, uint8_t* __restrict__ mbits) {
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [4][C] (unused when G.fixed)
  const bool fixed = G.fixed != 0;
  float p0[8], p1[8], p2[8], p3[8];           // scale, shift, scale2, shift2 of this thread's eight channels
  if (fixed) {
    const int c0 = (int)(((int64_t)blockIdx.x * THREADS + threadIdx.x) % G.cg) * 8;
    ld8f(scale + c0, p0); ld8f(shift + c0, p1); ld8f(scale2 + c0, p2); ld8f(shift2 + c0, p3);
  } else {
    for (int i = threadIdx.x; i < G.C; i += THREADS) {
      prm[i] = scale[i]; prm[G.C + i] = shift[i]; prm[2 * G.C + i] = scale2[i]; prm[3 * G.C + i] = shift2[i];
    }
    __syncthreads();
  }
  const int64_t total = G.M * G.cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    if (!fixed) {
      const int c0 = (int)(i % G.cg) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { p0[j] = prm[c0 + j]; p1[j] = prm[G.C + c0 + j]; p2[j] = prm[2 * G.C + c0 + j]; p3[j] = prm[3 * G.C + c0 + j]; }
    }
    float xv[8], rv[8];
    unpack8(ld16(x + i * 8, G.nt), xv);
    unpack8(ld16(x2 + i * 8, G.nt), rv);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      // the shortcut as its own batch norm would have stored it
      const uint32_t pk = pack2(fmaf(rv[j], p2[j], p3[j]), fmaf(rv[j + 1], p2[j + 1], p3[j + 1]));
      float v0 = fmaf(xv[j], p0[j], p1[j]);
      float v1 = fmaf(xv[j + 1], p0[j + 1], p1[j + 1]);
      v0 += bf_lo(pk); v1 += bf_hi(pk);
      o[j] = RELU ? fmaxf(v0, 0.f) : v0;
      o[j + 1] = RELU ? fmaxf(v1, 0.f) : v1;
    }
    uint4 out;
    out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
    st16(y + i * 8, out, G.nt);
    if (RELU && mbits) {
      uint32_t mb = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) mb |= (o[j] > 0.f ? 1u : 0u) << j;
      mbits[i] = (uint8_t)mb;
    }
  }
}

// One pass over dy for both batch norms: (sum dz, sum dz * xhat) and (sum dz, sum dz * xhat2), dz = dy where the ReLU
// bit is set.  Row partition and combine order are k_reduce's, so each pair of sums has k_reduce's bits.
__global__ __launch_bounds__(THREADS) void k_reduce_pair(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                          const uint8_t* __restrict__ mbits, const uint16_t* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                                          float* __restrict__ partial, float* __restrict__ partial2) {
  __shared__ float red[THREADS][25];
  const int tx = threadIdx.x % G.tpr, ty = threadIdx.x / G.tpr;
  const int cgi = blockIdx.y * G.tpr + tx;
  const bool c_ok = cgi < G.cg;
  const int64_t r0 = (int64_t)blockIdx.x * G.rows_per_part;
  int64_t r1 = r0 + G.rows_per_part;
  if (r1 > G.M) r1 = G.M;
  float q0[8], q1[8], q2[8], mu[8], is[8], mu2[8], is2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { q0[j] = q1[j] = q2[j] = 0.f; mu[j] = is[j] = mu2[j] = is2[j] = 0.f; }
  if (c_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = mean[cgi * 8 + j]; is[j] = invstd[cgi * 8 + j]; mu2[j] = mean2[cgi * 8 + j]; is2[j] = invstd2[cgi * 8 + j];
    }
    const int64_t rb = G.il ? (int64_t)blockIdx.x * G.rpb + ty : r0 + ty, re = G.il ? G.M : r1;
    const int64_t rstep = G.il ? (int64_t)G.parts * G.rpb : (int64_t)G.rpb;
#pragma unroll 2
    for (int64_t r = rb; r < re; r += rstep) {
      const int64_t off = r * G.C + (int64_t)cgi * 8;
      float xv[8], x2v[8], dv[8];
      unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
      unpack8(*reinterpret_cast<const uint4*>(x2 + off), x2v);
      unpack8(*reinterpret_cast<const uint4*>(dy + off), dv);
      const uint32_t mb = mbits[r * G.cg + cgi];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = ((mb >> j) & 1u) != 0u ? dv[j] : 0.f;
        q0[j] += dz;
        q1[j] = fmaf(dz, (xv[j] - mu[j]) * is[j], q1[j]);
        q2[j] = fmaf(dz, (x2v[j] - mu2[j]) * is2[j], q2[j]);
      }
    }
  }
  float own[24];          // (as in k_reduce: own sums in registers, the partner's read with one wait per round)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    own[j] = q0[j]; own[8 + j] = q1[j]; own[16 + j] = q2[j];
    red[threadIdx.x][j] = q0[j]; red[threadIdx.x][8 + j] = q1[j]; red[threadIdx.x][16 + j] = q2[j];
  }
  __syncthreads();
  for (int s = G.rpb >> 1; s > 0; s >>= 1) {
    if (ty < s) {
      float oth[24];
#pragma unroll
      for (int j = 0; j < 24; ++j) oth[j] = red[threadIdx.x + s * G.tpr][j];
#pragma unroll
      for (int j = 0; j < 24; ++j) own[j] += oth[j];
      if (s > 1) {
#pragma unroll
        for (int j = 0; j < 24; ++j) red[threadIdx.x][j] = own[j];
      }
    }
    if (s > 1) __syncthreads();
  }
  if (ty == 0 && c_ok) {
    float* p = partial + ((int64_t)blockIdx.x * 2) * G.C + cgi * 8;
    float* p2 = partial2 + ((int64_t)blockIdx.x * 2) * G.C + cgi * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = own[j]; p[G.C + j] = own[8 + j];
      p2[j] = own[j]; p2[G.C + j] = own[16 + j];
    }
  }
}

struct FinSet { const float* partial; const float* gamma; const float* invstd; float* dgamma; float* dbeta; float* coef; };

template <int CL>
__global__ __launch_bounds__(THREADS) void k_bwd_finalize_pair(Geom G, FinSet s0, FinSet s1) {
  const FinSet& f = blockIdx.y == 0 ? s0 : s1;
  double a0, a1; int c; bool leader;
  sum_partials<CL>(G, f.partial, a0, a1, c, leader);
  if (!leader) return;
  f.dbeta[c] = (float)a0;
  f.dgamma[c] = (float)a1;
  f.coef[c] = f.gamma[c] * f.invstd[c];
  f.coef[G.C + c] = (float)(a0 / (double)G.M);
  f.coef[2 * G.C + c] = (float)(a1 / (double)G.M);
}

// dx = a * (dz - b - xhat * c) and dx2 = a2 * (dz - b2 - xhat2 * c2) in one pass
__global__ __launch_bounds__(THREADS) void k_bwd_apply_pair(Geom G, const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                                             const uint8_t* __restrict__ mbits, const uint16_t* __restrict__ dy,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                                             const float* __restrict__ coef, const float* __restrict__ coef2,
                                                             uint16_t* __restrict__ dx, uint16_t* __restrict__ dx2) {
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [10][C]: mean, invstd, a, b, c of either side (unused when G.fixed)
  const bool fixed = G.fixed != 0;
  float q[10][8];
  if (fixed) {
    const int c0 = (int)(((int64_t)blockIdx.x * THREADS + threadIdx.x) % G.cg) * 8;
    ld8f(mean + c0, q[0]); ld8f(invstd + c0, q[1]); ld8f(coef + c0, q[2]); ld8f(coef + G.C + c0, q[3]); ld8f(coef + 2 * G.C + c0, q[4]);
    ld8f(mean2 + c0, q[5]); ld8f(invstd2 + c0, q[6]); ld8f(coef2 + c0, q[7]); ld8f(coef2 + G.C + c0, q[8]); ld8f(coef2 + 2 * G.C + c0, q[9]);
  } else {
    for (int i = threadIdx.x; i < G.C; i += THREADS) {
      prm[i] = mean[i]; prm[G.C + i] = invstd[i];
      prm[2 * G.C + i] = coef[i]; prm[3 * G.C + i] = coef[G.C + i]; prm[4 * G.C + i] = coef[2 * G.C + i];
      prm[5 * G.C + i] = mean2[i]; prm[6 * G.C + i] = invstd2[i];
      prm[7 * G.C + i] = coef2[i]; prm[8 * G.C + i] = coef2[G.C + i]; prm[9 * G.C + i] = coef2[2 * G.C + i];
    }
    __syncthreads();
  }
  const int64_t total = G.M * G.cg;
  const int64_t stride = (int64_t)gridDim.x * THREADS;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += stride) {
    if (!fixed) {
      const int c0 = (int)(i % G.cg) * 8;
#pragma unroll
      for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) q[k][j] = prm[k * G.C + c0 + j];
    }
    float xv[8], x2v[8], dv[8];
    unpack8(ld16(x + i * 8, G.nt), xv);
    unpack8(ld16(x2 + i * 8, G.nt), x2v);
    unpack8(ld16(dy + i * 8, G.nt), dv);
    const uint32_t mb = mbits[i];
    float o[8], o2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = ((mb >> j) & 1u) != 0u ? dv[j] : 0.f;
      const float xh = (xv[j] - q[0][j]) * q[1][j];
      o[j] = q[2][j] * (dz - q[3][j] - xh * q[4][j]);
      const float xh2 = (x2v[j] - q[5][j]) * q[6][j];
      o2[j] = q[7][j] * (dz - q[8][j] - xh2 * q[9][j]);
    }
    uint4 out;
    out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
    st16(dx + i * 8, out, G.nt);
    out.x = pack2(o2[0], o2[1]); out.y = pack2(o2[2], o2[3]); out.z = pack2(o2[4], o2[5]); out.w = pack2(o2[6], o2[7]);
    st16(dx2 + i * 8, out, G.nt);
  }
}

static Geom make_geom(int64_t m, int c) {
  Geom g;
  // measured in the ResNet-50 step (round 3): 0 -> 2 = -0.06 .. -0.10 ms, the convs gain too (less of their L2 evicted)
  g.nt = (int64_t)m * c * 2 >= (int64_t)RIGL_TUNE("bn_nt_mb", 0) * (1 << 20) ? RIGL_TUNE("bn_nt", 2) : 0;
  g.il = RIGL_TUNE("bn_il", 1);
  g.fixed = 0;
  g.M = m; g.C = c; g.cg = c / 8;
  int tpr = 1;
  while (tpr < g.cg && tpr < THREADS) tpr <<= 1;
  g.tpr = tpr; g.rpb = THREADS / tpr;
#ifndef RIGL_BN_ROWS_PER_LANE
#define RIGL_BN_ROWS_PER_LANE 8    // 16 left the 14x14 layers with 196 workgroups for 256 CUs (19.9 -> 18.1 us)
#endif
  int64_t parts = (m + (int64_t)g.rpb * RIGL_BN_ROWS_PER_LANE - 1) / ((int64_t)g.rpb * RIGL_BN_ROWS_PER_LANE);   // rows per row-lane
  const int64_t cap = RIGL_TUNE("bn_max_parts", MAX_PARTS);
  if (parts > cap) parts = cap;
  if (parts < 1) parts = 1;
  int64_t rpp = (m + parts - 1) / parts;
  rpp = (rpp + g.rpb - 1) / g.rpb * g.rpb;
  g.rows_per_part = rpp;
  g.parts = (int)((m + rpp - 1) / rpp);
  return g;
}

// (sets g.fixed: see Geom)
static unsigned apply_grid(Geom& g) {
// Grid of the element-wise apply passes: 2 x 16 B per thread, at most 16384 workgroups (4 per thread / 4096
// measured 0.8 % slower in the ResNet-50 step: more, shorter workgroups drain the tail of these streams sooner).
#ifndef RIGL_BN_APPLY_PER_THREAD
#define RIGL_BN_APPLY_PER_THREAD 2
#endif
#ifndef RIGL_BN_APPLY_CAP
#define RIGL_BN_APPLY_CAP 16384
#endif
  int64_t b = (g.M * g.cg + THREADS * RIGL_BN_APPLY_PER_THREAD - 1) / (THREADS * RIGL_BN_APPLY_PER_THREAD);
  if (b > RIGL_BN_APPLY_CAP) b = RIGL_BN_APPLY_CAP;
  if (b < 1) b = 1;
  g.fixed = (RIGL_TUNE("bn_regs", 1) != 0 && (b * THREADS) % g.cg == 0) ? 1 : 0;
  return (unsigned)b;
}

// Backward finalize for a caller that took the reductions itself (pool.hip: the fused stem tail).
void launch_bwd_finalize(int64_t m, int c, const float* partial, int parts, const float* gamma, const float* invstd,
                         float* dgamma, float* dbeta, float* coef, hipStream_t st) {
  Geom g = make_geom(m, c);
  g.parts = parts;
  if (parts > MAX_PARTS)
    hipLaunchKernelGGL(k_bwd_finalize<4>, dim3((unsigned)((c + 3) / 4)), dim3(THREADS), 0, st, g, partial, gamma, invstd,
                       dgamma, dbeta, coef);
  else
    hipLaunchKernelGGL(k_bwd_finalize<16>, dim3((unsigned)((c + 15) / 16)), dim3(THREADS), 0, st, g, partial, gamma, invstd,
                       dgamma, dbeta, coef);
}

// Statistics pass (unless the producer left partial sums) + finalize: everything of the forward but the apply pass.
static int fwd_statistics(Geom& g, int32_t c, const rigl_bf16* x, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                          float* save_invstd, float* save_scale, float* save_shift, const float* stats,
                          int32_t stats_parts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const float* partial = stats;
  if (!stats) {
    const size_t need = rigl_bn_workspace_bytes(g.M, c);
    if (!workspace || workspace_bytes < need) return fail(RIGL_EWORKSPACE, "rigl_bn_fwd: workspace %zu < %zu", workspace_bytes, need);
    float* ws_partial = static_cast<float*>(workspace);
    dim3 rgrid((unsigned)g.parts, (unsigned)((g.cg + g.tpr - 1) / g.tpr));
    hipLaunchKernelGGL((k_reduce<0, false, 0>), rgrid, dim3(THREADS), 0, st, g, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, ws_partial);
    partial = ws_partial;
  } else {
    g.parts = stats_parts;               // the producer's partial sums [stats_parts][2][C] replace the reduction pass
  }
  // parts > 1024 (the 56x56 layers: 3136 conv-epilogue partials per channel, 64-256 channels): one workgroup per channel --
  // with 4 channels per workgroup a 64-channel layer is 16 workgroups each walking 49 rows per lane in dependent batches
  // (11.8 us on average over the step's 25 such finalizes; round 3)
  if (g.parts > 1024 && RIGL_TUNE("bn_fin1", 1) != 0)
    hipLaunchKernelGGL(k_fwd_finalize<1>, dim3((unsigned)c), dim3(THREADS), 0, st, g, partial, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, save_scale, save_shift);
  else if (g.parts > 256)
    hipLaunchKernelGGL(k_fwd_finalize<4>, dim3((unsigned)((c + 3) / 4)), dim3(THREADS), 0, st, g, partial, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, save_scale, save_shift);
  else
    hipLaunchKernelGGL(k_fwd_finalize<16>, dim3((unsigned)((c + 15) / 16)), dim3(THREADS), 0, st, g, partial, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, save_scale, save_shift);
  return RIGL_OK;
}

}  // namespace kbn
}  // namespace rigl

extern "C" {

size_t rigl_bn_workspace_bytes(int64_t m, int32_t c) {
  if (m <= 0 || c <= 0) return 0;
  rigl::kbn::Geom g = rigl::kbn::make_geom(m, c);
  // partial sums [parts][2][C] + scale/shift or coef [3][C]
  return rigl::align_up((size_t)g.parts * 2 * c * 4, 256) + rigl::align_up((size_t)3 * c * 4, 256);
}

int rigl_bn_fwd_stats(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* residual, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      int32_t relu, rigl_bf16* y, float* save_mean, float* save_invstd, float* save_scale,
                      float* save_shift, const float* stats, int32_t stats_parts, uint8_t* relu_bits,
                      void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kbn;
  if (m <= 0 || c <= 0 || !x || !gamma || !beta || !y || !save_mean || !save_invstd || !save_scale || !save_shift)
    return fail(RIGL_EINVAL, "rigl_bn_fwd: bad arguments");
  if (c % 8) return fail(RIGL_EUNSUPPORTED, "rigl_bn_fwd: channels %% 8 != 0");
  if (2 * (size_t)c * 4 > 65536) return fail(RIGL_EUNSUPPORTED, "rigl_bn_fwd: too many channels for the LDS parameter cache");
  if (stats && stats_parts <= 0) return fail(RIGL_EINVAL, "rigl_bn_fwd_stats: stats_parts must be positive");
  hipStream_t st = as_stream(stream);
  Geom g = make_geom(m, c);
  int rc = fwd_statistics(g, c, x, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                          save_scale, save_shift, stats, stats_parts, workspace, workspace_bytes, st);
  if (rc) return rc;
  dim3 agrid(apply_grid(g));
  const size_t lds = g.fixed ? 0 : (size_t)2 * c * 4;
  if (relu && residual) hipLaunchKernelGGL((k_fwd_apply<true, true>), agrid, dim3(THREADS), lds, st, g, x, residual, save_scale, save_shift, y, relu_bits);
  else if (relu) hipLaunchKernelGGL((k_fwd_apply<true, false>), agrid, dim3(THREADS), lds, st, g, x, residual, save_scale, save_shift, y, relu_bits);
  else if (residual) hipLaunchKernelGGL((k_fwd_apply<false, true>), agrid, dim3(THREADS), lds, st, g, x, residual, save_scale, save_shift, y, nullptr);
  else hipLaunchKernelGGL((k_fwd_apply<false, false>), agrid, dim3(THREADS), lds, st, g, x, residual, save_scale, save_shift, y, nullptr);
  RIGL_CHECK_LAUNCH("rigl_bn_fwd");
  return RIGL_OK;
}

int rigl_bn_fwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* residual, const float* gamma,
                const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t relu,
                rigl_bf16* y, float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  return rigl_bn_fwd_stats(m, c, x, residual, gamma, beta, running_mean, running_var, momentum, eps, relu, y, save_mean,
                           save_invstd, save_scale, save_shift, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
}

int rigl_bn_fwd_statistics(int64_t m, int32_t c, const rigl_bf16* x, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                           float* save_invstd, float* save_scale, float* save_shift, const float* stats,
                           int32_t stats_parts, void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kbn;
  if (m <= 0 || c <= 0 || (!x && !stats) || !gamma || !beta || !save_mean || !save_invstd || !save_scale || !save_shift)
    return fail(RIGL_EINVAL, "rigl_bn_fwd_statistics: bad arguments");
  if (c % 8) return fail(RIGL_EUNSUPPORTED, "rigl_bn_fwd_statistics: channels %% 8 != 0");
  if (stats && stats_parts <= 0) return fail(RIGL_EINVAL, "rigl_bn_fwd_statistics: stats_parts must be positive");
  Geom g = make_geom(m, c);
  int rc = fwd_statistics(g, c, x, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                          save_scale, save_shift, stats, stats_parts, workspace, workspace_bytes, as_stream(stream));
  if (rc) return rc;
  RIGL_CHECK_LAUNCH("rigl_bn_fwd_statistics");
  return RIGL_OK;
}

int rigl_bn_add_bn_fwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* x2, const float* scale2,
                       const float* shift2, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int32_t relu, rigl_bf16* y, float* save_mean,
                       float* save_invstd, float* save_scale, float* save_shift, const float* stats,
                       int32_t stats_parts, uint8_t* relu_bits, void* workspace, size_t workspace_bytes,
                       rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kbn;
  if (m <= 0 || c <= 0 || !x || !x2 || !scale2 || !shift2 || !gamma || !beta || !y || !save_mean || !save_invstd ||
      !save_scale || !save_shift)
    return fail(RIGL_EINVAL, "rigl_bn_add_bn_fwd: bad arguments");
  if (c % 8) return fail(RIGL_EUNSUPPORTED, "rigl_bn_add_bn_fwd: channels %% 8 != 0");
  if (4 * (size_t)c * 4 > 65536) return fail(RIGL_EUNSUPPORTED, "rigl_bn_add_bn_fwd: too many channels for the LDS parameter cache");
  if (stats && stats_parts <= 0) return fail(RIGL_EINVAL, "rigl_bn_add_bn_fwd: stats_parts must be positive");
  hipStream_t st = as_stream(stream);
  Geom g = make_geom(m, c);
  int rc = fwd_statistics(g, c, x, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                          save_scale, save_shift, stats, stats_parts, workspace, workspace_bytes, st);
  if (rc) return rc;
  dim3 agrid(apply_grid(g));
  const size_t lds = g.fixed ? 0 : (size_t)4 * c * 4;
  if (relu) hipLaunchKernelGGL(k_fwd_apply_pair<true>, agrid, dim3(THREADS), lds, st, g, x, x2, save_scale, save_shift, scale2, shift2, y, relu_bits);
  else hipLaunchKernelGGL(k_fwd_apply_pair<false>, agrid, dim3(THREADS), lds, st, g, x, x2, save_scale, save_shift, scale2, shift2, y, nullptr);
  RIGL_CHECK_LAUNCH("rigl_bn_add_bn_fwd");
  return RIGL_OK;
}

size_t rigl_bn_add_bn_bwd_workspace_bytes(int64_t m, int32_t c) { return 2 * rigl_bn_workspace_bytes(m, c); }

int rigl_bn_add_bn_bwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* x2, const uint8_t* relu_bits,
                       const rigl_bf16* dy, const float* gamma, const float* save_mean, const float* save_invstd,
                       const float* gamma2, const float* save_mean2, const float* save_invstd2, rigl_bf16* dx,
                       rigl_bf16* dx2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, void* workspace,
                       size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kbn;
  if (m <= 0 || c <= 0 || !x || !x2 || !relu_bits || !dy || !gamma || !save_mean || !save_invstd || !gamma2 || !save_mean2 ||
      !save_invstd2 || !dx || !dx2 || !dgamma || !dbeta || !dgamma2 || !dbeta2)
    return fail(RIGL_EINVAL, "rigl_bn_add_bn_bwd: bad arguments");
  if (c % 8) return fail(RIGL_EUNSUPPORTED, "rigl_bn_add_bn_bwd: channels %% 8 != 0");
  const size_t lds = (size_t)10 * c * 4;
  if (lds > 160 * 1024 - 1024) return fail(RIGL_EUNSUPPORTED, "rigl_bn_add_bn_bwd: too many channels for the LDS parameter cache");
  const size_t need = rigl_bn_add_bn_bwd_workspace_bytes(m, c);
  if (!workspace || workspace_bytes < need) return fail(RIGL_EWORKSPACE, "rigl_bn_add_bn_bwd: workspace %zu < %zu", workspace_bytes, need);
  static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_apply_pair),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) == hipSuccess;
  if (lds > 65536 && !big_lds) return fail(RIGL_ELAUNCH, "rigl_bn_add_bn_bwd: cannot raise the dynamic LDS limit");
  hipStream_t st = as_stream(stream);
  Geom g = make_geom(m, c);
  const size_t half = rigl_bn_workspace_bytes(m, c);
  const size_t part_bytes = align_up((size_t)g.parts * 2 * c * 4, 256);
  char* w0 = static_cast<char*>(workspace);
  float* partial = reinterpret_cast<float*>(w0);
  float* coef = reinterpret_cast<float*>(w0 + part_bytes);
  float* partial2 = reinterpret_cast<float*>(w0 + half);
  float* coef2 = reinterpret_cast<float*>(w0 + half + part_bytes);
  dim3 rgrid((unsigned)g.parts, (unsigned)((g.cg + g.tpr - 1) / g.tpr));
  hipLaunchKernelGGL(k_reduce_pair, rgrid, dim3(THREADS), 0, st, g, x, x2, relu_bits, dy, save_mean, save_invstd, save_mean2,
                     save_invstd2, partial, partial2);
  const FinSet f0 = {partial, gamma, save_invstd, dgamma, dbeta, coef};
  const FinSet f1 = {partial2, gamma2, save_invstd2, dgamma2, dbeta2, coef2};
  hipLaunchKernelGGL(k_bwd_finalize_pair<16>, dim3((unsigned)((c + 15) / 16), 2), dim3(THREADS), 0, st, g, f0, f1);
  const dim3 pgrid(apply_grid(g));
  hipLaunchKernelGGL(k_bwd_apply_pair, pgrid, dim3(THREADS), g.fixed ? 0 : lds, st, g, x, x2, relu_bits, dy, save_mean, save_invstd,
                     save_mean2, save_invstd2, coef, coef2, dx, dx2);
  RIGL_CHECK_LAUNCH("rigl_bn_add_bn_bwd");
  return RIGL_OK;
}

int rigl_bn_bwd_stats(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* y, const uint8_t* relu_bits,
                      const rigl_bf16* dy, const float* gamma,
                      const float* save_mean, const float* save_invstd, const float* save_scale, const float* save_shift,
                      int32_t relu, rigl_bf16* dx, rigl_bf16* dresidual, float* dgamma, float* dbeta,
                      const float* stats, int32_t stats_parts, void* workspace,
                      size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kbn;
  if (m <= 0 || c <= 0 || !x || !dy || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta)
    return fail(RIGL_EINVAL, "rigl_bn_bwd: bad arguments");
  if (c % 8) return fail(RIGL_EUNSUPPORTED, "rigl_bn_bwd: channels %% 8 != 0");
  if (relu && !y && !relu_bits && (!save_scale || !save_shift))
    return fail(RIGL_EINVAL, "rigl_bn_bwd: relu needs y, relu_bits or scale/shift");
  if (7 * (size_t)c * 4 > 65536) return fail(RIGL_EUNSUPPORTED, "rigl_bn_bwd: too many channels for the LDS parameter cache");
  if (stats && stats_parts <= 0) return fail(RIGL_EINVAL, "rigl_bn_bwd_stats: stats_parts must be positive");
  const size_t need = rigl_bn_workspace_bytes(m, c);
  if (!workspace || workspace_bytes < need) return fail(RIGL_EWORKSPACE, "rigl_bn_bwd: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  Geom g = make_geom(m, c);
  float* partial = static_cast<float*>(workspace);
  float* coef = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)g.parts * 2 * c * 4, 256));
  dim3 rgrid((unsigned)g.parts, (unsigned)((g.cg + g.tpr - 1) / g.tpr));
  const int msk = relu_bits ? 2 : (y ? 1 : 0);
  const float* red = partial;
  if (stats) {
    // the producer of dy (a dgrad epilogue, rigl_masked_conv2d_bwd_bn) left sum dz / sum dz * xhat per row tile:
    // the reduction pass over dy and x is not run at all
    red = stats;
    g.parts = stats_parts;
  } else {
#define RIGL_BWD_REDUCE(R, K) hipLaunchKernelGGL((k_reduce<1, R, K>), rgrid, dim3(THREADS), 0, st, g, x, y, relu_bits, dy, save_mean, save_invstd, save_scale, save_shift, partial)
    if (!relu) RIGL_BWD_REDUCE(false, 0);
    else if (msk == 2) RIGL_BWD_REDUCE(true, 2);
    else if (msk == 1) RIGL_BWD_REDUCE(true, 1);
    else RIGL_BWD_REDUCE(true, 0);
#undef RIGL_BWD_REDUCE
  }
  if (g.parts > MAX_PARTS)      // (more partial rows than 512: a dgrad epilogue's, or a raised "bn_max_parts"; <16> is 0.7 us faster below it)
    hipLaunchKernelGGL(k_bwd_finalize<4>, dim3((unsigned)((c + 3) / 4)), dim3(THREADS), 0, st, g, red, gamma,
                       save_invstd, dgamma, dbeta, coef);
  else
    hipLaunchKernelGGL(k_bwd_finalize<16>, dim3((unsigned)((c + 15) / 16)), dim3(THREADS), 0, st, g, red, gamma,
                       save_invstd, dgamma, dbeta, coef);
  dim3 agrid(apply_grid(g));
  const size_t lds = g.fixed ? 0 : (size_t)7 * c * 4;
#define RIGL_BWD_APPLY(R, K, D) hipLaunchKernelGGL((k_bwd_apply<R, K, D>), agrid, dim3(THREADS), lds, st, g, x, y, relu_bits, dy, save_mean, save_invstd, save_scale, save_shift, coef, dx, dresidual)
  const bool dres = dresidual != nullptr;
  if (!relu) { if (dres) RIGL_BWD_APPLY(false, 0, true); else RIGL_BWD_APPLY(false, 0, false); }
  else if (msk == 2) { if (dres) RIGL_BWD_APPLY(true, 2, true); else RIGL_BWD_APPLY(true, 2, false); }
  else if (msk == 1) { if (dres) RIGL_BWD_APPLY(true, 1, true); else RIGL_BWD_APPLY(true, 1, false); }
  else { if (dres) RIGL_BWD_APPLY(true, 0, true); else RIGL_BWD_APPLY(true, 0, false); }
#undef RIGL_BWD_APPLY
  RIGL_CHECK_LAUNCH("rigl_bn_bwd");
  return RIGL_OK;
}

int rigl_bn_bwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* y, const uint8_t* relu_bits,
                const rigl_bf16* dy, const float* gamma,
                const float* save_mean, const float* save_invstd, const float* save_scale, const float* save_shift,
                int32_t relu, rigl_bf16* dx, rigl_bf16* dresidual, float* dgamma, float* dbeta, void* workspace,
                size_t workspace_bytes, rigl_stream_t stream) {
  return rigl_bn_bwd_stats(m, c, x, y, relu_bits, dy, gamma, save_mean, save_invstd, save_scale, save_shift, relu, dx,
                           dresidual, dgamma, dbeta, nullptr, 0, workspace, workspace_bytes, stream);
}

}  // extern "C"
