// Max pooling (forward + backward) for NHWC bf16 activations.
//
// Glue, not a graded hot-path row: the reference's ResNet-50 has exactly one
// tf.layers.max_pooling2d(pool_size=3, strides=2, padding='SAME') after the stem
// (rigl/imagenet_resnet/resnet_model.py:637-644), but at batch 128 its tensor
// is the largest activation of the network (128x112x112x64 = 205 MB) and the
// stock NHWC backward took 0.31 ms / step.  Both kernels are HBM-bound
// streams, one thread per (pixel, 8-channel group), 16 B per access:
//   forward : y = max over the window (first maximum in row-major window order,
//             like tf.nn.max_pool / torch), idx = r*kw+s of the winner (1 B/elem)
//   backward: gather -- dx[h,w] = sum of dy[ho,wo] over the <= ceil(kh/sh)*ceil(kw/sw)
//             windows that contain (h,w) and whose winner is (h,w).  No atomics,
//             deterministic, every dx element written exactly once.
// Padding is explicit (pad_top/left; windows are clipped to the image), which
// covers TF 'SAME' (asymmetric on even sizes) and 'VALID'.
#include "common.hpp"

namespace rigl {
namespace kpool {

constexpr int THREADS = 256;

struct Geom {
  int n, h, w, c, cg;
  int ho, wo, kh, kw, sh, sw, pt, pl;
};

__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ void unpack8(const uint4& v, float f[8]) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

__global__ __launch_bounds__(THREADS) void k_fwd(Geom G, const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                  uint8_t* __restrict__ idx) {
  const int64_t total = (int64_t)G.n * G.ho * G.wo * G.cg;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
    const int cgi = (int)(i % G.cg);
    int64_t p = i / G.cg;
    const int wo = (int)(p % G.wo); p /= G.wo;
    const int ho = (int)(p % G.ho);
    const int n = (int)(p / G.ho);
    uint32_t best[8];          // bf16 bit patterns of the running maxima
    float bestf[8];
    uint32_t arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = 0xFF80u; bestf[j] = -INFINITY; arg[j] = 0; }
    for (int r = 0; r < G.kh; ++r) {
      const int hi = ho * G.sh - G.pt + r;
      if ((unsigned)hi >= (unsigned)G.h) continue;
      for (int s = 0; s < G.kw; ++s) {
        const int wi = wo * G.sw - G.pl + s;
        if ((unsigned)wi >= (unsigned)G.w) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((int64_t)n * G.h + hi) * G.w + wi) * G.c + cgi * 8);
        float f[8];
        unpack8(v, f);
        const uint32_t raw[8] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16,
                                 v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > bestf[j]) { bestf[j] = f[j]; best[j] = raw[j]; arg[j] = (uint32_t)(r * G.kw + s); }
      }
    }
    uint4 o;
    o.x = best[0] | (best[1] << 16); o.y = best[2] | (best[3] << 16);
    o.z = best[4] | (best[5] << 16); o.w = best[6] | (best[7] << 16);
    *reinterpret_cast<uint4*>(y + i * 8) = o;
    uint2 a;
    a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = a;
  }
}

__global__ __launch_bounds__(THREADS) void k_bwd(Geom G, const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                  uint16_t* __restrict__ dx) {
  const int64_t total = (int64_t)G.n * G.h * G.w * G.cg;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
    const int cgi = (int)(i % G.cg);
    int64_t p = i / G.cg;
    const int w = (int)(p % G.w); p /= G.w;
    const int h = (int)(p % G.h);
    const int n = (int)(p / G.h);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int r = 0; r < G.kh; ++r) {
      const int th = h + G.pt - r;
      if (th < 0 || th % G.sh) continue;
      const int ho = th / G.sh;
      if (ho >= G.ho) continue;
      for (int s = 0; s < G.kw; ++s) {
        const int tw = w + G.pl - s;
        if (tw < 0 || tw % G.sw) continue;
        const int wo = tw / G.sw;
        if (wo >= G.wo) continue;
        const int64_t o = (((int64_t)n * G.ho + ho) * G.wo + wo) * G.cg + cgi;
        const uint2 a = *reinterpret_cast<const uint2*>(idx + o * 8);
        const uint4 v = *reinterpret_cast<const uint4*>(dy + o * 8);
        float f[8];
        unpack8(v, f);
        const uint32_t code = (uint32_t)(r * G.kw + s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t aj = ((j < 4 ? a.x : a.y) >> (8 * (j & 3))) & 0xFFu;
          if (aj == code) acc[j] += f[j];
        }
      }
    }
    uint4 out;
    out.x = f2bf(acc[0]) | (f2bf(acc[1]) << 16); out.y = f2bf(acc[2]) | (f2bf(acc[3]) << 16);
    out.z = f2bf(acc[4]) | (f2bf(acc[5]) << 16); out.w = f2bf(acc[6]) | (f2bf(acc[7]) << 16);
    *reinterpret_cast<uint4*>(dx + i * 8) = out;
  }
}

// The ResNet case (3x3 window, stride 2), one thread per 2x2 block of input pixels (in padded coordinates
// hp = h + pad_top in {2k, 2k+1}, wp likewise) and 8 channels.  Those four pixels can only be the winners of
// the four windows (k-1..k) x (l-1..l), so a thread loads 4 (argmax, dy) pairs for 4 outputs where the
// per-pixel gather loads up to 4 per output (2.25 on average) -- and all window arithmetic is compile-time
// (the generic kernel above spends its time in runtime divisions: 151 us for the 205 MB stem gradient).
// Each pixel still adds its candidates in (r, s) ascending order: bit-identical to the generic kernel.
__global__ __launch_bounds__(THREADS) void k_bwd_3x3s2(Geom G, int kb, int lb, const uint16_t* __restrict__ dy,
                                                        const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx) {
  const uint32_t total = (uint32_t)G.n * kb * lb * G.cg;        // < 2^31, checked by the caller
  for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < total; i += gridDim.x * THREADS) {
    const uint32_t cgi = i % (uint32_t)G.cg;
    uint32_t p = i / (uint32_t)G.cg;
    const int l = (int)(p % (uint32_t)lb); p /= (uint32_t)lb;
    const int k = (int)(p % (uint32_t)kb);
    const int n = (int)(p / (uint32_t)kb);
    uint2 av[2][2];
    float f[2][2][8];
    bool ok[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = k - 1 + a, wo = l - 1 + b;
        ok[a][b] = ho >= 0 && ho < G.ho && wo >= 0 && wo < G.wo;
        if (ok[a][b]) {
          const uint32_t o = ((uint32_t)(n * G.ho + ho) * (uint32_t)G.wo + (uint32_t)wo) * (uint32_t)G.cg + cgi;
          av[a][b] = *reinterpret_cast<const uint2*>(idx + (size_t)o * 8);
          unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)o * 8), f[a][b]);
        }
      }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int h = 2 * k + ph - G.pt;
      if (h < 0 || h >= G.h) continue;
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        const int w = 2 * l + pw - G.pl;
        if (w < 0 || w >= G.w) continue;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // even padded row: taps r = 0 (window row k) then r = 2 (window row k-1); odd: r = 1 (window row k)
#pragma unroll
        for (int ra = 0; ra < 2; ++ra) {
          if (ph == 1 && ra == 1) continue;
          const int r = ph + 2 * ra, wa = (ph == 0 && ra == 1) ? 0 : 1;
#pragma unroll
          for (int sb = 0; sb < 2; ++sb) {
            if (pw == 1 && sb == 1) continue;
            const int s2 = pw + 2 * sb, wb = (pw == 0 && sb == 1) ? 0 : 1;
            if (!ok[wa][wb]) continue;
            const uint32_t code = (uint32_t)(r * 3 + s2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const uint32_t aj = ((j < 4 ? av[wa][wb].x : av[wa][wb].y) >> (8 * (j & 3))) & 0xFFu;
              if (aj == code) acc[j] += f[wa][wb][j];
            }
          }
        }
        uint4 out;
        out.x = f2bf(acc[0]) | (f2bf(acc[1]) << 16); out.y = f2bf(acc[2]) | (f2bf(acc[3]) << 16);
        out.z = f2bf(acc[4]) | (f2bf(acc[5]) << 16); out.w = f2bf(acc[6]) | (f2bf(acc[7]) << 16);
        const uint32_t xi = ((uint32_t)(n * G.h + h) * (uint32_t)G.w + (uint32_t)w) * (uint32_t)G.cg + cgi;
        *reinterpret_cast<uint4*>(dx + (size_t)xi * 8) = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Stem tail: relu(batch_norm(x)) -> 3x3 / 2 max pooling without the activated tensor ever existing in HBM
// (resnet_model.py:631-644; at batch 128 it is 205 MB written by the batch norm and read back by the pooling, and in
// the backward the pooling's input gradient is another 205 MB written once and read twice by the batch norm).
//   forward : the pooling kernel above with a = bf16(relu(x * scale + shift)) formed on load -- the same rounding
//             point as the two-kernel form, so outputs and argmax bytes are bit-identical to it;
//   backward: the batch norm's two passes gather their input gradient (bf16-rounded, as the pooling backward would
//             have stored it) from dy / argmax on the fly: one thread per 2x2 block of pixels x 8 channels as in
//             k_bwd_3x3s2.  PASS 0 leaves sum dz, sum dz * xhat per workgroup (fixed order), PASS 1 writes dx.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// v_cvt_pk_bf16_f32: round-to-nearest-even pack of two floats (the software f2bf above is ~8 VALU operations per
// element, and these kernels are VALU-bound: 9 taps x 8 channels per output)
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__global__ __launch_bounds__(THREADS) void k_bn_relu_fwd(Geom G, const uint16_t* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          uint16_t* __restrict__ y, uint8_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float prm[];   // [2][C]
  for (int i = threadIdx.x; i < G.c; i += THREADS) { prm[i] = scale[i]; prm[G.c + i] = shift[i]; }
  __syncthreads();
  const int64_t total = (int64_t)G.n * G.ho * G.wo * G.cg;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * THREADS) {
    const int cgi = (int)(i % G.cg);
    int64_t p = i / G.cg;
    const int wo = (int)(p % G.wo); p /= G.wo;
    const int ho = (int)(p % G.ho);
    const int n = (int)(p / G.ho);
    float bestf[8];            // running maxima: bf16 values held as floats (their low 16 bits are zero)
    uint32_t arg[8];
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bestf[j] = -INFINITY; arg[j] = 0; sc[j] = prm[cgi * 8 + j]; sh[j] = prm[G.c + cgi * 8 + j]; }
    if (G.kh == 3 && G.kw == 3) {
      // the 3 x 3 window (the ImageNet stem): nine unconditional loads at clamped positions, all in flight, then the same
      // comparisons in the same order (a load under `continue` waits for itself: nine round trips per output)
      uint4 xr[9];
      bool vld[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = ho * G.sh - G.pt + t / 3, wi = wo * G.sw - G.pl + t % 3;
        vld[t] = (unsigned)hi < (unsigned)G.h && (unsigned)wi < (unsigned)G.w;
        const int hc = hi < 0 ? 0 : (hi >= G.h ? G.h - 1 : hi), wc = wi < 0 ? 0 : (wi >= G.w ? G.w - 1 : wi);
        xr[t] = *reinterpret_cast<const uint4*>(x + (((int64_t)n * G.h + hc) * G.w + wc) * G.c + cgi * 8);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (!vld[t]) continue;
        float f[8];
        unpack8(xr[t], f);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const uint32_t pk = pack2(fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f), fmaxf(fmaf(f[j + 1], sc[j + 1], sh[j + 1]), 0.f));
          const float a0 = bf_lo(pk), a1 = bf_hi(pk);
          if (a0 > bestf[j]) { bestf[j] = a0; arg[j] = (uint32_t)t; }
          if (a1 > bestf[j + 1]) { bestf[j + 1] = a1; arg[j + 1] = (uint32_t)t; }
        }
      }
    } else
    for (int r = 0; r < G.kh; ++r) {
      const int hi = ho * G.sh - G.pt + r;
      if ((unsigned)hi >= (unsigned)G.h) continue;
      for (int s = 0; s < G.kw; ++s) {
        const int wi = wo * G.sw - G.pl + s;
        if ((unsigned)wi >= (unsigned)G.w) continue;
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (((int64_t)n * G.h + hi) * G.w + wi) * G.c + cgi * 8), f);
        const uint32_t code = (uint32_t)(r * G.kw + s);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const uint32_t pk = pack2(fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f), fmaxf(fmaf(f[j + 1], sc[j + 1], sh[j + 1]), 0.f));
          const float a0 = bf_lo(pk), a1 = bf_hi(pk);
          if (a0 > bestf[j]) { bestf[j] = a0; arg[j] = code; }
          if (a1 > bestf[j + 1]) { bestf[j + 1] = a1; arg[j + 1] = code; }
        }
      }
    }
    uint4 o;
    o.x = (__float_as_uint(bestf[0]) >> 16) | (__float_as_uint(bestf[1]) & 0xFFFF0000u);
    o.y = (__float_as_uint(bestf[2]) >> 16) | (__float_as_uint(bestf[3]) & 0xFFFF0000u);
    o.z = (__float_as_uint(bestf[4]) >> 16) | (__float_as_uint(bestf[5]) & 0xFFFF0000u);
    o.w = (__float_as_uint(bestf[6]) >> 16) | (__float_as_uint(bestf[7]) & 0xFFFF0000u);
    *reinterpret_cast<uint4*>(y + i * 8) = o;
    uint2 a;
    a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = a;
  }
}

constexpr int STEM_PARTS = 1536;          // workgroups (= partial rows) of the reduction pass: one resident round (256 CUs x 6)

// Four channels per thread (8 B of x / dy, 4 argmax bytes): with eight the kernel needs 130 registers for its per-channel
// constants and window values and runs three waves per SIMD, too few for a gather that waits on twelve loads per item.
constexpr int SCH = 4;
template <int PASS>
__global__ __launch_bounds__(THREADS) void k_bn_relu_bwd(Geom G, int kb, int lb, const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ coef, float* __restrict__ partial,
                                                          uint16_t* __restrict__ dx) {
  __shared__ float red[THREADS / 64][64][2 * SCH + 1];
  // ng = C / 4 divides 64 (checked by the caller), so a thread keeps its channels over the whole grid-stride loop
  const uint32_t ng = (uint32_t)G.c / SCH;
  const uint32_t gi = threadIdx.x % ng;
  float mu[SCH], is[SCH], sc[SCH], sh[SCH], ca[SCH], cb[SCH], cc[SCH], q[2 * SCH];
#pragma unroll
  for (int j = 0; j < SCH; ++j) {
    const int c = (int)gi * SCH + j;
    mu[j] = mean[c]; is[j] = invstd[c]; sc[j] = scale[c]; sh[j] = shift[c];
    ca[j] = cb[j] = cc[j] = 0.f;
    if (PASS == 1) { ca[j] = coef[c]; cb[j] = coef[G.c + c]; cc[j] = coef[2 * G.c + c]; }
    q[j] = q[SCH + j] = 0.f;
  }
  const uint32_t total = (uint32_t)G.n * kb * lb * ng;          // < 2^31, checked by the caller
  for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < total; i += gridDim.x * THREADS) {
    uint32_t p = i / ng;
    const int l = (int)(p % (uint32_t)lb); p /= (uint32_t)lb;
    const int k = (int)(p % (uint32_t)kb);
    const int n = (int)(p / (uint32_t)kb);
    uint32_t av[2][2];
    float f[2][2][SCH];
    bool ok[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = k - 1 + a, wo = l - 1 + b;
        ok[a][b] = ho >= 0 && ho < G.ho && wo >= 0 && wo < G.wo;
        // unconditional loads at clamped positions (a load inside `if` waits for itself: the twelve loads of an item ran as
        // twelve round trips); a window that does not exist is never added (ok)
        const int hoc = ho < 0 ? 0 : (ho >= G.ho ? G.ho - 1 : ho), woc = wo < 0 ? 0 : (wo >= G.wo ? G.wo - 1 : wo);
        const uint32_t o = ((uint32_t)(n * G.ho + hoc) * (uint32_t)G.wo + (uint32_t)woc) * ng + gi;
        av[a][b] = *reinterpret_cast<const uint32_t*>(idx + (size_t)o * SCH);
        const uint2 v = *reinterpret_cast<const uint2*>(dy + (size_t)o * SCH);
        f[a][b][0] = bf_lo(v.x); f[a][b][1] = bf_hi(v.x); f[a][b][2] = bf_lo(v.y); f[a][b][3] = bf_hi(v.y);
      }
    uint2 xq[2][2];
    uint32_t xiq[2][2];
    bool xok[2][2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        const int h = 2 * k + ph - G.pt, w = 2 * l + pw - G.pl;
        xok[ph][pw] = h >= 0 && h < G.h && w >= 0 && w < G.w;
        const int hc = h < 0 ? 0 : (h >= G.h ? G.h - 1 : h), wc = w < 0 ? 0 : (w >= G.w ? G.w - 1 : w);
        xiq[ph][pw] = ((uint32_t)(n * G.h + hc) * (uint32_t)G.w + (uint32_t)wc) * ng + gi;
        xq[ph][pw] = *reinterpret_cast<const uint2*>(x + (size_t)xiq[ph][pw] * SCH);
      }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        if (!xok[ph][pw]) continue;
        const uint32_t xi = xiq[ph][pw];
        const uint2 xr = xq[ph][pw];
        const float xv[SCH] = {bf_lo(xr.x), bf_hi(xr.x), bf_lo(xr.y), bf_hi(xr.y)};
        float acc[SCH];
#pragma unroll
        for (int j = 0; j < SCH; ++j) acc[j] = 0.f;
        // even padded row: taps r = 0 (window row k) then r = 2 (window row k-1); odd: r = 1 (window row k)
#pragma unroll
        for (int ra = 0; ra < 2; ++ra) {
          if (ph == 1 && ra == 1) continue;
          const int r = ph + 2 * ra, wa = (ph == 0 && ra == 1) ? 0 : 1;
#pragma unroll
          for (int sb = 0; sb < 2; ++sb) {
            if (pw == 1 && sb == 1) continue;
            const int s2 = pw + 2 * sb, wb = (pw == 0 && sb == 1) ? 0 : 1;
            if (!ok[wa][wb]) continue;
            const uint32_t code = (uint32_t)(r * 3 + s2);
#pragma unroll
            for (int j = 0; j < SCH; ++j)
              if (((av[wa][wb] >> (8 * j)) & 0xFFu) == code) acc[j] += f[wa][wb][j];
          }
        }
        float o[SCH];
#pragma unroll
        for (int j = 0; j < SCH; j += 2) {
          const uint32_t pk = pack2(acc[j], acc[j + 1]);                   // the gradient the pooling backward would have stored
          const float da[2] = {bf_lo(pk), bf_hi(pk)};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int je = j + e;
            const float dz = fmaf(xv[je], sc[je], sh[je]) > 0.f ? da[e] : 0.f;
            const float xh = (xv[je] - mu[je]) * is[je];
            if (PASS == 0) { q[je] += dz; q[SCH + je] = fmaf(dz, xh, q[SCH + je]); }
            else o[je] = ca[je] * (dz - cb[je] - xh * cc[je]);
          }
        }
        if (PASS == 1) {
          uint2 out;
          out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]);
          *reinterpret_cast<uint2*>(dx + (size_t)xi * SCH) = out;
        }
      }
    }
  }
  if (PASS == 0) {
    // lanes ng apart hold the same channels: butterfly over them, one LDS row per (wave, group), waves in order
    for (uint32_t off = 32; off >= ng; off >>= 1)
#pragma unroll
      for (int j = 0; j < 2 * SCH; ++j) q[j] += __shfl_xor(q[j], (int)off, 64);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < ng)
#pragma unroll
      for (int j = 0; j < 2 * SCH; ++j) red[wave][lane][j] = q[j];
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ng * 2 * SCH; t += THREADS) {
      const uint32_t g2 = t / (2 * SCH), v = t % (2 * SCH);
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < THREADS / 64; ++wv) sum += red[wv][g2][v];
      partial[((int64_t)blockIdx.x * 2 + v / SCH) * G.c + g2 * SCH + v % SCH] = sum;
    }
  }
}

static int make_geom(const RiglConvDesc* d, Geom* g, const char* who) {
  if (!d) return fail(RIGL_EINVAL, "%s: NULL descriptor", who);
  if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->ho <= 0 || d->wo <= 0 || d->kh <= 0 || d->kw <= 0 ||
      d->stride_h <= 0 || d->stride_w <= 0 || d->pad_top < 0 || d->pad_left < 0)
    return fail(RIGL_EINVAL, "%s: non-positive dimension in descriptor", who);
  if (d->cin != d->cout) return fail(RIGL_EINVAL, "%s: pooling keeps the channel count (cin != cout)", who);
  if (d->cin % 8) return fail(RIGL_EUNSUPPORTED, "%s: channels %% 8 != 0", who);
  if (d->kh * d->kw > 255) return fail(RIGL_EUNSUPPORTED, "%s: window larger than 255 taps", who);
  if ((d->ho - 1) * d->stride_h - d->pad_top >= d->h || (d->wo - 1) * d->stride_w - d->pad_left >= d->w)
    return fail(RIGL_EINVAL, "%s: a window lies entirely outside the image", who);
  g->n = d->n; g->h = d->h; g->w = d->w; g->c = d->cin; g->cg = d->cin / 8;
  g->ho = d->ho; g->wo = d->wo; g->kh = d->kh; g->kw = d->kw; g->sh = d->stride_h; g->sw = d->stride_w;
  g->pt = d->pad_top; g->pl = d->pad_left;
  return RIGL_OK;
}

static unsigned grid_for(int64_t items) {
  int64_t b = (items + THREADS - 1) / THREADS;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static int stem_check(const Geom& g, const char* who) {
  if (g.kh != 3 || g.kw != 3 || g.sh != 2 || g.sw != 2)
    return fail(RIGL_EUNSUPPORTED, "%s: 3x3 window, stride 2 only", who);
  if (g.c / SCH > 64 || 64 % (g.c / SCH)) return fail(RIGL_EUNSUPPORTED, "%s: channels / 4 must divide 64", who);
  if ((int64_t)g.n * g.h * g.w * (g.c / SCH) >= (int64_t(1) << 31)) return fail(RIGL_EUNSUPPORTED, "%s: tensor too large", who);
  return RIGL_OK;
}

}  // namespace kpool
namespace kbn {
void launch_bwd_finalize(int64_t m, int c, const float* partial, int parts, const float* gamma, const float* invstd,
                         float* dgamma, float* dbeta, float* coef, hipStream_t st);      // bn.hip
}
}  // namespace rigl

extern "C" {

int rigl_bn_relu_maxpool_fwd(const RiglConvDesc* d, const rigl_bf16* x, const float* scale, const float* shift,
                             rigl_bf16* y, uint8_t* argmax, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kpool;
  Geom g;
  int rc = make_geom(d, &g, "rigl_bn_relu_maxpool_fwd");
  if (rc) return rc;
  if (!x || !scale || !shift || !y || !argmax) return fail(RIGL_EINVAL, "rigl_bn_relu_maxpool_fwd: NULL tensor");
  if (2 * (size_t)g.c * 4 > 65536) return fail(RIGL_EUNSUPPORTED, "rigl_bn_relu_maxpool_fwd: too many channels for the LDS parameter cache");
  hipLaunchKernelGGL(k_bn_relu_fwd, dim3(grid_for((int64_t)g.n * g.ho * g.wo * g.cg)), dim3(THREADS), (size_t)2 * g.c * 4,
                     as_stream(stream), g, x, scale, shift, y, argmax);
  RIGL_CHECK_LAUNCH("rigl_bn_relu_maxpool_fwd");
  return RIGL_OK;
}

size_t rigl_bn_relu_maxpool_bwd_workspace_bytes(const RiglConvDesc* d) {
  if (!d || d->cin <= 0) return 0;
  return rigl::align_up((size_t)rigl::kpool::STEM_PARTS * 2 * d->cin * 4, 256) + rigl::align_up((size_t)3 * d->cin * 4, 256);
}

int rigl_bn_relu_maxpool_bwd(const RiglConvDesc* d, const rigl_bf16* x, const rigl_bf16* dy, const uint8_t* argmax,
                             const float* gamma, const float* save_mean, const float* save_invstd,
                             const float* save_scale, const float* save_shift, rigl_bf16* dx, float* dgamma,
                             float* dbeta, void* workspace, size_t workspace_bytes, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kpool;
  Geom g;
  int rc = make_geom(d, &g, "rigl_bn_relu_maxpool_bwd");
  if (rc) return rc;
  if ((rc = stem_check(g, "rigl_bn_relu_maxpool_bwd"))) return rc;
  if (!x || !dy || !argmax || !gamma || !save_mean || !save_invstd || !save_scale || !save_shift || !dx || !dgamma || !dbeta)
    return fail(RIGL_EINVAL, "rigl_bn_relu_maxpool_bwd: NULL tensor");
  const size_t need = rigl_bn_relu_maxpool_bwd_workspace_bytes(d);
  if (!workspace || workspace_bytes < need)
    return fail(RIGL_EWORKSPACE, "rigl_bn_relu_maxpool_bwd: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(workspace);
  float* coef = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)STEM_PARTS * 2 * g.c * 4, 256));
  const int kb = (g.h + g.pt + 1) / 2, lb = (g.w + g.pl + 1) / 2;       // 2x2 blocks of padded coordinates
  const int64_t items = (int64_t)g.n * kb * lb * (g.c / SCH);
  int parts = (int)((items + THREADS - 1) / THREADS);
  if (parts > STEM_PARTS) parts = STEM_PARTS;
  hipLaunchKernelGGL(k_bn_relu_bwd<0>, dim3((unsigned)parts), dim3(THREADS), 0, st, g, kb, lb, x, dy, argmax, save_mean,
                     save_invstd, save_scale, save_shift, nullptr, partial, nullptr);
  kbn::launch_bwd_finalize((int64_t)g.n * g.h * g.w, g.c, partial, parts, gamma, save_invstd, dgamma, dbeta, coef, st);
  hipLaunchKernelGGL(k_bn_relu_bwd<1>, dim3(grid_for(items)), dim3(THREADS), 0, st, g, kb, lb, x, dy, argmax, save_mean,
                     save_invstd, save_scale, save_shift, coef, nullptr, dx);
  RIGL_CHECK_LAUNCH("rigl_bn_relu_maxpool_bwd");
  return RIGL_OK;
}

int rigl_maxpool_fwd(const RiglConvDesc* d, const rigl_bf16* x, rigl_bf16* y, uint8_t* argmax, rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kpool;
  Geom g;
  int rc = make_geom(d, &g, "rigl_maxpool_fwd");
  if (rc) return rc;
  if (!x || !y || !argmax) return fail(RIGL_EINVAL, "rigl_maxpool_fwd: NULL tensor");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_fwd, dim3(grid_for((int64_t)g.n * g.ho * g.wo * g.cg)), dim3(THREADS), 0, st, g, x, y, argmax);
  RIGL_CHECK_LAUNCH("rigl_maxpool_fwd");
  return RIGL_OK;
}

int rigl_maxpool_bwd(const RiglConvDesc* d, const rigl_bf16* dy, const uint8_t* argmax, rigl_bf16* dx,
                     rigl_stream_t stream) {
  using namespace rigl;
  using namespace rigl::kpool;
  Geom g;
  int rc = make_geom(d, &g, "rigl_maxpool_bwd");
  if (rc) return rc;
  if (!dy || !dx || !argmax) return fail(RIGL_EINVAL, "rigl_maxpool_bwd: NULL tensor");
  hipStream_t st = as_stream(stream);
  const int64_t items = (int64_t)g.n * g.h * g.w * g.cg;
  const int kb = (g.h + g.pt + 1) / 2, lb = (g.w + g.pl + 1) / 2;       // 2x2 blocks of padded coordinates
  if (g.kh == 3 && g.kw == 3 && g.sh == 2 && g.sw == 2 && items < (int64_t(1) << 31))
    hipLaunchKernelGGL(k_bwd_3x3s2, dim3(grid_for((int64_t)g.n * kb * lb * g.cg)), dim3(THREADS), 0, st, g, kb, lb, dy, argmax, dx);
  else
    hipLaunchKernelGGL(k_bwd, dim3(grid_for(items)), dim3(THREADS), 0, st, g, dy, argmax, dx);
  RIGL_CHECK_LAUNCH("rigl_maxpool_bwd");
  return RIGL_OK;
}

}  // extern "C"
