// K3 (masked fused SGD / Nesterov momentum), mask bitmap pack/unpack and the
// bf16 weight-shadow pack kernels.  All HBM-bound streaming kernels: 16 B per
// lane coalesced accesses, grid-stride, no LDS except for the OHWI transpose.
#include "common.hpp"

// Bit-exact parity with the reference arithmetic: every fp32 product / sum is
// rounded separately.  HIP's __fmul_rn/__fadd_rn are plain operators, so the
// default -ffp-contract=fast would fuse them into v_fma (1-ulp differences).
#pragma clang fp contract(off)

namespace rigl {
namespace k3 {

constexpr int BLOCK = 256;

__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);  // quiet NaN
  u += 0x7FFFu + ((u >> 16) & 1u);                                            // RNE
  return (uint16_t)(u >> 16);
}

// ---- mask bitmap -------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_mask_pack(const float* __restrict__ m, uint32_t* __restrict__ bits, int64_t n) {
  // one wave-lane per element, 64 elements -> two 32-bit words via ballot
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  const int64_t n_round = (n + 63) / 64 * 64;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += stride) {
    bool on = i < n && m[i] != 0.f;
    unsigned long long b = __ballot(on);
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (i - lane) >> 5;
    const int64_t n_words = (n + 31) >> 5;
    if (lane == 0 && w0 < n_words) bits[w0] = (uint32_t)b;
    if (lane == 32 && w0 + 1 < n_words) bits[w0 + 1] = (uint32_t)(b >> 32);
  }
}

__global__ __launch_bounds__(BLOCK) void k_mask_unpack(const uint32_t* __restrict__ bits, float* __restrict__ m, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride)
    m[i] = ((bits[i >> 5] >> (i & 31)) & 1u) ? 1.f : 0.f;
}

// ---- masked SGD / momentum -----------------------------------------------------
struct SgdArgs {
  int64_t n;
  float* w;
  float* mom;
  const float* g;
  const uint32_t* mask;
  float lr, mu, wd, gscale;
  int nesterov;
  uint16_t* shadow;
};

__device__ __forceinline__ void sgd_one(float& w, float& a, float g, bool on, const SgdArgs& A, bool has_mom) {
  // g_var = mask * (grad_scale * dense) + wd * w      (each op rounded: no FMA)
  float gm = on ? (A.gscale == 1.f ? g : __fmul_rn(g, A.gscale)) : 0.f;
  float gv = A.wd != 0.f ? __fadd_rn(gm, __fmul_rn(A.wd, w)) : gm;
  if (has_mom) {
    float an = __fadd_rn(__fmul_rn(a, A.mu), gv);       // accum = accum*momentum + grad
    float step;
    if (A.nesterov) step = __fadd_rn(__fmul_rn(gv, A.lr), __fmul_rn(__fmul_rn(an, A.mu), A.lr));
    else step = __fmul_rn(an, A.lr);
    a = an;
    w = __fsub_rn(w, step);
  } else {
    w = __fsub_rn(w, __fmul_rn(gv, A.lr));
  }
}

template <bool HAS_MOM, bool HAS_MASK, bool HAS_SHADOW>
__global__ __launch_bounds__(BLOCK) void k_sgd(SgdArgs A) {
  const int64_t nq = A.n >> 2;  // full quads
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t qi = (int64_t)blockIdx.x * BLOCK + threadIdx.x; qi < nq; qi += stride) {
    const int64_t e = qi << 2;
    float4 w = *reinterpret_cast<const float4*>(A.w + e);
    float4 g = *reinterpret_cast<const float4*>(A.g + e);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_MOM) a = *reinterpret_cast<const float4*>(A.mom + e);
    uint32_t nib = 0xFu;
    if (HAS_MASK) nib = (A.mask[e >> 5] >> (uint32_t)(e & 31)) & 0xFu;
    sgd_one(w.x, a.x, g.x, nib & 1u, A, HAS_MOM);
    sgd_one(w.y, a.y, g.y, nib & 2u, A, HAS_MOM);
    sgd_one(w.z, a.z, g.z, nib & 4u, A, HAS_MOM);
    sgd_one(w.w, a.w, g.w, nib & 8u, A, HAS_MOM);
    *reinterpret_cast<float4*>(A.w + e) = w;
    if (HAS_MOM) *reinterpret_cast<float4*>(A.mom + e) = a;
    if (HAS_SHADOW) {
      ushort4 s;
      s.x = (nib & 1u) ? f2bf(w.x) : (uint16_t)0;
      s.y = (nib & 2u) ? f2bf(w.y) : (uint16_t)0;
      s.z = (nib & 4u) ? f2bf(w.z) : (uint16_t)0;
      s.w = (nib & 8u) ? f2bf(w.w) : (uint16_t)0;
      *reinterpret_cast<ushort4*>(A.shadow + e) = s;
    }
  }
  // tail (n % 4 elements), one thread each
  const int64_t tail0 = nq << 2;
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < A.n - tail0) {
    const int64_t e = tail0 + threadIdx.x;
    float w = A.w[e], a = HAS_MOM ? A.mom[e] : 0.f;
    bool on = HAS_MASK ? ((A.mask[e >> 5] >> (uint32_t)(e & 31)) & 1u) : true;
    sgd_one(w, a, A.g[e], on, A, HAS_MOM);
    A.w[e] = w;
    if (HAS_MOM) A.mom[e] = a;
    if (HAS_SHADOW) A.shadow[e] = on ? f2bf(w) : (uint16_t)0;
  }
}

// ---- bf16 shadows of mask*W ------------------------------------------------------
// Layer = fp32 W[k][cout] (flat HWIO).  hwio[i] = bf16(mask_i ? W_i : 0);
// ohwi[co][kk] = the transpose.  One 64x64 tile per workgroup through LDS.
constexpr int MAX_PACK_LAYERS = 64;
struct PackLayer {
  const float* w;
  const uint32_t* mask;
  uint16_t* hwio;
  uint16_t* ohwi;
  int32_t k, cout;
};
struct PackTable {
  PackLayer l[MAX_PACK_LAYERS];
  uint32_t tile_begin[MAX_PACK_LAYERS + 1];
  int32_t n_layers;
};

__global__ __launch_bounds__(BLOCK) void k_pack(PackTable T) {
  __shared__ uint16_t tile[64][66];
  int li = 0;
  {
    int lo = 0, hi = T.n_layers - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (T.tile_begin[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    li = lo;
  }
  const PackLayer L = T.l[li];
  const uint32_t tl = blockIdx.x - T.tile_begin[li];
  const int tiles_c = (L.cout + 63) / 64;
  const int k0 = (int)(tl / tiles_c) * 64, c0 = (int)(tl % tiles_c) * 64;
  if ((L.cout & 3) == 0 && (L.k & 3) == 0) {
    // four consecutive output channels per lane (16-byte loads, 8-byte stores), the four row passes requested before the first is
    // used (unconditional loads at clamped positions: a load inside `if` waits for itself); the transposed copy leaves as four
    // consecutive reduction rows per lane.  Same conversions as the element-wise form below (the 7x7x3 stem keeps that one).
    __shared__ __attribute__((aligned(8))) uint16_t t4[64][68];
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
    float4 wv[4];
    uint32_t nib[4];
    bool ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int kk = k0 + rr + 16 * p, co = c0 + q * 4;
      ok[p] = kk < L.k && co < L.cout;
      const int64_t i = ok[p] ? (int64_t)kk * L.cout + co : 0;
      wv[p] = *reinterpret_cast<const float4*>(L.w + i);
      nib[p] = L.mask ? (L.mask[i >> 5] >> (uint32_t)(i & 31)) & 0xFu : 0xFu;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ushort4 v = make_ushort4(0, 0, 0, 0);
      if (ok[p]) {
        v.x = (nib[p] & 1u) ? f2bf(wv[p].x) : (uint16_t)0;
        v.y = (nib[p] & 2u) ? f2bf(wv[p].y) : (uint16_t)0;
        v.z = (nib[p] & 4u) ? f2bf(wv[p].z) : (uint16_t)0;
        v.w = (nib[p] & 8u) ? f2bf(wv[p].w) : (uint16_t)0;
        if (L.hwio) *reinterpret_cast<ushort4*>(L.hwio + (int64_t)(k0 + rr + 16 * p) * L.cout + c0 + q * 4) = v;
      }
      *reinterpret_cast<ushort4*>(&t4[rr + 16 * p][q * 4]) = v;
    }
    if (!L.ohwi) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int col = rr + 16 * p, co = c0 + col, kk = k0 + q * 4;
      if (co < L.cout && kk < L.k) {
        ushort4 v;
        v.x = t4[q * 4 + 0][col]; v.y = t4[q * 4 + 1][col]; v.z = t4[q * 4 + 2][col]; v.w = t4[q * 4 + 3][col];
        *reinterpret_cast<ushort4*>(L.ohwi + (int64_t)co * L.k + kk) = v;
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int kk = k0 + r, co = c0 + tx;
    uint16_t v = 0;
    if (kk < L.k && co < L.cout) {
      const int64_t i = (int64_t)kk * L.cout + co;
      bool on = L.mask ? ((L.mask[i >> 5] >> (uint32_t)(i & 31)) & 1u) : true;
      v = on ? f2bf(L.w[i]) : (uint16_t)0;
      if (L.hwio) L.hwio[i] = v;
    }
    tile[r][tx] = v;
  }
  if (!L.ohwi) return;
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int co = c0 + r, kk = k0 + tx;
    if (co < L.cout && kk < L.k) L.ohwi[(int64_t)co * L.k + kk] = tile[tx][r];
  }
}

}  // namespace k3
}  // namespace rigl

extern "C" {

int rigl_mask_pack(const float* mask01, uint32_t* bits, int64_t n, rigl_stream_t stream) {
  using namespace rigl;
  if (n < 0 || (n > 0 && (!mask01 || !bits))) return fail(RIGL_EINVAL, "rigl_mask_pack: bad arguments");
  if (n == 0) return RIGL_OK;
  int64_t blocks = ceil_div64(n, k3::BLOCK);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k3::k_mask_pack, dim3((unsigned)blocks), dim3(k3::BLOCK), 0, as_stream(stream), mask01, bits, n);
  RIGL_CHECK_LAUNCH("rigl_mask_pack");
  return RIGL_OK;
}

int rigl_mask_unpack(const uint32_t* bits, float* mask01, int64_t n, rigl_stream_t stream) {
  using namespace rigl;
  if (n < 0 || (n > 0 && (!mask01 || !bits))) return fail(RIGL_EINVAL, "rigl_mask_unpack: bad arguments");
  if (n == 0) return RIGL_OK;
  int64_t blocks = ceil_div64(n, k3::BLOCK);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k3::k_mask_unpack, dim3((unsigned)blocks), dim3(k3::BLOCK), 0, as_stream(stream), bits, mask01, n);
  RIGL_CHECK_LAUNCH("rigl_mask_unpack");
  return RIGL_OK;
}

int rigl_masked_sgd_momentum(int64_t n, float* w, float* momentum, const float* dense_grad,
                             const uint32_t* mask_bits, float lr, float mu, float weight_decay, float grad_scale,
                             int32_t nesterov, rigl_bf16* w_shadow, rigl_stream_t stream) {
  using namespace rigl;
  if (n < 0 || (n > 0 && (!w || !dense_grad))) return fail(RIGL_EINVAL, "rigl_masked_sgd_momentum: bad arguments");
  if (n == 0) return RIGL_OK;
  auto mis = [](const void* p, size_t a) { return p && (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; };
  if (mis(w, 16) || mis(momentum, 16) || mis(dense_grad, 16) || mis(w_shadow, 8) || mis(mask_bits, 4))
    return fail(RIGL_EINVAL, "rigl_masked_sgd_momentum: w/momentum/grad must be 16-byte aligned, shadow 8-byte");
  k3::SgdArgs A;
  A.n = n; A.w = w; A.mom = momentum; A.g = dense_grad; A.mask = mask_bits;
  A.lr = lr; A.mu = mu; A.wd = weight_decay; A.gscale = grad_scale; A.nesterov = nesterov; A.shadow = w_shadow;
  int64_t blocks = ceil_div64(ceil_div64(n, 4), k3::BLOCK);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipStream_t s = as_stream(stream);
  ProfScope prof(PROF_SGD, s);
  dim3 g((unsigned)blocks), b(k3::BLOCK);
  const int sel = (momentum ? 4 : 0) | (mask_bits ? 2 : 0) | (w_shadow ? 1 : 0);
  switch (sel) {
    case 0: hipLaunchKernelGGL((k3::k_sgd<false, false, false>), g, b, 0, s, A); break;
    case 1: hipLaunchKernelGGL((k3::k_sgd<false, false, true>), g, b, 0, s, A); break;
    case 2: hipLaunchKernelGGL((k3::k_sgd<false, true, false>), g, b, 0, s, A); break;
    case 3: hipLaunchKernelGGL((k3::k_sgd<false, true, true>), g, b, 0, s, A); break;
    case 4: hipLaunchKernelGGL((k3::k_sgd<true, false, false>), g, b, 0, s, A); break;
    case 5: hipLaunchKernelGGL((k3::k_sgd<true, false, true>), g, b, 0, s, A); break;
    case 6: hipLaunchKernelGGL((k3::k_sgd<true, true, false>), g, b, 0, s, A); break;
    default: hipLaunchKernelGGL((k3::k_sgd<true, true, true>), g, b, 0, s, A); break;
  }
  RIGL_CHECK_LAUNCH("rigl_masked_sgd_momentum");
  return RIGL_OK;
}

int rigl_pack_weights_batched(const RiglPackLayer* layers, int32_t n_layers, rigl_stream_t stream) {
  using namespace rigl;
  if (n_layers < 0 || (n_layers > 0 && !layers)) return fail(RIGL_EINVAL, "rigl_pack_weights_batched: bad arguments");
  hipStream_t s = as_stream(stream);
  ProfScope prof(PROF_PACK, s);
  for (int b0 = 0; b0 < n_layers; b0 += k3::MAX_PACK_LAYERS) {
    k3::PackTable T;
    const int cnt = n_layers - b0 < k3::MAX_PACK_LAYERS ? n_layers - b0 : k3::MAX_PACK_LAYERS;
    uint32_t tiles = 0;
    for (int i = 0; i < cnt; ++i) {
      const RiglPackLayer& l = layers[b0 + i];
      if (!l.w || l.k <= 0 || l.cout <= 0) return fail(RIGL_EINVAL, "rigl_pack_weights: layer %d: bad w/k/cout", b0 + i);
      T.l[i].w = l.w; T.l[i].mask = l.mask_bits; T.l[i].hwio = l.hwio; T.l[i].ohwi = l.ohwi;
      T.l[i].k = l.k; T.l[i].cout = l.cout;
      T.tile_begin[i] = tiles;
      tiles += (uint32_t)(((l.k + 63) / 64) * ((l.cout + 63) / 64));
    }
    T.tile_begin[cnt] = tiles;
    T.n_layers = cnt;
    if (tiles) hipLaunchKernelGGL(k3::k_pack, dim3(tiles), dim3(k3::BLOCK), 0, s, T);
  }
  RIGL_CHECK_LAUNCH("rigl_pack_weights");
  return RIGL_OK;
}

int rigl_pack_weights(const float* w, const uint32_t* mask_bits, int32_t k, int32_t cout, rigl_bf16* hwio,
                      rigl_bf16* ohwi, rigl_stream_t stream) {
  RiglPackLayer l;
  l.w = w; l.mask_bits = mask_bits; l.k = k; l.cout = cout; l.hwio = hwio; l.ohwi = ohwi;
  return rigl_pack_weights_batched(&l, 1, stream);
}

}  // extern "C"
