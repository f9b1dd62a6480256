// Probe: how fast can a CU pull operand tiles from L2 / Infinity Cache into LDS with `buffer_load ... lds`?
// K1's kernels move 16-24 KB per K-tile through this path and sit near 10-12 TB/s aggregate whatever their
// structure (profiles/r2: bench_kernels tables); this isolates the path: no LDS reads, no MFMAs.
//
// Each workgroup (256 threads) streams `rows` rows of `seg` contiguous bytes (64 = one 32-channel bf16 K-tile
// row, 128 = a 64-channel one) taken at a row pitch of `pitch` bytes from a buffer of `ws` bytes (L2- or
// MALL-resident), `depth` wave-instructions in flight per wave (counted vmcnt), into a 32 KB LDS ring.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_rate_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> __device__ __forceinline__ void wait_vmcnt();
#define W(N) template <> __device__ __forceinline__ void wait_vmcnt<N>() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); }
W(0) W(1) W(3) W(7) W(15)
#undef W

template <int SEG, int DEPTH>
__global__ __launch_bounds__(256) void k_stream(const unsigned char* __restrict__ src, unsigned ws_bytes, int pitch,
                                                int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 32 KB ring
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)ws_bytes, 0x00020000);
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per wave instruction
  // this workgroup walks rows  (blockIdx * 4 + wave) * RPI + i * gridDim * 4 * RPI ...  wrapped into the buffer
  const unsigned rows_mask = ws_bytes / (unsigned)pitch - 1u;    // (power of two: a modulo here cost ~40 VALU per load and capped the rate)
  unsigned row = ((unsigned)blockIdx.x * 4u + (unsigned)wave) * RPI + (unsigned)(lane / LPR);
  const unsigned step = gridDim.x * 4u * RPI;
  const unsigned col = (unsigned)(lane % LPR) * 16u;
  for (int i = 0; i < iters; ++i) {
    const unsigned r = row & rows_mask;
    const unsigned off = r * (unsigned)pitch + col;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((i & 7) * 4 + wave) * 1024), 16,
                                             (int)off, 0, 0, 0);
    wait_vmcnt<DEPTH - 1>();
    row += step;
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (tid == 0 && sink) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

template <int SEG, int DEPTH>
static double run(const unsigned char* buf, unsigned ws, int pitch, int blocks, int iters, unsigned* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<SEG, DEPTH>), dim3(blocks), dim3(256), 32768, 0, buf, ws, pitch, iters, sink);
  hipEventRecord(a);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_stream<SEG, DEPTH>), dim3(blocks), dim3(256), 32768, 0, buf, ws, pitch, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * blocks * 4.0 * 1024.0 * iters;
  return bytes / (ms * 1e-3) / 1e12;       // TB/s
}

int main() {
  unsigned char* buf;
  const unsigned WS_MAX = 512u << 20;
  hipMalloc(&buf, WS_MAX);
  hipMemset(buf, 1, WS_MAX);
  unsigned* sink;
  hipMalloc(&sink, 4096 * 4);
  printf("# seg = contiguous bytes per row, pitch = row pitch, ws = buffer walked (wraps), wg/cu, depth = instr in flight per wave\n");
  const int cus = 256;
  const unsigned wss[] = {2u << 20, 64u << 20, 512u << 20};      // inside one XCD's L2 / Infinity Cache / HBM
  const int pitches[] = {512, 2048};
  for (unsigned ws : wss)
    for (int pitch : pitches)
      for (int wgcu = 1; wgcu <= 3; ++wgcu) {
        const int blocks = cus * wgcu, iters = 2048 / wgcu;
        printf("ws %4u MB pitch %4d wg/cu %d :", ws >> 20, pitch, wgcu);
        printf("  seg64 d2 %5.2f d4 %5.2f d8 %5.2f d16 %5.2f |", run<64, 2>(buf, ws, pitch, blocks, iters, sink),
               run<64, 4>(buf, ws, pitch, blocks, iters, sink), run<64, 8>(buf, ws, pitch, blocks, iters, sink),
               run<64, 16>(buf, ws, pitch, blocks, iters, sink));
        printf("  seg128 d2 %5.2f d4 %5.2f d8 %5.2f d16 %5.2f TB/s\n", run<128, 2>(buf, ws, pitch, blocks, iters, sink),
               run<128, 4>(buf, ws, pitch, blocks, iters, sink), run<128, 8>(buf, ws, pitch, blocks, iters, sink),
               run<128, 16>(buf, ws, pitch, blocks, iters, sink));
        fflush(stdout);
      }
  return 0;
}
