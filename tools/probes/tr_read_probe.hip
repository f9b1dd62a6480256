// Probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used to design
// the wgrad LDS image).  LDS holds lds[i] = i; lane l reads 8 bytes at element
// 4*l, so out[l][e] names the (source lane, source element) it received.
// Build: hipcc -O3 --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = (uint16_t)v[i];
}
int main() {
  uint16_t* d; uint16_t h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      int src = h[l * 4 + e];
      printf("  (L%2d,e%d)", src / 4, src % 4);
      const int g = l & ~15, j = l & 15;
      if (src != (g + 4 * e + j / 4) * 4 + (j % 4)) ok = 0;
    }
    printf("\n");
  }
  printf("hypothesis out[j][e] = in[lane 4e + j/4][j%%4] per 16-lane group: %s\n", ok ? "HOLDS" : "FAILS");
  return 0;
}
