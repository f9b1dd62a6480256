// Probe (development aid): semantics of `buffer_load_dwordx4 ... lds` on gfx950.
//  (1) LDS destination = M0 base + lane*16 (contiguous per wave)?
//  (2) lanes whose buffer offset is out of range: is ZERO written, or is LDS left untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* p, unsigned bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4096];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xAAAAAAAAu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* dst = smem + wave * 1024;
  // lanes read REVERSED source chunks; odd lanes of wave 1 are out of range
  unsigned off = (unsigned)((wave * 64 + (63 - lane)) * 16);
  bool oob = (wave == 1) && (lane & 1);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, oob ? (int)0x80000000 : (int)off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;   // dword i; 16-byte chunk c holds 4c..4c+3
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 2048u /* only first 2 KB in range */, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  // expectation if linear: wave w, lane l -> LDS chunk (w*64 + l) holds source chunk (w*64 + 63 - l)
  int linear_ok = 1, oob_zero = 1, oob_untouched = 1, range_zero = 1;
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) {
    unsigned got = r[(w * 64 + l) * 4];
    unsigned src_chunk = w * 64 + 63 - l;
    bool in_range = src_chunk * 16 + 16 <= 2048;
    bool oob = (w == 1) && (l & 1);
    if (oob) { if (got != 0) oob_zero = 0; if (got != 0xAAAAAAAAu) oob_untouched = 0; }
    else if (in_range) { if (got != src_chunk * 4) linear_ok = 0; }
    else { if (got != 0) range_zero = 0; }
  }
  printf("lds_dma: linear_layout=%d oob_lane_writes_zero=%d oob_lane_untouched=%d beyond_num_records_zero=%d\n", linear_ok, oob_zero, oob_untouched, range_zero);
  printf("sample wave1: %08x %08x %08x %08x | wave2(first, out of 2KB range): %08x\n", r[256], r[260], r[264], r[268], r[512]);
  return 0;
}
