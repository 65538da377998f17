// Probe: semantics of buffer_load_dwordx4 ... lds on gfx950 (placement = M0 + lane*16 ? OOB lanes -> zeros ?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const unsigned* a, unsigned* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* s32 = (unsigned*)smem;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) s32[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int voff = (lane ^ 5) * 16 + wave * 1024;            // permuted source
  if ((lane % 7) == 3) voff = 0x7ffffff0;              // out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = s32[i];
}
int main() {
  const int n = 1024;  // dwords
  std::vector<unsigned> h(n); for (int i = 0; i < n; ++i) h[i] = i + 1;
  unsigned *a, *o; hipMalloc(&a, n * 4); hipMalloc(&o, n * 4);
  hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, a, o, n * 4);
  std::vector<unsigned> r(n); hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
  int bad_place = 0, oob_zero = 0, oob_untouched = 0, oob_other = 0;
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int d = 0; d < 4; ++d) {
    unsigned got = r[w * 256 + l * 4 + d];
    if ((l % 7) == 3) { if (got == 0) oob_zero++; else if (got == 0xDEADBEEFu) oob_untouched++; else oob_other++; }
    else { unsigned exp = (unsigned)(w * 256 + (l ^ 5) * 4 + d + 1); if (got != exp) bad_place++; }
  }
  printf("placement mismatches: %d ; OOB lanes: zero=%d untouched=%d other=%d\n", bad_place, oob_zero, oob_untouched, oob_other);
  return 0;
}
