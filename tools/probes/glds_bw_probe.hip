// Per-CU global->LDS delivery rate from L2-resident data with `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction),
// as a function of waves per CU and loads in flight per wave.  One workgroup per CU (256 blocks), no MFMA work.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int DEPTH>
__global__ void k(const char* src, int iters, unsigned span_mask, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  unsigned off = (blockIdx.x * 8191u + wave * 1024u) * 16u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // 8 rows x 128 B per piece, rows 2560 B apart (a K = 1280 operand)
      const unsigned vo = ((off + (lane >> 3) * 2560u + (lane & 7) * 16u)) & span_mask;
      lds_ptr_t dst = (lds_ptr_t)(smem + (wave * DEPTH + d) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)vo, 0, 0, 0);
      off += 128u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0 && iters < 0) sink[0] = smem[0];
}

template <int DEPTH>
void run(const char* src, int waves, float* sink) {
  const int iters = 2000 / DEPTH;
  const size_t lds = (size_t)waves * DEPTH * 1024;
  hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<DEPTH><<<256, waves * 64, lds>>>(src, iters, (2u << 20) - 1, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<DEPTH><<<256, waves * 64, lds>>>(src, iters, (2u << 20) - 1, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * waves * iters * DEPTH * 1024.0;
  printf("waves/CU=%2d in-flight/wave=%2d : %7.2f TB/s aggregate, %5.1f B/clk/CU @2.4GHz\n", waves, DEPTH, bytes / ms / 1e9,
         bytes / 256.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
  char* src; hipMalloc(&src, 4 << 20); hipMemset(src, 1, 4 << 20);
  float* sink; hipMalloc(&sink, 4);
  for (int waves : {4, 8, 16}) { run<1>(src, waves, sink); run<2>(src, waves, sink); run<4>(src, waves, sink); run<8>(src, waves, sink); }
  return 0;
}
