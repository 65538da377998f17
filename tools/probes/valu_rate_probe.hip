// Issue-rate probe for gfx950: cycles per wave64 instruction for the VALU ops the attention softmax uses
// (v_exp_f32, v_pk_fma_f32, v_cvt_pk_bf16_f32, v_max3_f32, v_perm_b32) and for v_mfma_f32_32x32x16_bf16,
// measured with s_memtime on one SIMD at 1 and 2 waves.   hipcc --offload-arch=gfx950 -O3 -o valu_probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

#define REP 64
template <int OP>
__global__ void probe(float* out, long long* cyc, int iters) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  f32x16 acc = {0}; f32x16 acc2 = {0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)i; b[i] = (__bf16)(float)(i + threadIdx.x); }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) { for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]); }
      if (OP == 1) { for (int i = 0; i < 8; i += 2) { f32x2 v = {x[i], x[i + 1]}; f32x2 m = {1.0001f, 0.9999f}; v = __builtin_elementwise_fma(v, m, m); x[i] = v[0]; x[i + 1] = v[1]; }
                     for (int i = 0; i < 8; i += 2) { f32x2 v = {x[i], x[i + 1]}; f32x2 m = {1.0002f, 0.9998f}; v = __builtin_elementwise_fma(v, m, m); x[i] = v[0]; x[i + 1] = v[1]; } }
      if (OP == 2) { for (int i = 0; i < 8; ++i) { f32x2 v = {x[i], x[(i + 1) & 7]}; bf2 c = __builtin_convertvector(v, bf2); x[i] = __builtin_bit_cast(float, c); } }
      if (OP == 3) { for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f); }
      if (OP == 4) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                     acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc2, 0, 0, 0);
                     acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                     acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc2, 0, 0, 0); }
      if (OP == 5) { for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_fmed3f(x[i], x[(i + 1) & 7], x[(i + 2) & 7]); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  // one workgroup on one CU; 256 threads = one wave per SIMD; 512 = two per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<1, 256 * waves_per_simd>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP><<<1, 256 * waves_per_simd>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * REP;
  printf("%-22s waves/SIMD=%d  counter ticks/instr=%.3f  wall ns/instr=%.3f\n", name, waves_per_simd, c / n, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_exp_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_cvt_pk_bf16_f32", w); run<3>("v_fma_f32", w);
    run<5>("v_med3_f32", w); run<4>("v_mfma_32x32x16_bf16", w);
  }
  return 0;
}
