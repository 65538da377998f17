// Exact-f32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate — bitwise a k-ordered fmaf chain,
// at the f32 VECTOR rate: 157 TFLOP/s peak, 1/16 of the bf16 MFMA).  This is the YARDSTICK of the path, not a product kernel:
// out[M, N] = A[M, K] . W[N, K]^T + bias[n], all f32, no 16-bit storage anywhere — so a test can separate "kernel arithmetic"
// from "storage rounding": the 16-bit and split-precision GEMMs of gemm.hip are checked against it on the same f32 operands
// (tests/test_split_gpu.py), and bench.py reports its rate next to theirs.  The reference computes these products with
// torch.nn.Linear / nn.Conv2d in fp32 (scripts/animation_gen.py:43-44; avgen/models/unets/utils.py:37-53).
//
// Tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 accumulator fragments of 32 x 32;
// LDS rows padded to 33 floats (conflict-free ds_read_b32 of a fragment column); single-buffered, two barriers per K tile.
#include "avsd_common.h"

namespace {

constexpr int FT = 128, FK = 32, FP = FK + 1;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, float* __restrict__ out, int ldc, int M, int N,
                                                       int K) {
  __shared__ float sA[FT * FP];
  __shared__ float sW[FT * FP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.y * FT, n0 = blockIdx.x * FT;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int lr = tid >> 3;            // 0..31: row inside a 32-row slab
  const int lk = (tid & 7) * 4;       // k offset of this thread's float4
  for (int k0 = 0; k0 < K; k0 += FK) {
    float4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i, n = n0 + lr + 32 * i, k = k0 + lk;
      ra[i] = (m < M && k < K) ? *reinterpret_cast<const float4*>(A + (int64_t)m * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[i] = (n < N && k < K) ? *reinterpret_cast<const float4*>(W + (int64_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                  // everybody is done reading the previous tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* da = sA + (lr + 32 * i) * FP + lk;
      float* dw = sW + (lr + 32 * i) * FP + lk;
      da[0] = ra[i].x; da[1] = ra[i].y; da[2] = ra[i].z; da[3] = ra[i].w;
      dw[0] = rw[i].x; dw[1] = rw[i].y; dw[2] = rw[i].z; dw[3] = rw[i].w;
    }
    __syncthreads();
    // operand layout of v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]
    const float* pa = sA + (wm * 64 + (lane & 31)) * FP + (lane >> 5);
    const float* pw = sW + (wn * 64 + (lane & 31)) * FP + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < FK; kk += 2) {
      const float a0 = pa[kk], a1 = pa[32 * FP + kk];
      const float w0 = pw[kk], w1 = pw[32 * FP + kk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1, acc[1][1], 0, 0, 0);
    }
  }
  // C/D layout: column j = lane & 31 (n), row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (m)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + wn * 64 + b * 32 + (lane & 31);
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) out[(int64_t)m * ldc + n] = acc[a][b][r] + bv;
      }
    }
}

}  // namespace

extern "C" int avsd_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                             int K, void* stream) {
  AVSD_REQUIRE(A && W && out && M > 0 && N > 0 && K > 0, "gemm_f32: bad arguments");
  AVSD_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldc >= N, "gemm_f32: K, lda, ldw must be multiples of 4");
  AVSD_REQUIRE(((uintptr_t)A | (uintptr_t)W) % 16 == 0, "gemm_f32: A and W must be 16-byte aligned");
  dim3 grid((unsigned)((N + FT - 1) / FT), (unsigned)((M + FT - 1) / FT));
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, lda, W, ldw, bias, out, ldc, M, N, K);
  AVSD_CHECK_LAUNCH("gemm_f32 launch");
  return AVSD_OK;
}
