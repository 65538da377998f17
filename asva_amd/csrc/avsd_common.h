// Shared device/host helpers for libavsd_hip.so (gfx950 only — no other target is built).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/avsd.h"

// ---- 16-bit storage type of activations and matrix weights --------------------------------------------------------
// One source tree, two builds (asva_amd/build.py): default = bfloat16 (libavsd_hip.so, BASELINE.json's dtype);
// -DAVSD_F16=1 = IEEE half (libavsd_hip_f16.so: the higher-precision mode, same MFMA rate).  Accumulation, softmax,
// normalisation statistics and every epilogue are f32 in both.  Only this block knows which one is compiled.
typedef unsigned short h16_t;  // raw 16-bit storage
#ifdef AVSD_F16
typedef _Float16 hw_h16;
#define AVSD_H16_ONE 0x3c00u
#define AVSD_PRECISION_NAME "fp16"
#else
typedef __bf16 hw_h16;
#define AVSD_H16_ONE 0x3f80u
#define AVSD_PRECISION_NAME "bf16"
#endif
typedef __attribute__((ext_vector_type(8))) hw_h16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing (host) -----------------------------------------------------------
void avsd_set_error(const char* fmt, ...);

#define AVSD_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      avsd_set_error(__VA_ARGS__);              \
      return AVSD_EINVAL;                       \
    }                                           \
  } while (0)

#define AVSD_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      avsd_set_error("%s: %s", what, hipGetErrorString(e__));                     \
      return AVSD_ELAUNCH;                                                        \
    }                                                                             \
  } while (0)

// ---- 16-bit <-> f32 (device) ------------------------------------------------------------
typedef hw_h16 hw_h16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
#ifdef AVSD_F16
__device__ __forceinline__ float h2f(h16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// low / high half of a packed pair -> f32 (v_cvt_f32_f16, the high half through SDWA)
__device__ __forceinline__ float lo2f(uint32_t u) { return (float)__builtin_bit_cast(hw_h16x2, u)[0]; }
__device__ __forceinline__ float hi2f(uint32_t u) { return (float)__builtin_bit_cast(hw_h16x2, u)[1]; }
__device__ __forceinline__ f32x16 mfma32x32x16(h16x8 a, h16x8 b, f32x16 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#else
__device__ __forceinline__ float h2f(h16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float lo2f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi2f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x16 mfma32x32x16(h16x8 a, h16x8 b, f32x16 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif

// c + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs, f32 result (v_dot2c_f32_bf16 / v_dot2c_f32_f16): no unpacking
__device__ __forceinline__ float dot2h(uint32_t a, uint32_t b, float c) {
#ifdef AVSD_F16
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(hw_h16x2, a), __builtin_bit_cast(hw_h16x2, b), c, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_h16x2, a), __builtin_bit_cast(hw_h16x2, b), c, false);
#endif
}

// two f32 -> one packed pair, round-to-nearest-even in both builds (v_cvt_pk_bf16_f32 / v_cvt_f16_f32 x2)
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_h16x2));
}
__device__ __forceinline__ h16_t f2h(float f) { return (h16_t)(pack2h(f, 0.f) & 0xffffu); }

// 8 packed 16-bit values in a uint4 -> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = lo2f(v.x); f[1] = hi2f(v.x);
  f[2] = lo2f(v.y); f[3] = hi2f(v.y);
  f[4] = lo2f(v.z); f[5] = hi2f(v.z);
  f[6] = lo2f(v.w); f[7] = hi2f(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2h(f[0], f[1]); v.y = pack2h(f[2], f[3]);
  v.z = pack2h(f[4], f[5]); v.w = pack2h(f[6], f[7]);
  return v;
}

// ---- split-precision ("x2") storage: a tensor is a PAIR of 16-bit planes with identical strides, main = round16(v) and
// rest = round16(v - main), `lo` = element offset from the main to the rest plane.  main + rest carries 16 (bf16) / 22 (fp16)
// significant bits; every kernel reconstructs v = main + rest in f32 (exact) and writes both planes back. ------------------
template <bool X2>
__device__ __forceinline__ void load8(const h16_t* p, int64_t lo, float* f) {
  unpack8(*reinterpret_cast<const uint4*>(p), f);
  if constexpr (X2) {
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(p + lo), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
  }
}
template <bool X2>
__device__ __forceinline__ void store8(h16_t* p, int64_t lo, const float* f) {
  const uint4 h = pack8(f);
  *reinterpret_cast<uint4*>(p) = h;
  if constexpr (X2) {
    float hf[8], r[8];
    unpack8(h, hf);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f[e] - hf[e];
    *reinterpret_cast<uint4*>(p + lo) = pack8(r);
  }
}
// one value: main plane word and the rest that rounding left
__device__ __forceinline__ void split2(float a, float b, uint32_t& main, uint32_t& rest) {
  main = pack2h(a, b);
  rest = pack2h(a - lo2f(main), b - hi2f(main));
}

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (~10 instructions): every
// GroupNorm / SiLU site shares this one form, so the kernels that apply it agree bit for bit
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf-GELU (torch F.gelu default; diffusers GEGLU): 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 — two orders below the 16-bit rounding of the result; libm's erff costs ~3x the instructions, and a
// GEGLU epilogue evaluates one per output element).  1 + erf(z) is formed without cancellation on both sides of 0.
// The reciprocal is the hardware's (v_rcp_f32, 1 ulp): an IEEE division costs 11 VALU instructions of the 26 this function
// compiled to, for an error a hundred times below the formula's own.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  const float pe = poly * __expf(-z * z);          // = 1 - erf(|z|)
  return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
