// Small / elementwise kernels: layout conversion at the NCFHW boundary, sinusoidal timestep
// embedding, skinny (M <= 32) linear layers, the fused guidance + scheduler update and the VAE
// post-processing.  All HBM- or latency-bound; none is worth an MFMA.
//
// Reference: audio_cond_unet_3d_condition.py:673-681 (time embedding), ff_spatio_temp_resnet_3d.py:170
// (time_emb_proj), ff_spatio_audio_temp_transformer_3d.py:348-349 (temporal position MLP),
// pipeline_audio_cond_animation.py:331-337 (latent duplication), :349-364 (guidance, scheduler.step on
// frames 1..), :206-213 (decode post-processing).  diffusers 0.29.2 get_timestep_embedding /
// PNDMScheduler.step_plms / DDIMScheduler.step supply the formulas.
#include "avsd_common.h"

namespace {

// dst_lo != 0: split-precision planes (main + rest, avsd_common.h)
__global__ void ncfhw_to_rows_kernel(const float* src, h16_t* dst, int64_t dst_lo, int B, int C, int F, int HW, int cpad,
                                     int rep, float scale) {
  // one thread per (rep, b, f, p); writes cpad channels
  const int64_t n = (int64_t)rep * B * F * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int f = (int)((i / HW) % F);
    const int b = (int)((i / ((int64_t)HW * F)) % B);
    h16_t* o = dst + i * cpad;
    for (int c = 0; c < cpad; ++c) {
      float v = 0.f;
      if (c < C) v = src[(((int64_t)b * C + c) * F + f) * HW + p] * scale;
      const h16_t m = f2h(v);
      o[c] = m;
      if (dst_lo) o[dst_lo + c] = f2h(v - h2f(m));
    }
  }
}

__global__ void rows_to_ncfhw_kernel(const float* src, int ld, float* dst, int B, int C, int F, int HW) {
  const int64_t n = (int64_t)B * C * F * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int f = (int)((i / HW) % F);
    const int c = (int)((i / ((int64_t)HW * F)) % C);
    const int b = (int)(i / ((int64_t)HW * F * C));
    dst[i] = src[(((int64_t)b * F + f) * HW + p) * ld + c];
  }
}

__global__ void timestep_embedding_kernel(const float* t, float* out, int n, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x;
  const float tv = t[i];
  for (int j = threadIdx.x; j < half; j += blockDim.x) {
    const float w = expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
    const float a = tv * w;
    out[(int64_t)i * dim + j] = cosf(a);          // flip_sin_to_cos: cos half first
    out[(int64_t)i * dim + half + j] = sinf(a);
  }
}

// out[m, n] = act_out(sum_k act_in(x[m, k]) * W[n, k] + bias[n]); one wave per n, 8 rows per block.y
__global__ void split_f32_kernel(const float* src, h16_t* dst, int64_t dst_lo, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = src[i];
    const h16_t m = f2h(v);
    dst[i] = m;
    if (dst_lo) dst[dst_lo + i] = f2h(v - h2f(m));
  }
}

template <bool X2>
__global__ __launch_bounds__(256) void linear_small_m_kernel(const float* x, const h16_t* W, int64_t w_lo, const float* bias,
                                                             float* out, int M, int N, int K, int ldw, int act_in,
                                                             int act_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int m0 = blockIdx.y * 8;
  const int mcount = min(8, M - m0);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const h16_t* wrow = W + (int64_t)n * ldw;
  for (int k = lane * 8; k < K; k += 512) {
    float w[8];
    load8<X2>(wrow + k, w_lo, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < mcount) {
        const float* xr = x + (int64_t)(m0 + i) * K + k;
        const float4 a0 = *reinterpret_cast<const float4*>(xr);
        const float4 a1 = *reinterpret_cast<const float4*>(xr + 4);
        float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = act_in ? silu_f(a[e]) : a[e];
          acc[i] = fmaf(xv, w[e], acc[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0 && i < mcount) {
      float r = v + (bias ? bias[n] : 0.f);
      if (act_out) r = silu_f(r);
      out[(int64_t)(m0 + i) * N + n] = r;
    }
  }
}

struct GuidedArgs {
  const float* noise_pred; float* eps_hist; const float* x_in; float* x_out;
  int n_branch, store_slot, n_hist;
  int hist_idx[4];
  float w[4];
  float g, g2, w_cur, ca, cb;
  int B, C, F, HW;
};

__global__ void guided_step_kernel(const GuidedArgs a) {
  const int64_t per = (int64_t)a.B * a.C * a.F * a.HW;   // elements of one latent tensor
  const int64_t fhw = (int64_t)a.F * a.HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    float eps = a.noise_pred[i];
    if (a.n_branch >= 2) {
      // e0 + g (e1 - e0) [+ g2 (e2 - e1)]: audio-only / text-only guidance, or the dual form of
      // pipeline_audio_cond_animation.py:349-353 with branches [uncond, text, text+audio], g = text scale, g2 = audio scale
      const float e1 = a.noise_pred[per + i];
      float gsum = eps + a.g * (e1 - eps);
      if (a.n_branch == 3) gsum = gsum + a.g2 * (a.noise_pred[2 * per + i] - e1);
      eps = gsum;
    }
    if (a.eps_hist && a.store_slot >= 0) a.eps_hist[(int64_t)a.store_slot * per + i] = eps;
    float e = a.w_cur * eps;
    for (int k = 0; k < a.n_hist; ++k) {
      // the freshly stored slot is read back from the register copy
      const float h = (a.hist_idx[k] == a.store_slot) ? eps : a.eps_hist[(int64_t)a.hist_idx[k] * per + i];
      e = fmaf(a.w[k], h, e);
    }
    const int f = (int)((i % fhw) / a.HW);
    const float xin = a.x_in[i];
    a.x_out[i] = (f == 0) ? xin : fmaf(a.ca, xin, a.cb * e);
  }
}

__global__ void vae_postprocess_kernel(const h16_t* src, int ld, int64_t src_lo, float* dst, int N, int HW) {
  const int64_t n = (int64_t)N * 3 * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % 3);
    const int im = (int)(i / ((int64_t)3 * HW));
    const int64_t o = ((int64_t)im * HW + p) * ld + c;
    const float x = h2f(src[o]) + (src_lo ? h2f(src[src_lo + o]) : 0.f);
    const float v = x * 0.5f + 0.5f;
    dst[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

// channels-last bf16 rows [N*HW][ld] -> uint8 frames (N, H, W, 3): (clamp(x/2+0.5, 0, 1) * 255) truncated, i.e. the
// pipeline's post-processing (:212) followed by generate_videos' `(video.permute(0,2,3,1) * 255).byte()` (:448)
__global__ void vae_postprocess_u8_kernel(const h16_t* src, int ld, int64_t src_lo, uint8_t* dst, int64_t npix) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    const h16_t* s = src + i * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = h2f(s[c]) + (src_lo ? h2f(s[src_lo + c]) : 0.f);
      const float v = fminf(fmaxf(x * 0.5f + 0.5f, 0.f), 1.f);
      dst[i * 3 + c] = (uint8_t)(v * 255.0f);
    }
  }
}

inline unsigned grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// dst[r][i] = src[i], 16 bytes per thread per round: each vector is read once and written `rep` times
__global__ void copy_rep_kernel(const uint4* src, uint4* dst, int64_t nvec, int rep) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    for (int r = 0; r < rep; ++r) dst[(int64_t)r * nvec + i] = v;
  }
}

// one thread per (block b, key slot j < lk_pad, 8-channel chunk): K rows copied as they are, V written transposed; the
// padding slots j >= lk are written as zeros on every call (the fused kernel multiplies p = 0 by the V^T padding, so it
// must be finite whatever the allocator handed out — a replaying host binds fresh memory)
__global__ void xattn_pack_kv_kernel(const h16_t* kv, int rows, int C, const int32_t* idx, int n_frames, int nk, int lk,
                                     h16_t* k_out, h16_t* vt_out, int lk_pad, int64_t total) {
  const int c8n = C / 8;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(t % c8n);
    const int j = (int)((t / c8n) % lk_pad);
    const int b = (int)(t / ((int64_t)c8n * lk_pad));
    if (j >= lk) {
      *reinterpret_cast<uint4*>(k_out + ((int64_t)b * lk_pad + j) * C + c8 * 8) = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) vt_out[((int64_t)b * C + c8 * 8 + e) * lk_pad + j] = 0;
      continue;
    }
    const int n = idx ? b / n_frames : b;
    const int key = idx ? idx[(b % n_frames) * nk + j] : j;
    const h16_t* src = kv + ((int64_t)n * rows + key) * 2 * C + c8 * 8;
    *reinterpret_cast<uint4*>(k_out + ((int64_t)b * lk_pad + j) * C + c8 * 8) = *reinterpret_cast<const uint4*>(src);
    const uint4 v = *reinterpret_cast<const uint4*>(src + C);
    const h16_t* ve = reinterpret_cast<const h16_t*>(&v);
#pragma unroll
    for (int e = 0; e < 8; ++e) vt_out[((int64_t)b * C + c8 * 8 + e) * lk_pad + j] = ve[e];
  }
}

}  // namespace

extern "C" int avsd_ncfhw_to_rows_x2(const float* src, void* dst, int64_t dst_lo, int B, int C, int F, int HW, int cpad, int rep,
                                     float scale, void* stream) {
  AVSD_REQUIRE(src && dst && B > 0 && C > 0 && F > 0 && HW > 0 && cpad >= C && rep >= 1, "ncfhw_to_rows: bad arguments");
  const int64_t n = (int64_t)rep * B * F * HW;
  hipLaunchKernelGGL(ncfhw_to_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, (h16_t*)dst, dst_lo, B, C, F, HW, cpad, rep, scale);
  AVSD_CHECK_LAUNCH("ncfhw_to_rows launch");
  return AVSD_OK;
}

extern "C" int avsd_ncfhw_to_rows(const float* src, void* dst, int B, int C, int F, int HW, int cpad, int rep,
                                  float scale, void* stream) {
  return avsd_ncfhw_to_rows_x2(src, dst, 0, B, C, F, HW, cpad, rep, scale, stream);
}

extern "C" int avsd_split_f32(const float* src, void* dst, int64_t dst_lo, int64_t n, void* stream) {
  AVSD_REQUIRE(src && dst && n > 0, "split_f32: bad arguments");
  hipLaunchKernelGGL(split_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     (h16_t*)dst, dst_lo, n);
  AVSD_CHECK_LAUNCH("split_f32 launch");
  return AVSD_OK;
}

extern "C" int avsd_rows_to_ncfhw(const float* src, int ld, float* dst, int B, int C, int F, int HW, void* stream) {
  AVSD_REQUIRE(src && dst && B > 0 && C > 0 && F > 0 && HW > 0 && ld >= C, "rows_to_ncfhw: bad arguments");
  const int64_t n = (int64_t)B * C * F * HW;
  hipLaunchKernelGGL(rows_to_ncfhw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, ld, dst, B, C, F, HW);
  AVSD_CHECK_LAUNCH("rows_to_ncfhw launch");
  return AVSD_OK;
}

extern "C" int avsd_timestep_embedding(const float* t, float* out, int n, int dim, void* stream) {
  AVSD_REQUIRE(t && out && n > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t,
                     out, n, dim);
  AVSD_CHECK_LAUNCH("timestep_embedding launch");
  return AVSD_OK;
}

extern "C" int avsd_linear_small_m_x2(const float* x, const void* W, int64_t w_lo, const float* bias, float* out, int M, int N, int K,
                                      int ldw, int act_in, int act_out, void* stream) {
  AVSD_REQUIRE(x && W && out, "linear_small_m: null pointer");
  AVSD_REQUIRE(M > 0 && M <= 64 && N > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldw >= K && w_lo % 8 == 0,
               "linear_small_m: need 0 < M <= 64, K %% 8 == 0, ldw %% 8 == 0 (M=%d N=%d K=%d ldw=%d)", M, N, K, ldw);
  dim3 grid((unsigned)((N + 3) / 4), (unsigned)((M + 7) / 8));
  if (w_lo != 0)
    hipLaunchKernelGGL(linear_small_m_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       (const h16_t*)W, w_lo, bias, out, M, N, K, ldw, act_in, act_out);
  else
    hipLaunchKernelGGL(linear_small_m_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       (const h16_t*)W, w_lo, bias, out, M, N, K, ldw, act_in, act_out);
  AVSD_CHECK_LAUNCH("linear_small_m launch");
  return AVSD_OK;
}

extern "C" int avsd_linear_small_m(const float* x, const void* W, const float* bias, float* out, int M, int N, int K,
                                   int ldw, int act_in, int act_out, void* stream) {
  return avsd_linear_small_m_x2(x, W, 0, bias, out, M, N, K, ldw, act_in, act_out, stream);
}

extern "C" int avsd_guided_step(const float* noise_pred, int n_branch, float g, float g2, float* eps_hist, int store_slot,
                                float w_cur, const int32_t* hist_idx, const float* w, int n_hist, const float* x_in,
                                float* x_out, float ca, float cb, int B, int C, int F, int HW, void* stream) {
  AVSD_REQUIRE(noise_pred && x_in && x_out, "guided_step: null pointer");
  AVSD_REQUIRE(n_branch >= 1 && n_branch <= 3, "guided_step: n_branch must be 1, 2 or 3");
  AVSD_REQUIRE(n_hist >= 0 && n_hist <= 4, "guided_step: n_hist must be in [0, 4]");
  AVSD_REQUIRE((n_hist == 0 && store_slot < 0) || eps_hist, "guided_step: history requested without eps_hist");
  AVSD_REQUIRE(n_hist == 0 || (hist_idx && w), "guided_step: null history tables");
  AVSD_REQUIRE(B > 0 && C > 0 && F > 0 && HW > 0, "guided_step: bad shape");
  GuidedArgs a;
  a.noise_pred = noise_pred; a.eps_hist = eps_hist; a.x_in = x_in; a.x_out = x_out;
  a.n_branch = n_branch; a.store_slot = store_slot; a.n_hist = n_hist;
  for (int k = 0; k < 4; ++k) { a.hist_idx[k] = k < n_hist ? hist_idx[k] : 0; a.w[k] = k < n_hist ? w[k] : 0.f; }
  a.g = g; a.g2 = g2; a.w_cur = w_cur; a.ca = ca; a.cb = cb;
  a.B = B; a.C = C; a.F = F; a.HW = HW;
  const int64_t n = (int64_t)B * C * F * HW;
  hipLaunchKernelGGL(guided_step_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  AVSD_CHECK_LAUNCH("guided_step launch");
  return AVSD_OK;
}

extern "C" int avsd_vae_postprocess_x2(const void* src, int ld, int64_t src_lo, float* dst, int N, int HW, void* stream) {
  AVSD_REQUIRE(src && dst && N > 0 && HW > 0 && ld >= 3, "vae_postprocess: bad arguments");
  const int64_t n = (int64_t)N * 3 * HW;
  hipLaunchKernelGGL(vae_postprocess_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const h16_t*)src, ld, src_lo, dst, N, HW);
  AVSD_CHECK_LAUNCH("vae_postprocess launch");
  return AVSD_OK;
}

extern "C" int avsd_vae_postprocess(const void* src, int ld, float* dst, int N, int HW, void* stream) {
  return avsd_vae_postprocess_x2(src, ld, 0, dst, N, HW, stream);
}

extern "C" int avsd_vae_postprocess_u8_x2(const void* src, int ld, int64_t src_lo, void* dst, int N, int HW, void* stream) {
  AVSD_REQUIRE(src && dst && N > 0 && HW > 0 && ld >= 3, "vae_postprocess_u8: bad arguments");
  const int64_t n = (int64_t)N * HW;
  hipLaunchKernelGGL(vae_postprocess_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const h16_t*)src, ld, src_lo, (uint8_t*)dst, n);
  AVSD_CHECK_LAUNCH("vae_postprocess_u8 launch");
  return AVSD_OK;
}

extern "C" int avsd_vae_postprocess_u8(const void* src, int ld, void* dst, int N, int HW, void* stream) {
  return avsd_vae_postprocess_u8_x2(src, ld, 0, dst, N, HW, stream);
}

extern "C" int avsd_copy(const void* src, void* dst, int64_t bytes, int rep, void* stream) {
  AVSD_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && rep >= 1, "copy: need non-null pointers, bytes %% 16 == 0, rep >= 1");
  AVSD_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "copy: pointers must be 16-byte aligned");
  const int64_t nvec = bytes / 16;
  hipLaunchKernelGGL(copy_rep_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const uint4*)src, (uint4*)dst, nvec, rep);
  AVSD_CHECK_LAUNCH("copy launch");
  return AVSD_OK;
}

extern "C" int avsd_xattn_pack_kv(const void* kv, int n_kv, int rows, int C, const int32_t* idx, int n_frames, int nk,
                                  void* k_out, void* vt_out, int lk_pad, void* stream) {
  AVSD_REQUIRE(kv && k_out && vt_out && n_kv > 0 && rows > 0 && C > 0 && C % 8 == 0, "xattn_pack_kv: bad arguments");
  AVSD_REQUIRE(!idx || (n_frames > 0 && nk > 0 && nk <= rows), "xattn_pack_kv: a gather list needs n_frames > 0 and 0 < nk <= rows");
  const int lk = idx ? nk : rows;
  AVSD_REQUIRE(lk_pad >= lk, "xattn_pack_kv: lk_pad (%d) < keys per block (%d)", lk_pad, lk);
  const int nb = idx ? n_kv * n_frames : n_kv;
  const int64_t total = (int64_t)nb * lk_pad * (C / 8);
  hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3(grid_for(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const h16_t*)kv, rows, C, idx, n_frames, nk, lk, (h16_t*)k_out, (h16_t*)vt_out, lk_pad, total);
  AVSD_CHECK_LAUNCH("xattn_pack_kv launch");
  return AVSD_OK;
}
