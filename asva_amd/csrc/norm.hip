// HBM-bound normalisation kernels for gfx950: GroupNorm (stats + apply), LayerNorm, row softmax.
// All loads/stores are 16-byte vectors of 8 bf16 channels, coalesced along the channels-last row.
//
// Reference: torch.nn.GroupNorm at ff_spatio_temp_resnet_3d.py:130,146 (5-D input: statistics pooled
// over frames, H, W), audio_cond_unet_3d_condition.py:445 (conv_norm_out),
// ff_spatio_audio_temp_transformer_3d.py:62 (per-frame, eps 1e-6); torch.nn.LayerNorm at
// ff_spatio_audio_temp_transformer_3d.py:199-275; biased variance, eps inside the sqrt.
#include "avsd_common.h"

namespace {

// ---- GroupNorm statistics -----------------------------------------------------------------------
// Three launches, all deterministic (no atomics):
//   gn_stats_kernel    grid (nchunks, nb): thread -> (channel vector t % nvec, position t / nvec), striding ppb
//                      positions over its chunk of rows; per-thread sums are laid out in LDS [ppb][2C] and the
//                      first `groups` threads fold (positions x channels-of-group) -> partial[nb][nchunks][g][2]
//   gn_finalize_kernel grid nb: 8 lanes per group tree-reduce the chunks in double -> stat[nb][g] = (mean, rstd)
//   gn_apply_kernel    streams rows: y = act(x * scale[c] + shift[c]) with scale/shift built once per block in LDS
__global__ void gn_stats_kernel(const bf16_t* x1, int ld1, int c1, const bf16_t* x2, int ld2, int c2,
                                int rows_per_batch, int groups, float* partial, int nchunks, int nvec,
                                int ppb) {
  extern __shared__ float sgn[];  // [ppb][2*C]: sums | sums of squares
  const int C = c1 + c2;
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  const int b = blockIdx.y;
  const int chunk_rows = (rows_per_batch + nchunks - 1) / nchunks;
  const int r0 = chunk * chunk_rows;
  const int r1 = min(rows_per_batch, r0 + chunk_rows);
  const int vec = tid % nvec;
  const int pos0 = tid / nvec;
  const int c0 = vec * 8;

  if (pos0 < ppb) {
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    const bf16_t* base;
    int ld;
    if (c0 < c1) { base = x1 + c0; ld = ld1; } else { base = x2 + (c0 - c1); ld = ld2; }
    base += (int64_t)b * rows_per_batch * ld;
    int r = r0 + pos0;
    for (; r + 3 * ppb < r1; r += 4 * ppb) {          // 4 independent loads in flight
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)(r + u * ppb) * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]); }
      }
    }
    for (; r < r1; r += ppb) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(base + (int64_t)r * ld), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]); }
    }
    float* dst = sgn + (size_t)pos0 * 2 * C + c0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dst[e] = s[e]; dst[C + e] = ss[e]; }
  }
  __syncthreads();
  if (tid < groups) {
    const int cg = C / groups;
    float a = 0.f, q = 0.f;
    for (int pp = 0; pp < ppb; ++pp) {
      const float* row = sgn + (size_t)pp * 2 * C + tid * cg;
      for (int c = 0; c < cg; ++c) { a += row[c]; q += row[C + c]; }
    }
    float* o = partial + (((int64_t)b * nchunks + chunk) * groups + tid) * 2;
    o[0] = a;
    o[1] = q;
  }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* partial, int nchunks, int groups, int rows_per_batch,
                                                           int cg, float eps, float* stat) {
  // 16 lanes per group, 4 independent accumulator pairs per lane so the loads overlap
  const int b = blockIdx.x;
  const int g = threadIdx.x >> 4;
  const int sub = threadIdx.x & 15;
  double a0 = 0.0, q0 = 0.0, a1 = 0.0, q1 = 0.0;
  if (g < groups) {
    const float2* base = reinterpret_cast<const float2*>(partial) + (int64_t)b * nchunks * groups + g;
    int k = sub;
    for (; k + 16 < nchunks; k += 32) {
      const float2 u = base[(int64_t)k * groups];
      const float2 v = base[(int64_t)(k + 16) * groups];
      a0 += (double)u.x; q0 += (double)u.y;
      a1 += (double)v.x; q1 += (double)v.y;
    }
    if (k < nchunks) {
      const float2 u = base[(int64_t)k * groups];
      a0 += (double)u.x; q0 += (double)u.y;
    }
  }
  double a = a0 + a1, q = q0 + q1;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (g < groups && sub == 0) {
    const double n = (double)rows_per_batch * cg;
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[((int64_t)b * groups + g) * 2 + 0] = (float)mean;
    stat[((int64_t)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- GroupNorm apply (+ optional SiLU, + channel concat) ----------------------------------------
__global__ void gn_apply_kernel(const bf16_t* x1, int ld1, int c1, const bf16_t* x2, int ld2, int c2,
                                int rows_per_batch, int groups, const float* stat,
                                const float* gamma, const float* beta, int act, bf16_t* y,
                                int ldy, int rows_per_block) {
  extern __shared__ float sap[];  // [C] scale | [C] shift
  const int C = c1 + c2;
  float* sscale = sap;
  float* sshift = sap + C;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int cg = C / groups;
  for (int c = tid; c < C; c += blockDim.x) {
    const int g = c / cg;
    const float mean = stat[((int64_t)b * groups + g) * 2 + 0];
    const float rstd = stat[((int64_t)b * groups + g) * 2 + 1];
    const float sc = rstd * gamma[c];
    sscale[c] = sc;
    sshift[c] = beta[c] - mean * sc;
  }
  __syncthreads();

  const int nvec = C / 8;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows_per_batch, r0 + rows_per_block);
  const int total = (r1 - r0) * nvec;
  for (int idx0 = tid; idx0 < total; idx0 += 2 * blockDim.x) {
    uint4 v[2];
    int cc[2];
    int64_t gr[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = idx0 + u * blockDim.x;
      ok[u] = idx < total;
      const int r = r0 + (ok[u] ? idx / nvec : 0);
      cc[u] = ok[u] ? (idx % nvec) * 8 : 0;
      gr[u] = (int64_t)b * rows_per_batch + r;
      const bf16_t* src = (cc[u] < c1) ? x1 + gr[u] * ld1 + cc[u] : x2 + gr[u] * ld2 + (cc[u] - c1);
      v[u] = ok[u] ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(f[e], sscale[cc[u] + e], sshift[cc[u] + e]);
        f[e] = act ? silu_f(t) : t;
      }
      *reinterpret_cast<uint4*>(y + gr[u] * ldy + cc[u]) = pack8(f);
    }
  }
}

// ---- LayerNorm: one wave per row, up to 4 vectors (C <= 2048) per lane in registers -------------
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* x, int ldx, bf16_t* y, int ldy, int M,
                                                        int C, const float* gamma, const float* beta,
                                                        float eps, const float* pos, int hw, int frames) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = C / 8;
  const float* prow = pos ? pos + (int64_t)((row / hw) % frames) * C : nullptr;
  float f[4][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      unpack8(*reinterpret_cast<const uint4*>(x + (int64_t)row * ldx + v * 8), f[i]);
      if (prow) {
        const float4 p0 = *reinterpret_cast<const float4*>(prow + v * 8);
        const float4 p1 = *reinterpret_cast<const float4*>(prow + v * 8 + 4);
        f[i][0] += p0.x; f[i][1] += p0.y; f[i][2] += p0.z; f[i][3] += p0.w;
        f[i][4] += p1.x; f[i][5] += p1.y; f[i][6] += p1.z; f[i][7] += p1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += f[i][e];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; sq = fmaf(d, d, sq); }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + v * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + v * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + v * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(beta + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf((f[i][e] - mean) * rstd, g[e], bb[e]);
      *reinterpret_cast<uint4*>(y + (int64_t)row * ldy + v * 8) = pack8(o);
    }
  }
}

// ---- row softmax: S f32 -> P bf16, one wave per row ---------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* S, int lds, bf16_t* P, int ldp, int rows,
                                                           int L) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + (int64_t)row * lds;
  float mx = -1e30f;
  for (int j = lane * 4; j < L; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + j);
    mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane * 4; j < L; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + j);
    sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
  }
  const float inv = 1.0f / wave_sum(sum);
  bf16_t* o = P + (int64_t)row * ldp;
  for (int j = lane * 4; j < L; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + j);
    uint2 st;
    st.x = pack2bf(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv);
    st.y = pack2bf(__expf(v.z - mx) * inv, __expf(v.w - mx) * inv);
    *reinterpret_cast<uint2*>(o + j) = st;
  }
}

void gn_geometry(int C, int* nvec, int* ppb, int* threads) {
  *nvec = C / 8;
  int p = 256 / *nvec;
  if (p < 1) p = 1;
  *ppb = p;
  int t = (*nvec * p + 63) / 64 * 64;
  *threads = t;
}

}  // namespace

extern "C" int avsd_groupnorm_nchunks(int nb, int rows_per_batch, int channels) {
  (void)channels;
  if (nb <= 0 || rows_per_batch <= 0) return 1;
  int n = 1024 / nb;                 // ~1024 blocks in flight
  int cap = rows_per_batch / 16;     // >= 16 rows per chunk
  if (n > cap) n = cap;
  if (n > 512) n = 512;
  if (n < 1) n = 1;
  return n;
}

// floats of scratch the stats + apply pair needs: partial[nb][nchunks][groups][2] then stat[nb][groups][2]
extern "C" int avsd_groupnorm_scratch_floats(int nb, int nchunks, int groups) {
  return nb * nchunks * groups * 2 + nb * groups * 2;
}

static int gn_check(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                    int rows_per_batch, int groups, int nchunks) {
  AVSD_REQUIRE(x1 && c1 > 0 && c1 % 8 == 0 && ld1 % 8 == 0 && ld1 >= c1, "groupnorm: bad first source (c1=%d ld1=%d)", c1, ld1);
  AVSD_REQUIRE(c2 >= 0 && c2 % 8 == 0 && (c2 == 0 || (x2 && ld2 % 8 == 0 && ld2 >= c2)), "groupnorm: bad second source (c2=%d ld2=%d)", c2, ld2);
  const int C = c1 + c2;
  AVSD_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "groupnorm: channels (%d) not divisible by groups (%d)", C, groups);
  AVSD_REQUIRE(C <= 4096, "groupnorm: at most 4096 channels (got %d)", C);
  AVSD_REQUIRE(nb > 0 && rows_per_batch > 0 && nchunks > 0 && nchunks <= rows_per_batch, "groupnorm: bad batch geometry nb=%d rows=%d nchunks=%d", nb, rows_per_batch, nchunks);
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_stats(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                                    int rows_per_batch, int groups, float eps, float* scratch, int nchunks,
                                    void* stream) {
  int rc = gn_check(x1, ld1, c1, x2, ld2, c2, nb, rows_per_batch, groups, nchunks);
  if (rc) return rc;
  AVSD_REQUIRE(scratch, "groupnorm_stats: null scratch buffer");
  const int C = c1 + c2;
  int nvec, ppb, threads;
  gn_geometry(C, &nvec, &ppb, &threads);
  const size_t lds = (size_t)ppb * 2 * C * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)nchunks, (unsigned)nb), dim3((unsigned)threads), lds, s,
                     (const bf16_t*)x1, ld1, c1, (const bf16_t*)x2, ld2, c2, rows_per_batch, groups, scratch, nchunks, nvec, ppb);
  AVSD_CHECK_LAUNCH("groupnorm_stats launch");
  float* stat = scratch + (size_t)nb * nchunks * groups * 2;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)nb), dim3(1024), 0, s, scratch, nchunks, groups, rows_per_batch,
                     C / groups, eps, stat);
  AVSD_CHECK_LAUNCH("groupnorm_finalize launch");
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_apply(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                                    int rows_per_batch, int groups, const float* scratch, int nchunks,
                                    const float* gamma, const float* beta, int act, void* y, int ldy,
                                    void* stream) {
  int rc = gn_check(x1, ld1, c1, x2, ld2, c2, nb, rows_per_batch, groups, nchunks);
  if (rc) return rc;
  AVSD_REQUIRE(scratch && gamma && beta && y, "groupnorm_apply: null pointer");
  const int C = c1 + c2;
  AVSD_REQUIRE(ldy % 8 == 0 && ldy >= C, "groupnorm_apply: bad ldy %d", ldy);
  int rows_per_block = 16;
  while ((int64_t)nb * ((rows_per_batch + rows_per_block - 1) / rows_per_block) > 4096) rows_per_block *= 2;
  const int nblk = (rows_per_batch + rows_per_block - 1) / rows_per_block;
  const size_t lds = (size_t)2 * C * sizeof(float);
  const float* stat = scratch + (size_t)nb * nchunks * groups * 2;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)nblk, (unsigned)nb), dim3(256), lds,
                     reinterpret_cast<hipStream_t>(stream), (const bf16_t*)x1, ld1, c1, (const bf16_t*)x2, ld2, c2,
                     rows_per_batch, groups, stat, gamma, beta, act, (bf16_t*)y, ldy, rows_per_block);
  AVSD_CHECK_LAUNCH("groupnorm_apply launch");
  return AVSD_OK;
}

extern "C" int avsd_layernorm(const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma,
                              const float* beta, float eps, const float* pos, int hw, int frames, void* stream) {
  AVSD_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
  AVSD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "layernorm: C (%d) must be a multiple of 8, <= 2048", C);
  AVSD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "layernorm: bad strides");
  if (pos) AVSD_REQUIRE(hw > 0 && frames > 0, "layernorm: pos needs hw and frames");
  if (!pos) { hw = 1; frames = 1; }
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const bf16_t*)x, ldx, (bf16_t*)y, ldy, M, C, gamma, beta, eps, pos, hw, frames);
  AVSD_CHECK_LAUNCH("layernorm launch");
  return AVSD_OK;
}

extern "C" int avsd_softmax_rows(const float* S, int lds, void* P, int ldp, int rows, int L, void* stream) {
  AVSD_REQUIRE(S && P && rows > 0 && L > 0, "softmax_rows: bad arguments");
  AVSD_REQUIRE(L % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0, "softmax_rows: L, lds, ldp must be multiples of 4");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), S, lds, (bf16_t*)P, ldp, rows, L);
  AVSD_CHECK_LAUNCH("softmax_rows launch");
  return AVSD_OK;
}
