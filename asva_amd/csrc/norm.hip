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
// Two launches, both deterministic (no atomics):
//   gn_stats_kernel    grid (nchunks, nb): thread -> (channel vector t % nvec, position t / nvec), striding ppb
//                      positions over its chunk of rows; per-thread sums are laid out in LDS [ppb][2C] and the
//                      first `groups` threads fold (positions x channels-of-group) -> partial[nb][nchunks][g][2]
//   gn_apply_kernel    every block folds the chunk partials of its batch in double -> (mean, rstd) per group ->
//                      scale/shift per channel in LDS, then streams its rows: y = act(x * scale[c] + shift[c])
// X2: split-precision planes (avsd_common.h): values are main + rest, lo1 / lo2 the offsets to the rest planes
template <bool X2>
__global__ void gn_stats_kernel(const h16_t* x1, int ld1, int c1, int64_t lo1, const h16_t* x2, int ld2, int c2, int64_t lo2,
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
    const h16_t* base;
    int ld;
    int64_t lo;
    if (c0 < c1) { base = x1 + c0; ld = ld1; lo = lo1; } else { base = x2 + (c0 - c1); ld = ld2; lo = lo2; }
    base += (int64_t)b * rows_per_batch * ld;
    int r = r0 + pos0;
    for (; r + 3 * ppb < r1; r += 4 * ppb) {          // 4 independent loads in flight
      uint4 v[4], w[X2 ? 4 : 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)(r + u * ppb) * ld);
        if constexpr (X2) w[u] = *reinterpret_cast<const uint4*>(base + lo + (int64_t)(r + u * ppb) * ld);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8(v[u], f);
        if constexpr (X2) {
          float g[8];
          unpack8(w[u], g);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += g[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]); }
      }
    }
    for (; r < r1; r += ppb) {
      float f[8];
      load8<X2>(base + (int64_t)r * ld, lo, f);
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

// ---- GroupNorm apply (+ optional SiLU, + channel concat) ---------------------------------------------------------
// grid (blocks per batch, nb).  Every block first folds the chunk partials of ITS batch into (mean, rstd) per group —
// up to 64 lanes per group, double accumulation, fixed order — and builds scale[c] = rstd * gamma, shift[c] =
// beta - mean * scale in LDS (a separate finalize launch cost a full ~5 us kernel boundary for a few KB of work), then
// streams its share of the batch's rows: y = act(x * scale + shift), 2 vectors in flight per thread.
template <bool X2>
__global__ __launch_bounds__(1024) void gn_apply_kernel(const h16_t* x1, int ld1, int c1, int64_t lo1, const h16_t* x2, int ld2, int c2,
                                                       int64_t lo2, int rows_per_batch, const float* partial, int nchunks, int groups,
                                                       float eps, const float* gamma, const float* beta, int act,
                                                       h16_t* y, int ldy, int64_t loy) {
  extern __shared__ float gn_ss[];   // [C] scale | [C] shift
  __shared__ float smean[64], srstd[64];
  const int C = c1 + c2;
  const int cg = C / groups;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  {
    const int lpg = min(1024 / groups, 64);  // lanes per group: a power of two, at most one wave (host-checked)
    const int g = min(tid / lpg, groups - 1), sub = tid % lpg;
    const bool gvalid = tid / lpg < groups;
    double a = 0.0, q = 0.0;
    const float2* base = reinterpret_cast<const float2*>(partial) + (int64_t)b * nchunks * groups + g;
    int k = sub;
    for (; k + 15 * lpg < nchunks; k += 16 * lpg) {    // sixteen independent loads in flight: one round trip for
      float2 u[16];                                     // the 128 chunks of a pooled (F, H, W) batch
#pragma unroll
      for (int t = 0; t < 16; ++t) u[t] = base[(int64_t)(k + t * lpg) * groups];
#pragma unroll
      for (int t = 0; t < 16; ++t) { a += (double)u[t].x; q += (double)u[t].y; }
    }
    for (; k + 3 * lpg < nchunks; k += 4 * lpg) {
      float2 u[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) u[t] = base[(int64_t)(k + t * lpg) * groups];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a += (double)u[t].x; q += (double)u[t].y; }
    }
    for (; k < nchunks; k += lpg) {
      const float2 u = base[(int64_t)k * groups];
      a += (double)u.x;
      q += (double)u.y;
    }
    for (int off = lpg >> 1; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      q += __shfl_xor(q, off, 64);
    }
    if (sub == 0 && gvalid) {
      const double n = (double)rows_per_batch * cg;
      const double mean = a / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      smean[g] = (float)mean;
      srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024) {
    const int gg = c / cg;
    const float sc = srstd[gg] * gamma[c];
    gn_ss[c] = sc;
    gn_ss[C + c] = beta[c] - smean[gg] * sc;
  }
  __syncthreads();

  const int nvec = C / 8;
  const int rpb = (rows_per_batch + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb;
  const int r1 = min(rows_per_batch, r0 + rpb);
  const int64_t row_base = (int64_t)b * rows_per_batch;
  const int total = max(r1 - r0, 0) * nvec;
  constexpr int NV = 2;                 // vectors in flight per thread
  for (int i0 = tid; i0 < total; i0 += NV * 1024) {
    uint4 v[NV], w[X2 ? NV : 1];
    int cc[NV];
    int64_t gr[NV];
    bool ok[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int i = i0 + u * 1024;
      ok[u] = i < total;
      const int r = ok[u] ? i / nvec : 0;
      gr[u] = row_base + r0 + r;
      cc[u] = ok[u] ? (i - r * nvec) * 8 : 0;
      const h16_t* src = (cc[u] < c1) ? x1 + gr[u] * ld1 + cc[u] : x2 + gr[u] * ld2 + (cc[u] - c1);
      v[u] = ok[u] ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
      if constexpr (X2) w[u] = ok[u] ? *reinterpret_cast<const uint4*>(src + ((cc[u] < c1) ? lo1 : lo2)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (!ok[u]) continue;
      const float* ss = gn_ss + cc[u];
      const float4 s0 = *reinterpret_cast<const float4*>(ss), s1 = *reinterpret_cast<const float4*>(ss + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(ss + C), h1 = *reinterpret_cast<const float4*>(ss + C + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      float f[8];
      unpack8(v[u], f);
      if constexpr (X2) {
        float g[8];
        unpack8(w[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += g[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(f[e], sc[e], sh[e]);
        f[e] = act ? silu_f(t) : t;
      }
      store8<X2>(y + gr[u] * ldy + cc[u], loy, f);
    }
  }
}

// ---- LayerNorm row statistics, pre-folded: the K / 32 (sum, sumsq) pairs a ROWSTATS producer wrote per row -> one pair per row,
// added in the order the consumers' own fold uses (ascending block), so a consumer handed the folded pair (ln_nblk = 1) computes
// bit-identical mean / rstd.  Worth a launch where a consumer would re-fold the same rows in many column tiles (the GEGLU projection:
// 20-80 column tiles, +8-10 us per launch, tools/geglu_probe.py).
__global__ __launch_bounds__(256) void ln_fold_kernel(const float2* stats, int M, int nblk, float2* out) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float2* st = stats + (int64_t)m * nblk;
  float sm = 0.f, sq = 0.f;
  int j = 0;
  for (; j + 10 <= nblk; j += 10) {
    float2 t[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) t[u] = st[j + u];
#pragma unroll
    for (int u = 0; u < 10; ++u) { sm += t[u].x; sq += t[u].y; }
  }
  for (; j < nblk; ++j) { const float2 t = st[j]; sm += t.x; sq += t.y; }
  out[m] = make_float2(sm, sq);
}

// ---- LayerNorm: LPR lanes per row (power of two), 64/LPR rows per wave, up to 8 vectors per lane in registers.
// C = 320/640/1280 -> LPR = 8/16/32 with exactly 5 vectors per lane: every lane busy, 5 loads in flight per lane.
template <int LPR, bool X2>
__global__ __launch_bounds__(256) void layernorm_kernel(const h16_t* x, int ldx, int64_t lox, h16_t* y, int ldy, int64_t loy, int M,
                                                        int C, const float* gamma, const float* beta,
                                                        float eps, const float* pos, int hw, int frames) {
  constexpr int RPW = 64 / LPR;                 // rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool rvalid = row < M;
  const int nvec = C / 8;
  const float* prow = (pos && rvalid) ? pos + (int64_t)((row / hw) % frames) * C : nullptr;
  float f[8][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = sub + LPR * i;
    if (rvalid && v < nvec) {
      load8<X2>(x + (int64_t)row * ldx + v * 8, lox, f[i]);
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
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = sub + LPR * i;
    if (rvalid && v < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; sq = fmaf(d, d, sq); }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = sub + LPR * i;
    if (rvalid && v < nvec) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + v * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + v * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + v * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(beta + v * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf((f[i][e] - mean) * rstd, g[e], bb[e]);
      store8<X2>(y + (int64_t)row * ldy + v * 8, loy, o);
    }
  }
}

template <int LPR>
void launch_layernorm(const h16_t* x, int ldx, int64_t lox, h16_t* y, int ldy, int64_t loy, int M, int C, const float* gamma,
                      const float* beta, float eps, const float* pos, int hw, int frames, hipStream_t s) {
  const int rows_per_block = 4 * (64 / LPR);
  const dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block));
  if (lox != 0)
    hipLaunchKernelGGL((layernorm_kernel<LPR, true>), grid, dim3(256), 0, s, x, ldx, lox, y, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames);
  else
    hipLaunchKernelGGL((layernorm_kernel<LPR, false>), grid, dim3(256), 0, s, x, ldx, lox, y, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames);
}

// ---- row softmax: S f32 -> P bf16, one wave per row ---------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* S, int lds, h16_t* P, int ldp, int64_t p_lo, int rows,
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
  h16_t* o = P + (int64_t)row * ldp;
  for (int j = lane * 4; j < L; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + j);
    uint2 st, sr;
    split2(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv, st.x, sr.x);
    split2(__expf(v.z - mx) * inv, __expf(v.w - mx) * inv, st.y, sr.y);
    *reinterpret_cast<uint2*>(o + j) = st;
    if (p_lo) *reinterpret_cast<uint2*>(o + p_lo + j) = sr;      // split-precision planes
  }
}

// (round 6, profiles/r6_pmc_groupnorm.md: the pooled ResBlock norms at 32 x 32 run 512 statistics waves on 1024 SIMDs, 52 vector loads per
//  wave in dependent batches of 4, 66 % of the wave cycles parked.  Spreading a chunk over up to 1024 threads — 2 batches per thread instead of
//  8 — was SLOWER on every shape, 20.4 -> 23.3 us for that pair, 12.0 -> 16.3 at 16 x 16: the 32 threads that fold the per-thread sums out of
//  LDS then walk 4x the positions, serially, behind the barrier.  256 threads stay.)
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
  if (n > 64) n = 64;                // every apply workgroup folds the nchunks partials of its batch (256 chunks measured 4 % SLOWER over
                                     // the step's norms, round 5: the apply's fold grows faster than the statistics pass shrinks)
  if (n < 1) n = 1;
  return n;
}

// floats of scratch the stats + apply pair needs: partial[nb][nchunks][groups][2]
extern "C" int avsd_groupnorm_scratch_floats(int nb, int nchunks, int groups, int channels) {
  (void)channels;
  return nb * nchunks * groups * 2;
}

static int gn_check(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                    int rows_per_batch, int groups, int nchunks) {
  AVSD_REQUIRE(x1 && c1 > 0 && c1 % 8 == 0 && ld1 % 8 == 0 && ld1 >= c1, "groupnorm: bad first source (c1=%d ld1=%d)", c1, ld1);
  AVSD_REQUIRE(c2 >= 0 && c2 % 8 == 0 && (c2 == 0 || (x2 && ld2 % 8 == 0 && ld2 >= c2)), "groupnorm: bad second source (c2=%d ld2=%d)", c2, ld2);
  const int C = c1 + c2;
  AVSD_REQUIRE(groups >= 4 && groups <= 64 && (groups & (groups - 1)) == 0 && C % groups == 0,
               "groupnorm: groups (%d) must be a power of two in 4..64 dividing the channels (%d)", groups, C);
  AVSD_REQUIRE(C <= 4096, "groupnorm: at most 4096 channels (got %d)", C);
  AVSD_REQUIRE(nb > 0 && rows_per_batch > 0 && nchunks > 0 && nchunks <= rows_per_batch, "groupnorm: bad batch geometry nb=%d rows=%d nchunks=%d", nb, rows_per_batch, nchunks);
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_stats_x2(const void* x1, int ld1, int c1, int64_t lo1, const void* x2, int ld2, int c2, int64_t lo2,
                                       int nb, int rows_per_batch, int groups, float* scratch, int nchunks, void* stream) {
  int rc = gn_check(x1, ld1, c1, x2, ld2, c2, nb, rows_per_batch, groups, nchunks);
  if (rc) return rc;
  AVSD_REQUIRE(scratch, "groupnorm_stats: null pointer");
  AVSD_REQUIRE(lo1 % 8 == 0 && lo2 % 8 == 0 && (c2 == 0 || (lo1 != 0) == (lo2 != 0)), "groupnorm_stats: plane offsets must be multiples of 8, both sources split or neither");
  const int C = c1 + c2;
  int nvec, ppb, threads;
  gn_geometry(C, &nvec, &ppb, &threads);
  const size_t lds = (size_t)ppb * 2 * C * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (lo1 != 0)
    hipLaunchKernelGGL(gn_stats_kernel<true>, dim3((unsigned)nchunks, (unsigned)nb), dim3((unsigned)threads), lds, s, (const h16_t*)x1,
                       ld1, c1, lo1, (const h16_t*)x2, ld2, c2, lo2, rows_per_batch, groups, scratch, nchunks, nvec, ppb);
  else
    hipLaunchKernelGGL(gn_stats_kernel<false>, dim3((unsigned)nchunks, (unsigned)nb), dim3((unsigned)threads), lds, s, (const h16_t*)x1,
                       ld1, c1, lo1, (const h16_t*)x2, ld2, c2, lo2, rows_per_batch, groups, scratch, nchunks, nvec, ppb);
  AVSD_CHECK_LAUNCH("groupnorm_stats launch");
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_stats(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                                    int rows_per_batch, int groups, float* scratch, int nchunks, void* stream) {
  return avsd_groupnorm_stats_x2(x1, ld1, c1, 0, x2, ld2, c2, 0, nb, rows_per_batch, groups, scratch, nchunks, stream);
}

extern "C" int avsd_ln_fold(const float* stats, int M, int nblk, float* out, void* stream) {
  AVSD_REQUIRE(stats && out && M > 0 && nblk > 0, "ln_fold: bad arguments (M %d, nblk %d)", M, nblk);
  hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2*>(stats), M, nblk, reinterpret_cast<float2*>(out));
  AVSD_CHECK_LAUNCH("ln_fold launch");
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_apply(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb,
                                    int rows_per_batch, int groups, const float* gamma, const float* beta, float eps,
                                    const float* scratch, int nchunks, int act, void* y, int ldy, void* stream) {
  return avsd_groupnorm_apply_x2(x1, ld1, c1, 0, x2, ld2, c2, 0, nb, rows_per_batch, groups, gamma, beta, eps, scratch, nchunks, act,
                                 y, ldy, 0, stream);
}

extern "C" int avsd_groupnorm_apply_x2(const void* x1, int ld1, int c1, int64_t lo1, const void* x2, int ld2, int c2, int64_t lo2,
                                       int nb, int rows_per_batch, int groups, const float* gamma, const float* beta, float eps,
                                       const float* scratch, int nchunks, int act, void* y, int ldy, int64_t loy, void* stream) {
  int rc = gn_check(x1, ld1, c1, x2, ld2, c2, nb, rows_per_batch, groups, nchunks);
  if (rc) return rc;
  AVSD_REQUIRE(lo1 % 8 == 0 && lo2 % 8 == 0 && loy % 8 == 0 && (lo1 != 0) == (loy != 0) && (c2 == 0 || (lo1 != 0) == (lo2 != 0)),
               "groupnorm_apply: plane offsets must be multiples of 8; sources and output all split or none");
  AVSD_REQUIRE(scratch && y && gamma && beta, "groupnorm_apply: null pointer");
  const int C = c1 + c2;
  AVSD_REQUIRE(ldy % 8 == 0 && ldy >= C, "groupnorm_apply: bad ldy %d", ldy);
  // ~512 workgroups of 1024 threads over the batches (every workgroup re-folds the partials of its batch, so fewer,
  // larger workgroups), each with at least 2048 vectors to stream
  const int64_t vec_per_batch = (int64_t)rows_per_batch * (C / 8);
  int bpb = 512 / nb;
  if (bpb < 1) bpb = 1;
  const int64_t cap = (vec_per_batch + 2047) / 2048;
  if (bpb > cap) bpb = (int)cap;
  if (bpb > rows_per_batch) bpb = rows_per_batch;
  if (lo1 != 0)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)bpb, (unsigned)nb), dim3(1024), (size_t)2 * C * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), (const h16_t*)x1, ld1, c1, lo1, (const h16_t*)x2, ld2, c2, lo2,
                       rows_per_batch, scratch, nchunks, groups, eps, gamma, beta, act, (h16_t*)y, ldy, loy);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)bpb, (unsigned)nb), dim3(1024), (size_t)2 * C * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), (const h16_t*)x1, ld1, c1, lo1, (const h16_t*)x2, ld2, c2, lo2,
                       rows_per_batch, scratch, nchunks, groups, eps, gamma, beta, act, (h16_t*)y, ldy, loy);
  AVSD_CHECK_LAUNCH("groupnorm_apply launch");
  return AVSD_OK;
}

extern "C" int avsd_layernorm(const void* x, int ldx, void* y, int ldy, int M, int C, const float* gamma,
                              const float* beta, float eps, const float* pos, int hw, int frames, void* stream) {
  return avsd_layernorm_x2(x, ldx, 0, y, ldy, 0, M, C, gamma, beta, eps, pos, hw, frames, stream);
}

extern "C" int avsd_layernorm_x2(const void* x, int ldx, int64_t lox, void* y, int ldy, int64_t loy, int M, int C, const float* gamma,
                                 const float* beta, float eps, const float* pos, int hw, int frames, void* stream) {
  AVSD_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
  AVSD_REQUIRE(lox % 8 == 0 && loy % 8 == 0 && (lox != 0) == (loy != 0), "layernorm: plane offsets must be multiples of 8, input and output both split or neither");
  AVSD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 4096, "layernorm: C (%d) must be a multiple of 8, <= 4096", C);
  AVSD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "layernorm: bad strides");
  if (pos) AVSD_REQUIRE(hw > 0 && frames > 0, "layernorm: pos needs hw and frames");
  if (!pos) { hw = 1; frames = 1; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nvec = C / 8;
  const h16_t* xi = (const h16_t*)x;
  h16_t* yo = (h16_t*)y;
  if (nvec <= 8) launch_layernorm<1>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else if (nvec <= 16) launch_layernorm<2>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else if (nvec <= 32) launch_layernorm<4>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else if (nvec <= 64) launch_layernorm<8>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else if (nvec <= 128) launch_layernorm<16>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else if (nvec <= 256) launch_layernorm<32>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  else launch_layernorm<64>(xi, ldx, lox, yo, ldy, loy, M, C, gamma, beta, eps, pos, hw, frames, s);
  AVSD_CHECK_LAUNCH("layernorm launch");
  return AVSD_OK;
}

extern "C" int avsd_softmax_rows(const float* S, int lds, void* P, int ldp, int rows, int L, void* stream) {
  return avsd_softmax_rows_x2(S, lds, P, ldp, 0, rows, L, stream);
}

extern "C" int avsd_softmax_rows_x2(const float* S, int lds, void* P, int ldp, int64_t p_lo, int rows, int L, void* stream) {
  AVSD_REQUIRE(S && P && rows > 0 && L > 0, "softmax_rows: bad arguments");
  AVSD_REQUIRE(L % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0 && p_lo % 4 == 0, "softmax_rows: L, lds, ldp must be multiples of 4");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), S, lds, (h16_t*)P, ldp, p_lo, rows, L);
  AVSD_CHECK_LAUNCH("softmax_rows launch");
  return AVSD_OK;
}
