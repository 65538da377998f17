// Pieces shared by the GEMM family (gemm.hip) and the fused kernels built on its main loop (xattn.hip):
// the f32 epilogue, the LayerNorm-fold row statistics, counted vmcnt waits and the LDS-direct tile addressing.
#pragma once
#include <utility>

#include "avsd_common.h"

namespace {

constexpr int BK = 64;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order: the index is a compile-time constant inside f
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LayerNorm statistics of row m of A folded from the producer's per-32-column (sum, sumsq) pairs -> (rstd, mean * rstd)
__device__ __forceinline__ void ln_row_stats(const avsd_gemm_desc& p, int m, int64_t bz, float& rstd, float& mr) {
  const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + ((int64_t)bz * (p.batch_stride_a / p.lda) + m) * p.ln_nblk;
  float sm = 0.f, sq = 0.f;
  int j = 0;
  for (; j + 10 <= p.ln_nblk; j += 10) {      // ten independent loads in flight (C = 320 k -> 10 k pairs)
    float2 t[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) t[u] = st[j + u];
#pragma unroll
    for (int u = 0; u < 10; ++u) { sm += t[u].x; sq += t[u].y; }
  }
  for (; j < p.ln_nblk; ++j) {
    const float2 t = st[j];
    sm += t.x;
    sq += t.y;
  }
  const float inv_k = 1.0f / (float)p.K;       // the ln_nblk pairs of a row cover its K columns (32 each, or pre-folded: avsd_ln_fold)
  const float mean = sm * inv_k;
  const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
  rstd = rsqrtf(var + p.ln_eps);
  mr = mean * rstd;
}

// frame table row of output row m (stats_pos / ln_rowvec, include/avsd.h): [pos_frames][N], frame = (m / pos_hw) % pos_frames
__device__ __forceinline__ const float* pos_row(const avsd_gemm_desc& p, const float* table, int m) {
  return table + (int64_t)((m / p.pos_hw) % p.pos_frames) * p.N;
}

// (sum, sum of squares) of the 16 rounded values a lane has just stored (8 packed pairs, columns ncol .. ncol + 15 in order),
// optionally of (value + pos[column]) — AVSD_GEMM_ROWSTATS with stats_pos
__device__ __forceinline__ void rowstats16(const unsigned (&w8)[8], const float* pos16, float& sm, float& sq) {
  float pv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pv[i] = 0.f;
  if (pos16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(pos16 + 4 * j);
      pv[4 * j] = t.x; pv[4 * j + 1] = t.y; pv[4 * j + 2] = t.z; pv[4 * j + 3] = t.w;
    }
  }
  sm = 0.f;
  sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float lo = lo2f(w8[i]) + pv[2 * i], hi = hi2f(w8[i]) + pv[2 * i + 1];
    sm += lo + hi;
    sq = fmaf(lo, lo, fmaf(hi, hi, sq));
  }
}

// ---- compile-time extras of the shared epilogue (template parameter EX) -------------------------------------------------------------
// Round 6 first added both as run-time flags of the one epilogue: the bf16 step lost 1.6 %, the VAE decode 4.5 % against the round-5
// tree on the same box (profiles/r6_epilogue_ab.txt) and the 768-thread tiles spilled 10-87 registers (tools/resource_usage.py) — the
// shared epilogue is compiled into ~290 kernels that sit at their register budgets.  EX = 0 is the round-5 code, untouched.
//   EPI_REST  AVSD_GEMM_OUT_REST (include/avsd.h): the 16-bit output also gets its rest plane round16(v - main) at out + out_lo — the
//             (main, rest) pair a three-pass product reads, made by the producer instead of by an avsd_split_f32 pass over the f32
//             master (68 launches per step in round 5).  Instantiated in the IEEE-half build only (the per-layer precision plan is
//             fp16 storage, asva_amd/precision.py) by the kernels that produce residual-stream tensors: gemm2_kernel, gemm4_kernel.
//   EPI_SHUF  AVSD_GEMM_CONV3 with ups = 2: GEMM row m = input pixel (img, y, x), GEMM column n = (parity dy dx, channel co) is element
//             co of OUTPUT pixel (img, 2y + dy, 2x + dx) of the [n_img * 2 hs * 2 ws][ld] result (shuf_row: the output row of parity
//             (0, 0); shuf_col: a column — a whole 32-column fragment lies inside one parity — as (rows to add, channel)); the rest
//             plane is a run-time option there.  Instantiated by the CONV3 form of gemm2_kernel only.
constexpr int EPI_PLAIN = 0, EPI_REST = 1, EPI_SHUF = 2;
__device__ __forceinline__ unsigned rest2h(float v0, float v1, unsigned mainw) { return pack2h(v0 - lo2f(mainw), v1 - hi2f(mainw)); }
__device__ __forceinline__ int64_t shuf_row(const avsd_gemm_desc& p, int m) { return 4 * (int64_t)m - 2 * (m % p.ws); }
__device__ __forceinline__ void shuf_col(const avsd_gemm_desc& p, int n, int& radd, int& col) {
  const int cout = p.N >> 2, par = n / cout;
  col = n - par * cout;
  radd = (par >> 1) * 2 * p.ws + (par & 1);
}
template <int EX>
__device__ __forceinline__ bool epi_rest(const avsd_gemm_desc& p) {
  if constexpr (EX == EPI_REST) return true;
  else if constexpr (EX == EPI_SHUF) return (p.flags & AVSD_GEMM_OUT_REST) != 0;
  else return false;
}

// Big tiles hold 128-160 accumulator registers: letting the compiler batch the loads of ALL fragments of a term would spill.
// A scheduling fence after each fragment caps the batch at one fragment's loads (4-8 in flight), still one wait per fragment
// instead of one per vector.
template <int FN, int FM>
__device__ __forceinline__ void epilogue_fence() {
  if constexpr (FN * FM >= 8) __builtin_amdgcn_sched_barrier(0);
}

// acc += R[m][n] for every fragment: R f32 [M][ldr] or 16-bit (wide form when the fragment allows it, see epilogue)
template <int FN, int FM>
__device__ __forceinline__ void epilogue_add_residual(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], const void* R, int ldr, bool f32,
                                                      const int (&mld)[FM], int n_base, int hsel, int nmax, int64_t bo, bool wide_ok) {
  if (f32) {
    const float* Rf = reinterpret_cast<const float*>(R) + bo;
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int a = 0; a < FN; ++a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 rr = *reinterpret_cast<const float4*>(Rf + (int64_t)mld[b] * ldr + min(n_base + a * 32 + 8 * q + hsel, nmax));
          acc[a][b][4 * q + 0] += rr.x; acc[a][b][4 * q + 1] += rr.y; acc[a][b][4 * q + 2] += rr.z; acc[a][b][4 * q + 3] += rr.w;
        }
        epilogue_fence<FN, FM>();
      }
    return;
  }
  const h16_t* Rh = reinterpret_cast<const h16_t*>(R) + bo;
#pragma unroll
  for (int b = 0; b < FM; ++b)
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nb = n_base + a * 32;
      if (wide_ok && nb + 32 <= p.N) {
        const h16_t* rp = Rh + (int64_t)mld[b] * ldr + nb + 4 * hsel;      // this lane's 16 columns: nb (lanes 0-31) or nb + 16
        const uint4 lo = *reinterpret_cast<const uint4*>(rp);
        const uint4 hi = *reinterpret_cast<const uint4*>(rp + 8);
        const unsigned s0[2] = {lo.x, lo.y}, s1[2] = {lo.z, lo.w}, s2[2] = {hi.x, hi.y}, s3[2] = {hi.z, hi.w};
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto e = __builtin_amdgcn_permlane32_swap(s0[d], s1[d], false, false);   // -> quads 0 and 2
          const auto o = __builtin_amdgcn_permlane32_swap(s2[d], s3[d], false, false);   // -> quads 1 and 3
          acc[a][b][0 + 2 * d] += lo2f(e[0]); acc[a][b][1 + 2 * d] += hi2f(e[0]);
          acc[a][b][8 + 2 * d] += lo2f(e[1]); acc[a][b][9 + 2 * d] += hi2f(e[1]);
          acc[a][b][4 + 2 * d] += lo2f(o[0]); acc[a][b][5 + 2 * d] += hi2f(o[0]);
          acc[a][b][12 + 2 * d] += lo2f(o[1]); acc[a][b][13 + 2 * d] += hi2f(o[1]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint2 rr = *reinterpret_cast<const uint2*>(Rh + (int64_t)mld[b] * ldr + min(nb + 8 * q + hsel, nmax));
          acc[a][b][4 * q + 0] += lo2f(rr.x); acc[a][b][4 * q + 1] += hi2f(rr.x);
          acc[a][b][4 * q + 2] += lo2f(rr.y); acc[a][b][4 * q + 3] += hi2f(rr.y);
        }
      }
      epilogue_fence<FN, FM>();
    }
}

// ---- shared f32 epilogue: lane holds row m = m_base + 32 b + (lane & 31) of row fragment b, and columns
// n = n_base + 32 a + 8 q + 4 (lane >> 5) + {0..3} (quad q) of column fragment a (see the header of gemm.hip) ----------
// Residuals are 16-bit, or f32 under AVSD_GEMM_RES1_F32 / RES2_F32 (the f32 residual stream); with `out_master` the
// un-rounded f32 result is stored next to the 16-bit one.
//
// Structure: the accumulators are updated IN PLACE, one epilogue term at a time over all fragments (scale / LayerNorm
// fold, bias, row vector, GELU, residual 1, residual 2), and only then stored.  Every term's loads are independent of
// each other and no store sits between them, so the compiler issues them back to back and waits once per term — a
// fragment-at-a-time epilogue (load, wait, add, store, next fragment) serialises one L2 round trip per vector behind
// uniform branches and possibly-aliasing stores (it was 40-60 dependent round trips per workgroup; rocprof:
// SQ_WAIT_ANY 60-67 % of the wave cycles of the short-K GEMMs).  Loads are made unconditional by clamping their row /
// column to the last valid one; only the stores are guarded.  The f32 operation order per element is unchanged.
template <int FN, int FM, int EX = EPI_PLAIN>
__device__ __forceinline__ void epilogue_by_term(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], int m_base, int n_base,
                                                 int lane, int64_t bz, const float (&pre_ln)[2 * FM], bool have_pre, int mstride = 32) {
  const int frow = lane & 31;
  const bool geglu = (p.flags & AVSD_GEMM_GEGLU) != 0;
  const bool out_f32 = (p.flags & AVSD_GEMM_OUT_F32) != 0;
  const bool gelu = (p.flags & AVSD_GEMM_GELU) != 0;
  const bool lnfuse = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  const bool rowstats = (p.flags & AVSD_GEMM_ROWSTATS) != 0;
  const bool r1f = (p.flags & AVSD_GEMM_RES1_F32) != 0, r2f = (p.flags & AVSD_GEMM_RES2_F32) != 0;
  const int hsel = (lane >> 5) * 4;
  const int64_t bo = bz * p.batch_stride_out;
  int mrow[FM], mld[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    mrow[b] = m_base + b * mstride + frow;
    mld[b] = min(mrow[b], p.M - 1);           // row used for loads: always valid
  }
  const int nmax = p.N - 4;                   // last valid quad start (N % 4 == 0)
  // 16-bit residuals / output of a fragment fully inside N with 16-byte-aligned rows go through the wide form: the two
  // lanes that share a row (l, l ^ 32) trade halves with v_permlane32_swap so each reads / stores 32 contiguous bytes —
  // two dwordx4 per fragment instead of four dwordx2 scattered 8 bytes apart (store issue, not bandwidth, bounds the tail)
  const bool wide_ok = !geglu && !out_f32 && (p.ldc & 7) == 0 && (!p.res1 || r1f || (p.ldr1 & 7) == 0) &&
                       (!p.res2 || r2f || (p.ldr2 & 7) == 0);

  // ---- 1. alpha, LayerNorm fold: v = rstd * (alpha acc) - mean rstd colsum[n] ---------------------------------------
  if (lnfuse) {
    float rstd[FM], mr[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      if (have_pre) { rstd[b] = pre_ln[2 * b]; mr[b] = pre_ln[2 * b + 1]; }
      else ln_row_stats(p, mld[b], bz, rstd[b], mr[b]);
    }
    const float* lrv[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) lrv[b] = p.ln_rowvec ? pos_row(p, p.ln_rowvec, mld[b]) : nullptr;
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nc = min(n_base + a * 32 + 8 * q + hsel, nmax);
        const float4 cs = *reinterpret_cast<const float4*>(p.ln_colsum + nc);
        const float c4[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);        // LayerNorm(A + pos): W'.pos[frame] joins the product inside the rstd scaling
          if (p.ln_rowvec) pv = *reinterpret_cast<const float4*>(lrv[b] + nc);
          const float p4[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[a][b][4 * q + i] = fmaf(p.alpha * acc[a][b][4 * q + i] + p4[i], rstd[b], -mr[b] * c4[i]);
        }
        if (q == 3) epilogue_fence<FN, FM>();
      }
  } else {
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] *= p.alpha;
  }
  // ---- 2. bias[n] ------------------------------------------------------------------------------------------------------
  if (p.bias) {
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + min(n_base + a * 32 + 8 * q + hsel, nmax));
        const float c4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[a][b][4 * q + i] += c4[i];
        if (q == 3) epilogue_fence<FN, FM>();
      }
  }
  // ---- 3. rowvec[m / rows_per_vec][n] (time embedding) ----------------------------------------------------------------
  if (p.rowvec) {
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const float* rv = p.rowvec + (int64_t)(mld[b] / p.rows_per_vec) * p.ldv;
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bb = *reinterpret_cast<const float4*>(rv + min(n_base + a * 32 + 8 * q + hsel, nmax));
          acc[a][b][4 * q + 0] += bb.x; acc[a][b][4 * q + 1] += bb.y; acc[a][b][4 * q + 2] += bb.z; acc[a][b][4 * q + 3] += bb.w;
          if (q == 3) epilogue_fence<FN, FM>();
        }
    }
  }
  // ---- 4. GELU (not GEGLU: that pairs value and gate at the store) ----------------------------------------------------
  if (gelu) {
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = gelu_erf_f(acc[a][b][r]);
  }
  // ---- 5. residuals -----------------------------------------------------------------------------------------------------
  if (p.res1) epilogue_add_residual<FN, FM>(p, acc, p.res1, p.ldr1, r1f, mld, n_base, hsel, nmax, bo, wide_ok);
  if (p.res2) epilogue_add_residual<FN, FM>(p, acc, p.res2, p.ldr2, r2f, mld, n_base, hsel, nmax, bo, wide_ok);

  // ---- 6. stores --------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = mrow[b];
    if (m >= p.M) continue;
    float2* rs_out = rowstats ? reinterpret_cast<float2*>(p.rowstats) + ((int64_t)bz * p.M + m) * (p.N >> 5) : nullptr;
    int64_t orow = bo + (int64_t)m * p.ldc;
    float* mrow_p = p.out_master ? p.out_master + bo + (int64_t)m * p.ldm : nullptr;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nb = n_base + a * 32;  // first packed column of this fragment
      if (nb >= p.N) continue;
      if constexpr (EX == EPI_SHUF) {       // this fragment's parity moves it `radd` output rows down and to channel `ob` of its pixel
        int radd, ob;
        shuf_col(p, nb, radd, ob);
        const int64_t mo = shuf_row(p, m) + radd;
        orow = bo + mo * p.ldc + (ob - nb);
        mrow_p = p.out_master ? p.out_master + bo + mo * p.ldm + (ob - nb) : nullptr;
      }
      if (geglu) {
        // GEGLU: packed 32-row block = [16 value rows | 16 gate rows]; quads 0,1 hold values, 2,3 their gates.
        float g[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) g[q][i] = acc[a][b][4 * q + i] * gelu_erf_f(acc[a][b][4 * (q + 2) + i]);
        if (!out_f32 && (p.ldc & 7) == 0) {
          // lanes l / l ^ 32 hold output columns {0-3, 8-11} / {4-7, 12-15} of the 16-column block: swap so that each stores
          // 8 contiguous columns (one dwordx4 instead of two dwordx2)
          unsigned e0[2], e1[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto e = __builtin_amdgcn_permlane32_swap(pack2h(g[0][2 * d], g[0][2 * d + 1]), pack2h(g[1][2 * d], g[1][2 * d + 1]), false, false);
            e0[d] = e[0]; e1[d] = e[1];
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + orow + (nb >> 1) + 2 * hsel) = make_uint4(e0[0], e0[1], e1[0], e1[1]);
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int64_t o = orow + (nb >> 1) + hsel + 8 * q;
            if (out_f32) {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(g[q][0], g[q][1], g[q][2], g[q][3]);
            } else {
              uint2 st;
              st.x = pack2h(g[q][0], g[q][1]);
              st.y = pack2h(g[q][2], g[q][3]);
              *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + o) = st;
            }
          }
        }
        continue;
      }
      if (mrow_p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + hsel;
          if (n < p.N) *reinterpret_cast<float4*>(mrow_p + n) = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
      }
      if (wide_ok && nb + 32 <= p.N) {
        unsigned x[2][2], y[2][2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto e = __builtin_amdgcn_permlane32_swap(pack2h(acc[a][b][0 + 2 * d], acc[a][b][1 + 2 * d]), pack2h(acc[a][b][8 + 2 * d], acc[a][b][9 + 2 * d]), false, false);
          const auto o = __builtin_amdgcn_permlane32_swap(pack2h(acc[a][b][4 + 2 * d], acc[a][b][5 + 2 * d]), pack2h(acc[a][b][12 + 2 * d], acc[a][b][13 + 2 * d]), false, false);
          x[0][d] = e[0]; x[1][d] = e[1];
          y[0][d] = o[0]; y[1][d] = o[1];
        }
        h16_t* op = reinterpret_cast<h16_t*>(p.out) + orow + nb + 4 * hsel;
        *reinterpret_cast<uint4*>(op) = make_uint4(x[0][0], x[0][1], x[1][0], x[1][1]);
        *reinterpret_cast<uint4*>(op + 8) = make_uint4(y[0][0], y[0][1], y[1][0], y[1][1]);
        if constexpr (EX != EPI_PLAIN) {
          if (epi_rest<EX>(p)) {       // the rest plane of the same 32 bytes: round16(v - main), traded between the two lanes like the main words
            unsigned xr[2][2], yr[2][2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const unsigned m0 = pack2h(acc[a][b][0 + 2 * d], acc[a][b][1 + 2 * d]), m2 = pack2h(acc[a][b][8 + 2 * d], acc[a][b][9 + 2 * d]);
              const unsigned m1 = pack2h(acc[a][b][4 + 2 * d], acc[a][b][5 + 2 * d]), m3 = pack2h(acc[a][b][12 + 2 * d], acc[a][b][13 + 2 * d]);
              const auto e = __builtin_amdgcn_permlane32_swap(rest2h(acc[a][b][0 + 2 * d], acc[a][b][1 + 2 * d], m0), rest2h(acc[a][b][8 + 2 * d], acc[a][b][9 + 2 * d], m2), false, false);
              const auto o = __builtin_amdgcn_permlane32_swap(rest2h(acc[a][b][4 + 2 * d], acc[a][b][5 + 2 * d], m1), rest2h(acc[a][b][12 + 2 * d], acc[a][b][13 + 2 * d], m3), false, false);
              xr[0][d] = e[0]; xr[1][d] = e[1];
              yr[0][d] = o[0]; yr[1][d] = o[1];
            }
            *reinterpret_cast<uint4*>(op + p.out_lo) = make_uint4(xr[0][0], xr[0][1], xr[1][0], xr[1][1]);
            *reinterpret_cast<uint4*>(op + p.out_lo + 8) = make_uint4(yr[0][0], yr[0][1], yr[1][0], yr[1][1]);
          }
        }
        if (rs_out) {
          // (sum, sum of squares) of the 16 rounded values this lane just stored (+ stats_pos), plus the partner lane's 16
          float sm, sq;
          const unsigned w8[8] = {x[0][0], x[0][1], x[1][0], x[1][1], y[0][0], y[0][1], y[1][0], y[1][1]};
          rowstats16(w8, p.stats_pos ? pos_row(p, p.stats_pos, m) + nb + 4 * hsel : nullptr, sm, sq);
          const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
          const auto u = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
          // lanes 0-31: t = (own, partner's); fixed order low half + high half on both lanes
          if (hsel == 0) rs_out[nb >> 5] = make_float2(__uint_as_float(t[0]) + __uint_as_float(t[1]),
                                                       __uint_as_float(u[0]) + __uint_as_float(u[1]));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + hsel;
          if (n >= p.N) continue;
          if (out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow + n) = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
          } else {
            uint2 st;
            st.x = pack2h(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
            st.y = pack2h(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + orow + n) = st;
            if constexpr (EX != EPI_PLAIN) {
              if (epi_rest<EX>(p)) {
                uint2 sr;
                sr.x = rest2h(acc[a][b][4 * q], acc[a][b][4 * q + 1], st.x);
                sr.y = rest2h(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3], st.y);
                *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + p.out_lo + orow + n) = sr;
              }
            }
          }
        }
      }
    }
  }
}

// ---- fragment-at-a-time form of the epilogue (big tiles): lane holds row m = m_base + (lane&31) of fragment b, and columns
// n = n_base + 32*a + 8q + 4*(lane>>5) + {0..3} of fragment a (see header of this file) ----------------
// Residuals are 16-bit, or f32 under AVSD_GEMM_RES1_F32 / RES2_F32 (the f32 residual stream); with `out_master` the
// un-rounded f32 result is stored next to the 16-bit one.
template <int FN, int FM, int EX = EPI_PLAIN>
__device__ __forceinline__ void epilogue_by_fragment(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], int m_base, int n_base,
                                         int lane, int64_t bz, const float (&pre_ln)[2 * FM], bool have_pre, int mstride = 32) {
  const int frow = lane & 31;
  const bool geglu = (p.flags & AVSD_GEMM_GEGLU) != 0;
  const bool out_f32 = (p.flags & AVSD_GEMM_OUT_F32) != 0;
  const bool gelu = (p.flags & AVSD_GEMM_GELU) != 0;
  const bool lnfuse = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  const bool rowstats = (p.flags & AVSD_GEMM_ROWSTATS) != 0;
  const bool r1f = (p.flags & AVSD_GEMM_RES1_F32) != 0, r2f = (p.flags & AVSD_GEMM_RES2_F32) != 0;
  const h16_t* R1 = reinterpret_cast<const h16_t*>(p.res1);
  const h16_t* R2 = reinterpret_cast<const h16_t*>(p.res2);
  const float* R1f = reinterpret_cast<const float*>(p.res1);
  const float* R2f = reinterpret_cast<const float*>(p.res2);
  const int hsel = (lane >> 5) * 4;
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = m_base + b * mstride + frow;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ldv : nullptr;
    const float* lrv = (lnfuse && p.ln_rowvec) ? pos_row(p, p.ln_rowvec, m) : nullptr;
    // AVSD_GEMM_LNFUSE: LayerNorm statistics of this lane's row of A (rstd, mean * rstd)
    float ln_rstd = 1.f, ln_mr = 0.f;
    if (lnfuse) {
      if (have_pre) { ln_rstd = pre_ln[2 * b]; ln_mr = pre_ln[2 * b + 1]; }
      else ln_row_stats(p, m, bz, ln_rstd, ln_mr);
    }
    float2* rs_out = rowstats ? reinterpret_cast<float2*>(p.rowstats) + ((int64_t)bz * p.M + m) * (p.N >> 5) : nullptr;
    int64_t orow = bz * p.batch_stride_out + (int64_t)m * p.ldc;
    float* mrow = p.out_master ? p.out_master + bz * p.batch_stride_out + (int64_t)m * p.ldm : nullptr;
    // alpha * acc (+ LayerNorm fold) + bias + rowvec (+ GELU) for quad q of fragment (a, b) -> v[0..3]
    auto head = [&](int a, int q, int n, float (&v)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = p.alpha * acc[a][b][4 * q + i];
      if (lnfuse) {
        const float4 cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
        if (lrv) {
          const float4 pv = *reinterpret_cast<const float4*>(lrv + n);
          v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
        }
        v[0] = fmaf(v[0], ln_rstd, -ln_mr * cs.x); v[1] = fmaf(v[1], ln_rstd, -ln_mr * cs.y);
        v[2] = fmaf(v[2], ln_rstd, -ln_mr * cs.z); v[3] = fmaf(v[3], ln_rstd, -ln_mr * cs.w);
      }
      if (p.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (rv) {
        const float4 bb = *reinterpret_cast<const float4*>(rv + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
    };
    auto add_f32 = [&](const float* R, int ldr, int n, float (&v)[4]) {
      const float4 rr = *reinterpret_cast<const float4*>(R + bz * p.batch_stride_out + (int64_t)m * ldr + n);
      v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    };
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nb = n_base + a * 32;  // first packed column of this fragment
      if (nb >= p.N) continue;
      if constexpr (EX == EPI_SHUF) {       // this fragment's parity moves it `radd` output rows down and to channel `ob` of its pixel
        int radd, ob;
        shuf_col(p, nb, radd, ob);
        const int64_t mo = shuf_row(p, m) + radd;
        orow = bz * p.batch_stride_out + mo * p.ldc + (ob - nb);
        mrow = p.out_master ? p.out_master + bz * p.batch_stride_out + mo * p.ldm + (ob - nb) : nullptr;
      }
      // 16-bit output, fragment fully inside N, 16-byte-aligned rows: the two lanes that share a row (l, l ^ 32) trade
      // halves with v_permlane32_swap so each stores (and reads 16-bit residuals as) 32 contiguous bytes — two dwordx4
      // per fragment instead of four dwordx2 scattered 8 bytes apart (store issue, not bandwidth, bounds this tail).
      const bool wide = !geglu && !out_f32 && nb + 32 <= p.N && (p.ldc & 7) == 0 && (!R1 || r1f || (p.ldr1 & 7) == 0) &&
                        (!R2 || r2f || (p.ldr2 & 7) == 0);
      if (wide) {
        float v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          head(a, q, nb + 8 * q + hsel, v[q]);
          if (gelu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[q][i] = gelu_erf_f(v[q][i]);
          }
        }
        const int ncol = nb + 4 * hsel;   // this lane's 16 columns after the swap: nb (lanes 0-31) or nb + 16
        auto add_res = [&](const h16_t* R, int ldr) {
          const h16_t* rp = R + bz * p.batch_stride_out + (int64_t)m * ldr + ncol;
          const uint4 lo = *reinterpret_cast<const uint4*>(rp);
          const uint4 hi = *reinterpret_cast<const uint4*>(rp + 8);
          const unsigned s0[2] = {lo.x, lo.y}, s1[2] = {lo.z, lo.w}, s2[2] = {hi.x, hi.y}, s3[2] = {hi.z, hi.w};
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto e = __builtin_amdgcn_permlane32_swap(s0[d], s1[d], false, false);   // -> quads 0 and 2
            const auto o = __builtin_amdgcn_permlane32_swap(s2[d], s3[d], false, false);   // -> quads 1 and 3
            v[0][2 * d] += lo2f(e[0]); v[0][2 * d + 1] += hi2f(e[0]);
            v[2][2 * d] += lo2f(e[1]); v[2][2 * d + 1] += hi2f(e[1]);
            v[1][2 * d] += lo2f(o[0]); v[1][2 * d + 1] += hi2f(o[0]);
            v[3][2 * d] += lo2f(o[1]); v[3][2 * d + 1] += hi2f(o[1]);
          }
        };
        if (R1) {
          if (r1f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) add_f32(R1f, p.ldr1, nb + 8 * q + hsel, v[q]);
          } else add_res(R1, p.ldr1);
        }
        if (R2) {
          if (r2f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) add_f32(R2f, p.ldr2, nb + 8 * q + hsel, v[q]);
          } else add_res(R2, p.ldr2);
        }
        if (mrow) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(mrow + nb + 8 * q + hsel) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
        }
        unsigned x[2][2], y[2][2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto e = __builtin_amdgcn_permlane32_swap(pack2h(v[0][2 * d], v[0][2 * d + 1]), pack2h(v[2][2 * d], v[2][2 * d + 1]), false, false);
          const auto o = __builtin_amdgcn_permlane32_swap(pack2h(v[1][2 * d], v[1][2 * d + 1]), pack2h(v[3][2 * d], v[3][2 * d + 1]), false, false);
          x[0][d] = e[0]; x[1][d] = e[1];
          y[0][d] = o[0]; y[1][d] = o[1];
        }
        h16_t* op = reinterpret_cast<h16_t*>(p.out) + orow + ncol;
        *reinterpret_cast<uint4*>(op) = make_uint4(x[0][0], x[0][1], x[1][0], x[1][1]);
        *reinterpret_cast<uint4*>(op + 8) = make_uint4(y[0][0], y[0][1], y[1][0], y[1][1]);
        if constexpr (EX != EPI_PLAIN) {
          if (epi_rest<EX>(p)) {       // the rest plane of the same 32 bytes: round16(v - main), traded between the two lanes like the main words
            unsigned xr[2][2], yr[2][2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const auto e = __builtin_amdgcn_permlane32_swap(rest2h(v[0][2 * d], v[0][2 * d + 1], pack2h(v[0][2 * d], v[0][2 * d + 1])),
                                                              rest2h(v[2][2 * d], v[2][2 * d + 1], pack2h(v[2][2 * d], v[2][2 * d + 1])), false, false);
              const auto o = __builtin_amdgcn_permlane32_swap(rest2h(v[1][2 * d], v[1][2 * d + 1], pack2h(v[1][2 * d], v[1][2 * d + 1])),
                                                              rest2h(v[3][2 * d], v[3][2 * d + 1], pack2h(v[3][2 * d], v[3][2 * d + 1])), false, false);
              xr[0][d] = e[0]; xr[1][d] = e[1];
              yr[0][d] = o[0]; yr[1][d] = o[1];
            }
            *reinterpret_cast<uint4*>(op + p.out_lo) = make_uint4(xr[0][0], xr[0][1], xr[1][0], xr[1][1]);
            *reinterpret_cast<uint4*>(op + p.out_lo + 8) = make_uint4(yr[0][0], yr[0][1], yr[1][0], yr[1][1]);
          }
        }
        if (rs_out) {
          // (sum, sum of squares) of the 16 rounded values this lane just stored (+ stats_pos), plus the partner lane's 16
          float sm, sq;
          const unsigned w8[8] = {x[0][0], x[0][1], x[1][0], x[1][1], y[0][0], y[0][1], y[1][0], y[1][1]};
          rowstats16(w8, p.stats_pos ? pos_row(p, p.stats_pos, m) + nb + 4 * hsel : nullptr, sm, sq);
          const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
          const auto u = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
          // lanes 0-31: t = (own, partner's); fixed order low half + high half on both lanes
          if (hsel == 0) rs_out[nb >> 5] = make_float2(__uint_as_float(t[0]) + __uint_as_float(t[1]),
                                                       __uint_as_float(u[0]) + __uint_as_float(u[1]));
        }
      } else if (!geglu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + hsel;
          if (n >= p.N) continue;
          float v[4];
          head(a, q, n, v);
          if (gelu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
          }
          if (R1) {
            if (r1f) add_f32(R1f, p.ldr1, n, v);
            else {
              const uint2 rr = *reinterpret_cast<const uint2*>(R1 + bz * p.batch_stride_out + (int64_t)m * p.ldr1 + n);
              v[0] += lo2f(rr.x); v[1] += hi2f(rr.x); v[2] += lo2f(rr.y); v[3] += hi2f(rr.y);
            }
          }
          if (R2) {
            if (r2f) add_f32(R2f, p.ldr2, n, v);
            else {
              const uint2 rr = *reinterpret_cast<const uint2*>(R2 + bz * p.batch_stride_out + (int64_t)m * p.ldr2 + n);
              v[0] += lo2f(rr.x); v[1] += hi2f(rr.x); v[2] += lo2f(rr.y); v[3] += hi2f(rr.y);
            }
          }
          if (mrow) *reinterpret_cast<float4*>(mrow + n) = make_float4(v[0], v[1], v[2], v[3]);
          if (out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 st;
            st.x = pack2h(v[0], v[1]);
            st.y = pack2h(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + orow + n) = st;
            if constexpr (EX != EPI_PLAIN) {
              if (epi_rest<EX>(p)) {
                uint2 sr;
                sr.x = rest2h(v[0], v[1], st.x);
                sr.y = rest2h(v[2], v[3], st.y);
                *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + p.out_lo + orow + n) = sr;
              }
            }
          }
        }
      } else {
        // GEGLU: packed 32-row block = [16 value rows | 16 gate rows]; quads 0,1 hold values, 2,3 their gates.
        float v[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float gate[4];
          head(a, q, nb + 8 * q + hsel, v[q]);
          head(a, q + 2, nb + 8 * q + hsel + 16, gate);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[q][i] *= gelu_erf_f(gate[i]);
        }
        const int no = (nb >> 1) + hsel;   // output feature of quad 0; quad 1 sits 8 further
        if (!out_f32 && (p.ldc & 7) == 0) {
          // lanes l / l ^ 32 hold output columns {0-3, 8-11} / {4-7, 12-15} of the 16-column block: swap so that each stores
          // 8 contiguous columns (one dwordx4 instead of two dwordx2)
          unsigned e0[2], e1[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto e = __builtin_amdgcn_permlane32_swap(pack2h(v[0][2 * d], v[0][2 * d + 1]), pack2h(v[1][2 * d], v[1][2 * d + 1]), false, false);
            e0[d] = e[0]; e1[d] = e[1];
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + orow + (nb >> 1) + 2 * hsel) = make_uint4(e0[0], e0[1], e1[0], e1[1]);
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int64_t o = orow + no + 8 * q;
            if (out_f32) {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
            } else {
              uint2 st;
              st.x = pack2h(v[q][0], v[q][1]);
              st.y = pack2h(v[q][2], v[q][3]);
              *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + o) = st;
            }
          }
        }
      }
    }
  }
}

// Tiles with up to 4 fragments per wave (<= 64 accumulator registers) take the term-at-a-time form: batched loads, measured
// 7-10 % faster on the K = 320 / 640 linear layers.  Bigger wave tiles hold 128-160 accumulator registers and the batched
// form spills them (4-5x slower, measured): they keep the fragment-at-a-time form, which needs one fragment of temporaries.
// (TIGHT: the kernel runs under a reduced register budget — the loader-wave tiles of 768 threads get 168 registers.)
// mstride: rows between the row fragments of a wave (32: consecutive rows; the 2-D convolution tiles pass the image width —
// fragment b is the tile's b-th image row).
template <int FN, int FM, bool TIGHT = false, int EX = EPI_PLAIN>
__device__ __forceinline__ void epilogue(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], int m_base, int n_base,
                                         int lane, int64_t bz, const float (&pre_ln)[2 * FM], bool have_pre, int mstride = 32) {
  if constexpr (FN * FM <= (TIGHT ? 2 : 4)) epilogue_by_term<FN, FM, EX>(p, acc, m_base, n_base, lane, bz, pre_ln, have_pre, mstride);
  else epilogue_by_fragment<FN, FM, EX>(p, acc, m_base, n_base, lane, bz, pre_ln, have_pre, mstride);
}

// Wave tiles of 8+ fragments whose epilogue the compiler does NOT unroll (hipcc 7.2: "loop not unrolled" on the 5 x 2 and 4 x 2
// fragment forms of conv3r.hip) index the accumulator array at run time: the whole array then lives in scratch memory — 160 registers
// stored behind the main loop and read back piecewise, 1.2 KB per lane and ~360 scratch instructions per wave in the 256 x 160
// resident-convolution tile (tools/resource_usage.py; profiles/r5_resource_usage.txt).  This form hands the shared epilogue ONE fragment
// at a time with compile-time indices, so every accumulator is consumed from the register it was computed in.  Same terms in the same
// order per element; the LayerNorm-fold row statistics are folded once per row band and handed to the fragments.
template <int FN, int FM, bool TIGHT = false, int EX = EPI_PLAIN>
__device__ __forceinline__ void epilogue_each(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], int m_base, int n_base, int lane, int64_t bz,
                                              const float (&pre_ln)[2 * FM], bool have_pre, int mstride = 32) {
  const bool lnfuse = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  static_for<FM>([&](auto b_c) {
    constexpr int B = decltype(b_c)::value;
    // LayerNorm fold: the row statistics of this lane's row, folded ONCE per row band
    float pl[2] = {1.f, 0.f};
    if (lnfuse) {
      if (have_pre) { pl[0] = pre_ln[2 * B]; pl[1] = pre_ln[2 * B + 1]; }
      else ln_row_stats(p, min(m_base + B * mstride + (lane & 31), p.M - 1), bz, pl[0], pl[1]);
    }
    // one fragment at a time (a whole band in the term-at-a-time form was tried: with 160 accumulators live beside its batched loads the
    // 256 x 160 convolution tile spilled 1552 registers)
    static_for<FN>([&](auto a_c) {
      constexpr int A = decltype(a_c)::value;
      __builtin_amdgcn_sched_barrier(0);          // fragments are independent: keep each one's loads and stores together
      f32x16 one[1][1];
      one[0][0] = acc[A][B];
      epilogue<1, 1, TIGHT, EX>(p, one, m_base + B * mstride, n_base + 32 * A, lane, bz, pl, lnfuse, mstride);
    });
  });
}

// ---- AVSD_GEMM_X2 epilogue: same terms and f32 order as above; 16-bit residuals are read as main + rest, the result is
// written as main = round16(v), rest = round16(v - main) (+ the un-rounded f32 value to `out_master`: the three-pass products of
// the per-layer precision plan feed a residual add AND another three-pass product); LayerNorm row statistics are taken from
// main + rest (what the consumer reconstructs).  SHUF: the per-pixel output scatter of AVSD_GEMM_CONV3 with ups = 2 (see EPI_SHUF).  Fragment-at-a-time, 8-byte stores (the precise tier trades the wide-store form for one code path).
template <int FN, int FM, bool SHUF = false>
__device__ __forceinline__ void epilogue_x2(const avsd_gemm_desc& p, f32x16 (&acc)[FN][FM], int m_base, int n_base,
                                            int lane, int64_t bz, const float (&pre_ln)[2 * FM], bool have_pre) {
  const int frow = lane & 31;
  const bool geglu = (p.flags & AVSD_GEMM_GEGLU) != 0;
  const bool out_f32 = (p.flags & AVSD_GEMM_OUT_F32) != 0;
  const bool gelu = (p.flags & AVSD_GEMM_GELU) != 0;
  const bool lnfuse = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  const bool rowstats = (p.flags & AVSD_GEMM_ROWSTATS) != 0;
  const bool r1f = (p.flags & AVSD_GEMM_RES1_F32) != 0, r2f = (p.flags & AVSD_GEMM_RES2_F32) != 0;
  const h16_t* R1 = reinterpret_cast<const h16_t*>(p.res1);
  const h16_t* R2 = reinterpret_cast<const h16_t*>(p.res2);
  h16_t* O = reinterpret_cast<h16_t*>(p.out);
  const int hsel = (lane >> 5) * 4;
  const int64_t bo = bz * p.batch_stride_out;
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = m_base + b * 32 + frow;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ldv : nullptr;
    float ln_rstd = 1.f, ln_mr = 0.f;
    if (lnfuse) {
      if (have_pre) { ln_rstd = pre_ln[2 * b]; ln_mr = pre_ln[2 * b + 1]; }
      else ln_row_stats(p, m, bz, ln_rstd, ln_mr);
    }
    float2* rs_out = rowstats ? reinterpret_cast<float2*>(p.rowstats) + ((int64_t)bz * p.M + m) * (p.N >> 5) : nullptr;
    int64_t orow = bo + (int64_t)m * p.ldc;
    float* mrow = p.out_master ? p.out_master + bo + (int64_t)m * p.ldm : nullptr;
    auto head = [&](int a, int q, int n, float (&v)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = p.alpha * acc[a][b][4 * q + i];
      if (lnfuse) {
        const float4 cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
        v[0] = fmaf(v[0], ln_rstd, -ln_mr * cs.x); v[1] = fmaf(v[1], ln_rstd, -ln_mr * cs.y);
        v[2] = fmaf(v[2], ln_rstd, -ln_mr * cs.z); v[3] = fmaf(v[3], ln_rstd, -ln_mr * cs.w);
      }
      if (p.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (rv) {
        const float4 bb = *reinterpret_cast<const float4*>(rv + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
    };
    auto add_res = [&](const h16_t* R, int64_t rlo, int ldr, bool f32, int n, float (&v)[4]) {
      if (f32) {
        const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(R) + bo + (int64_t)m * ldr + n);
        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
      } else {
        const h16_t* rp = R + bo + (int64_t)m * ldr + n;
        const uint2 a = *reinterpret_cast<const uint2*>(rp);
        const uint2 r = *reinterpret_cast<const uint2*>(rp + rlo);
        v[0] += lo2f(a.x) + lo2f(r.x); v[1] += hi2f(a.x) + hi2f(r.x);
        v[2] += lo2f(a.y) + lo2f(r.y); v[3] += hi2f(a.y) + hi2f(r.y);
      }
    };
    // main / rest words of 4 values -> the two planes; returns (sum, sum of squares) of the values as the consumer sees them
    auto store4 = [&](int64_t o, const float (&v)[4], float& sm, float& sq) {
      uint2 st, sr;
      split2(v[0], v[1], st.x, sr.x);
      split2(v[2], v[3], st.y, sr.y);
      *reinterpret_cast<uint2*>(O + o) = st;
      *reinterpret_cast<uint2*>(O + p.out_lo + o) = sr;
      const float w[4] = {lo2f(st.x) + lo2f(sr.x), hi2f(st.x) + hi2f(sr.x), lo2f(st.y) + lo2f(sr.y), hi2f(st.y) + hi2f(sr.y)};
#pragma unroll
      for (int i = 0; i < 4; ++i) { sm += w[i]; sq = fmaf(w[i], w[i], sq); }
    };
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nb = n_base + a * 32;
      if (nb >= p.N) continue;
      if constexpr (SHUF) {
        int radd, ob;
        shuf_col(p, nb, radd, ob);
        const int64_t mo = shuf_row(p, m) + radd;
        orow = bo + mo * p.ldc + (ob - nb);
        mrow = p.out_master ? p.out_master + bo + mo * p.ldm + (ob - nb) : nullptr;
      }
      if (!geglu) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + hsel;
          if (n >= p.N) continue;
          float v[4];
          head(a, q, n, v);
          if (gelu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
          }
          if (R1) add_res(R1, p.res1_lo, p.ldr1, r1f, n, v);
          if (R2) add_res(R2, p.res2_lo, p.ldr2, r2f, n, v);
          if (mrow) *reinterpret_cast<float4*>(mrow + n) = make_float4(v[0], v[1], v[2], v[3]);
          if (out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow + n) = make_float4(v[0], v[1], v[2], v[3]);
          else store4(orow + n, v, sm, sq);
        }
        if (rs_out) {        // ROWSTATS: N % 32 == 0, so the fragment is whole; this lane's 16 values + the partner lane's 16
          const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
          const auto u = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
          if (hsel == 0) rs_out[nb >> 5] = make_float2(__uint_as_float(t[0]) + __uint_as_float(t[1]),
                                                       __uint_as_float(u[0]) + __uint_as_float(u[1]));
        }
      } else {
        // GEGLU: packed 32-row block = [16 value rows | 16 gate rows]; quads 0,1 hold values, 2,3 their gates
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[4], gate[4];
          head(a, q, nb + 8 * q + hsel, v);
          head(a, q + 2, nb + 8 * q + hsel + 16, gate);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] *= gelu_erf_f(gate[i]);
          const int64_t o = orow + (nb >> 1) + hsel + 8 * q;
          float sm = 0.f, sq = 0.f;
          if (out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
          else store4(o, v, sm, sq);
        }
      }
    }
  }
}

// s_waitcnt vmcnt(N) with N a compile-time constant (0..63), everything else unconstrained.  gfx9 encoding of the
// immediate: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14, expcnt (7 = no wait) in 6:4, lgkmcnt (15) in 11:8.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
  asm volatile("" ::: "memory");
}

// waits until at most min(MAXA, ahead) tiles of LPT loads each are still in flight (ahead is wave-uniform)
template <int MAXA, int LPT>
__device__ __forceinline__ void wait_tiles_ahead(int ahead) {
  if constexpr (MAXA <= 0) {
    wait_vmcnt<0>();
  } else {
    if (ahead >= MAXA) wait_vmcnt<MAXA * LPT>();
    else wait_tiles_ahead<MAXA - 1, LPT>(ahead);
  }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Work item `wg` (XCD-contiguous: each of the 8 XCDs owns one contiguous range) -> output tile (tm, tn).  Bands of the major
// dimension (M by default, N under AVSD_GEMM_XCD_N: the operand of the band is what the XCD's L2 fetches once); on big grids the
// ~32 workgroups an XCD runs at a time cover a G x (32 / G) BLOCK of tiles instead of one row of 32, so the L2 fetches
// G + 32 / G operand panels per K tile instead of 33 (measured on the asm tiles: 8192^3 1090 -> 850 us, 6144 x 5120 x 640
// 64 -> 54 us; G = 4 .. 8 equal, profiles/r4_raster_probe.txt).  `g_override` > 0 replaces G (probe knob, avsd_gemm_desc.raster_g: 0..64, checked by avsd_gemm_bf16).
__device__ __forceinline__ void tile_of_item(int wg, int ntm, int ntn, bool nmaj, int g_override, int& tm, int& tn) {
  const int nmajor = nmaj ? ntn : ntm, nminor = nmaj ? ntm : ntn;      // wg = major * nminor + minor
  int tmaj, tmin;
  if (nminor >= 16 && nmajor >= 8) {
    const int G = g_override > 0 ? g_override : 8;
    const int gsize = G * nminor, grp = wg / gsize, first = grp * G, in = wg - grp * gsize;
    const int gm = min(nmajor - first, G);
    tmaj = first + in % gm;
    tmin = in / gm;
  } else {
    tmaj = wg / nminor;
    tmin = wg - tmaj * nminor;
  }
  tn = nmaj ? tmaj : tmin;
  tm = nmaj ? tmin : tmaj;
}


// ---- LDS-direct K tiles (gemm2_kernel and the fused kernels) ----------------------------------------------------------
// Tile image: rows of 64 16-bit values = 128 B, two rows = one 256-B line L; the 16-byte slot index inside a line is
// XOR-ed with (L & 15).  A 1-KiB piece = 4 lines, written lane-linearly by one buffer_load_dwordx4 ... lds: lane l
// lands in line L = 4 * piece + l / 16, slot l % 16, and therefore fetches row 2L + (x >> 3), k-chunk (x & 7) * 8 with
// x = (l % 16) ^ (L & 15).  Fragment reads undo the permutation: row r, 16-byte chunk c -> line r >> 1,
// slot ((r & 1) << 3 | c) ^ ((r >> 1) & 15).
__device__ __forceinline__ void piece_row_chunk(int piece, int lane, int& row, int& kchunk) {
  const int L = piece * 4 + (lane >> 4);
  const int x = (lane & 15) ^ (L & 15);
  row = 2 * L + (x >> 3);
  kchunk = (x & 7) * 8;
}
__device__ __forceinline__ int frag_offset(int r, int c) {   // byte offset of (row r, 16-byte chunk c) inside a tile image
  return (r >> 1) * 256 + ((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) * 16;
}

}  // namespace
