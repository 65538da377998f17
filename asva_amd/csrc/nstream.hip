// A-resident, N-streaming GEMM for the wide short-K projections (gfx950): tile id AVSD_GEMM_TILE_NSTREAM of avsd_gemm_bf16.
//
// The GEGLU projection of a transformer block (ff_spatio_audio_temp_transformer_3d.py:276, 361-371) is 24576 x 2560 x 320 at the
// 32 x 32 level: 40 GFLOP with K = 5 tiles of 64.  As a tiled GEMM it is 960-3840 workgroups that each load an A tile and a W tile,
// run 5-10 K tiles and drain through an erf-GELU epilogue — prologue and epilogue in series with a main loop too short to hide
// either (79-95 us on every tile of both families, the library 52-59 us for the plain product).  Here the loop nest
// is turned around:
//   * a workgroup owns a band of BM = 32 RF rows and keeps ALL of its A (BM x K, 61 KB at K = 320, 121 KB at K = 640) in LDS for
//     the whole launch — A is read from memory once per launch (once per N split);
//   * its 8 waves are INDEPENDENT after that one barrier: wave w walks the 32-column fragments w, w + 8, ... of the workgroup's N
//     range, for each of them the full K, with RF accumulator fragments (all rows of the band);
//   * W never touches LDS: it is stored in MFMA-fragment order (AVSD_GEMM_W_FRAG: [N / 32][K / 16][64 lanes][8 values], packed once
//     with the weights), so a wave's operand of one k-step is ONE coalesced 1-KB load straight into the registers the MFMA reads,
//     prefetched D k-steps ahead through a register ring that runs on across fragment boundaries — no prologue per fragment;
//   * two waves share a SIMD and drift apart on their own (nothing staggers them): while one drains its
//     accumulators through the epilogue (LayerNorm fold, bias, GEGLU's erf, 16-bit stores — VALU and memory), the other owns the
//     MFMA pipe.  No barrier, no LDS traffic besides the A fragment reads (3 ds_read_b128 per 3 MFMAs per wave).
// The epilogue is the shared one (gemm_common.h): same terms, same f32 order per element as every other tile; K is summed in
// ascending order, so results are bit-identical to the LDS-direct tiles on their UNROTATED walk (this tile ignores AVSD_GEMM_KROT).  The LayerNorm statistics of a wave's rows are folded once
// per launch (the rows never change), from the producer's K / 32 partial pairs — no avsd_ln_fold launch in front of it.
// PLAIN single-source descriptors, K = 320 or 640, N % 32 == 0, no split-K, no AVSD_GEMM_X2.  Wave w owns fragments w, w + 8, ...: with
// fewer than 8 fragments (N < 256) or N / 32 not a multiple of 8 some waves simply run fewer (or no) fragments — correct, just not what the
// tile is for; the host rule (ops.nstream_supported) only picks it at N >= 4 K, N % 256 == 0.
#include "gemm_common.h"

#ifdef NS_STAMPS     // probe build (tools/nstream_probe.py --stamps): cycle stamps of 4 workgroups x 8 waves x (start, per fragment: loop start / loop end / epilogue end)
__device__ unsigned long long ns_stamps[4 * 8 * 64];
extern "C" int avsd_nstream_debug_read(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ns_stamps), sizeof(ns_stamps)); }
#define NS_STAMP(i) do { if (sb >= 0 && lane == 0) ns_stamps[(sb * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NS_STAMP(i) do { } while (0)
#endif

namespace {

typedef unsigned u32x4n __attribute__((ext_vector_type(4)));

// FAST: the GEGLU projection's own epilogue, written out (LayerNorm fold + bias + value * gelu(gate), 16-bit stores): the same f32
// operations per element as epilogue_by_term, with the per-column operands (colsum, bias) requested BEFORE the fragment's K loop so the
// epilogue waits for nothing.  Every other flag combination runs the shared epilogue one fragment at a time (the generic instantiation).
// (Parking a wave's results in an LDS slab and writing whole 128-byte rows — runs of 4 adjacent fragments per wave — was built and measured
// slower, 70.8 -> 75.2 us at C = 320: the extra LDS round trip costs more issue slots than the narrower stores)
template <int K, int RF, bool FAST>
__global__ __launch_bounds__(512, 1) void nstream_kernel(const avsd_gemm_desc p, const int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smn[];
  constexpr int KS = K / 16;                 // k-steps (one 32x32x16 MFMA per row fragment each)
  constexpr int PITCH = K * 2 + 16;          // LDS row pitch: 164 / 324 dwords = 36 / 4 mod 64 -> ds_read_b128 conflict-free
  constexpr int BM = 32 * RF;
  constexpr int D = 10;                      // k-steps of W in flight per wave (divides KS = 20 / 40)
  constexpr int AD = 2;                      // A fragments are read AD k-steps ahead (AD + 1 register sets)
  static_assert(KS % D == 0, "the ring index must not depend on the fragment");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int band = blockIdx.x / nsplit, sp = blockIdx.x - band * nsplit;
  const int m0 = band * BM;
#ifdef NS_STAMPS
  const int sb = blockIdx.x == 0 ? 0 : blockIdx.x == 77 ? 1 : blockIdx.x == 150 ? 2 : blockIdx.x == 255 ? 3 : -1;
#endif
  NS_STAMP(0);

  // LayerNorm fold, first half: the K / 32 (or 1) partial pairs of this lane's RF rows are requested together, ahead of the A band
  // (ln_row_stats would take one round trip per row fragment, in series: 6000 of the prologue's 9000 cycles)
  constexpr int NPAIR = K / 32;              // (even: 10 / 20 pairs = 5 / 10 16-byte vectors per row)
  float4 lnq[RF][NPAIR / 2];
  const bool lnfuse = (p.flags & AVSD_GEMM_LNFUSE) != 0, ln1 = p.ln_nblk == 1;
  if (lnfuse) {
#pragma unroll
    for (int b = 0; b < RF; ++b) {
      const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (int64_t)min(m0 + 32 * b + (lane & 31), p.M - 1) * p.ln_nblk;
      if (ln1) {
        const float2 t = st[0];
        lnq[b][0] = make_float4(t.x, t.y, 0.f, 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < NPAIR / 2; ++j) lnq[b][j] = reinterpret_cast<const float4*>(st)[j];
      }
    }
  }
  // ---- this wave's fragments -------------------------------------------------------------------------------------------------
  // wave w takes the 32-column fragments w, w + 8, ... of the workgroup's N range
  const int nfr_wg = (p.N / 32) / nsplit;           // fragments of this workgroup's N range
  const int T = wave < nfr_wg ? (nfr_wg - wave + 7) / 8 : 0;       // ... of this wave (a range that is no multiple of 8 leaves the last waves one short)
  const int fbase = sp * nfr_wg;
  auto frag_of = [&](int t) -> int { return fbase + min(wave + 8 * t, nfr_wg - 1); };
  const u32x4n* wbase = reinterpret_cast<const u32x4n*>(p.W) + lane;
  const unsigned char* arow = smn + (lane & 31) * PITCH + (lane >> 5) * 16;
  const int hsel = (lane >> 5) * 4;

  // the first D k-steps of W are requested before the A band: they land while it is staged
  u32x4n wq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) wq[d] = __builtin_nontemporal_load(wbase + ((int64_t)frag_of(0) * KS + d) * 64);
  // ---- A band -> LDS, once -------------------------------------------------------------------------------------------------
  {
    constexpr int V = K / 8;                 // 16-byte vectors per row
    const h16_t* A = reinterpret_cast<const h16_t*>(p.A);
    constexpr int NV = (BM * V + 511) / 512;  // vectors per thread: all requested before the first is written (one round trip)
    uint4 x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = min(tid + i * 512, BM * V - 1), r = v / V, c = v - r * V;
      x[i] = *reinterpret_cast<const uint4*>(A + (int64_t)min(m0 + r, p.M - 1) * p.lda + c * 8);      // (rows past M: a copy of the last row, never stored)
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * 512, r = v / V, c = v - r * V;
      if (v < BM * V) *reinterpret_cast<uint4*>(smn + r * PITCH + c * 16) = x[i];
    }
  }
  // ... second half: (rstd, mean * rstd), the arithmetic of ln_row_stats (gemm_common.h) in the same order
  float pre_ln[2 * RF];
#pragma unroll
  for (int b = 0; b < RF; ++b) {
    pre_ln[2 * b] = 1.f; pre_ln[2 * b + 1] = 0.f;
    if (lnfuse) {
      float sm = 0.f, sq = 0.f;
      if (ln1) { sm += lnq[b][0].x; sq += lnq[b][0].y; }
      else {
#pragma unroll
        for (int j = 0; j < NPAIR / 2; ++j) { sm += lnq[b][j].x; sq += lnq[b][j].y; sm += lnq[b][j].z; sq += lnq[b][j].w; }
      }
      const float inv_k = 1.0f / (float)p.K;
      const float mean = sm * inv_k;
      const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
      pre_ln[2 * b] = rsqrtf(var + p.ln_eps);
      pre_ln[2 * b + 1] = mean * pre_ln[2 * b];
    }
  }
  __syncthreads();
  NS_STAMP(1);

  for (int t = 0; t < T; ++t) {
    const int f = frag_of(t);
    const int fnext = t + 1 < T ? frag_of(t + 1) : f;        // past the end: re-read this fragment (never consumed)
    const u32x4n* wcur = wbase + (int64_t)f * KS * 64;
    const u32x4n* wnxt = wbase + (int64_t)fnext * KS * 64;
    f32x16 acc[1][RF];
#pragma unroll
    for (int b = 0; b < RF; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
    float4 cs4[4], bi4[4];
    if constexpr (FAST) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cs4[q] = *reinterpret_cast<const float4*>(p.ln_colsum + f * 32 + 8 * q + hsel);
        bi4[q] = *reinterpret_cast<const float4*>(p.bias + f * 32 + 8 * q + hsel);
      }
    }
    // the scheduling barriers keep hipcc from hoisting all 3 KS fragment reads of the unrolled loop to its top (it did: 235 spilled registers)
    h16x8 af[AD + 1][RF];
#pragma unroll
    for (int j = 0; j < AD; ++j)
#pragma unroll
      for (int b = 0; b < RF; ++b) af[j][b] = *reinterpret_cast<const h16x8*>(arow + b * 32 * PITCH + j * 32);
    NS_STAMP(2 + 3 * t);
    __builtin_amdgcn_s_setprio(1);         // the wave in its MFMA phase goes first
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      const h16x8 wf = __builtin_bit_cast(h16x8, wq[ks % D]);
      if (ks + AD < KS) {
#pragma unroll
        for (int b = 0; b < RF; ++b) af[(ks + AD) % (AD + 1)][b] = *reinterpret_cast<const h16x8*>(arow + b * 32 * PITCH + (ks + AD) * 32);
      }
#pragma unroll
      for (int b = 0; b < RF; ++b) acc[0][b] = mfma32x32x16(wf, af[ks % (AD + 1)][b], acc[0][b], 0, 0, 0);
      wq[ks % D] = __builtin_nontemporal_load(ks + D < KS ? wcur + (ks + D) * 64 : wnxt + (ks + D - KS) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    NS_STAMP(3 + 3 * t);
    if constexpr (FAST) {
#pragma unroll
      for (int b = 0; b < RF; ++b) {
        const int row = 32 * b + (lane & 31);
        const float rstd = pre_ln[2 * b], mr = pre_ln[2 * b + 1];
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float c4[4] = {cs4[q].x, cs4[q].y, cs4[q].z, cs4[q].w}, b4[4] = {bi4[q].x, bi4[q].y, bi4[q].z, bi4[q].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) v[4 * q + i] = fmaf(p.alpha * acc[0][b][4 * q + i] + 0.f, rstd, -mr * c4[i]) + b4[i];
        }
        // GEGLU: quads 0, 1 hold 8 values, quads 2, 3 their gates; lanes l / l ^ 32 swap halves so each stores 8 contiguous columns
        unsigned e0[2], e1[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const float g00 = v[2 * d] * gelu_erf_f(v[8 + 2 * d]), g01 = v[2 * d + 1] * gelu_erf_f(v[8 + 2 * d + 1]);
          const float g10 = v[4 + 2 * d] * gelu_erf_f(v[12 + 2 * d]), g11 = v[4 + 2 * d + 1] * gelu_erf_f(v[12 + 2 * d + 1]);
          const auto e = __builtin_amdgcn_permlane32_swap(pack2h(g00, g01), pack2h(g10, g11), false, false);
          e0[d] = e[0]; e1[d] = e[1];
        }
        const uint4 piece = make_uint4(e0[0], e0[1], e1[0], e1[1]);          // out columns 16 f + 2 hsel ... + 8
        if (m0 + row < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(p.out) + (int64_t)(m0 + row) * p.ldc + f * 16 + 2 * hsel) = piece;
      }
    } else {
      // (opaque copies: everything the epilogue derives from the band's rows is loop-invariant here, and hipcc hoists it all out of the
      //  fragment loop — row pointers, frame divisions, 64-bit addresses of every optional operand — 231 registers spilled across the loop)
      int m0v = m0, lanev = lane;
      asm volatile("" : "+s"(m0v), "+v"(lanev));
      epilogue_each<1, RF>(p, acc, m0v, f * 32, lanev, 0, pre_ln, true);
    }
    NS_STAMP(4 + 3 * t);
  }
}

template <int K, int RF, bool FAST>
int launch_nstream(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 32 * RF;
  constexpr size_t lds = (size_t)BM * (K * 2 + 16);
  static_assert(lds <= 160 * 1024, "the A band does not fit the 160 KB of LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nstream_kernel<K, RF, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm/nstream: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int nbands = (d.M + BM - 1) / BM;
  // N splits: enough workgroups for the 256 CUs, every wave at least one fragment
  const int nfr = d.N / 32;
  int nsplit = 1;
  for (int c : {1, 2, 3, 4, 5, 6, 8, 10}) {          // the smallest split of N that fills the 256 CUs and leaves every wave of a workgroup a fragment
    if (nfr % c != 0 || nfr / c < 8) continue;
    nsplit = c;
    if (nbands * c >= 256) break;
  }
  hipLaunchKernelGGL((nstream_kernel<K, RF, FAST>), dim3((unsigned)(nbands * nsplit)), dim3(512), lds, s, d, nsplit);
  AVSD_CHECK_LAUNCH("gemm/nstream launch");
  return AVSD_OK;
}

}  // namespace

int avsd_gemm_dispatch_nstream(const avsd_gemm_desc& d, hipStream_t s) {
  AVSD_REQUIRE(d.mode == AVSD_GEMM_PLAIN && !d.A2 && d.batch == 1 && d.split_k <= 1 && !(d.flags & AVSD_GEMM_X2),
               "gemm/nstream: PLAIN single-source descriptors, no batching, no split-K, no AVSD_GEMM_X2");
  AVSD_REQUIRE(d.flags & AVSD_GEMM_W_FRAG, "gemm/nstream: W must be in MFMA-fragment order (AVSD_GEMM_W_FRAG)");
  AVSD_REQUIRE((d.K == 320 || d.K == 640) && d.N % 32 == 0, "gemm/nstream: K = 320 or 640 and N %% 32 == 0 (got K %d, N %d)", d.K, d.N);
  AVSD_REQUIRE(d.lda % 8 == 0 && !d.stats_pos && !d.ln_rowvec, "gemm/nstream: lda %% 8 == 0, no position tables");
  // the GEGLU projection of a transformer block and nothing else in its epilogue: the written-out form
  const bool fast = (d.flags & AVSD_GEMM_GEGLU) && (d.flags & AVSD_GEMM_LNFUSE) && d.bias && !d.rowvec && !d.res1 && !d.res2 && !d.out_master &&
                    !(d.flags & (AVSD_GEMM_OUT_F32 | AVSD_GEMM_GELU | AVSD_GEMM_ROWSTATS)) && d.ldc % 8 == 0;
  if (d.K == 320) return fast ? launch_nstream<320, 3, true>(d, s) : launch_nstream<320, 3, false>(d, s);
  return fast ? launch_nstream<640, 3, true>(d, s) : launch_nstream<640, 3, false>(d, s);
}
