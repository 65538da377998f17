// Big-tile bf16 GEMM with a HAND-SCHEDULED main loop (gfx950): 4 waves, one per SIMD, each owning a (32 FM) x (32 FN) block of
// the output in accumulator registers (256 AGPRs at FM = FN = 4: a 256 x 256 tile per workgroup), operands staged
// global -> VGPR -> LDS -> fragment registers.  Tile ids AVSD_GEMM_TILE_ASM_FIRST.. of avsd_gemm_bf16.
//
// Why another main loop: the LDS-direct tiles of gemm.hip let hipcc schedule the K loop.  With one wave per SIMD (what a
// 128 x 128 accumulator block costs in registers) nothing hides a wave's own issue stalls — the compiler clusters the loads of
// a K tile, the fragment reads and the MFMAs into phases, and a ~60-180-cycle LDS-DMA issue sits in front of matrix work
// (0.9-1.08 PFLOP/s at 8192^3 in every compiler-scheduled form).  Here the loop body is a fixed sequence of
// `asm volatile` statements, one instruction each, in issue order: every 32-cycle MFMA carries at most one memory instruction
// pair in its shadow (one ds_read_b128, or one ds_write_b128 + one buffer_load_dwordx4), waits are counted (two K tiles of
// global loads stay in flight across the barrier), and there is ONE s_barrier per K tile.  The compiler allocates the
// registers (operands of the statements) and generates the address arithmetic between them; it neither reorders nor waits.
//
// Pipeline per K tile t (tile t lives in LDS stage t & 1, fragments of k-step s in register set s & 1):
//   k-steps 0..2: FM FN MFMAs each; the first FM + FN slots read the fragments of the next k-step, the following slots write
//                 tile t+1 from staging registers G[(t+1) & 1] to the other LDS stage and re-issue those registers' global
//                 loads for tile t+3
//   barrier       (all fragment reads of tile t and all writes of tile t+1 are done)
//   k-step 3:     FM FN MFMAs; the first slots read k-step 0 of tile t+1
// Tiles past the end of K re-read the last tile (never consumed), so the loop has no tail code.
//
// Same math, operand layout, XCD banding, split-K slabs and epilogue as gemm2_kernel (gemm.hip); f32 order per element: K ascending
// -> bit-identical to the LDS-direct tiles ON THE UNROTATED WALK ONLY (AVSD_GEMM_KROT clear; AVSD_KROT=0 on the host).  With the rotated
// walk — the host's default wherever 2 N K <= 16 MB — row band tm starts at K tile (tm nk) / ntm: the summation order of an output
// element then depends on the tile height BM, on M (how many clips share the launch) and on the table's tile pick, so results agree
// across those only to f32 rounding.  They stay bit-repeatable run to run and rank to rank (same table, same shapes: the witness-clip
// checksum of bench.py / asva_amd/dist.py checks exactly that).
// Replaces (reference file:line): as gemm.hip — nn.Linear / 1x1 nn.Conv2d at avgen/models/unets/utils.py:123-131,159;
// ff_spatio_audio_temp_transformer_3d.py:66,92,276,361-371 (the GEGLU projection is the widest GEMM of a step).
#include <utility>

#include "gemm_common.h"

int avsd_gemm_splitk_reduce(const avsd_gemm_desc& d, hipStream_t s);   // gemm.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 144;          // LDS row pitch: 64 values + 8 pad = 144 B -> ds_read_b128 / ds_write_b128 conflict-free

#ifdef AVSD_F16
#define AVSD_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define AVSD_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif

#include "gemm4_loops.inc"

// The loop leaves the accumulators in AGPRs ("=&a" outputs).  Read through the compiler, all of them are copied to VGPRs right behind
// the loop (the value's register class is decided at its definition), which the 256 x 256 tile cannot hold: 100-132 registers went to
// scratch, 25 us per tile.  Read through this instead, a fragment leaves the AGPRs where the epilogue consumes it.
__device__ __forceinline__ f32x16 from_agpr(const f32x16& a) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float f;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(f) : "a"(a[r]));
    v[r] = f;
  }
  return v;
}

// X2 (AVSD_GEMM_X2, split precision): every operand a (main, rest) pair of planes — an LDS stage holds [A | W | A rest | W rest], the rest
// planes come through second buffer descriptors with the same offsets, every fragment pair takes three MFMAs (Wr.A, W.Ar, W.A).
// NS = K tiles of global loads in flight per workgroup (staging register sets of the generated loop): 2 in every shipped tile
// EX (gemm_common.h): EPI_REST = this kernel also stores the rest plane of its 16-bit output (AVSD_GEMM_OUT_REST; IEEE-half build only)
template <int FM, int FN, int MODE, bool X2 = false, int NS = 2, int EX = EPI_PLAIN>
__global__ __launch_bounds__(256, 1) void gemm4_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem4[];
  constexpr int BM = 64 * FM, BN = 64 * FN;
  constexpr int A_BYTES = BM * ROWB;
  static_assert(MODE == AVSD_GEMM_PLAIN || MODE == AVSD_GEMM_TMIX, "gemm4: PLAIN or TMIX operands");
  constexpr int NA = BM / 32;
  constexpr int STAGE = (BM + BN) * ROWB * (X2 ? 2 : 1);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  const int ntm = (p.M + BM - 1) / BM;
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  const int nsplit = p.split_k > 1 ? p.split_k : 1;
  int wg, ksplit;
  {   // XCD-contiguous (tile, K-slice) work items, as gemm2_kernel
    const int total = nwg * nsplit;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    wg = c / nsplit;
    ksplit = c - wg * nsplit;
  }
  int tm, tn;
  tile_of_item(wg, ntm, ntn, (p.flags & AVSD_GEMM_XCD_N) != 0, p.raster_g, tm, tn);
  const int nk_all = p.K / BK;
  const int per_split = (nk_all + nsplit - 1) / nsplit;
  const int kt0 = ksplit * per_split;
  const int nk = max(min(nk_all, kt0 + per_split) - kt0, 0);

  // buffer descriptors (scalar registers): num_records = the rows that exist, so rows past M / N read as zeros
  // (TMIX: every source row of a valid output row is a valid row; rows past M get an offset past num_records below)
  const unsigned long long pa = (unsigned long long)p.A, pw = (unsigned long long)p.W;
  const u32x4 rsA = {(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, (unsigned)p.M * (unsigned)p.lda * 2u, 0x00020000u};
  const u32x4 rsW = {(unsigned)pw, (unsigned)(pw >> 32) & 0xffffu, (unsigned)p.N * (unsigned)p.ldw * 2u, 0x00020000u};
  const unsigned long long par = pa + (X2 ? (unsigned long long)p.a_lo * 2ull : 0ull), pwr = pw + (X2 ? (unsigned long long)p.w_lo * 2ull : 0ull);
  const u32x4 rsAr = {(unsigned)par, (unsigned)(par >> 32) & 0xffffu, rsA[2], 0x00020000u};
  const u32x4 rsWr = {(unsigned)pwr, (unsigned)(pwr >> 32) & 0xffffu, rsW[2], 0x00020000u};

  // Rotated K walk (AVSD_GEMM_KROT): workgroups that share a column band of W (same tn, different tm) start at different K tiles of the
  // slice and wrap around.  Walking K in lockstep, all of them wait for the SAME weight tile at the same time — inside a denoising
  // step the weights come from HBM (2.3 GB stream through the caches between two uses), so the launch runs as one chain of HBM round
  // trips with only (column bands x prefetch depth) distinct tiles in flight; rotated, every tile a workgroup needs was fetched by
  // its neighbour a few tiles earlier and all of W is requested in the first few tile times.  The f32 summation order of a row band
  // then starts at its kst (deterministic; results differ from the unrotated walk in the last bits).
  using Loop = G4Loop<FM, FN, NS, MODE == AVSD_GEMM_TMIX, X2>;
  const int kst = (Loop::rot && (p.flags & AVSD_GEMM_KROT) && nk > 1 && ntm > 1) ? (int)(((long long)tm * nk) / ntm) : 0;
  const int kfirst = kt0 + kst;
  // staging: thread t moves the 16-byte vector (row (t >> 3) + 32 i, k-chunk t & 7) of each operand tile; fragment row
  // wm/wn * (32 F) + 32 b + (lane & 31), 16-byte chunk 2 ks + (lane >> 5)
  const int srow = tid >> 3, sch = tid & 7;
  G4Args ga;
  ga.va0 = (unsigned)((tm * BM + srow) * p.lda + kfirst * BK + sch * 8) * 2u;
  ga.vw0 = (unsigned)((tn * BN + srow) * p.ldw + kfirst * BK + sch * 8) * 2u;
  ga.wr0 = (unsigned)(srow * ROWB + sch * 16);
  ga.rda0 = (unsigned)((wm * 32 * FM + (lane & 31)) * ROWB + (lane >> 5) * 16);
  ga.rdw0 = (unsigned)(A_BYTES + (wn * 32 * FN + (lane & 31)) * ROWB + (lane >> 5) * 16);
  ga.rsA = rsA; ga.rsW = rsW; ga.rsAr = rsAr; ga.rsWr = rsWr;
  ga.kst = (unsigned)kst; ga.winc = 128u - 128u * (unsigned)nk;
  ga.sa = 32u * (unsigned)p.lda * 2u; ga.sw = 32u * (unsigned)p.ldw * 2u;
  ga.nk = (unsigned)nk; ga.kt0 = (unsigned)kt0; ga.tps = 0u; ga.tps2 = 0u;

  // LayerNorm fold with pre-folded statistics (one (sum, sumsq) pair per row, avsd_ln_fold): the pairs of this lane's FM rows are
  // requested BEFORE the main loop (older than every load the loop issues, so its counted waits stay valid) and folded after it
  const bool ln_pre = (p.flags & AVSD_GEMM_LNFUSE) != 0 && p.ln_nblk == 1;
  float2 ln_raw[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    ln_raw[b] = make_float2(0.f, 0.f);
    const int m = tm * BM + wm * 32 * FM + b * 32 + (lane & 31);
    if (ln_pre && m < p.M) ln_raw[b] = reinterpret_cast<const float2*>(p.ln_stats)[m];
  }

  f32x16 acc[FN][FM];
  if constexpr (MODE == AVSD_GEMM_TMIX) {
    // temporal-mix A operand (utils.py:43-53): K segment s of output row (b, f, p) reads row (b, {0, max(f - 1, 0), f}[s], p).  Per 16-byte
    // vector of this thread: its byte offset in the first tile it loads and the jumps it takes, on top of the regular +128 bytes per
    // K tile, when the tile index crosses a segment boundary (d01, d12) or wraps from the last tile of the slice to its first (dw).
    // The asm loop fetches them from the second LDS stage (unused until its first write, which follows these reads in the same
    // wave's in-order LDS queue).
    const int tps = p.cseg / BK;                                   // K tiles per segment
    const int seg0 = min(kfirst / tps, 2), col0 = (kfirst - seg0 * tps) * BK;
    const int ke = kt0 + nk - 1;                                   // last / first tile of the slice: the wrap jump
    const int seg_e = min(max(ke, 0) / tps, 2), col_e = (ke - seg_e * tps) * BK;
    const int seg_s = min(kt0 / tps, 2), col_s = (kt0 - seg_s * tps) * BK;
    unsigned* tbl = reinterpret_cast<unsigned*>(smem4 + STAGE) + tid * (4 * NA);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = tm * BM + srow + 32 * i;
      unsigned v = 0xC0000000u, d01 = 0u, d12 = 0u, dw = 0u;       // rows past M: far past num_records, and they stay there
      if (m < p.M) {
        const int f = (m / p.hw) % p.frames;
        const int r0 = m - f * p.hw, r1 = f > 0 ? m - p.hw : m;
        const int o[3] = {r0 * p.lda, r1 * p.lda, m * p.lda};
        v = (unsigned)(o[seg0] + col0 + sch * 8) * 2u;
        d01 = (unsigned)(o[1] - o[0] - p.cseg) * 2u;
        d12 = (unsigned)(o[2] - o[1] - p.cseg) * 2u;
        dw = (unsigned)(o[seg_s] + col_s - o[seg_e] - col_e - BK) * 2u;
      }
      tbl[i] = v; tbl[NA + i] = d01; tbl[2 * NA + i] = d12; tbl[3 * NA + i] = dw;
    }
    ga.va0 = (unsigned)(STAGE + tid * (4 * NA) * 4);
    ga.tps = (unsigned)tps; ga.tps2 = 2u * (unsigned)tps;
  }
  // every K slice holds at least one tile (avsd_gemm_dispatch_asm refuses a split that leaves an empty slice): the accumulators
  // are ONLY the loop's AGPR outputs — a zero-filled alternative merges with them in VGPRs, all FM * FN * 16 copied at the join
  // (the 256 x 256 tile spilled 132 of them to scratch: 25 us per tile in front of its epilogue)
  Loop::run(acc, ga);

  const int m_base = tm * BM + wm * 32 * FM, n_base = tn * BN + wn * 32 * FN;
  if (p.split_k > 1) {
    float* ws = p.splitk_ws + (int64_t)ksplit * p.M * p.N;
    const int hsel = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = m_base + b * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        const f32x16 v = from_agpr(acc[a][b]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n_base + a * 32 + 8 * q + hsel;
          if (n < p.N)
            *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
      }
    }
    return;
  }
  // the shared epilogue, one 32-row fragment band at a time (a 16-fragment instantiation does not unroll: the accumulators would
  // go through scratch memory); a band of FN <= 5 fragments takes the term-at-a-time form with batched operand loads
  static_for<FM>([&p, &acc, &ln_raw, ln_pre, m_base, n_base, lane](auto b_c) {
    constexpr int B = decltype(b_c)::value;
    // bands are independent, and with 512 registers per lane the scheduler would interleave them (all FM * FN * 16 accumulators
    // copied out of the AGPRs up front: 132 spilled registers in the 256 x 256 tile); keep each band's instructions together
    __builtin_amdgcn_sched_barrier(0);
    f32x16 band[FN][1];
#pragma unroll
    for (int a = 0; a < FN; ++a) band[a][0] = from_agpr(acc[a][B]);
    float pre_ln[2] = {1.f, 0.f};
    if (ln_pre) {          // same arithmetic as ln_row_stats (gemm_common.h) on one pair
      float sm = ln_raw[B].x, sq = ln_raw[B].y;
      asm volatile("" : "+v"(sm), "+v"(sq));      // the fold (and the wait for the pair) stays behind the main loop
      const float inv_k = 1.0f / (float)p.K;
      const float mean = sm * inv_k;
      const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
      pre_ln[0] = rsqrtf(var + p.ln_eps);
      pre_ln[1] = mean * pre_ln[0];
    }
    if constexpr (X2) epilogue_x2<FN, 1>(p, band, m_base + 32 * B, n_base, lane, 0, pre_ln, ln_pre);
    else if constexpr (FN == 5) epilogue_by_term<FN, 1, EX>(p, band, m_base + 32 * B, n_base, lane, 0, pre_ln, ln_pre);   // (as the fused cross-attention tile)
    else epilogue<FN, 1, false, EX>(p, band, m_base + 32 * B, n_base, lane, 0, pre_ln, ln_pre);
  });
}

template <int FM, int FN, int MODE, bool X2 = false, int NS = 2, int EX = EPI_PLAIN>
int launch4_ex(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 64 * FM, BN = 64 * FN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * ROWB * (X2 ? 2 : 1);
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<FM, FN, MODE, X2, NS, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm4: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const int nsplit = d.split_k > 1 ? d.split_k : 1;
  dim3 grid((unsigned)(ntm * ntn), (unsigned)nsplit, 1);
  hipLaunchKernelGGL((gemm4_kernel<FM, FN, MODE, X2, NS, EX>), grid, dim3(256), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm4 launch");
  if (nsplit > 1) return avsd_gemm_splitk_reduce(d, s);
  return AVSD_OK;
}
// (IEEE-half build: a one-pass launch with AVSD_GEMM_OUT_REST runs the EPI_REST instantiation of its tile — separate kernels)
template <int FM, int FN, int MODE, bool X2 = false, int NS = 2>
int launch4(const avsd_gemm_desc& d, hipStream_t s) {
#ifdef AVSD_F16
  if constexpr (!X2) {
    if (d.flags & AVSD_GEMM_OUT_REST) return launch4_ex<FM, FN, MODE, X2, NS, EPI_REST>(d, s);
  }
#endif
  return launch4_ex<FM, FN, MODE, X2, NS>(d, s);
}

}  // namespace

int avsd_gemm_dispatch_asm(const avsd_gemm_desc& d, hipStream_t s) {
  AVSD_REQUIRE((d.mode == AVSD_GEMM_PLAIN || d.mode == AVSD_GEMM_TMIX) && !d.A2 && d.batch == 1 && d.K % 64 == 0,
               "gemm/asm tiles: PLAIN single-source or TMIX operands with K %% 64 == 0 (got mode %d, K %d)", d.mode, d.K);
  AVSD_REQUIRE((double)d.M * d.lda * 2.0 < 1073741824.0 && (double)d.N * d.ldw * 2.0 < 1073741824.0, "gemm/asm tiles: operands must be < 1 GiB");
  AVSD_REQUIRE(d.split_k <= 1 || (d.splitk_ws && d.split_k <= d.K / 64 && !(d.flags & AVSD_GEMM_GEGLU)), "gemm/asm tiles: bad split_k %d", d.split_k);
  if (d.split_k > 1) {
    const int nk_all = d.K / 64, per = (nk_all + d.split_k - 1) / d.split_k;
    AVSD_REQUIRE((d.split_k - 1) * per < nk_all, "gemm/asm tiles: split_k %d leaves an empty slice of the %d K tiles", d.split_k, nk_all);
  }
  const int k = d.tile - AVSD_GEMM_TILE_ASM_FIRST;
  const bool tmix = d.mode == AVSD_GEMM_TMIX;
  if (tmix) AVSD_REQUIRE(d.cseg % 64 == 0 && !(d.flags & AVSD_GEMM_LNFUSE), "gemm/asm tiles: TMIX needs cseg %% 64 == 0 (got %d)", d.cseg);
  if (d.flags & AVSD_GEMM_X2) {
    // (plane offsets, out_lo / res*_lo: checked by avsd_gemm_bf16 before it dispatches here)
    switch (k) {
      case 3: return tmix ? launch4<2, 2, AVSD_GEMM_TMIX, true>(d, s) : launch4<2, 2, AVSD_GEMM_PLAIN, true>(d, s);     // 128 x 128, 144 KB
      case 4: return tmix ? launch4<2, 1, AVSD_GEMM_TMIX, true>(d, s) : launch4<2, 1, AVSD_GEMM_PLAIN, true>(d, s);     // 128 x 64, 108 KB
      case 5: return tmix ? launch4<1, 2, AVSD_GEMM_TMIX, true>(d, s) : launch4<1, 2, AVSD_GEMM_PLAIN, true>(d, s);     // 64 x 128
      case 6: return tmix ? launch4<1, 1, AVSD_GEMM_TMIX, true>(d, s) : launch4<1, 1, AVSD_GEMM_PLAIN, true>(d, s);     // 64 x 64, 72 KB
      default: AVSD_REQUIRE(false, "gemm/asm tiles: tile %d has no split-precision form (63..66 do)", d.tile);
    }
  }
  if (tmix) {
    switch (k) {
      case 1: return launch4<4, 2, AVSD_GEMM_TMIX>(d, s);
      case 2: return launch4<2, 4, AVSD_GEMM_TMIX>(d, s);
      case 3: return launch4<2, 2, AVSD_GEMM_TMIX>(d, s);
      case 4: return launch4<2, 1, AVSD_GEMM_TMIX>(d, s);
      case 5: return launch4<1, 2, AVSD_GEMM_TMIX>(d, s);
      case 6: return launch4<1, 1, AVSD_GEMM_TMIX>(d, s);
      case 7: return launch4<2, 5, AVSD_GEMM_TMIX>(d, s);
      default: AVSD_REQUIRE(false, "gemm/asm tiles: tile %d has no TMIX form (61..67 do)", d.tile);
    }
  }
  switch (k) {
    case 0: return launch4<4, 4, AVSD_GEMM_PLAIN>(d, s);     // 256 x 256, 128 x 128 per wave, 144 KB
    case 1: return launch4<4, 2, AVSD_GEMM_PLAIN>(d, s);     // 256 x 128, 108 KB
    case 2: return launch4<2, 4, AVSD_GEMM_PLAIN>(d, s);     // 128 x 256
    case 3: return launch4<2, 2, AVSD_GEMM_PLAIN>(d, s);     // 128 x 128, 72 KB: two workgroups per CU
    case 4: return launch4<2, 1, AVSD_GEMM_PLAIN>(d, s);     // 128 x 64, 54 KB
    case 5: return launch4<1, 2, AVSD_GEMM_PLAIN>(d, s);     // 64 x 128
    case 6: return launch4<1, 1, AVSD_GEMM_PLAIN>(d, s);     // 64 x 64, 36 KB: four workgroups per CU
    case 7: return launch4<2, 5, AVSD_GEMM_PLAIN>(d, s);     // 128 x 320, 64 x 160 per wave, 126 KB: N = 320 in one column tile
    default: AVSD_REQUIRE(false, "gemm/asm tiles: unknown tile %d", d.tile);
  }
}
