// Row-panel GEMM for the short-K linear layers of the 32 x 32 level (gfx950) — tile id AVSD_GEMM_TILE_ROWPANEL of avsd_gemm_bf16,
// mode AVSD_GEMM_PLAIN, single source, K <= 320.
//
// Why another GEMM loop: at C = 320 a linear layer has FIVE K tiles.  gemm2_kernel cuts M x N into tiles and every workgroup
// pays a prologue (first loads from a cold start), five K tiles and an epilogue; the activation tile is fetched again by every
// column tile (20 times for the GEGLU projection, N = 2560) and so are its LayerNorm statistics.  The layers are bound by the
// global -> LDS path and by those per-tile fixed costs, not by the MFMA pipe (tools/step_vs_blas.py: 457 TFLOP/s on the GEGLU
// projection, 267-310 on the 320 x 320 projections).
// Here a workgroup OWNS a panel of 96 rows — 24576 rows of one clip are exactly 256 panels, one per CU, one wave of workgroups —
// stages the panel's activation (96 x K, <= 60 KB) ONCE, keeps it resident, and walks N in steps of 320 columns: only the
// 320 x 32 weight half-tiles stream through a four-stage ring (three in flight: a ring of two whole K tiles left one 40-KB
// tile in flight at a time — 2.9 us per K tile, issue + landing serialised, profiles/r3_rowpanel_probe.txt), without a drain
// between column steps — the next step's weights are in flight while this step's epilogue runs.  LayerNorm statistics of the panel's rows are folded once.
// Global -> LDS bytes per output element: (1/BM + 1/BN) * 2K with BM = BN = 128 becomes (1/96) * 2K + A once.
//
// Geometry: 6 MFMA waves = 3 (rows) x 2 (columns), wave tile 32 x 160 (5 fragments of v_mfma_f32_32x32x16), + 4 loader waves
// that own all global -> LDS traffic (gemm2_kernel's loader-wave scheme: they wait for their loads, everybody meets at one
// barrier per half K tile).  LDS: A panel = K/64 tile images of [96][64] (gemm_common.h image), W ring 4 x [320][32]: 60 + 80 KB.
// Epilogue: the shared f32 epilogue of gemm_common.h per column step (bias, row vector, LayerNorm fold, GELU / GEGLU,
// residuals, f32 master, row statistics) — same f32 order per element as every other tile; the K order is the plain one, so
// results are bit-identical to the gemm2 tiles.
//
// Replaces (reference file:line): the nn.Linear calls of BasicTransformerBlock at C = 320 — to_q / to_k / to_v / to_out
// (avgen/models/unets/utils.py:123-131,159), ff.net.0.proj (ff_spatio_audio_temp_transformer_3d.py:361-371), proj_in /
// proj_out (…transformer_3d.py:66,92).
#include "gemm_common.h"

namespace {

constexpr unsigned OOBP = 0x80000000u;
constexpr int RP_BM = 96, RP_BN = 320, RP_KMAX = 320;
constexpr int RP_WM = 3, RP_WN = 2, RP_LW = 4;
constexpr int RP_A_IMG = RP_BM * 128;                 // one K tile of the panel: 12 KiB
constexpr int RP_HK = 32;                             // the weight ring holds HALF K tiles: 320 rows x 32 values = 20 KiB
constexpr int RP_W_BYTES = RP_BN * RP_HK * 2;
constexpr int RP_STAGES = 4;
constexpr int RP_PA = (RP_BM / 8) / RP_LW;            // 3 A pieces per loader wave per K tile
constexpr int RP_PW = (RP_W_BYTES / 1024) / RP_LW;    // 5 W pieces per loader wave per half tile
static_assert(RP_PA * RP_LW * 8 == RP_BM && RP_PW * RP_LW * 1024 == RP_W_BYTES, "pieces must split evenly over the loader waves");

// Half-tile image: rows of 32 values = 64 B, four rows per 256-byte line L; the 16-byte slot of (row r, chunk c in 0..3) is
// ((r & 3) << 2) | (c ^ (L & 3)): the 16-lane groups of ds_read_b128 (32 consecutive rows from a multiple of 32, one chunk)
// touch four lines with distinct L & 3 -> 16 distinct slots.  A 1-KiB piece = 4 lines = 16 rows, written lane-linearly.
__device__ __forceinline__ void half_piece_row_chunk(int piece, int lane, int& row, int& kchunk) {
  const int L = piece * 4 + (lane >> 4);
  const int s = lane & 15;
  row = 4 * L + (s >> 2);
  kchunk = ((s & 3) ^ (L & 3)) * 8;
}

__global__ __launch_bounds__(64 * (RP_WM * RP_WN + RP_LW)) void rowpanel_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smemp[];
  constexpr int NC = RP_WM * RP_WN;
  constexpr int FN = RP_BN / RP_WN / 32;               // 5
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave_all >= NC;
  const int wave = is_loader ? wave_all - NC : 0;
  const int wm = wave_all % RP_WM;
  const int wn = (wave_all / RP_WM) % RP_WN;

  const int m0 = blockIdx.x * RP_BM;
  const int nkt = (p.K + BK - 1) / BK;                 // <= 5
  const int nh = 2 * nkt;                              // half tiles per column step
  const int nsteps = (p.N + RP_BN - 1) / RP_BN;
  const int T = nsteps * nh;
  // panels walk the column steps in rotated order: 256 workgroups streaming the SAME weight rows in lockstep queue up on the L2
  // channels that hold them (measured: 14 GB/s per CU); rotated, neighbours read different slices at any moment
  const int step0 = blockIdx.x % nsteps;
  const int a_bytes = nkt * RP_A_IMG;
  unsigned char* ring = smemp + a_bytes;

  if (is_loader) {
    const h16_t* Ab = reinterpret_cast<const h16_t*>(p.A);
    const h16_t* Wb = reinterpret_cast<const h16_t*>(p.W);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);
    // the panel, all K tiles
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
      for (int j = 0; j < RP_PA; ++j) {
        int row, kc;
        piece_row_chunk(wave + j * RP_LW, lane, row, kc);
        const int m = m0 + row, k = kt * BK + kc;
        const unsigned vo = (m < p.M && k < p.K) ? (unsigned)(m * p.lda + k) * 2u : OOBP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smemp + kt * RP_A_IMG + (wave + j * RP_LW) * 1024), 16, (int)vo, 0, 0, 0);
      }
    }
    int wrow[RP_PW], wkc[RP_PW];
#pragma unroll
    for (int j = 0; j < RP_PW; ++j) half_piece_row_chunk(wave + j * RP_LW, lane, wrow[j], wkc[j]);
    int i_t = 0, i_step = step0, i_h = 0;
    auto issue_w = [&]() {
      unsigned char* sb = ring + (i_t % RP_STAGES) * RP_W_BYTES;
#pragma unroll
      for (int j = 0; j < RP_PW; ++j) {
        const int n = i_step * RP_BN + wrow[j], k = i_h * RP_HK + wkc[j];
        const unsigned vo = (n < p.N && k < p.K) ? (unsigned)(n * p.ldw + k) * 2u : OOBP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sb + (wave + j * RP_LW) * 1024), 16, (int)vo, 0, 0, 0);
      }
      ++i_t;
      if (++i_h == nh) { i_h = 0; if (++i_step == nsteps) i_step = 0; }
    };
#pragma unroll
    for (int s = 0; s < RP_STAGES - 1; ++s)
      if (s < T) issue_w();
    for (int t = 0; t < T; ++t) {
      wait_tiles_ahead<RP_STAGES - 2, RP_PW>(T - 1 - t);   // half tile t (and, at t = 0, the panel) have landed
      __builtin_amdgcn_s_barrier();
      if (t + RP_STAGES - 1 < T) issue_w();                  // into the stage every MFMA wave finished reading before this barrier
    }
    return;
  }

  // ---- MFMA waves -----------------------------------------------------------------------------------------------------
  const int r = wm * 32 + (lane & 31);
  const int a_line = (r >> 1) * 256, a_sw = (r >> 1) & 15, a_hi = (r & 1) << 3;
  int w_off[FN], w_sw[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int rr = wn * (RP_BN / RP_WN) + a * 32 + (lane & 31);
    w_off[a] = (rr >> 2) * 256 + ((rr & 3) << 6);
    w_sw[a] = (rr >> 2) & 3;
  }
  const int chalf = lane >> 5;
  // LayerNorm statistics of this lane's row: once per panel
  float pre_ln[2] = {1.f, 0.f};
  const bool lnf = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  if (lnf && m0 + r < p.M) ln_row_stats(p, m0 + r, 0, pre_ln[0], pre_ln[1]);

  int t = 0;
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    const int step = (step0 + sidx) % nsteps;
    f32x16 acc[FN][1];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][0][q] = 0.f;
    for (int h = 0; h < nh; ++h, ++t) {
      __builtin_amdgcn_s_barrier();
      const unsigned char* sA = smemp + (h >> 1) * RP_A_IMG;
      const unsigned char* sW = ring + (t % RP_STAGES) * RP_W_BYTES;
#pragma unroll
      for (int ks = 0; ks < RP_HK / 16; ++ks) {
        const int cw = ks * 2 + chalf;                 // chunk inside the half tile
        const int ca = (h & 1) * 4 + cw;               // chunk inside the panel's K tile
        const h16x8 xf = *reinterpret_cast<const h16x8*>(sA + a_line + (((a_hi | ca) ^ a_sw) << 4));
        h16x8 wf[FN];
#pragma unroll
        for (int a = 0; a < FN; ++a)
          wf[a] = *reinterpret_cast<const h16x8*>(sW + w_off[a] + ((cw ^ w_sw[a]) << 4));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FN; ++a) acc[a][0] = mfma32x32x16(wf[a], xf, acc[a][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    epilogue<FN, 1>(p, acc, m0 + wm * 32, step * RP_BN + wn * (RP_BN / RP_WN), lane, 0, pre_ln, lnf);
  }
}

}  // namespace

extern "C" int avsd_gemm_rowpanel_supported(int M, int N, int K) {
  return (M > 0 && N > 0 && K > 0 && K <= RP_KMAX && K % 8 == 0 && N % 32 == 0) ? RP_BM : 0;
}

int avsd_gemm_dispatch_rowpanel(const avsd_gemm_desc& d, hipStream_t s) {
  AVSD_REQUIRE(d.mode == AVSD_GEMM_PLAIN && !d.A2 && d.batch == 1 && d.split_k <= 1 && !(d.flags & AVSD_GEMM_X2),
               "gemm/rowpanel: single-source PLAIN operands only (no split_k, batching, split precision)");
  AVSD_REQUIRE(avsd_gemm_rowpanel_supported(d.M, d.N, d.K) != 0, "gemm/rowpanel: K <= %d and N %% 32 == 0 (got N %d, K %d)", RP_KMAX, d.N, d.K);
  AVSD_REQUIRE((double)d.M * d.lda * 2.0 < 2147483648.0 && (double)d.N * d.ldw * 2.0 < 2147483648.0, "gemm/rowpanel: operands must be < 2 GiB");
  const size_t lds = (size_t)((d.K + BK - 1) / BK) * RP_A_IMG + RP_STAGES * RP_W_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowpanel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((RP_KMAX / BK) * RP_A_IMG + RP_STAGES * RP_W_BYTES));
    if (e != hipSuccess) {
      avsd_set_error("rowpanel: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(rowpanel_kernel, dim3((unsigned)((d.M + RP_BM - 1) / RP_BM)), dim3(64 * (RP_WM * RP_WN + RP_LW)), lds, s, d);
  AVSD_CHECK_LAUNCH("rowpanel launch");
  return AVSD_OK;
}
