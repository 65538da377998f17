// 256 x 256 x 64 GEMM tile with a phase-interleaved main loop (gfx950) — tile id AVSD_GEMM_TILE_8PHASE of avsd_gemm_bf16.
//
// Why another main loop: gemm2_kernel's big tiles run every wave through "barrier, read fragments, multiply" in lockstep,
// so the LDS reads of a K tile (75 % of its MFMA time at 256 x 256) and the matrix work overlap only as far as hipcc
// interleaves them inside one wave.  Here the two waves of a SIMD are held half a phase apart by the barriers themselves:
// while one multiplies a quadrant of its output (8 MFMAs, 256 cycles) the other fetches the fragments of its next
// quadrant and issues its share of the LDS-DMA for a later K tile (the schedule of the CDNA programming guide's 8-phase
// template, rebuilt on this library's tile image, loaders and epilogue).
//
// Geometry: 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 of the output = 4 x 2 fragments of v_mfma_f32_32x32x16.
// A K tile (64 deep) is split into four HALF-TILES of 128 rows x 128 B = 16 KiB, by the quadrant that consumes them:
//     HA0 = the first 64 rows of both wave rows (row fragments 0, 1)      HW0 = the first 32 columns of the four wave columns
//     HA1 = the last 64 rows                                              HW1 = the last 32 columns
// each a tile image in the layout of gemm_common.h (piece_row_chunk / frag_offset).  A K tile takes 4 phases:
//     phase   reads (ds_read_b128)      multiplies          issues half-tile (LDS-DMA, 2 loads per lane)
//       1     HA0 (8) + HW0 (4)         (rows 0, cols 0)     HW1 of K tile +1
//       2     HW1 (4)                   (rows 0, cols 1)     HA1 of K tile +1
//       3     HA1 (8)                   (rows 1, cols 1)     HA0 of K tile +2
//       4     —  (HW0 stays in regs)    (rows 1, cols 0)     HW0 of K tile +2
// and a phase is  [reads, issue, counted vmcnt] s_barrier [lgkmcnt(0), 8 MFMAs] s_barrier.  Wave row 1 runs one barrier
// behind wave row 0, so on every SIMD one wave is in its MFMA block while the other is in its read block.
// LDS: a ring of 8 half-tile slots (128 KiB); half-tile i (i = 4 * K tile + {HA0, HW0, HW1, HA1}) lives in slot i % 8 and is
// issued in phase i - 6.
//   RAW: after issuing in phase g a wave waits until at most 4 half-tiles (8 loads) are in flight, i.e. half-tiles <= g + 2
//        have landed, BEFORE the first barrier of phase g; they are read in phase g + 1 or later, which every wave enters
//        behind that barrier (wave row 1: behind the next one).
//   WAR: slot reuse distance: a half-tile is re-issued >= 2 phases after the phase of its last read (HA0: read in phase 1,
//        slot re-issued in phase 3; HW0: 1 -> 4; HW1: 2 -> 5; HA1: 3 -> 6), so the late wave row's reads (retired by the
//        lgkmcnt(0) behind ITS first barrier of that phase) are complete before any wave reaches the issuing phase.
// Only PLAIN single-source and CONV3 (cin % 64 == 0) operands, no split-K.
//
// MEASURED (tools/p8_probe.py, profiles/r3_8phase_probe.txt): bit-identical to the other tiles, 1.02-1.08 PFLOP/s at 8192^3 — the
// same as the 256 x 128 tile with loader waves (1.02-1.04) — and 0.5-0.8x of the tuned tiles on every shape the UNet and the VAE
// contain (one 128-KiB workgroup per CU; 4 phases of prologue per tile).  The ablations (G8_ABL below) say where the time goes at
// 8192^3: barriers + MFMA blocks alone (no fragment reads, no LDS-DMA) run at 1.44 PF; the fragment reads cost 6 %, the
// LDS-DMA issue 23 % (two 1-KiB pieces per wave per phase, ~100 cycles each in front of the wave's next read block); rotating
// four accumulators instead of two, dropping s_setprio, global_load_lds instead of buffer_load ... lds and a branch-free
// steady-state loop change nothing or lose.  The tile is kept selectable (descriptor field `tile`) and tested, but the tuner
// does not list it: no shape of this workload prefers it.
#include <type_traits>

#include "gemm_common.h"

#ifndef G8_ABL
#define G8_ABL 0      // timing probes only (results are garbage): 1 no fragment reads, 2 no LDS-DMA in the loop, 4 no MFMAs, 8 no s_setprio
#endif

namespace {

constexpr unsigned OOB8 = 0x80000000u;
constexpr int HALF_BYTES = 128 * 128;

template <int MODE>
__global__ __launch_bounds__(512) void gemm8p_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem8[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  int wg;
  {
    const int total = ntm * ntn;
    const int bid = blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const bool nmaj = (p.flags & AVSD_GEMM_XCD_N) != 0;
  const int tn = nmaj ? wg / ntm : wg % ntn;
  const int tm = nmaj ? wg % ntm : wg / ntn;
  const int64_t bz = blockIdx.z;

  const h16_t* Ab = reinterpret_cast<const h16_t*>(p.A) + bz * p.batch_stride_a;
  const h16_t* Wb = reinterpret_cast<const h16_t*>(p.W) + bz * p.batch_stride_w;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);

  // ---- loader state: this lane's two pieces (q = wave, wave + 8) of every half-tile type --------------------------------
  // image row r' of a half-tile -> tile row:  HA0 / HA1: (r' / 64) * 128 + r' % 64 (+ 64);  HW0 / HW1: (r' / 32) * 64 + r' % 32 (+ 32)
  int kc[2];
  int a_off[2][2];           // [HA0 / HA1][piece]: PLAIN: element offset of (row, kc); CONV3: image index * hs * ws (see below)
  bool a_ok[2][2];
  int a_hb[2][2], a_wb[2][2];   // CONV3: top-left input coordinate of the output pixel
  unsigned w_off[2][2];      // [HW0 / HW1][piece]: byte offset, OOB8 when the row is past N
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int L = (wave + 8 * j) * 4 + (lane >> 4);
    const int x = (lane & 15) ^ (L & 15);
    const int ri = 2 * L + (x >> 3);
    kc[j] = (x & 7) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = tm * 256 + (ri >> 6) * 128 + (ri & 63) + 64 * h;
      a_ok[h][j] = m < p.M;
      if (MODE == AVSD_GEMM_PLAIN) {
        a_off[h][j] = m * p.lda + kc[j];
        a_hb[h][j] = a_wb[h][j] = 0;
      } else {
        const int hw = p.ho * p.wo;
        const int img = m / hw, rem = m - img * hw;
        const int oh = rem / p.wo, ow = rem - oh * p.wo;
        a_off[h][j] = img;
        a_hb[h][j] = oh * p.stride - p.pad;
        a_wb[h][j] = ow * p.stride - p.pad;
      }
      const int n = tn * 256 + (ri >> 5) * 64 + (ri & 31) + 32 * h;
      w_off[h][j] = n < p.N ? (unsigned)(n * p.ldw + kc[j]) * 2u : OOB8;
    }
  }
  const int hin = p.hs << p.ups, win = p.ws << p.ups;

  const int nk = (p.K + BK - 1) / BK;
  const int nhalf = 4 * nk;
  // half-tile i: K tile i >> 2, type T = i & 3 in need order (HA0, HW0, HW1, HA1), slot i & 7.  T is a compile-time tag
  // (std::integral_constant): the per-type address arrays are then indexed statically and stay in registers.
  auto issue = [&](auto tag, int kt, int slot) {
    constexpr int T = decltype(tag)::value;
    const int kbase = kt * BK;
    const bool full = kbase + BK <= p.K;          // wave-uniform: only the last K tile of a ragged K checks per vector
    unsigned char* sb = smem8 + slot * HALF_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      lds_ptr_t dst = (lds_ptr_t)(sb + (wave + 8 * j) * 1024);
      if constexpr (T == 0 || T == 3) {
        constexpr int h = T == 3;
        unsigned vo;
        if (MODE == AVSD_GEMM_PLAIN) {
          vo = (a_ok[h][j] && (full || kbase + kc[j] < p.K)) ? (unsigned)(a_off[h][j] + kbase) * 2u : OOB8;
        } else {
          const int tap = kbase / p.cin, c0 = kbase - tap * p.cin;      // wave-uniform
          const int kh = tap / 3, kw = tap - kh * 3;
          const int hi = a_hb[h][j] + kh, wi = a_wb[h][j] + kw;
          const bool ok = a_ok[h][j] && (full || tap < 9) && (unsigned)hi < (unsigned)hin && (unsigned)wi < (unsigned)win;
          vo = ok ? (unsigned)(((a_off[h][j] * p.hs + (hi >> p.ups)) * p.ws + (wi >> p.ups)) * p.lda + c0 + kc[j]) * 2u : OOB8;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, (int)vo, 0, 0, 0);
      } else {
        constexpr int h = T == 2;
        const unsigned vo = (w_off[h][j] != OOB8 && (full || kbase + kc[j] < p.K)) ? w_off[h][j] + (unsigned)kbase * 2u : OOB8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, (int)vo, 0, 0, 0);
      }
    }
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read addressing inside a half-tile image
  int a_line[2], a_sw[2], a_hi[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wr * 64 + b * 32 + (lane & 31);
    a_line[b] = (r >> 1) * 256;
    a_sw[b] = (r >> 1) & 15;
    a_hi[b] = (r & 1) << 3;
  }
  int w_line, w_sw, w_hi;
  {
    const int r = wc * 32 + (lane & 31);
    w_line = (r >> 1) * 256;
    w_sw = (r >> 1) & 15;
    w_hi = (r & 1) << 3;
  }
  const int chalf = lane >> 5;

  // prologue: half-tiles 0..5 in flight, 0 and 1 landed
  issue(T0{}, 0, 0); issue(T1{}, 0, 1); issue(T2{}, 0, 2); issue(T3{}, 0, 3);
  if (nk > 1) { issue(T0{}, 1, 4); issue(T1{}, 1, 5); }
  float pre_ln[8] = {};
  const bool pre = (p.flags & AVSD_GEMM_LNFUSE) != 0;
  if (pre) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = tm * 256 + wr * 128 + b * 32 + (lane & 31);
      pre_ln[2 * b] = 1.f; pre_ln[2 * b + 1] = 0.f;
      if (m < p.M) ln_row_stats(p, m, bz, pre_ln[2 * b], pre_ln[2 * b + 1]);
    }
  }
  wait_tiles_ahead<4, 2>(nhalf - 2);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one barrier behind

  h16x8 xf[2][4], wf0[4], wf1[4];
  auto read_a = [&](const unsigned char* s) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + chalf;
#pragma unroll
      for (int b = 0; b < 2; ++b) xf[b][ks] = *reinterpret_cast<const h16x8*>(s + a_line[b] + (((a_hi[b] | c) ^ a_sw[b]) << 4));
    }
  };
  auto read_w = [&](const unsigned char* s, h16x8 (&wf)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + chalf;
      wf[ks] = *reinterpret_cast<const h16x8*>(s + w_line + (((w_hi | c) ^ w_sw) << 4));
    }
  };
  // the second half of a phase: [counted vmcnt] barrier, fragments landed, 8 MFMAs of quadrant (mh, a), barrier
  // (phase g = 4 kt + P issues half-tile g + 6: type (P + 2) % 4 of K tile kt + 1 (P < 2) or kt + 2)
  // (steady: every K tile but the last two — the half-tile to issue exists and 4 stay in flight, no tail arithmetic)
  auto block = [&](auto ptag, int kt, auto mtag, auto atag, const h16x8 (&wf)[4], auto steady) {
    constexpr int P = decltype(ptag)::value;
    constexpr int mh = decltype(mtag)::value, a = decltype(atag)::value;
    constexpr bool STEADY = decltype(steady)::value;
    const int g = 4 * kt + P;
    if constexpr (STEADY) {
      if (!(G8_ABL & 2)) issue(std::integral_constant<int, (P + 2) & 3>{}, kt + (P < 2 ? 1 : 2), ((kt & 1) * 4 + P + 6) & 7);
      wait_vmcnt<8>();
    } else {
      if (!(G8_ABL & 2) && g + 6 < nhalf) issue(std::integral_constant<int, (P + 2) & 3>{}, kt + (P < 2 ? 1 : 2), (g + 6) & 7);
      wait_tiles_ahead<4, 2>(nhalf - g - 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(G8_ABL & 8)) __builtin_amdgcn_s_setprio(1);
    if (!(G8_ABL & 4)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][mh * 2 + b] = mfma32x32x16(wf[ks], xf[b][ks], acc[a][mh * 2 + b], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][mh * 2 + b][0] += (float)wf[P & 3][0] + (float)xf[b][P & 3][0];
    }
    if (!(G8_ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  auto ktile = [&](int kt, auto steady) {
    const unsigned char* sb = smem8 + (kt & 1) * 4 * HALF_BYTES;
    const bool rd = !(G8_ABL & 1) || kt == 0;
    if (rd) read_w(sb + 1 * HALF_BYTES, wf0);   // phase 1: HW0 first (its 4 reads retire first), then HA0
    __builtin_amdgcn_sched_barrier(0);
    if (rd) read_a(sb);
    block(T0{}, kt, T0{}, T0{}, wf0, steady);
    if (rd) read_w(sb + 2 * HALF_BYTES, wf1);   // phase 2: HW1
    block(T1{}, kt, T0{}, T1{}, wf1, steady);
    if (rd) read_a(sb + 3 * HALF_BYTES);        // phase 3: HA1
    block(T2{}, kt, T1{}, T1{}, wf1, steady);
    block(T3{}, kt, T1{}, T0{}, wf0, steady);   // phase 4: HW0 is still in registers
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{});
  for (; kt < nk; ++kt) ktile(kt, std::false_type{});
  if (wr == 0) __builtin_amdgcn_s_barrier();     // every wave has executed the same number of barriers

  epilogue<2, 4>(p, acc, tm * 256 + wr * 128, tn * 256 + wc * 64, lane, bz, pre_ln, pre);
}

template <int MODE>
int launch8p(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr size_t lds = 8 * HALF_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm8p: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + 255) / 256, ntn = (d.N + 255) / 256;
  hipLaunchKernelGGL((gemm8p_kernel<MODE>), dim3((unsigned)(ntm * ntn), 1, (unsigned)d.batch), dim3(512), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm8p launch");
  return AVSD_OK;
}

}  // namespace

// called by avsd_gemm_bf16 for tile AVSD_GEMM_TILE_8PHASE (the entry point has validated the descriptor)
int avsd_gemm_dispatch_8phase(const avsd_gemm_desc& d, hipStream_t s) {
  if (d.mode == AVSD_GEMM_PLAIN) return launch8p<AVSD_GEMM_PLAIN>(d, s);
  return launch8p<AVSD_GEMM_CONV3>(d, s);
}
