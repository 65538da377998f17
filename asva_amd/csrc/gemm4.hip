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
// -> bit-identical to the LDS-direct tiles.
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

#define G4_MFMA(ACC, WF, XF) asm volatile(AVSD_MFMA_OP " %0, %1, %2, %0" : "+a"(ACC) : "v"(WF), "v"(XF))
#define G4_DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define G4_DSWRITE(ADDR, SRC, OFF) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(ADDR), "v"(SRC), "n"(OFF) : "memory")
#define G4_GLOAD(DST, VOFF, RSRC) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(DST) : "v"(VOFF), "s"(RSRC) : "memory")
#define G4_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory")
#define G4_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory")
#define G4_BARRIER() asm volatile("s_barrier" ::: "memory")

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order: the index is a compile-time constant inside f
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int FM, int FN, int MODE>
__global__ __launch_bounds__(256, 1) void gemm4_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem4[];
  constexpr int BM = 64 * FM, BN = 64 * FN;
  constexpr int NA = BM / 32, NW = BN / 32;          // 16-byte global loads per thread per K tile (A rows, W rows)
  constexpr int NL = NA + NW;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int NMF = FM * FN;                        // MFMAs per k-step
  constexpr int NFR = FM + FN;                        // fragment reads per k-step
  constexpr int WPK = (NL + 2) / 3;                   // write + reload pairs per k-step (k-steps 0..2)
  static_assert(MODE == AVSD_GEMM_PLAIN, "gemm4: PLAIN operands");
  static_assert(2 * NL - 1 < 64 && WPK < 16, "vmcnt is a 6-bit, lgkmcnt a 4-bit counter");
  static_assert((NA > NW ? NA : NW) * 32 * ROWB < 65536, "ds offset field is 16 bits");

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
  const bool nmaj = (p.flags & AVSD_GEMM_XCD_N) != 0;
  const int tn = nmaj ? wg / ntm : wg % ntn;
  const int tm = nmaj ? wg % ntm : wg / ntn;
  const int nk_all = p.K / BK;
  const int per_split = (nk_all + nsplit - 1) / nsplit;
  const int kt0 = ksplit * per_split;
  const int nk = max(min(nk_all, kt0 + per_split) - kt0, 0);

  // buffer descriptors in scalar registers (operands of the load statements)
  const unsigned long long pa = (unsigned long long)p.A, pw = (unsigned long long)p.W;
  const u32x4 rsA = {(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  const u32x4 rsW = {(unsigned)pw, (unsigned)(pw >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};

  // ---- staging: thread t moves the 16-byte vector (row (t >> 3) + 32 i, k-chunk t & 7) of each operand tile ---------------
  const int srow = tid >> 3, sch = tid & 7;
  unsigned va[NA], vw[NW];                 // byte offsets of this thread's vectors in the NEXT tile to load
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = tm * BM + srow + 32 * i;
    va[i] = m < p.M ? (unsigned)(m * p.lda + kt0 * BK + sch * 8) * 2u : 0x80000000u;       // rows past M: zero-filled by the bounds check
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int n = tn * BN + srow + 32 * i;
    vw[i] = n < p.N ? (unsigned)(n * p.ldw + kt0 * BK + sch * 8) * 2u : 0x80000000u;
  }
  // LDS byte addresses per stage (the 16-bit offset field of the ds instructions cannot span a 72-KB stage): vector i of a tile
  // at + i * 32 * ROWB; fragment row wm/wn * (32 F) + 32 b + (lane & 31), 16-byte chunk 2 ks + (lane >> 5)
  unsigned wr_a[2], wr_w[2], rd_a[2], rd_w[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    wr_a[st] = (unsigned)(st * STAGE + srow * ROWB + sch * 16);
    wr_w[st] = wr_a[st] + A_BYTES;
    rd_a[st] = (unsigned)(st * STAGE + (wm * 32 * FM + (lane & 31)) * ROWB + (lane >> 5) * 16);
    rd_w[st] = (unsigned)(st * STAGE + A_BYTES + (wn * 32 * FN + (lane & 31)) * ROWB + (lane >> 5) * 16);
  }

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 g[2][NL];                         // staging registers of two K tiles in flight
  h16x8 xf[2][FM], wf[2][FN];             // fragments of two k-steps

  int t_load = 0;                         // index (from kt0) of the next tile to load
  // after the loads of a tile are issued its offsets advance by one K tile — except past the end of K, where the last tile
  // is loaded again (never consumed: keeps the loop uniform and every address inside the tensors)
  auto advance = [&]() {
    ++t_load;
    const unsigned inc = t_load < nk ? BK * 2u : 0u;
#pragma unroll
    for (int i = 0; i < NA; ++i) va[i] += inc;
#pragma unroll
    for (int i = 0; i < NW; ++i) vw[i] += inc;
  };
  auto load_one = [&g, &va, &vw, &rsA, &rsW](auto s_c, auto j_c) {
    constexpr int S = decltype(s_c)::value, J = decltype(j_c)::value;
    if constexpr (J < NA) G4_GLOAD(g[S][J], va[J], rsA);
    else G4_GLOAD(g[S][J], vw[J - NA], rsW);
  };
  auto write_one = [&g, &wr_a, &wr_w](auto s_c, auto st_c, auto j_c) {
    constexpr int S = decltype(s_c)::value, ST = decltype(st_c)::value, J = decltype(j_c)::value;
    if constexpr (J < NA) G4_DSWRITE(wr_a[ST], g[S][J], J * 32 * ROWB);
    else G4_DSWRITE(wr_w[ST], g[S][J], (J - NA) * 32 * ROWB);
  };
  // fragment read R (0 .. FM + FN - 1) of k-step KS from LDS stage ST into fragment set FS
  auto frag_read = [&xf, &wf, &rd_a, &rd_w](auto fs_c, auto st_c, auto ks_c, auto r_c) {
    constexpr int FS = decltype(fs_c)::value, ST = decltype(st_c)::value, KS = decltype(ks_c)::value, R = decltype(r_c)::value;
    if constexpr (R < FM) G4_DSREAD(xf[FS][R], rd_a[ST], R * 32 * ROWB + KS * 32);
    else G4_DSREAD(wf[FS][R - FM], rd_w[ST], (R - FM) * 32 * ROWB + KS * 32);
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;

  // ---- prologue: tiles 0 and 1 in flight, tile 0 to LDS stage 0, tile 2 issued, fragments of k-step 0 ------------------------
  if (nk > 0) {
  static_for<NL>([&](auto j) { load_one(c0{}, j); });
  advance();
  static_for<NL>([&](auto j) { load_one(c1{}, j); });
  advance();
  G4_WAIT_VM(NL);                               // tile 0 landed (tile 1 may still be in flight)
  static_for<NL>([&](auto j) { write_one(c0{}, c0{}, j); });
  static_for<NL>([&](auto j) { load_one(c0{}, j); });
  advance();
  G4_WAIT_LGKM(0);
  G4_BARRIER();
  static_for<NFR>([&](auto r) { frag_read(c0{}, c0{}, c0{}, r); });
  G4_WAIT_LGKM(0);

  // One K tile: CS = LDS stage of the tile being multiplied (its successor goes to CS ^ 1 from staging set GS).
  auto tile = [&acc, &xf, &wf, &load_one, &write_one, &frag_read, &advance](auto cs_c, auto gs_c) {
    constexpr int CS = decltype(cs_c)::value;
    static_for<4>([&acc, &xf, &wf, &load_one, &write_one, &frag_read, &advance, cs_c, gs_c](auto ks_c) {
      constexpr int CS = decltype(cs_c)::value;
      constexpr int KS = decltype(ks_c)::value;
      constexpr int FS = KS & 1;
      constexpr int NWR = KS < 3 ? (NL - KS * WPK < WPK ? (NL - KS * WPK > 0 ? NL - KS * WPK : 0) : WPK) : 0;   // writes in this k-step
      constexpr int NF = NFR + NWR;                 // memory "fillers" of this k-step: the fragment reads first, then the write + reload pairs
      static_for<NMF>([&acc, &xf, &wf, &load_one, &write_one, &frag_read, cs_c, gs_c, ks_c](auto i_c) {
        constexpr int CS = decltype(cs_c)::value, KS = decltype(ks_c)::value, FS = KS & 1;
        constexpr int I = decltype(i_c)::value;
        G4_MFMA(acc[I / FM][I % FM], wf[FS][I / FM], xf[FS][I % FM]);
        static_for<NF>([&load_one, &write_one, &frag_read, cs_c, gs_c, ks_c, i_c](auto f_c) {
          constexpr int CS = decltype(cs_c)::value, KS = decltype(ks_c)::value, FS = KS & 1, I = decltype(i_c)::value;
          constexpr int Fi = decltype(f_c)::value;
          constexpr int SLOT = NF <= NMF ? Fi : Fi * NMF / NF;      // one per MFMA shadow while they fit, else spread evenly
          if constexpr (SLOT == I) {
            if constexpr (Fi < NFR) {
              // fragments of the next k-step: same stage for k-steps 1..3, the other stage (tile t+1, k-step 0) in k-step 3
              if constexpr (KS < 3) frag_read(std::integral_constant<int, FS ^ 1>{}, cs_c, std::integral_constant<int, KS + 1>{}, f_c);
              else frag_read(std::integral_constant<int, FS ^ 1>{}, std::integral_constant<int, CS ^ 1>{}, c0{}, f_c);
            } else {
              constexpr int J = KS * WPK + (Fi - NFR);
              G4_WAIT_VM(2 * NL - 1);             // the oldest load in flight (vector J of tile t+1) has landed
              write_one(gs_c, std::integral_constant<int, CS ^ 1>{}, std::integral_constant<int, J>{});
              load_one(gs_c, std::integral_constant<int, J>{});
            }
          }
        });
      });
      if constexpr (KS == 2) {
        advance();                              // (all NL reloads of tile t+3 are issued by now)
        G4_WAIT_LGKM(0);
        G4_BARRIER();
      } else if constexpr (KS == 3) {
        G4_WAIT_LGKM(0);
      } else {
        G4_WAIT_LGKM(NWR);                      // the NFR fragment reads are older than this k-step's NWR writes
      }
    });
  };

  int t = 0;
  for (; t + 1 < nk; t += 2) {
    tile(c0{}, c1{});
    tile(c1{}, c0{});
  }
  if (t < nk) tile(c0{}, c1{});
  G4_WAIT_VM(0);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> any other reader

  const int m_base = tm * BM + wm * 32 * FM, n_base = tn * BN + wn * 32 * FN;
  if (p.split_k > 1) {
    float* ws = p.splitk_ws + (int64_t)ksplit * p.M * p.N;
    const int hsel = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = m_base + b * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n_base + a * 32 + 8 * q + hsel;
          if (n < p.N)
            *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) =
                make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
    }
    return;
  }
  const float pre_ln[2 * FM] = {};
  epilogue<FN, FM>(p, acc, m_base, n_base, lane, 0, pre_ln, false);
}

template <int FM, int FN, int MODE>
int launch4(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 64 * FM, BN = 64 * FN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * ROWB;
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<FM, FN, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm4: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const int nsplit = d.split_k > 1 ? d.split_k : 1;
  dim3 grid((unsigned)(ntm * ntn), (unsigned)nsplit, 1);
  hipLaunchKernelGGL((gemm4_kernel<FM, FN, MODE>), grid, dim3(256), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm4 launch");
  if (nsplit > 1) return avsd_gemm_splitk_reduce(d, s);
  return AVSD_OK;
}

}  // namespace

int avsd_gemm_dispatch_asm(const avsd_gemm_desc& d, hipStream_t s) {
  AVSD_REQUIRE(d.mode == AVSD_GEMM_PLAIN && !d.A2 && d.batch == 1 && !(d.flags & AVSD_GEMM_X2) && d.K % 64 == 0,
               "gemm/asm tiles: PLAIN single-source 16-bit operands with K %% 64 == 0 (got mode %d, K %d)", d.mode, d.K);
  AVSD_REQUIRE((double)d.M * d.lda * 2.0 < 2147483648.0 && (double)d.N * d.ldw * 2.0 < 2147483648.0, "gemm/asm tiles: operands must be < 2 GiB");
  AVSD_REQUIRE(d.split_k <= 1 || (d.splitk_ws && d.split_k <= d.K / 64 && !(d.flags & AVSD_GEMM_GEGLU)), "gemm/asm tiles: bad split_k %d", d.split_k);
  switch (d.tile - AVSD_GEMM_TILE_ASM_FIRST) {
    case 0: return launch4<4, 4, AVSD_GEMM_PLAIN>(d, s);     // 256 x 256, 128 x 128 per wave, 144 KB
    case 1: return launch4<4, 2, AVSD_GEMM_PLAIN>(d, s);     // 256 x 128, 108 KB
    case 2: return launch4<2, 4, AVSD_GEMM_PLAIN>(d, s);     // 128 x 256
    case 3: return launch4<2, 2, AVSD_GEMM_PLAIN>(d, s);     // 128 x 128, 72 KB: two workgroups per CU
    default: AVSD_REQUIRE(false, "gemm/asm tiles: unknown tile %d", d.tile);
  }
}
