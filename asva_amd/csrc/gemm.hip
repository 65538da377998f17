// bf16 MFMA GEMM family for gfx950: plain linear / 1x1 conv, temporal-mix linear, implicit-GEMM
// 3x3 convolution — one kernel template, three A-loaders, one fused f32 epilogue.
//
// Math is out^T = W . A'^T: the weight tile is the MFMA "A" operand and the activation tile the
// "B" operand of v_mfma_f32_32x32x16_bf16, so every lane ends up holding 4 CONSECUTIVE output
// columns of one output row per accumulator quad -> 8-byte bf16 stores / 8-byte residual loads.
//
// Tile: BM x BN x 64, 256 threads = 4 waves (2 along M x 2 along N), register-staged double
// buffer in LDS (one barrier per K tile), rows padded to 72 elements (144 B): the 16-lane groups
// of ds_read_b128 then fall on 16 distinct 4-bank slots -> conflict free.
//
// Replaces (reference file:line): nn.Linear / nn.Conv2d calls at avgen/models/unets/utils.py:37-38,
// 53,123-131,159; ff_spatio_audio_temp_transformer_3d.py:66,92,276; ff_spatio_temp_resnet_3d.py:132,
// 148,159 (see include/avsd.h for the per-mode mapping).
#include "avsd_common.h"
#include "gemm_common.h"

namespace {

constexpr int LDS_STRIDE = BK + 8;  // elements

struct RowInfo {
  // PLAIN: o0 = m*lda, o1 = m*lda2.  TMIX: o0/o1/o2 = source-row offsets of the 3 segments.
  // CONV3: o0 = image index, hb/wb = top-left input coordinate of the 3x3 window.
  int64_t o0, o1, o2;
  int hb, wb;
  bool valid;
};

template <int MODE>
__device__ __forceinline__ RowInfo make_row(const avsd_gemm_desc& p, int m) {
  RowInfo r;
  r.valid = m < p.M;
  r.o0 = r.o1 = r.o2 = 0;
  r.hb = r.wb = 0;
  if (!r.valid) return r;
  if (MODE == AVSD_GEMM_PLAIN) {
    r.o0 = (int64_t)m * p.lda;
    r.o1 = (int64_t)m * p.lda2;
  } else if (MODE == AVSD_GEMM_TMIX) {
    const int f = (m / p.hw) % p.frames;
    r.o0 = (int64_t)(m - f * p.hw) * p.lda;           // frame 0
    r.o1 = (int64_t)(f > 0 ? m - p.hw : m) * p.lda;   // previous frame (clamped)
    r.o2 = (int64_t)m * p.lda;                         // current frame
  } else {
    const int per = p.ho * p.wo;
    const int n = m / per;
    const int rem = m - n * per;
    const int oh = rem / p.wo;
    const int ow = rem - oh * p.wo;
    r.o0 = n;
    r.hb = oh * p.stride - p.pad;
    r.wb = ow * p.stride - p.pad;
  }
  return r;
}

template <int MODE>
__device__ __forceinline__ uint4 load_a(const avsd_gemm_desc& p, const h16_t* A, const h16_t* A2,
                                        const RowInfo& r, int k0) {
  uint4 z = make_uint4(0, 0, 0, 0);
  if (!r.valid || k0 >= p.K) return z;
  const h16_t* ptr;
  if (MODE == AVSD_GEMM_PLAIN) {
    ptr = (k0 < p.k_split) ? (A + r.o0 + k0) : (A2 + r.o1 + (k0 - p.k_split));
  } else if (MODE == AVSD_GEMM_TMIX) {
    const int seg = k0 / p.cseg;
    const int kk = k0 - seg * p.cseg;
    const int64_t o = r.o2 + (seg == 0 ? r.o0 - r.o2 : 0) + (seg == 1 ? r.o1 - r.o2 : 0);
    ptr = A + o + kk;
  } else {
    const int tap = k0 / p.cin;
    const int c = k0 - tap * p.cin;
    const int kh = tap / 3;
    const int kw = tap - kh * 3;
    const int hi = r.hb + kh;
    const int wi = r.wb + kw;
    const int hin = p.hs << p.ups;
    const int win = p.ws << p.ups;
    if (tap >= 9 || hi < 0 || hi >= hin || wi < 0 || wi >= win) return z;
    const int64_t pix = ((int64_t)r.o0 * p.hs + (hi >> p.ups)) * p.ws + (wi >> p.ups);
    ptr = A + pix * p.lda + c;
  }
  return *reinterpret_cast<const uint4*>(ptr);
}

template <int BM, int BN, int MODE, int EX = EPI_PLAIN>
__global__ __launch_bounds__(256) void gemm_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16_t* sA = reinterpret_cast<h16_t*>(smem);      // [2][BM][LDS_STRIDE]
  h16_t* sW = sA + 2 * BM * LDS_STRIDE;             // [2][BN][LDS_STRIDE]

  constexpr int NA = BM / 32;  // 16-byte vectors per thread per A tile
  constexpr int NW = BN / 32;
  constexpr int FM = BM / 64;  // 32-wide fragments per wave along M
  constexpr int FN = BN / 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1;
  const int wn = wave >> 1;

  // ---- block -> tile, XCD-aware: consecutive tiles (same A rows) share one XCD's L2 ----------
  const int ntm = (p.M + BM - 1) / BM;
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  int wg;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Each XCD owns a contiguous range of `wg`.  Default: row-major tiles, so an XCD covers a band of M and its L2
  // fetches that band of A once while EVERY XCD streams all of W.  AVSD_GEMM_XCD_N: column-major, an XCD covers a
  // band of N — W is fetched once chip-wide and A by every XCD (the host picks whichever operand is larger).
  const bool nmaj = (p.flags & AVSD_GEMM_XCD_N) != 0;
  const int tn = nmaj ? wg / ntm : wg % ntn;
  const int tm = nmaj ? wg % ntm : wg / ntn;

  const int64_t bz = blockIdx.z;
  const h16_t* A = reinterpret_cast<const h16_t*>(p.A) + bz * p.batch_stride_a;
  const h16_t* A2 = p.A2 ? reinterpret_cast<const h16_t*>(p.A2) + bz * p.batch_stride_a : nullptr;
  const h16_t* W = reinterpret_cast<const h16_t*>(p.W) + bz * p.batch_stride_w;

  // ---- per-thread staging rows -----------------------------------------------------------------
  const int kv = (tid & 7) * 8;   // k offset of this thread's vector inside a K tile
  const int srow = tid >> 3;      // 0..31
  RowInfo ra[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) ra[i] = make_row<MODE>(p, tm * BM + srow + 32 * i);
  int64_t wo[NW];
  bool wv[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int n = tn * BN + srow + 32 * i;
    wv[i] = n < p.N;
    wo[i] = (int64_t)n * p.ldw;
  }

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  uint4 rga[NA], rgw[NW];
  const int nk = (p.K + BK - 1) / BK;

  auto gload = [&](int kt) {
    const int k0 = kt * BK + kv;
#pragma unroll
    for (int i = 0; i < NA; ++i) rga[i] = load_a<MODE>(p, A, A2, ra[i], k0);
#pragma unroll
    for (int i = 0; i < NW; ++i)
      rgw[i] = (wv[i] && k0 < p.K) ? *reinterpret_cast<const uint4*>(W + wo[i] + k0) : make_uint4(0, 0, 0, 0);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<uint4*>(sA + (buf * BM + srow + 32 * i) * LDS_STRIDE + kv) = rga[i];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      *reinterpret_cast<uint4*>(sW + (buf * BN + srow + 32 * i) * LDS_STRIDE + kv) = rgw[i];
  };

  gload(0);
  lstore(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const h16_t* bA = sA + (buf * BM + wm * (BM / 2) + frow) * LDS_STRIDE + fk;
    const h16_t* bW = sW + (buf * BN + wn * (BN / 2) + frow) * LDS_STRIDE + fk;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      h16x8 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const h16x8*>(bA + b * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const h16x8*>(bW + a * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma32x32x16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  const float no_pre[2 * FM] = {};
  epilogue<FN, FM, false, EX>(p, acc, tm * BM + wm * (BM / 2), tn * BN + wn * (BN / 2), lane, bz, no_pre, false);
}

template <int BM, int BN, int MODE, int EX = EPI_PLAIN>
int launch(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(h16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, MODE, EX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  dim3 grid((unsigned)(ntm * ntn), 1, (unsigned)d.batch);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE, EX>), grid, dim3(256), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm launch");
  return AVSD_OK;
}


// =====================================================================================================
// v2 main loop: LDS-direct loads (buffer_load_dwordx4 ... lds), STAGES-deep LDS ring, counted vmcnt.
//
//  * every wave-instruction moves 64 lanes x 16 B = 1 KiB from global straight into LDS at
//    M0 + lane*16 (no VGPR staging, asynchronous); out-of-range lanes (row >= M, conv padding, K tail)
//    use a voffset beyond num_records and the hardware writes zeros (probed: tools/probes/glds_probe.hip)
//  * LDS tile image: rows unpadded (128 B per row of 64 bf16), two rows = one 256-B line L; the 16-byte
//    slot index inside a line is XOR-ed with (L & 15).  The permutation is applied on the SOURCE side
//    (which (row, k-chunk) a lane fetches) and again on the fragment read, so the LDS destination stays
//    lane-linear as the instruction requires, and ds_read_b128's 16-lane groups hit 16 distinct slots.
//  * tile kt+STAGES-1 is issued while tile kt is consumed: s_waitcnt vmcnt(N) with N = loads of the
//    tiles still allowed in flight, then ONE s_barrier per K tile (raw, so the compiler adds no vmcnt(0)).
// =====================================================================================================
constexpr unsigned OOB = 0x80000000u;   // every tensor handed to v2 is < 2 GiB (checked on the host)

struct RowInfo32 {
  int o0, o1, o2;   // element offsets (PLAIN: m*lda, m*lda2; TMIX: three source rows; CONV3: image index)
  int hb, wb;
  bool valid;
};

template <int MODE>
__device__ __forceinline__ RowInfo32 make_row32(const avsd_gemm_desc& p, int m) {
  RowInfo32 r;
  r.valid = m < p.M;
  r.o0 = r.o1 = r.o2 = r.hb = r.wb = 0;
  if (!r.valid) return r;
  if (MODE == AVSD_GEMM_PLAIN) {
    r.o0 = m * p.lda;
    r.o1 = m * p.lda2;
  } else if (MODE == AVSD_GEMM_TMIX) {
    const int f = (m / p.hw) % p.frames;
    r.o0 = (m - f * p.hw) * p.lda;
    r.o1 = (f > 0 ? m - p.hw : m) * p.lda;
    r.o2 = m * p.lda;
  } else {
    const int per = p.ho * p.wo;
    const int n = m / per;
    const int rem = m - n * per;
    const int oh = rem / p.wo;
    r.o0 = n;
    r.hb = oh * p.stride - p.pad;
    r.wb = (rem - oh * p.wo) * p.stride - p.pad;
  }
  return r;
}

// byte offset of the 16-byte vector (row r, k0..k0+7) inside the A buffer, or OOB
template <int MODE>
__device__ __forceinline__ unsigned a_voffset(const avsd_gemm_desc& p, const RowInfo32& r, int k0) {
  if (!r.valid || k0 >= p.K) return OOB;
  if (MODE == AVSD_GEMM_PLAIN) {
    return (unsigned)((k0 < p.k_split) ? (r.o0 + k0) : (r.o1 + (k0 - p.k_split))) * 2u;
  } else if (MODE == AVSD_GEMM_TMIX) {
    const int seg = k0 / p.cseg;
    const int kk = k0 - seg * p.cseg;
    // (two independent selects against 0: a 3-way select over struct fields is lowered to an indexed stack load)
    const int o = r.o2 + (seg == 0 ? r.o0 - r.o2 : 0) + (seg == 1 ? r.o1 - r.o2 : 0);
    return (unsigned)(o + kk) * 2u;
  } else {
    const int tap = k0 / p.cin;
    const int c = k0 - tap * p.cin;
    const int kh = tap / 3;
    const int kw = tap - kh * 3;
    const int hi = r.hb + kh;
    const int wi = r.wb + kw;
    if (tap >= 9 || hi < 0 || hi >= (p.hs << p.ups) || wi < 0 || wi >= (p.ws << p.ups)) return OOB;
    return (unsigned)(((r.o0 * p.hs + (hi >> p.ups)) * p.ws + (wi >> p.ups)) * p.lda + c) * 2u;
  }
}

// LW > 0 adds LW loader waves to the WM x WN MFMA waves: they alone issue the global->LDS loads (and own the
// per-piece address state), so the ~100-cycle issue cost of each 1-KiB LDS-DMA piece (every wave of the block pushing
// its pieces into the texture path right after the barrier) no longer sits in front of the MFMA waves' matrix work.
// All waves meet at the one barrier per K tile; loaders wait for their loads to land before it.
// X2 (AVSD_GEMM_X2, split precision): every operand is a (main, rest) pair of planes.  A stage holds both planes of both
// operands ([A | W | A rest | W rest]; the rest planes are fetched with the same per-lane offsets through a second buffer
// descriptor), and every fragment pair contributes three MFMAs: W.A + Wr.A + W.Ar.
// waves per SIMD the tile's LDS footprint admits, capped at 3: the second __launch_bounds__ argument (HIP: minimum waves per execution
// unit).  Without it the 128 x 128 x 4-wave tile (64 KB: two workgroups per CU) compiled to 193 + 64 = 257 registers — one wave per
// SIMD, half its occupancy — after an unrelated epilogue change.
constexpr int gemm2_min_waves(int bm, int bn, int stages, bool x2, int waves) {
  const int lds = stages * (bm + bn) * 128 * (x2 ? 2 : 1);
  const int wgs = (160 * 1024) / lds < 1 ? 1 : (160 * 1024) / lds;
  const int w = (wgs * waves + 3) / 4;
  return w > 3 ? 3 : w;
}
// MODE_SUBPIX: the sub-pixel form of nearest-2x upsample + 3x3 convolution (AVSD_GEMM_CONV3 descriptors with ups = 2, include/avsd.h) — an
// A-loader mode of its own, so that the CONV3 kernels keep the code (and the registers) they had.
// EX (gemm_common.h): EPI_REST = this kernel also stores the rest plane of its 16-bit output (AVSD_GEMM_OUT_REST; IEEE-half build only).
constexpr int MODE_SUBPIX = 3;
template <int BM, int BN, int WM, int WN, int STAGES, int MODE, int LW = 0, bool X2 = false, int EX = EPI_PLAIN>
__global__ __launch_bounds__(64 * (WM * WN + LW), gemm2_min_waves(BM, BN, STAGES, X2, WM * WN + LW)) void gemm2_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem2[];
  constexpr int NC = WM * WN;               // MFMA waves
  constexpr int NWAVES = LW > 0 ? LW : NC;  // waves that issue loads
  constexpr int A_BYTES = BM * 128;
  constexpr int W_BYTES = BN * 128;
  constexpr int PLANE_BYTES = A_BYTES + W_BYTES;
  constexpr int STAGE_BYTES = PLANE_BYTES * (X2 ? 2 : 1);
  constexpr int PA = (BM / 8) / NWAVES;   // 1-KiB pieces of the A tile per loading wave
  constexpr int PW = (BN / 8) / NWAVES;
  static_assert(PA * NWAVES * 8 == BM && PW * NWAVES * 8 == BN, "tile rows must split evenly into 1-KiB pieces per wave");
  constexpr int LPT = (PA + PW) * (X2 ? 2 : 1);   // loads per tile per loading wave
  static_assert((STAGES - 2) * LPT < 64, "vmcnt is a 6-bit counter");
  constexpr int FM = BM / WM / 32;
  constexpr int FN = BN / WN / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LW == 0 || wave_all >= NC;
  const int wave = LW == 0 ? wave_all : (wave_all >= NC ? wave_all - NC : 0);   // index among the loading waves
  const int wm = wave_all % WM;
  const int wn = (wave_all / WM) % WN;

  const int ntm = (p.M + BM - 1) / BM;
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  const int nsplit = p.split_k > 1 ? p.split_k : 1;
  // Workgroups go to the 8 XCDs round-robin by their LINEAR id (x fastest, then y; tools/probes/xcd_map_probe.hip):
  // give every XCD one contiguous range of (tile, K-slice) work items, the slices of a tile adjacent, so the banding
  // below also holds for split-K launches (gridDim.x = tiles, gridDim.y = slices).
  int wg, ksplit;
  {
    const int total = nwg * nsplit;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    wg = c / nsplit;
    ksplit = c - wg * nsplit;
  }
  // Each XCD owns a contiguous range of `wg`.  Default: row-major tiles, so an XCD covers a band of M and its L2
  // fetches that band of A once while EVERY XCD streams all of W.  AVSD_GEMM_XCD_N: column-major, an XCD covers a
  // band of N — W is fetched once chip-wide and A by every XCD (the host picks whichever operand is larger).
  int tm, tn;
  tile_of_item(wg, ntm, ntn, (p.flags & AVSD_GEMM_XCD_N) != 0, p.raster_g, tm, tn);
  const int64_t bz = blockIdx.z;
  // split-K: this workgroup owns K tiles [kt0, kt1)
  const int nk_all = (p.K + BK - 1) / BK;
  const int per_split = (nk_all + nsplit - 1) / nsplit;
  const int kt0 = ksplit * per_split;
  const int kt1 = min(nk_all, kt0 + per_split);

  const h16_t* Ab = reinterpret_cast<const h16_t*>(p.A) + bz * p.batch_stride_a;
  const h16_t* A2b = p.A2 ? reinterpret_cast<const h16_t*>(p.A2) + bz * p.batch_stride_a : Ab;
  const h16_t* Wb = reinterpret_cast<const h16_t*>(p.W) + bz * p.batch_stride_w;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)A2b, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);
  // rest planes (X2): same offsets, other base
  const __amdgpu_buffer_rsrc_t rsAr = __builtin_amdgcn_make_buffer_rsrc((void*)(Ab + (X2 ? p.a_lo : 0)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2r = __builtin_amdgcn_make_buffer_rsrc((void*)(A2b + (X2 ? (p.A2 ? p.a2_lo : p.a_lo) : 0)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsWr = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + (X2 ? p.w_lo : 0)), 0, 0x7fffffff, 0x00020000);

  // ---- this lane's (row, k-chunk) for each of its pieces: line L = piece*4 + lane/16, slot = lane%16 ----
  // (scalars in separate statically-indexed arrays: a struct array selected by a runtime field goes to scratch)
  int ro0[PA], ro1[PA], ro2[PA], rhb[PA], rwb[PA], kca[PA];
  bool rvalid[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int L = (wave + j * NWAVES) * 4 + (lane >> 4);
    const int x = (lane & 15) ^ (L & 15);
    const RowInfo32 r = make_row32<MODE>(p, tm * BM + 2 * L + (x >> 3));
    ro0[j] = r.o0; ro1[j] = r.o1; ro2[j] = r.o2; rhb[j] = r.hb; rwb[j] = r.wb; rvalid[j] = r.valid;
    kca[j] = (x & 7) * 8;
  }
  int wo[PW], kcw[PW];
  bool wv[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int L = (wave + j * NWAVES) * 4 + (lane >> 4);
    const int x = (lane & 15) ^ (L & 15);
    const int n = tn * BN + 2 * L + (x >> 3);
    wv[j] = n < p.N;
    kcw[j] = (x & 7) * 8;
    wo[j] = n * p.ldw + kcw[j];
  }

  // K-tile decode state of the NEXT tile to issue, kept wave-uniform (scalar registers) and advanced
  // incrementally.  Per piece, (abase, aok) = element offset of the lane's vector at channel/segment offset 0
  // and its validity; they change only when the tile enters a new conv tap / temporal segment / second
  // source (`rebase`, a wave-uniform branch), so the per-tile address work is one add per load.
  // Fast path needs every K tile inside one tap / segment (cin % 64 == 0, cseg % 64 == 0).
  const bool fast = (MODE == AVSD_GEMM_PLAIN) || (MODE == AVSD_GEMM_TMIX && p.cseg % BK == 0) ||
                    ((MODE == AVSD_GEMM_CONV3 || MODE == MODE_SUBPIX) && p.cin % BK == 0);
  // Rotated K walk (AVSD_GEMM_KROT, see gemm4.hip): row bands (tm) that share a column band of W start at different K tiles of
  // the slice and wrap — the weights of the low-resolution layers stream from HBM inside a step, and a lockstep walk is one chain
  // of round trips.  Deterministic; only the f32 summation order of a band rotates.
  const int nk_slice = max(kt1 - kt0, 0);
  const int krot = ((p.flags & AVSD_GEMM_KROT) && nk_slice > 1 && ntm > 1) ? (int)(((long long)tm * nk_slice) / ntm) : 0;
  // MODE_SUBPIX: a 2 x 2 kernel on the ORIGINAL image per output-pixel parity (dy, dx); this column tile lies inside one parity
  // (cout % BN == 0, checked by launch2), whose input window starts at (y + dy - 1, x + dx - 1): the gather of a 2 x 2 convolution
  // with padding (1 - dy, 1 - dx)
  constexpr bool subpix = MODE == MODE_SUBPIX;
  constexpr int KS = subpix ? 2 : 3;                  // kernel rows / columns walked by the tap decode
  const int ush = subpix ? 0 : p.ups;
  const int sub_par = subpix ? (tn * BN) / (p.N >> 2) : 0;
  const int sub_dy = sub_par >> 1, sub_dx = sub_par & 1;
  int i_kbase = 0;          // first k of the tile
  int i_c0 = 0;             // PLAIN: = kbase; CONV3: channel offset inside the tap; TMIX: offset inside the segment
  int i_kh = 0, i_kw = 0, i_seg = 0;
  bool i_second = false;
  auto seek = [&](int kbase) {            // decode state of the tile that starts at element `kbase` of K
    i_kbase = kbase;
    if (MODE == AVSD_GEMM_PLAIN) {
      i_c0 = i_kbase;
      i_second = p.A2 != nullptr && i_kbase >= p.k_split;
    } else if (MODE == AVSD_GEMM_TMIX) {
      i_seg = i_kbase / p.cseg;
      i_c0 = i_kbase - i_seg * p.cseg;
    } else {
      const int tap = i_kbase / p.cin;
      i_c0 = i_kbase - tap * p.cin;
      i_kh = tap / KS;
      i_kw = tap - i_kh * KS;
    }
  };
  seek((kt0 + krot) * BK);
  const int hin = p.hs << ush, win = p.ws << ush;
  int abase[PA];
  bool aok[PA];

  auto rebase = [&]() {
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      if (MODE == AVSD_GEMM_PLAIN) {
        abase[j] = (i_second ? ro1[j] - p.k_split : ro0[j]) + kca[j];
        aok[j] = rvalid[j];
      } else if (MODE == AVSD_GEMM_TMIX) {
        const int o = ro2[j] + (i_seg == 0 ? ro0[j] - ro2[j] : 0) + (i_seg == 1 ? ro1[j] - ro2[j] : 0);
        abase[j] = o + kca[j];
        aok[j] = rvalid[j] && i_seg < 3;
      } else {
        const int hi = rhb[j] + i_kh + sub_dy;
        const int wi = rwb[j] + i_kw + sub_dx;
        aok[j] = rvalid[j] && i_kh < KS && (unsigned)hi < (unsigned)hin && (unsigned)wi < (unsigned)win;
        abase[j] = ((ro0[j] * p.hs + (hi >> ush)) * p.ws + (wi >> ush)) * p.lda + kca[j];
      }
    }
  };
  rebase();

  auto issue = [&](int stage) {
    unsigned char* sb = smem2 + stage * STAGE_BYTES;
    const int kbase = i_kbase;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      unsigned vo;
      if (!fast) {
        RowInfo32 r;
        r.o0 = ro0[j]; r.o1 = ro1[j]; r.o2 = ro2[j]; r.hb = rhb[j]; r.wb = rwb[j]; r.valid = rvalid[j];
        vo = a_voffset<MODE>(p, r, kbase + kca[j]);
      } else {
        const bool ok = aok[j] && (kbase + kca[j] < p.K);
        vo = ok ? (unsigned)(abase[j] + i_c0) * 2u : OOB;
      }
      lds_ptr_t dst = (lds_ptr_t)(sb + (wave + j * NWAVES) * 1024);
      if (i_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, (int)vo, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, (int)vo, 0, 0, 0);
      if constexpr (X2) {
        lds_ptr_t dr = (lds_ptr_t)(sb + PLANE_BYTES + (wave + j * NWAVES) * 1024);
        if (i_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2r, dr, 16, (int)vo, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAr, dr, 16, (int)vo, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const unsigned vo = (wv[j] && kbase + kcw[j] < p.K) ? (unsigned)(wo[j] + kbase) * 2u : OOB;
      lds_ptr_t dst = (lds_ptr_t)(sb + A_BYTES + (wave + j * NWAVES) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, (int)vo, 0, 0, 0);
      if constexpr (X2) {
        lds_ptr_t dr = (lds_ptr_t)(sb + PLANE_BYTES + A_BYTES + (wave + j * NWAVES) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWr, dr, 16, (int)vo, 0, 0, 0);
      }
    }
    // advance the decode state to the next K tile (rotated walk: from the last tile of the slice back to its first)
    i_kbase += BK;
    if (krot != 0 && i_kbase >= kt1 * BK) {
      seek(kt0 * BK);
      rebase();
    } else
    if (MODE == AVSD_GEMM_PLAIN) {
      i_c0 = i_kbase;
      if (!i_second && i_kbase >= p.k_split && i_kbase < p.K) { i_second = true; rebase(); }
    } else if (MODE == AVSD_GEMM_TMIX) {
      i_c0 += BK;
      if (i_c0 >= p.cseg) { i_c0 = 0; ++i_seg; rebase(); }
    } else {
      i_c0 += BK;
      if (i_c0 >= p.cin) {
        i_c0 = 0;
        if (++i_kw == KS) { i_kw = 0; ++i_kh; }
        rebase();
      }
    }
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read addressing: row r of a tile -> line r>>1, slot ((r&1)<<3 | chunk) ^ (line & 15)
  int a_line[FM], a_sw[FM], a_hi[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int r = wm * (BM / WM) + b * 32 + (lane & 31);
    a_line[b] = (r >> 1) * 256;
    a_sw[b] = (r >> 1) & 15;
    a_hi[b] = (r & 1) << 3;
  }
  int w_line[FN], w_sw[FN], w_hi[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int r = wn * (BN / WN) + a * 32 + (lane & 31);
    w_line[a] = (r >> 1) * 256;
    w_sw[a] = (r >> 1) & 15;
    w_hi[a] = (r & 1) << 3;
  }
  const int chalf = lane >> 5;

  const int nk = max(kt1 - kt0, 0);
  auto wait_stage = [&](int kt) {   // tiles kt+1 .. kt+STAGES-2 may stay in flight (fewer at the end of the K range)
    wait_tiles_ahead<STAGES - 2, LPT>(nk - 1 - kt);
  };
  if (is_loader) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) issue(s);
  }
  // LayerNorm statistics of this wave's rows, fetched while the prologue tiles are in flight (their round trip would
  // otherwise sit in the epilogue)
  float pre_ln[2 * FM] = {};
  const bool pre = (p.flags & AVSD_GEMM_LNFUSE) != 0 && !(LW > 0 && is_loader);
  if (pre) {
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = tm * BM + wm * (BM / WM) + b * 32 + (lane & 31);
      pre_ln[2 * b] = 1.f; pre_ln[2 * b + 1] = 0.f;
      if (m < p.M) ln_row_stats(p, m, bz, pre_ln[2 * b], pre_ln[2 * b + 1]);
    }
  }
  if (LW > 0 && is_loader) {
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt);
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES);
    }
    return;
  }

  for (int kt = 0; kt < nk; ++kt) {
    if (LW == 0) wait_stage(kt);
    __builtin_amdgcn_s_barrier();
    if (LW == 0 && kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES);

    const unsigned char* sA = smem2 + (kt % STAGES) * STAGE_BYTES;
    const unsigned char* sW = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + chalf;
      h16x8 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b)
        xf[b] = *reinterpret_cast<const h16x8*>(sA + a_line[b] + (((a_hi[b] | c) ^ a_sw[b]) << 4));
#pragma unroll
      for (int a = 0; a < FN; ++a)
        wf[a] = *reinterpret_cast<const h16x8*>(sW + w_line[a] + (((w_hi[a] | c) ^ w_sw[a]) << 4));
      if constexpr (X2) {
        h16x8 xr[FM], wr[FN];
#pragma unroll
        for (int b = 0; b < FM; ++b)
          xr[b] = *reinterpret_cast<const h16x8*>(sA + PLANE_BYTES + a_line[b] + (((a_hi[b] | c) ^ a_sw[b]) << 4));
#pragma unroll
        for (int a = 0; a < FN; ++a)
          wr[a] = *reinterpret_cast<const h16x8*>(sW + PLANE_BYTES + w_line[a] + (((w_hi[a] | c) ^ w_sw[a]) << 4));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FM; ++b) {
            acc[a][b] = mfma32x32x16(wr[a], xf[b], acc[a][b], 0, 0, 0);
            acc[a][b] = mfma32x32x16(wf[a], xr[b], acc[a][b], 0, 0, 0);
            acc[a][b] = mfma32x32x16(wf[a], xf[b], acc[a][b], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      } else {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma32x32x16(wf[a], xf[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  if (p.split_k > 1) {
    // raw f32 partial tile -> ws[split][m][n]; the reduce kernel applies the epilogue
    float* ws = p.splitk_ws + (int64_t)ksplit * p.M * p.N;
    const int hsel = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = tm * BM + wm * (BM / WM) + b * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = tn * BN + wn * (BN / WN) + a * 32 + 8 * q + hsel;
          if (n < p.N)
            *reinterpret_cast<float4*>(ws + (int64_t)m * p.N + n) =
                make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
    }
    // splitk_reduce_kernel folds the slabs and applies the epilogue.  The fold below is NEVER taken (avsd_gemm_bf16 refuses a split_k
    // beyond the K tiles): it is what is left of round 3's in-launch reduction, kept because of what it does to the register allocator —
    // with a path that redefines every accumulator between the main loop and the epilogue, hipcc 7.2 gives the same kernels 12-56
    // fewer VGPRs (128x128x8w: 201 -> 168 = 3 waves per SIMD, 256x128 + loader waves: 137 -> 125 = 4, 256x320: 96 spilled
    // registers -> 4); without it the VAE decode lost 12 % and cfg 4 10 % between round 3 and round 4 (found with
    // -Rpass-analysis=kernel-resource-usage against the round-3 tree; same-box A/B in profiles/r4_regalloc_ab.txt).
    if (p.split_k != 0x7fffffff) return;
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int sl = 0; sl < nsplit; ++sl) {
      const float* slab = p.splitk_ws + (int64_t)sl * p.M * p.N;
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int m = min(tm * BM + wm * (BM / WM) + b * 32 + (lane & 31), p.M - 1);
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = min(tn * BN + wn * (BN / WN) + a * 32 + 8 * q + hsel, p.N - 4);
            const float4 v = *reinterpret_cast<const float4*>(slab + (int64_t)m * p.N + n);
            acc[a][b][4 * q] += v.x; acc[a][b][4 * q + 1] += v.y; acc[a][b][4 * q + 2] += v.z; acc[a][b][4 * q + 3] += v.w;
          }
      }
    }
  }
  if constexpr (subpix) {     // the same epilogue with the per-pixel output scatter compiled in (gemm_common.h EPI_SHUF)
    if constexpr (X2) epilogue_x2<FN, FM, true>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
    else epilogue_each<FN, FM, (64 * (WM * WN + LW) > 512), EPI_SHUF>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
  } else if constexpr (X2) epilogue_x2<FN, FM>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
  else if constexpr (EX == EPI_REST) {     // the same forms as below with the rest-plane store compiled in
    if constexpr (FN * FM >= 10) epilogue_each<FN, FM, (64 * (WM * WN + LW) > 512), EPI_REST>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
    else epilogue<FN, FM, (64 * (WM * WN + LW) > 512), EPI_REST>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
  } else if constexpr (FN * FM >= 10)      // 64 x 160 wave tiles (tile 19): the 10-fragment epilogue is not unrolled -> accumulators in scratch (gemm_common.h)
    epilogue_each<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
  else epilogue<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, tm * BM + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, bz, pre_ln, pre);
}

// out = epilogue(sum_s ws[s]) for split-K launches: one thread per 4 consecutive columns.  S = the slice count (2 / 4 / 8:
// all slab loads of a thread are issued before the first add — the kernel is one memory round trip deep instead of S;
// 7.4 -> ~4 us per launch, 73 launches per step) or 0 (any count, one load at a time).  The sum runs in slice order.
// EXTRA: the rest plane of a one-pass output (AVSD_GEMM_OUT_REST) and the per-pixel output scatter of AVSD_GEMM_CONV3 with ups = 2
// (gemm_common.h) — run-time options of a SEPARATE instantiation, so that the reduce of every other split-K launch keeps its registers
template <int S, bool EXTRA = false>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const avsd_gemm_desc p) {
  const int nq = p.N / 4;
  const int64_t total = (int64_t)p.M * nq;
  const h16_t* R1 = reinterpret_cast<const h16_t*>(p.res1);
  const h16_t* R2 = reinterpret_cast<const h16_t*>(p.res2);
  const bool x2 = (p.flags & AVSD_GEMM_X2) != 0;
  const bool out_rest = EXTRA && (p.flags & AVSD_GEMM_OUT_REST) != 0;     // one-pass product, rest plane of the output wanted (not part of the row statistics)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / nq);
    const int n = (int)(i - (int64_t)m * nq) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // epilogue operands first: their loads share the round trip of the slab loads
    float4 pre_bias = make_float4(0.f, 0.f, 0.f, 0.f), pre_vec = pre_bias, pre_r1f = pre_bias, pre_r2f = pre_bias;
    uint2 pre_r1 = make_uint2(0, 0), pre_r1lo = pre_r1, pre_r2 = pre_r1, pre_r2lo = pre_r1;
    if (p.bias) pre_bias = *reinterpret_cast<const float4*>(p.bias + n);
    if (p.rowvec) pre_vec = *reinterpret_cast<const float4*>(p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ldv + n);
    if (R1) {
      if (p.flags & AVSD_GEMM_RES1_F32) {
        pre_r1f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res1) + (int64_t)m * p.ldr1 + n);
      } else {
        pre_r1 = *reinterpret_cast<const uint2*>(R1 + (int64_t)m * p.ldr1 + n);
        if (x2) pre_r1lo = *reinterpret_cast<const uint2*>(R1 + p.res1_lo + (int64_t)m * p.ldr1 + n);
      }
    }
    if (R2) {
      if (p.flags & AVSD_GEMM_RES2_F32) {
        pre_r2f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res2) + (int64_t)m * p.ldr2 + n);
      } else {
        pre_r2 = *reinterpret_cast<const uint2*>(R2 + (int64_t)m * p.ldr2 + n);
        if (x2) pre_r2lo = *reinterpret_cast<const uint2*>(R2 + p.res2_lo + (int64_t)m * p.ldr2 + n);
      }
    }
    if constexpr (S > 0) {
      float4 sv[S];
#pragma unroll
      for (int s = 0; s < S; ++s) sv[s] = *reinterpret_cast<const float4*>(p.splitk_ws + ((int64_t)s * p.M + m) * p.N + n);
#pragma unroll
      for (int s = 0; s < S; ++s) { acc.x += sv[s].x; acc.y += sv[s].y; acc.z += sv[s].z; acc.w += sv[s].w; }
    } else {
      for (int s = 0; s < p.split_k; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p.splitk_ws + ((int64_t)s * p.M + m) * p.N + n);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    float v[4] = {p.alpha * acc.x, p.alpha * acc.y, p.alpha * acc.z, p.alpha * acc.w};
    if (p.flags & AVSD_GEMM_LNFUSE) {
      const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (int64_t)m * p.ln_nblk;
      float sm = 0.f, sq = 0.f;
      for (int j = 0; j < p.ln_nblk; ++j) { const float2 t = st[j]; sm += t.x; sq += t.y; }
      const float inv_k = 1.0f / (float)p.K;
      const float mean = sm * inv_k;
      const float rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
      const float4 cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
      if (p.ln_rowvec) {
        const float4 pv = *reinterpret_cast<const float4*>(pos_row(p, p.ln_rowvec, m) + n);
        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
      }
      v[0] = fmaf(v[0], rstd, -mean * rstd * cs.x); v[1] = fmaf(v[1], rstd, -mean * rstd * cs.y);
      v[2] = fmaf(v[2], rstd, -mean * rstd * cs.z); v[3] = fmaf(v[3], rstd, -mean * rstd * cs.w);
    }
    if (p.bias) { v[0] += pre_bias.x; v[1] += pre_bias.y; v[2] += pre_bias.z; v[3] += pre_bias.w; }
    if (p.rowvec) { v[0] += pre_vec.x; v[1] += pre_vec.y; v[2] += pre_vec.z; v[3] += pre_vec.w; }
    if (p.flags & AVSD_GEMM_GELU) {
      for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
    }
    if (R1) {
      if (p.flags & AVSD_GEMM_RES1_F32) {
        v[0] += pre_r1f.x; v[1] += pre_r1f.y; v[2] += pre_r1f.z; v[3] += pre_r1f.w;
      } else {
        v[0] += lo2f(pre_r1.x); v[1] += hi2f(pre_r1.x); v[2] += lo2f(pre_r1.y); v[3] += hi2f(pre_r1.y);
        if (x2) { v[0] += lo2f(pre_r1lo.x); v[1] += hi2f(pre_r1lo.x); v[2] += lo2f(pre_r1lo.y); v[3] += hi2f(pre_r1lo.y); }
      }
    }
    if (R2) {
      if (p.flags & AVSD_GEMM_RES2_F32) {
        v[0] += pre_r2f.x; v[1] += pre_r2f.y; v[2] += pre_r2f.z; v[3] += pre_r2f.w;
      } else {
        v[0] += lo2f(pre_r2.x); v[1] += hi2f(pre_r2.x); v[2] += lo2f(pre_r2.y); v[3] += hi2f(pre_r2.y);
        if (x2) { v[0] += lo2f(pre_r2lo.x); v[1] += hi2f(pre_r2lo.x); v[2] += lo2f(pre_r2lo.y); v[3] += hi2f(pre_r2lo.y); }
      }
    }
    int64_t mo = m;
    int no = n;
    if (EXTRA && p.mode == AVSD_GEMM_CONV3 && p.ups == 2) {      // (input pixel, parity x channel) -> (output pixel, channel), gemm_common.h EPI_SHUF
      int radd;
      shuf_col(p, n, radd, no);
      mo = shuf_row(p, m) + radd;
    }
    if (p.out_master) *reinterpret_cast<float4*>(p.out_master + mo * p.ldm + no) = make_float4(v[0], v[1], v[2], v[3]);
    const int64_t o = mo * p.ldc + no;
    if (p.flags & AVSD_GEMM_OUT_F32) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2 st;
      st.x = pack2h(v[0], v[1]);
      st.y = pack2h(v[2], v[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + o) = st;
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;     // x2: what the rest plane adds to the stored value
      if (x2 || out_rest) {
        uint2 sr;
        sr.x = pack2h(v[0] - lo2f(st.x), v[1] - hi2f(st.x));
        sr.y = pack2h(v[2] - lo2f(st.y), v[3] - hi2f(st.y));
        *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(p.out) + p.out_lo + o) = sr;
        if (x2) { e0 = lo2f(sr.x); e1 = hi2f(sr.x); e2 = lo2f(sr.y); e3 = hi2f(sr.y); }
      }
      if (p.flags & AVSD_GEMM_ROWSTATS) {
        // 8 consecutive threads hold one 32-column block of row m (N % 32 == 0, 256 % 8 == 0): fold in a fixed order
        float a0 = lo2f(st.x) + e0, a1 = hi2f(st.x) + e1;
        float a2 = lo2f(st.y) + e2, a3 = hi2f(st.y) + e3;
        if (p.stats_pos) {
          const float4 pv = *reinterpret_cast<const float4*>(pos_row(p, p.stats_pos, m) + n);
          a0 += pv.x; a1 += pv.y; a2 += pv.z; a3 += pv.w;
        }
        float sm = (a0 + a1) + (a2 + a3);
        float sq = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, a3 * a3)));
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
          sm += __shfl_xor(sm, off, 64);
          sq += __shfl_xor(sq, off, 64);
        }
        if ((threadIdx.x & 7) == 0)
          reinterpret_cast<float2*>(p.rowstats)[(int64_t)m * (p.N >> 5) + (n >> 5)] = make_float2(sm, sq);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int STAGES, int MODE, int LW = 0, bool X2 = false, int EX = EPI_PLAIN>
int launch2(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128 * (X2 ? 2 : 1);
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
  if (MODE == MODE_SUBPIX && (d.N / 4) % BN != 0) {
    avsd_set_error("gemm/conv3: ups = 2 needs whole column tiles per output-pixel parity: cout (%d) %% %d != 0 for this tile", d.N / 4, BN);
    return AVSD_EINVAL;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<BM, BN, WM, WN, STAGES, MODE, LW, X2, EX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm2: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const int nsplit = d.split_k > 1 ? d.split_k : 1;
  dim3 grid((unsigned)(ntm * ntn), (unsigned)nsplit, (unsigned)d.batch);
  hipLaunchKernelGGL((gemm2_kernel<BM, BN, WM, WN, STAGES, MODE, LW, X2, EX>), grid, dim3(64 * (WM * WN + LW)), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm2 launch");
  if (nsplit > 1) {
    const int64_t total = (int64_t)d.M * (d.N / 4);
    int64_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    if (MODE == MODE_SUBPIX || (d.flags & AVSD_GEMM_OUT_REST)) {      // the reduce with the rest-plane store / output scatter (any slice count)
      hipLaunchKernelGGL((splitk_reduce_kernel<0, true>), dim3((unsigned)g), dim3(256), 0, s, d);
    } else
    switch (nsplit) {
      case 2: hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3((unsigned)g), dim3(256), 0, s, d); break;
      case 4: hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)g), dim3(256), 0, s, d); break;
      case 8: hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3((unsigned)g), dim3(256), 0, s, d); break;
      case 3: hipLaunchKernelGGL(splitk_reduce_kernel<3>, dim3((unsigned)g), dim3(256), 0, s, d); break;
      case 5: hipLaunchKernelGGL(splitk_reduce_kernel<5>, dim3((unsigned)g), dim3(256), 0, s, d); break;
      default: hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    }
    AVSD_CHECK_LAUNCH("gemm split-K reduce launch");
  }
  return AVSD_OK;
}

template <int MODE, int EX = EPI_PLAIN>
int dispatch_tile(const avsd_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {
    case 1: return launch<128, 128, MODE, EX>(d, s);
    case 2: return launch<128, 64, MODE, EX>(d, s);
    case 3: return launch<64, 64, MODE, EX>(d, s);
    // v2 (LDS-direct ring): BM, BN, waves M x N, stages
    case 4: return launch2<128, 64, 2, 2, 3, MODE, 0, false, EX>(d, s);
    case 6: return launch2<128, 128, 2, 4, 3, MODE, 0, false, EX>(d, s);
    case 7: return launch2<64, 64, 2, 2, 4, MODE, 0, false, EX>(d, s);
    case 9: return launch2<256, 128, 4, 2, 3, MODE, 0, false, EX>(d, s);
    case 11: return launch2<128, 128, 2, 2, 2, MODE, 0, false, EX>(d, s);   // 64 KB: two 4-wave blocks per CU, 64x64 wave tiles
    case 12: return launch2<128, 64, 2, 2, 2, MODE, 0, false, EX>(d, s);    // 48 KB: three blocks per CU
    case 13: return launch2<64, 64, 2, 2, 2, MODE, 0, false, EX>(d, s);     // 32 KB: five blocks per CU (short-K GEMMs)
    case 14: return launch2<256, 128, 4, 2, 2, MODE, 0, false, EX>(d, s);   // 96 KB
    // full-row tiles for N = 320 / 640 / 1280 layers: the activation tile is fetched from L2 once per block
    case 17: return launch2<128, 320, 4, 2, 2, MODE, 0, false, EX>(d, s);   // 112 KB, 8 waves, 32x160 wave tiles
    // 256-row tiles for the widest layers: (BM + BN) / (BM * BN) global->LDS bytes per MFMA is what the texture path pays
    case 19: return launch2<256, 320, 4, 2, 2, MODE, 0, false, EX>(d, s);   // 144 KB, 8 waves, 64x160 wave tiles
    // the same tiles with 2 extra loader waves (LW): the MFMA waves issue no loads
    case 20: return launch2<256, 128, 4, 2, 3, MODE, 4, false, EX>(d, s);
    case 24: return launch2<128, 64, 2, 2, 3, MODE, 2, false, EX>(d, s);
    case 25: return launch2<64, 64, 2, 2, 4, MODE, 2, false, EX>(d, s);
    // deep rings for the weight-streaming low-resolution layers (M = 384 / 1536, K up to 23040): what bounds them is the
    // bytes in flight per CU against the ~2 us HBM round trip, so the ring takes all of LDS
    case 30: return launch2<128, 128, 2, 4, 4, MODE, 0, false, EX>(d, s);     // 128 KB
    case 31: return launch2<128, 256, 2, 4, 3, MODE, 0, false, EX>(d, s);     // 144 KB, weight-heavy tile
    // 256 x 160: the geometry the resident convolution does best with on the N = 320 layers (conv3r.hip tile 43) — N = 320 /
    // 640 / 960 / 1280 in whole column tiles, 98 FLOP per staged byte (128 x 128: 64)
    case AVSD_GEMM_TILE_256x160_8W: return launch2<256, 160, 8, 1, 3, MODE, 4, false, EX>(d, s);   // 156 KB, 32x160 wave tiles, 8 MFMA + 4 loader waves
    // (built, measured and dropped — never the tuner's pick on MI355X: 128x128 x 3 on 4 waves (5), 128x64 x 4 (10), the 128x320 / 64x320
    //  4-wave full-row tiles (15, 16), 256x256 (18), 128x128 / 256x64 / 64x320 with loader waves (26-28), the 5-deep 128x128 ring (29),
    //  96x320 x 2 (33), 256x160 on 4 MFMA waves (39); round 4, <= 2 table entries each once the asm tiles existed: 256x64 (8), the
    //  2-stage 256x128 / 3-stage 128x128 / 128x320 with loader waves (21-23), the 96x320 full-row tile (32))
    default:
      avsd_set_error("gemm: tile id %d is not built (gemm.hip dispatch_tile)", tile);
      return AVSD_EINVAL;
  }
}

// split-precision tiles: the subset whose doubled stage still fits LDS
template <int MODE>
int dispatch_tile_x2(const avsd_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {
    case 7: return launch2<64, 64, 2, 2, 4, MODE, 0, true>(d, s);      // 128 KB
    case 11: return launch2<128, 128, 2, 2, 2, MODE, 0, true>(d, s);   // 128 KB
    case 13: return launch2<64, 64, 2, 2, 2, MODE, 0, true>(d, s);     // 64 KB: two blocks per CU
    case 24: return launch2<128, 64, 2, 2, 3, MODE, 2, true>(d, s);    // 144 KB, loader waves
    case 25: return launch2<64, 64, 2, 2, 4, MODE, 2, true>(d, s);     // 128 KB, loader waves
    // split-precision-only shapes (ids above the 16-bit range): what a doubled 2-stage ring still holds
    case 34: return launch2<128, 160, 4, 1, 2, MODE, 0, true>(d, s);   // 144 KB: N = 320 layers in two column tiles, 32x160 wave tiles
    case 35: return launch2<256, 64, 4, 2, 2, MODE, 0, true>(d, s);    // 160 KB, 8 waves
    case 36: return launch2<128, 192, 2, 2, 2, MODE, 0, true>(d, s);   // 160 KB, 64x96 wave tiles
    default:
      avsd_set_error("gemm: AVSD_GEMM_X2 runs on tiles 7, 11, 13, 24, 25, 34, 35, 36 (got %d)", tile);
      return AVSD_EINVAL;
  }
}

// Wave-quantised cost model: 2 resident blocks per CU, relative per-tile MFMA efficiency.
int pick_tile(int M, int N, int batch) {
  static int num_cu = 0;
  if (num_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      num_cu = prop.multiProcessorCount;
    if (num_cu <= 0) num_cu = 256;
  }
  const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
  const double eff[3] = {1.0, 0.85, 0.62};
  const int slots[3] = {2 * num_cu, 3 * num_cu, 4 * num_cu};
  int best = 3;
  double best_cost = 1e300;
  for (int t = 0; t < 3; ++t) {
    const long blocks = (long)((M + bm[t] - 1) / bm[t]) * ((N + bn[t] - 1) / bn[t]) * batch;
    const long waves = (blocks + slots[t] - 1) / slots[t];
    // fractional last wave still costs a full tile time; earlier waves run `slots` tiles each
    const double cost = (double)waves * bm[t] * bn[t] / eff[t] * slots[t];
    if (cost < best_cost) { best_cost = cost; best = t + 1; }
  }
  return best;
}

}  // namespace

// The three A-loader modes instantiate ~90 kernels each: the build compiles this file once per mode (-DAVSD_GEMM_TU=0/1/2,
// only that mode's dispatcher) and once for the entry point (-DAVSD_GEMM_TU=3), in parallel; without the macro one
// translation unit carries everything.
int avsd_gemm_dispatch_plain(const avsd_gemm_desc& d, int tile, hipStream_t s);
int avsd_gemm_dispatch_tmix(const avsd_gemm_desc& d, int tile, hipStream_t s);
int avsd_gemm_dispatch_conv3(const avsd_gemm_desc& d, int tile, hipStream_t s);
// (IEEE-half build: a launch with AVSD_GEMM_OUT_REST runs the EPI_REST instantiation of its tile — separate kernels, so the ones every
//  other launch runs keep their registers; profiles/r6_epilogue_ab.txt)
#ifdef AVSD_F16
#define AVSD_DISPATCH_TILE(MODE) ((d.flags & AVSD_GEMM_OUT_REST) ? dispatch_tile<MODE, EPI_REST>(d, tile, s) : dispatch_tile<MODE>(d, tile, s))
#else
#define AVSD_DISPATCH_TILE(MODE) dispatch_tile<MODE>(d, tile, s)
#endif
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 0
int avsd_gemm_dispatch_plain(const avsd_gemm_desc& d, int tile, hipStream_t s) { return AVSD_DISPATCH_TILE(AVSD_GEMM_PLAIN); }
#endif
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 1
int avsd_gemm_dispatch_tmix(const avsd_gemm_desc& d, int tile, hipStream_t s) { return AVSD_DISPATCH_TILE(AVSD_GEMM_TMIX); }
#endif
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 2
int avsd_gemm_dispatch_conv3(const avsd_gemm_desc& d, int tile, hipStream_t s) { return AVSD_DISPATCH_TILE(AVSD_GEMM_CONV3); }
#endif
// sub-pixel upsample convolution (CONV3 descriptors with ups = 2): its own A-loader mode, one-pass and split-precision tiles
int avsd_gemm_dispatch_subpix(const avsd_gemm_desc& d, int tile, hipStream_t s);
int avsd_gemm_dispatch_x2_subpix(const avsd_gemm_desc& d, int tile, hipStream_t s);
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 6
int avsd_gemm_dispatch_subpix(const avsd_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {       // (the tiles the tuner is offered for these layers, asva_amd/ops.py SUBPIX_TILES)
    case 4: return launch2<128, 64, 2, 2, 3, MODE_SUBPIX>(d, s);
    case 6: return launch2<128, 128, 2, 4, 3, MODE_SUBPIX>(d, s);
    case 9: return launch2<256, 128, 4, 2, 3, MODE_SUBPIX>(d, s);
    case 11: return launch2<128, 128, 2, 2, 2, MODE_SUBPIX>(d, s);
    case 13: return launch2<64, 64, 2, 2, 2, MODE_SUBPIX>(d, s);
    case 17: return launch2<128, 320, 4, 2, 2, MODE_SUBPIX>(d, s);
    case 20: return launch2<256, 128, 4, 2, 3, MODE_SUBPIX, 4>(d, s);
    case 24: return launch2<128, 64, 2, 2, 3, MODE_SUBPIX, 2>(d, s);
    case 25: return launch2<64, 64, 2, 2, 4, MODE_SUBPIX, 2>(d, s);
    case 30: return launch2<128, 128, 2, 4, 4, MODE_SUBPIX>(d, s);
    case AVSD_GEMM_TILE_256x160_8W: return launch2<256, 160, 8, 1, 3, MODE_SUBPIX, 4>(d, s);
    default:
      avsd_set_error("gemm/conv3: ups = 2 runs on tiles 4, 6, 9, 11, 13, 17, 20, 24, 25, 30, 38 (got %d)", tile);
      return AVSD_EINVAL;
  }
}
int avsd_gemm_dispatch_x2_subpix(const avsd_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {
    case 7: return launch2<64, 64, 2, 2, 4, MODE_SUBPIX, 0, true>(d, s);
    case 11: return launch2<128, 128, 2, 2, 2, MODE_SUBPIX, 0, true>(d, s);
    case 13: return launch2<64, 64, 2, 2, 2, MODE_SUBPIX, 0, true>(d, s);
    case 24: return launch2<128, 64, 2, 2, 3, MODE_SUBPIX, 2, true>(d, s);
    case 25: return launch2<64, 64, 2, 2, 4, MODE_SUBPIX, 2, true>(d, s);
    case 34: return launch2<128, 160, 4, 1, 2, MODE_SUBPIX, 0, true>(d, s);
    case 35: return launch2<256, 64, 4, 2, 2, MODE_SUBPIX, 0, true>(d, s);
    default:
      avsd_set_error("gemm/conv3: ups = 2 with AVSD_GEMM_X2 runs on tiles 7, 11, 13, 24, 25, 34, 35 (got %d)", tile);
      return AVSD_EINVAL;
  }
}
#endif

int avsd_gemm_dispatch_x2_plain(const avsd_gemm_desc& d, int tile, hipStream_t s);
int avsd_gemm_dispatch_x2_tmix(const avsd_gemm_desc& d, int tile, hipStream_t s);
int avsd_gemm_dispatch_x2_conv3(const avsd_gemm_desc& d, int tile, hipStream_t s);
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 4
int avsd_gemm_dispatch_x2_plain(const avsd_gemm_desc& d, int tile, hipStream_t s) { return dispatch_tile_x2<AVSD_GEMM_PLAIN>(d, tile, s); }
int avsd_gemm_dispatch_x2_tmix(const avsd_gemm_desc& d, int tile, hipStream_t s) { return dispatch_tile_x2<AVSD_GEMM_TMIX>(d, tile, s); }
#endif
#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 5
int avsd_gemm_dispatch_x2_conv3(const avsd_gemm_desc& d, int tile, hipStream_t s) { return dispatch_tile_x2<AVSD_GEMM_CONV3>(d, tile, s); }
#endif

#if !defined(AVSD_GEMM_TU) || AVSD_GEMM_TU == 3
int avsd_gemm_dispatch_conv3r(const avsd_gemm_desc& d, hipStream_t s);      // conv3r.hip
int avsd_gemm_dispatch_asm(const avsd_gemm_desc& d, hipStream_t s);         // gemm4.hip
int avsd_gemm_dispatch_nstream(const avsd_gemm_desc& d, hipStream_t s);     // nstream.hip

// the reduce / epilogue launch of a split-K GEMM, for kernels outside this file that write the same slabs (conv3r.hip)
int avsd_gemm_splitk_reduce(const avsd_gemm_desc& d, hipStream_t s) {
  const int64_t total = (int64_t)d.M * (d.N / 4);
  int64_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (d.flags & AVSD_GEMM_OUT_REST) {
    hipLaunchKernelGGL((splitk_reduce_kernel<0, true>), dim3((unsigned)g), dim3(256), 0, s, d);
  } else
  switch (d.split_k) {
    case 2: hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    case 4: hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    case 8: hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    case 3: hipLaunchKernelGGL(splitk_reduce_kernel<3>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    case 5: hipLaunchKernelGGL(splitk_reduce_kernel<5>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    case 10: hipLaunchKernelGGL(splitk_reduce_kernel<10>, dim3((unsigned)g), dim3(256), 0, s, d); break;
    default: hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((unsigned)g), dim3(256), 0, s, d); break;
  }
  AVSD_CHECK_LAUNCH("gemm split-K reduce launch");
  return AVSD_OK;
}

extern "C" int avsd_gemm_bf16(const avsd_gemm_desc* dp, void* stream) {
  AVSD_REQUIRE(dp != nullptr, "gemm: null descriptor");
  avsd_gemm_desc d = *dp;
  AVSD_REQUIRE(d.A && d.W && d.out, "gemm: A, W and out must be non-null");
  AVSD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", d.M, d.N, d.K);
  AVSD_REQUIRE(!(d.flags & AVSD_GEMM_W_FRAG) || d.tile == AVSD_GEMM_TILE_NSTREAM, "gemm: fragment-ordered W (AVSD_GEMM_W_FRAG) is read by tile %d only (got %d)", AVSD_GEMM_TILE_NSTREAM, d.tile);
  if (d.flags & AVSD_GEMM_W_FRAG) d.ldw = d.K;
  AVSD_REQUIRE(d.K % 8 == 0 && d.ldw % 8 == 0 && d.ldw >= d.K, "gemm: K (%d) and ldw (%d) must be multiples of 8, ldw >= K", d.K, d.ldw);
  AVSD_REQUIRE(d.N % 4 == 0 && d.ldc % 4 == 0, "gemm: N (%d) and ldc (%d) must be multiples of 4", d.N, d.ldc);
  AVSD_REQUIRE(d.lda % 8 == 0, "gemm: lda (%d) must be a multiple of 8", d.lda);
  if (d.batch <= 0) d.batch = 1;
  AVSD_REQUIRE(d.raster_g >= 0 && d.raster_g <= 64, "gemm: raster_g (%d) must be 0 (library default) .. 64", d.raster_g);
  if (d.flags & AVSD_GEMM_GEGLU) AVSD_REQUIRE(d.N % 32 == 0, "gemm: GEGLU needs N %% 32 == 0 (got %d)", d.N);
  AVSD_REQUIRE(!((d.flags & AVSD_GEMM_GEGLU) && (d.flags & AVSD_GEMM_GELU)), "gemm: GEGLU and GELU are exclusive");
  if (d.flags & AVSD_GEMM_ROWSTATS) {
    AVSD_REQUIRE(d.rowstats, "gemm: ROWSTATS without a rowstats buffer");
    AVSD_REQUIRE(!(d.flags & (AVSD_GEMM_GEGLU | AVSD_GEMM_OUT_F32)) && d.N % 32 == 0 && d.ldc % 8 == 0 &&
                     (!d.res1 || d.ldr1 % 8 == 0) && (!d.res2 || d.ldr2 % 8 == 0),
                 "gemm: ROWSTATS needs bf16 output, N %% 32 == 0 and 16-byte-aligned rows");
  }
  if (d.flags & AVSD_GEMM_LNFUSE) {
    AVSD_REQUIRE(d.ln_stats && d.ln_colsum && d.ln_nblk > 0 && d.mode == AVSD_GEMM_PLAIN && !d.A2,
                 "gemm: LNFUSE needs ln_stats, ln_colsum, ln_nblk and a single-source PLAIN operand");
    AVSD_REQUIRE(d.ln_nblk * 32 == d.K || d.ln_nblk == 1, "gemm: LNFUSE statistics are K / 32 pairs per row or one pre-folded pair (got %d for K = %d)", d.ln_nblk, d.K);
    AVSD_REQUIRE(d.batch == 1 || d.batch_stride_a % d.lda == 0, "gemm: LNFUSE batch stride must be whole rows");
  }
  if (d.stats_pos || d.ln_rowvec) {       // LayerNorm(x + pos[frame]) fold (include/avsd.h)
    AVSD_REQUIRE(d.pos_hw > 0 && d.pos_frames > 0 && d.batch == 1 && !(d.flags & AVSD_GEMM_X2),
                 "gemm: stats_pos / ln_rowvec need pos_hw > 0, pos_frames > 0, batch 1 and no AVSD_GEMM_X2 (got hw %d, frames %d)", d.pos_hw, d.pos_frames);
    AVSD_REQUIRE(!d.stats_pos || (d.flags & AVSD_GEMM_ROWSTATS), "gemm: stats_pos without AVSD_GEMM_ROWSTATS");
    AVSD_REQUIRE(!d.ln_rowvec || (d.flags & AVSD_GEMM_LNFUSE), "gemm: ln_rowvec without AVSD_GEMM_LNFUSE");
  }
  if (d.res1) AVSD_REQUIRE(d.ldr1 % 4 == 0, "gemm: ldr1 must be a multiple of 4");
  if (d.res2) AVSD_REQUIRE(d.ldr2 % 4 == 0, "gemm: ldr2 must be a multiple of 4");
  if (d.out_master) AVSD_REQUIRE(d.ldm % 4 == 0 && d.ldm >= ((d.mode == AVSD_GEMM_CONV3 && d.ups == 2) ? d.N / 4 : d.N) && !(d.flags & AVSD_GEMM_GEGLU),
                                 "gemm: out_master needs ldm %% 4 == 0, ldm >= N (ups = 2: cout) and no GEGLU");
  if (d.rowvec) AVSD_REQUIRE(d.rows_per_vec > 0 && d.ldv % 4 == 0, "gemm: rowvec needs rows_per_vec > 0 and ldv %% 4 == 0");
  if (d.flags & AVSD_GEMM_OUT_REST) {
#ifndef AVSD_F16
    AVSD_REQUIRE(d.mode == AVSD_GEMM_CONV3 && d.ups == 2,
                 "gemm: OUT_REST is compiled into the IEEE-half build only (libavsd_hip_f16.so: the per-layer precision plan is fp16 storage)");
#endif
    AVSD_REQUIRE(!(d.flags & (AVSD_GEMM_X2 | AVSD_GEMM_OUT_F32 | AVSD_GEMM_GEGLU)) && d.out_lo != 0 && (d.out_lo & 7) == 0 && d.batch == 1 &&
                     d.tile != AVSD_GEMM_TILE_NSTREAM && !(d.tile >= AVSD_GEMM_TILE_CONV3R_FIRST && d.tile <= AVSD_GEMM_TILE_CONV3R2D_LAST),
                 "gemm: OUT_REST is for one-pass 16-bit outputs (no X2 / OUT_F32 / GEGLU / batch, not on tiles 40..54 / 70) and needs out_lo %% 8 == 0, != 0");
  }
  if (d.mode == AVSD_GEMM_PLAIN) {
    if (!d.A2) d.k_split = d.K;
    AVSD_REQUIRE(d.k_split % 8 == 0 && d.k_split <= d.K, "gemm: k_split (%d) must be a multiple of 8 and <= K", d.k_split);
    if (d.A2) AVSD_REQUIRE(d.lda2 % 8 == 0, "gemm: lda2 must be a multiple of 8");
  } else if (d.mode == AVSD_GEMM_TMIX) {
    AVSD_REQUIRE(d.cseg > 0 && d.cseg % 8 == 0 && d.K == 3 * d.cseg, "gemm/tmix: K (%d) must equal 3*cseg (%d)", d.K, d.cseg);
    AVSD_REQUIRE(d.hw > 0 && d.frames > 0 && d.M % (d.hw * d.frames) == 0, "gemm/tmix: M (%d) must be a multiple of frames*hw (%d*%d)", d.M, d.frames, d.hw);
  } else if (d.mode == AVSD_GEMM_CONV3) {
    AVSD_REQUIRE(d.stride == 1 || d.stride == 2, "gemm/conv3: stride must be 1 or 2");
    if (d.ups == 2) {      // sub-pixel upsample convolution: rows = input pixels, columns = (parity, cout), K = 4 taps x cin
      AVSD_REQUIRE(d.cin > 0 && d.cin % 64 == 0 && d.K == 4 * d.cin, "gemm/conv3: ups = 2 needs K (%d) == 4*cin (%d), cin %% 64 == 0", d.K, d.cin);
      AVSD_REQUIRE(d.stride == 1 && d.pad == 1 && d.hs > 0 && d.ws > 0 && d.ho == d.hs && d.wo == d.ws && d.M % (d.hs * d.ws) == 0,
                   "gemm/conv3: ups = 2 needs stride 1, pad 1, (ho, wo) = (hs, ws) and whole images");
      AVSD_REQUIRE(d.N % 4 == 0 && (d.N / 4) % 64 == 0 && d.ldc >= d.N / 4, "gemm/conv3: ups = 2 needs N = 4 * cout, cout %% 64 == 0, ldc >= cout");
      AVSD_REQUIRE(!d.A2 && !d.res1 && !d.res2 && !d.rowvec && !(d.flags & (AVSD_GEMM_GEGLU | AVSD_GEMM_GELU | AVSD_GEMM_ROWSTATS | AVSD_GEMM_LNFUSE)) && d.batch == 1,
                   "gemm/conv3: ups = 2 takes bias / out_master / the rest plane only");
      AVSD_REQUIRE((double)d.M * 4.0 * d.ldc < 2147483648.0, "gemm/conv3: ups = 2 output too large");
      AVSD_REQUIRE(d.tile >= 4 && !(d.tile >= AVSD_GEMM_TILE_CONV3R_FIRST && d.tile <= AVSD_GEMM_TILE_CONV3R2D_LAST) && !(d.tile >= AVSD_GEMM_TILE_ASM_FIRST),
                   "gemm/conv3: ups = 2 runs on the LDS-direct tiles (4..38), got %d", d.tile);
    } else {
    AVSD_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.K == 9 * d.cin, "gemm/conv3: K (%d) must equal 9*cin (%d), cin %% 8 == 0", d.K, d.cin);
    AVSD_REQUIRE(d.ups == 0 || d.ups == 1, "gemm/conv3: ups must be 0, 1 or 2");
    AVSD_REQUIRE(d.pad == 0 || d.pad == 1, "gemm/conv3: pad must be 0 or 1");
    AVSD_REQUIRE(d.hs > 0 && d.ws > 0 && d.ho > 0 && d.wo > 0 && d.M % (d.ho * d.wo) == 0, "gemm/conv3: bad image geometry");
    const int hin = d.hs << d.ups, win = d.ws << d.ups;
    AVSD_REQUIRE(d.ho == (hin + 2 - 3) / d.stride + 1 && d.wo == (win + 2 - 3) / d.stride + 1, "gemm/conv3: (ho,wo)=(%d,%d) inconsistent with input (%d,%d) stride %d", d.ho, d.wo, hin, win, d.stride);
    }
  } else {
    AVSD_REQUIRE(false, "gemm: unknown mode %d", d.mode);
  }
  if ((d.tile >= AVSD_GEMM_TILE_CONV3R_FIRST && d.tile <= AVSD_GEMM_TILE_CONV3R_LAST) ||
      (d.tile >= AVSD_GEMM_TILE_CONV3R2D_FIRST && d.tile <= AVSD_GEMM_TILE_CONV3R2D_LAST)) {
    if (d.split_k > 1) AVSD_REQUIRE(d.splitk_ws != nullptr, "gemm: split_k needs a workspace");
    return avsd_gemm_dispatch_conv3r(d, reinterpret_cast<hipStream_t>(stream));
  }
  // every kernel below reads the A operand of a convolution / temporal mix from ONE buffer: a second source would be ignored silently
  AVSD_REQUIRE(d.mode == AVSD_GEMM_PLAIN || !d.A2, "gemm: a two-source A needs the PLAIN mode or a resident convolution tile (40..54), got mode %d tile %d", d.mode, d.tile);
  if (d.tile == AVSD_GEMM_TILE_NSTREAM) return avsd_gemm_dispatch_nstream(d, reinterpret_cast<hipStream_t>(stream));
  const bool asm_tile = d.tile >= AVSD_GEMM_TILE_ASM_FIRST && d.tile <= AVSD_GEMM_TILE_ASM_LAST;
  if (asm_tile && !(d.flags & AVSD_GEMM_X2)) return avsd_gemm_dispatch_asm(d, reinterpret_cast<hipStream_t>(stream));
  if (d.split_k > 1 && !asm_tile) {
    AVSD_REQUIRE(d.splitk_ws != nullptr, "gemm: split_k needs a workspace");
    AVSD_REQUIRE(!(d.flags & AVSD_GEMM_GEGLU) && d.batch == 1, "gemm: split_k cannot be combined with GEGLU or batching");
    AVSD_REQUIRE((d.tile >= 4 && d.tile <= ((d.flags & AVSD_GEMM_X2) ? AVSD_GEMM_MAX_TILE_X2 : AVSD_GEMM_MAX_TILE)) ||
                     (!(d.flags & AVSD_GEMM_X2) && d.tile == AVSD_GEMM_TILE_256x160_8W),
                 "gemm: split_k needs an LDS-direct tile (4..33; split precision: ..36), got %d", d.tile);
    AVSD_REQUIRE(d.split_k <= (d.K + 63) / 64, "gemm: split_k (%d) exceeds the number of K tiles", d.split_k);
  }
  if (d.flags & AVSD_GEMM_X2) {
    AVSD_REQUIRE(d.a_lo != 0 && d.w_lo != 0 && (!d.A2 || d.a2_lo != 0), "gemm/x2: A, A2 and W need their rest-plane offsets");
    AVSD_REQUIRE((d.flags & AVSD_GEMM_OUT_F32) || d.out_lo != 0, "gemm/x2: a 16-bit output needs out_lo");
    AVSD_REQUIRE(!d.res1 || (d.flags & AVSD_GEMM_RES1_F32) || d.res1_lo != 0, "gemm/x2: a 16-bit res1 needs res1_lo");
    AVSD_REQUIRE(!d.res2 || (d.flags & AVSD_GEMM_RES2_F32) || d.res2_lo != 0, "gemm/x2: a 16-bit res2 needs res2_lo");
    AVSD_REQUIRE(((d.a_lo | d.a2_lo | d.w_lo | d.out_lo | d.res1_lo | d.res2_lo) & 7) == 0, "gemm/x2: plane offsets must be multiples of 8 elements");
    AVSD_REQUIRE(d.mode != AVSD_GEMM_PLAIN || !d.A2 || d.k_split % 64 == 0, "gemm/x2: a two-source A needs k_split %% 64 == 0 (got %d)", d.k_split);
    const double a_rows = d.mode == AVSD_GEMM_CONV3 ? (double)(d.M / (d.ho * d.wo)) * d.hs * d.ws : (double)d.M;
    AVSD_REQUIRE(a_rows * d.lda * 2.0 < 2147483648.0 && (double)d.N * d.ldw * 2.0 < 2147483648.0, "gemm/x2: operands must be < 2 GiB per plane");
    hipStream_t sx = reinterpret_cast<hipStream_t>(stream);
    if (asm_tile) return avsd_gemm_dispatch_asm(d, sx);
    switch (d.mode) {
      case AVSD_GEMM_PLAIN: return avsd_gemm_dispatch_x2_plain(d, d.tile, sx);
      case AVSD_GEMM_TMIX: return avsd_gemm_dispatch_x2_tmix(d, d.tile, sx);
      default: return d.ups == 2 ? avsd_gemm_dispatch_x2_subpix(d, d.tile, sx) : avsd_gemm_dispatch_x2_conv3(d, d.tile, sx);
    }
  }
  int tile = d.tile;
  // v2 tiles (>= 4) address A/W with 32-bit byte offsets: fall back to v1 for tensors >= 2 GiB
  {
    const double a_rows = d.mode == AVSD_GEMM_CONV3 ? (double)(d.M / (d.ho * d.wo)) * d.hs * d.ws : (double)d.M;
    const bool big = a_rows * d.lda * 2.0 >= 2147483648.0 || (double)d.N * d.ldw * 2.0 >= 2147483648.0 ||
                     (d.A2 && (double)d.M * d.lda2 * 2.0 >= 2147483648.0);
    AVSD_REQUIRE(!(big && d.mode == AVSD_GEMM_CONV3 && d.ups == 2), "gemm/conv3: ups = 2 needs operands < 2 GiB");
    if (big && (tile < 1 || tile > 3)) tile = (d.N > 64 && d.M > 2048) ? 1 : 3;
    // the LDS-direct loader picks the source buffer (A or A2) per 64-wide K tile: a split point inside a tile
    // needs the per-vector select of the register-staged kernel
    if (d.mode == AVSD_GEMM_PLAIN && d.A2 && d.k_split % 64 != 0 && (tile < 1 || tile > 3)) {
      AVSD_REQUIRE(d.split_k <= 1, "gemm: split_k with a two-source A needs k_split %% 64 == 0 (got %d)", d.k_split);
      tile = (d.N > 64 && d.M > 2048) ? 2 : 3;
    }
  }
  if (tile < 1 || (tile > AVSD_GEMM_MAX_TILE && tile != AVSD_GEMM_TILE_256x160_8W)) tile = pick_tile(d.M, (d.flags & AVSD_GEMM_GEGLU) ? d.N : d.N, d.batch);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d.mode) {
    case AVSD_GEMM_PLAIN: return avsd_gemm_dispatch_plain(d, tile, s);
    case AVSD_GEMM_TMIX: return avsd_gemm_dispatch_tmix(d, tile, s);
    default: return d.ups == 2 ? avsd_gemm_dispatch_subpix(d, tile, s) : avsd_gemm_dispatch_conv3(d, tile, s);
  }
}
#endif  // AVSD_GEMM_TU
