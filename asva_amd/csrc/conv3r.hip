// 3x3 / stride 1 / pad 1 implicit-GEMM convolution with the INPUT TILE RESIDENT IN LDS (gfx950) — tile ids
// AVSD_GEMM_TILE_CONV3R_FIRST.. of avsd_gemm_bf16, mode AVSD_GEMM_CONV3.
//
// Why another convolution loop: gemm2_kernel<CONV3> walks K = 9 * cin tap-major and gathers, for every 64-wide K tile, a
// fresh BM x 64 activation tile from global memory — every input element travels L2 -> LDS nine times (once per tap), and at
// BM >= BN those gathers are more than half of the LDS-DMA pieces a workgroup issues.  The global -> LDS path (issue cost of
// the 1-KiB pieces and ~50 B/clk/CU of delivery) is what bounds these kernels, not HBM and not the MFMA pipe.
// Here K is walked CHUNK-major: for a chunk of 64 input channels the workgroup stages the BM output pixels' rows PLUS one
// image row of halo above and below ((BM + 2 ws) x 64 values) ONCE, and all nine taps read their fragments from that one
// image with a per-lane row shift (kh - 1) * ws + (kw - 1); lanes whose tap falls outside the image read a zeroed LDS line.
// Only the BN x 64 weight tile of each (chunk, tap) is streamed per K tile: (BM + BN) * 128 B per K tile become
// BN * 128 + (BM + 2 ws) * 128 / 9.
//
// Geometry: a tile is BM consecutive output rows m = (image, y, x) with BM % ws == 0, and either (hs * ws) % BM == 0 (a band
// of whole image rows inside one image) or BM % (hs * ws) == 0 (several whole images); 2 ws <= 64 halo rows.  The staged rows
// are the global rows [m0 - ws, m0 + BM + ws) (rows outside the tensor are zero-filled by the buffer bounds check; rows of a
// neighbouring image are staged but never read: the tap-validity mask of the reading lane excludes them).
// A image in LDS (own swizzle, shift-invariant): row i, 16-byte chunk c at (i >> 1) * 256 + ((i & 1) << 7) + ((c ^ ((i >> 1) & 7)) << 4)
// — the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots for ANY row shift (the parity bit is not mixed into the
// XOR, unlike the weight-tile image of gemm_common.h).  W ring: tile images exactly as gemm2_kernel.
// Loads: A chunk c + 1 is issued in the iteration of (chunk c, tap 0), after that iteration's barrier (every wave is done
// with chunk c - 1, whose buffer it overwrites), ahead of the W tile issued in the same iteration; vmcnt retires in order, so
// the counted wait for a W tile also covers every A piece issued before it.
// Split-K: over chunks (split_k <= cin / 64); partial tiles + splitk_reduce_kernel as gemm2_kernel (same f32 slice order).
// The f32 summation order inside a slice is chunk-major / tap-minor — different from the tap-major tiles, so results
// differ from them in the last bits of the f32 accumulator (like any two tiles with different K orders).
//
// Replaces (reference file:line): the nn.Conv2d(3x3, padding=1) inside FFInflatedConv3d, avgen/models/unets/utils.py:37-38,
// as called by ff_spatio_temp_resnet_3d.py:132,148 (ResBlock conv1 / conv2) and audio_cond_unet_3d_condition.py conv_in / conv_out.
#include "gemm_common.h"

int avsd_gemm_splitk_reduce(const avsd_gemm_desc& d, hipStream_t s);   // gemm.hip (entry-point translation unit)

namespace {

constexpr unsigned OOBR = 0x80000000u;
constexpr int HALO_MAX = 64;       // staged halo rows: 2 * ws

// waits until at most nw * PW (+ PAC when `a`) loads are still in flight; nw <= MAXW, wave-uniform
template <int MAXW, int PW, int PAC>
__device__ __forceinline__ void wait_ring(int nw, bool a) {
  if constexpr (MAXW <= 0) {
    if (a) wait_vmcnt<PAC>();
    else wait_vmcnt<0>();
  } else {
    if (nw >= MAXW) {
      if (a) wait_vmcnt<MAXW * PW + PAC>();
      else wait_vmcnt<MAXW * PW>();
    } else {
      wait_ring<MAXW - 1, PW, PAC>(nw, a);
    }
  }
}

template <int BM, int BN, int WM, int WN, int STAGES, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void conv3r_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smemr[];
  constexpr int NC = WM * WN;
  constexpr int NWAVES = LW > 0 ? LW : NC;
  constexpr int AR_MAX = BM + HALO_MAX;
  constexpr int PAC = (AR_MAX / 8 + NWAVES - 1) / NWAVES;     // A pieces per loading wave per chunk
  constexpr int A_BYTES = PAC * NWAVES * 1024;
  constexpr int W_BYTES = BN * 128;
  constexpr int PW = (BN / 8) / NWAVES;
  static_assert(PW * NWAVES * 8 == BN, "weight tile rows must split evenly into 1-KiB pieces per loading wave");
  static_assert((STAGES - 2) * PW + PAC < 64, "vmcnt is a 6-bit counter");
  static_assert(STAGES >= 2 && STAGES <= 9, "ring depth");
  constexpr int FM = BM / WM / 32;
  constexpr int FN = BN / WN / 32;
  constexpr int ZERO_OFF = 2 * A_BYTES + STAGES * W_BYTES;    // one zeroed 256-byte line

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LW == 0 || wave_all >= NC;
  const int wave = LW == 0 ? wave_all : (wave_all >= NC ? wave_all - NC : 0);
  const int wm = wave_all % WM;
  const int wn = (wave_all / WM) % WN;

  const int ntm = (p.M + BM - 1) / BM;
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  const int nsplit = p.split_k > 1 ? p.split_k : 1;
  int wg, ksplit;
  {   // XCD-contiguous work items, as gemm2_kernel
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
  const int m0 = tm * BM;
  const int ws = p.ws;
  const int ar = BM + 2 * ws;                 // staged rows

  // this slice's chunks [c0, c1) of 64 input channels
  const int nchunks = p.cin >> 6;
  const int per_split = (nchunks + nsplit - 1) / nsplit;
  const int c0 = ksplit * per_split;
  const int c1 = min(nchunks, c0 + per_split);
  const int nch = max(c1 - c0, 0);
  const int nk = nch * 9;
  // Rotated chunk walk (AVSD_GEMM_KROT, see gemm4.hip): row bands that share a column band of W start at different channel chunks of
  // the slice and wrap, so the cold weight stream is requested in parallel instead of as one chain of HBM round trips.  The walk is
  // indexed by q = 0 .. nch - 1 (the A double buffer alternates on q), the chunk it stands for is chunk_of(q).
  const int crot = ((p.flags & AVSD_GEMM_KROT) && nch > 1 && ntm > 1) ? (int)(((long long)tm * nch) / ntm) : 0;
  auto chunk_of = [&](int q) { const int c = q + crot; return c0 + (c >= nch ? c - nch : c); };

  const h16_t* Ab = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* Wb = reinterpret_cast<const h16_t*>(p.W);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);
  // second source: the input is the channel concat [A (k_split channels) | A2]
  const bool two = p.A2 != nullptr;
  const int csplit = two ? (p.k_split >> 6) : nchunks;
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(two ? p.A2 : p.A), 0, 0x7fffffff, 0x00020000);

  if (tid < 64) *reinterpret_cast<unsigned*>(smemr + ZERO_OFF + tid * 4) = 0u;
  __syncthreads();                            // (before any load is in flight: this waits for vmcnt(0) too)

  // ---- loader state -------------------------------------------------------------------------------------------------
  // A piece j of this wave: line L = piece * 4 + lane / 16, slot s = lane % 16 -> staged row 2 L + (s >> 3), chunk (s & 7) ^ (L & 7)
  // (row and global row advance by 8 * NWAVES per piece; the chunk slot ch does not depend on j: 4 * NWAVES % 8 == 0)
  static_assert((4 * NWAVES) % 8 == 0, "piece stride must keep the swizzle phase");
  const int a_row0 = 2 * (wave * 4 + (lane >> 4)) + ((lane & 15) >> 3);
  const int a_ch8 = (((lane & 15) & 7) ^ ((wave * 4 + (lane >> 4)) & 7)) * 8;
  const int a_g0 = m0 - ws + a_row0;
  int wo[PW];
  bool wv[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int L = (wave + j * NWAVES) * 4 + (lane >> 4);
    const int x = (lane & 15) ^ (L & 15);
    const int n = tn * BN + 2 * L + (x >> 3);
    wv[j] = n < p.N;
    wo[j] = n * p.ldw + (x & 7) * 8;
  }
  auto issue_a = [&](int q) {
    unsigned char* ab = smemr + (q & 1) * A_BYTES;
    const int chunk = chunk_of(q);
    const bool first = chunk < csplit;
    const int ld = first ? p.lda : p.lda2;
    const int coff = a_ch8 + (first ? chunk : chunk - csplit) * 64;
#pragma unroll
    for (int j = 0; j < PAC; ++j) {
      const int g = a_g0 + j * 8 * NWAVES;
      const bool ok = a_row0 + j * 8 * NWAVES < ar && g >= 0 && g < p.M;
      const unsigned vo = ok ? (unsigned)(g * ld + coff) * 2u : OOBR;
      lds_ptr_t dst = (lds_ptr_t)(ab + (wave + j * NWAVES) * 1024);
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, (int)vo, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, (int)vo, 0, 0, 0);
    }
  };
  int i_t = 0, i_q = 0, i_tap = 0;            // next W tile to issue
  auto issue_w = [&]() {
    unsigned char* sb = smemr + 2 * A_BYTES + (i_t % STAGES) * W_BYTES;
    const int kbase = i_tap * p.cin + chunk_of(i_q) * 64;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const unsigned vo = wv[j] ? (unsigned)(wo[j] + kbase) * 2u : OOBR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sb + (wave + j * NWAVES) * 1024), 16, (int)vo, 0, 0, 0);
    }
    ++i_t;
    if (++i_tap == 9) { i_tap = 0; ++i_q; }
  };
  // loads that may stay in flight behind W tile kt: the W tiles of iterations kt-STAGES+2 .. kt-1 and the A chunk one of
  // them issued (an iteration j >= 0 with tap(j) == 0 and a chunk after its own)
  auto wait_tile = [&](int kt) {
    const int nw = min(STAGES - 2, nk - 1 - kt);
    bool a = false;
#pragma unroll
    for (int d = 1; d <= STAGES - 2; ++d) {
      const int j = kt - d;
      if (j >= 0 && j % 9 == 0 && j / 9 + 1 < nch) a = true;
    }
    wait_ring<STAGES - 2, PW, PAC>(nw, a);
  };

  if (is_loader && nk > 0) {
    issue_a(0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) issue_w();
  }
  if (LW > 0 && is_loader) {
    int tap = 0, q = 0;
    for (int kt = 0; kt < nk; ++kt) {
      wait_tile(kt);
      __builtin_amdgcn_s_barrier();
      if (tap == 0 && q + 1 < nch) issue_a(q + 1);
      if (kt + STAGES - 1 < nk) issue_w();
      if (++tap == 9) { tap = 0; ++q; }
    }
    return;
  }

  // ---- MFMA waves -----------------------------------------------------------------------------------------------------
  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // output row of fragment b held by this lane, its staged row at tap (1, 1) and the 9-bit tap-validity mask
  int rloc[FM];
  unsigned tmask[FM];
  {
    const int per = p.hs * ws;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int r = wm * (BM / WM) + b * 32 + (lane & 31);
      const int m = m0 + r;
      rloc[b] = r;
      const int pix = m % per;
      const int y = pix / ws, x = pix - y * ws;
      unsigned mk = 0;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int yy = y + kh - 1, xx = x + kw - 1;
          if (m < p.M && yy >= 0 && yy < p.hs && xx >= 0 && xx < ws) mk |= 1u << (kh * 3 + kw);
        }
      tmask[b] = mk;
    }
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

  int tap = 0, kh = 0, kw = 0, q = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (LW == 0) wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    if (LW == 0) {
      if (tap == 0 && q + 1 < nch) issue_a(q + 1);
      if (kt + STAGES - 1 < nk) issue_w();
    }
    // this tap's fragment rows: staged row i = r + kh * ws + kw - 1
    const int shift = kh * ws + kw - 1;
    const int abuf = (q & 1) * A_BYTES;
    int xa_base[FM], xa_sw[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int i = rloc[b] + shift;
      const bool ok = (tmask[b] >> tap) & 1u;
      xa_base[b] = ok ? abuf + (i >> 1) * 256 + ((i & 1) << 7) : ZERO_OFF;
      xa_sw[b] = ok ? (i >> 1) & 7 : 0;
    }
    const unsigned char* sW = smemr + 2 * A_BYTES + (kt % STAGES) * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + chalf;
      h16x8 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b)
        xf[b] = *reinterpret_cast<const h16x8*>(smemr + xa_base[b] + ((c ^ xa_sw[b]) << 4));
#pragma unroll
      for (int a = 0; a < FN; ++a)
        wf[a] = *reinterpret_cast<const h16x8*>(sW + w_line[a] + (((w_hi[a] | c) ^ w_sw[a]) << 4));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma32x32x16(wf[a], xf[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    ++tap;
    if (++kw == 3) { kw = 0; ++kh; }
    if (tap == 9) { tap = 0; kh = 0; kw = 0; ++q; }
  }

  if (p.split_k > 1) {
    float* wsl = p.splitk_ws + (int64_t)ksplit * p.M * p.N;
    const int hsel = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = m0 + wm * (BM / WM) + b * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = tn * BN + wn * (BN / WN) + a * 32 + 8 * q + hsel;
          if (n < p.N)
            *reinterpret_cast<float4*>(wsl + (int64_t)m * p.N + n) =
                make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
    }
    return;
  }
  if constexpr (FN * FM >= 8) {      // (the resident tiles never fold a LayerNorm: ops.gemm offers them only when ln is None)
    const float pre_ln[2 * FM] = {};
    epilogue_each<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, m0 + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, 0, pre_ln, false);
  } else {
    const float pre_ln[2 * FM] = {};
    epilogue<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, m0 + wm * (BM / WM), tn * BN + wn * (BN / WN), lane, 0, pre_ln, false);
  }
}

// ---- 2-D tiles: TH image rows x 32 pixels --------------------------------------------------------------------------------
// The full-width row bands above need (BM + 2 ws) staged rows: fine up to 32-pixel-wide images, too much LDS beyond.  Wider images
// (the VAE decoder's 64 ... 256-pixel levels, cfg 4's 64 x 64 latents) take rectangular tiles: BM = 32 TH output pixels = TH image rows
// of 32 pixels, staged as the (TH + 2) x 34 rectangle around them — positions outside the image are zero-filled AT LOAD TIME (buffer
// bounds check), so the taps need no validity mask at all.  Staged position i = yy * 34 + xx lives where "row i" lives in the image of
// the 1-D kernel (same shift-invariant swizzle); the fragment of tile row ty reads i = (ty + kh) * 34 + tx + kw.  A wave's row
// fragment b is tile row (wm * FM + b): its output rows are 32 consecutive pixels, fragments of a wave lie one IMAGE ROW apart — the
// shared epilogue takes that distance as `mstride`.  Everything else (chunk-major K, double-buffered chunks, weight ring, loader waves,
// split-K over chunks, second source) is the 1-D kernel.
constexpr int TWP = 34;       // staged row pitch: 32 pixels + one halo pixel either side

template <int BM, int BN, int WM, int WN, int STAGES, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void conv3r2d_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem2d[];
  constexpr int NC = WM * WN;
  constexpr int NWAVES = LW > 0 ? LW : NC;
  constexpr int TH = BM / 32;
  constexpr int SR = (TH + 2) * TWP;                          // staged positions
  constexpr int PAC = ((SR + 7) / 8 + NWAVES - 1) / NWAVES;
  constexpr int A_BYTES = PAC * NWAVES * 1024;
  constexpr int W_BYTES = BN * 128;
  constexpr int PW = (BN / 8) / NWAVES;
  static_assert(PW * NWAVES * 8 == BN, "weight tile rows must split evenly into 1-KiB pieces per loading wave");
  static_assert((STAGES - 2) * PW + PAC < 64, "vmcnt is a 6-bit counter");
  static_assert(TH % WM == 0, "tile rows must split evenly over the wave rows");
  constexpr int FM = TH / WM;
  constexpr int FN = BN / WN / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LW == 0 || wave_all >= NC;
  const int wave = LW == 0 ? wave_all : (wave_all >= NC ? wave_all - NC : 0);
  const int wm = wave_all % WM;
  const int wn = (wave_all / WM) % WN;

  const int H = p.hs, W = p.ws;
  const int tpr = W / 32, tpi = (H / TH) * tpr;               // tiles per image row band / per image
  const int ntm = (p.M / (H * W)) * tpi;
  const int ntn = (p.N + BN - 1) / BN;
  const int nwg = ntm * ntn;
  const int nsplit = p.split_k > 1 ? p.split_k : 1;
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
  int tm, tn;
  tile_of_item(wg, ntm, ntn, (p.flags & AVSD_GEMM_XCD_N) != 0, p.raster_g, tm, tn);
  const int img = tm / tpi;
  const int trem = tm - img * tpi;
  const int ty0 = (trem / tpr) * TH, tx0 = (trem % tpr) * 32;

  const int nchunks = p.cin >> 6;
  const int per_split = (nchunks + nsplit - 1) / nsplit;
  const int c0 = ksplit * per_split;
  const int c1 = min(nchunks, c0 + per_split);
  const int nk = max(c1 - c0, 0) * 9;

  const bool two = p.A2 != nullptr;
  const int csplit = two ? (p.k_split >> 6) : nchunks;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(two ? p.A2 : p.A), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);

  // loader state: piece j of this wave covers staged positions 8 * piece .. + 7; this lane's first position (the others follow
  // 8 * NWAVES apart) — the global pixel of a position is decoded when a chunk is issued (once per nine K tiles), not kept
  const int a_i0 = 2 * (wave * 4 + (lane >> 4)) + ((lane & 15) >> 3);
  static_assert((4 * NWAVES) % 8 == 0, "piece stride must keep the swizzle phase");
  const int a_ch8 = (((lane & 15) & 7) ^ ((wave * 4 + (lane >> 4)) & 7)) * 8;
  int wo[PW];
  bool wv[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int L = (wave + j * NWAVES) * 4 + (lane >> 4);
    const int x = (lane & 15) ^ (L & 15);
    const int n = tn * BN + 2 * L + (x >> 3);
    wv[j] = n < p.N;
    wo[j] = n * p.ldw + (x & 7) * 8;
  }
  auto issue_a = [&](int chunk) {
    unsigned char* ab = smem2d + (chunk & 1) * A_BYTES;
    const bool first = chunk < csplit;
    const int ld = first ? p.lda : p.lda2;
    const int coff = a_ch8 + (first ? chunk : chunk - csplit) * 64;
#pragma unroll
    for (int j = 0; j < PAC; ++j) {
      const int i = a_i0 + j * 8 * NWAVES;
      const int yy = i / TWP, xx = i - yy * TWP;
      const int gy = ty0 - 1 + yy, gx = tx0 - 1 + xx;
      const bool ok = i < SR && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const unsigned vo = ok ? (unsigned)(((img * H + gy) * W + gx) * ld + coff) * 2u : OOBR;
      lds_ptr_t dst = (lds_ptr_t)(ab + (wave + j * NWAVES) * 1024);
      if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, (int)vo, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, (int)vo, 0, 0, 0);
    }
  };
  int i_t = 0, i_chunk = c0, i_tap = 0;
  auto issue_w = [&]() {
    unsigned char* sb = smem2d + 2 * A_BYTES + (i_t % STAGES) * W_BYTES;
    const int kbase = i_tap * p.cin + i_chunk * 64;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const unsigned vo = wv[j] ? (unsigned)(wo[j] + kbase) * 2u : OOBR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(sb + (wave + j * NWAVES) * 1024), 16, (int)vo, 0, 0, 0);
    }
    ++i_t;
    if (++i_tap == 9) { i_tap = 0; ++i_chunk; }
  };
  auto wait_tile = [&](int kt) {
    const int nw = min(STAGES - 2, nk - 1 - kt);
    bool a = false;
#pragma unroll
    for (int d = 1; d <= STAGES - 2; ++d) {
      const int j = kt - d;
      if (j >= 0 && j % 9 == 0 && c0 + j / 9 + 1 < c1) a = true;
    }
    wait_ring<STAGES - 2, PW, PAC>(nw, a);
  };

  if (is_loader && nk > 0) {
    issue_a(c0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) issue_w();
  }
  if (LW > 0 && is_loader) {
    int tap = 0, chunk = c0;
    for (int kt = 0; kt < nk; ++kt) {
      wait_tile(kt);
      __builtin_amdgcn_s_barrier();
      if (tap == 0 && chunk + 1 < c1) issue_a(chunk + 1);
      if (kt + STAGES - 1 < nk) issue_w();
      if (++tap == 9) { tap = 0; ++chunk; }
    }
    return;
  }

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  int ibase[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) ibase[b] = (wm * FM + b) * TWP + (lane & 31);
  int w_line[FN], w_sw[FN], w_hi[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int r = wn * (BN / WN) + a * 32 + (lane & 31);
    w_line[a] = (r >> 1) * 256;
    w_sw[a] = (r >> 1) & 15;
    w_hi[a] = (r & 1) << 3;
  }
  const int chalf = lane >> 5;

  int tap = 0, kh = 0, kw = 0, chunk = c0;
  for (int kt = 0; kt < nk; ++kt) {
    if (LW == 0) wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    if (LW == 0) {
      if (tap == 0 && chunk + 1 < c1) issue_a(chunk + 1);
      if (kt + STAGES - 1 < nk) issue_w();
    }
    const int shift = kh * TWP + kw;
    const int abuf = (chunk & 1) * A_BYTES;
    int xa_base[FM], xa_sw[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int i = ibase[b] + shift;
      xa_base[b] = abuf + (i >> 1) * 256 + ((i & 1) << 7);
      xa_sw[b] = (i >> 1) & 7;
    }
    const unsigned char* sW = smem2d + 2 * A_BYTES + (kt % STAGES) * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + chalf;
      h16x8 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b)
        xf[b] = *reinterpret_cast<const h16x8*>(smem2d + xa_base[b] + ((c ^ xa_sw[b]) << 4));
#pragma unroll
      for (int a = 0; a < FN; ++a)
        wf[a] = *reinterpret_cast<const h16x8*>(sW + w_line[a] + (((w_hi[a] | c) ^ w_sw[a]) << 4));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma32x32x16(wf[a], xf[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    ++tap;
    if (++kw == 3) { kw = 0; ++kh; }
    if (tap == 9) { tap = 0; kh = 0; kw = 0; ++chunk; }
  }

  const int m_wave = (img * H + ty0 + wm * FM) * W + tx0;       // output row of (fragment 0, lane 0) of this wave
  if (p.split_k > 1) {
    float* wsl = p.splitk_ws + (int64_t)ksplit * p.M * p.N;
    const int hsel = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int m = m_wave + b * W + (lane & 31);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = tn * BN + wn * (BN / WN) + a * 32 + 8 * q + hsel;
          if (n < p.N)
            *reinterpret_cast<float4*>(wsl + (int64_t)m * p.N + n) =
                make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
    }
    return;
  }
  if constexpr (FN * FM >= 8) {
    const float pre_ln[2 * FM] = {};
    epilogue_each<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, m_wave, tn * BN + wn * (BN / WN), lane, 0, pre_ln, false, W);
  } else {
    const float pre_ln[2 * FM] = {};
    epilogue<FN, FM, (64 * (WM * WN + LW) > 512)>(p, acc, m_wave, tn * BN + wn * (BN / WN), lane, 0, pre_ln, false, W);
  }
}

template <int BM, int BN, int WM, int WN, int STAGES, int LW>
int launch_r2d(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr int NWAVES = LW > 0 ? LW : WM * WN;
  constexpr int TH = BM / 32;
  constexpr int PAC = (((TH + 2) * TWP + 7) / 8 + NWAVES - 1) / NWAVES;
  constexpr size_t lds = (size_t)2 * PAC * NWAVES * 1024 + (size_t)STAGES * BN * 128;
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
  AVSD_REQUIRE(d.ws % 32 == 0 && d.hs % TH == 0, "gemm/conv3r2d: a %d x 32-pixel tile needs image height %% %d == 0 and width %% 32 == 0 (image %d x %d)", TH, TH, d.hs, d.ws);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3r2d_kernel<BM, BN, WM, WN, STAGES, LW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("conv3r2d: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M / (d.hs * d.ws)) * (d.hs / TH) * (d.ws / 32), ntn = (d.N + BN - 1) / BN;
  const int nsplit = d.split_k > 1 ? d.split_k : 1;
  dim3 grid((unsigned)(ntm * ntn), (unsigned)nsplit, 1);
  hipLaunchKernelGGL((conv3r2d_kernel<BM, BN, WM, WN, STAGES, LW>), grid, dim3(64 * (WM * WN + LW)), lds, s, d);
  AVSD_CHECK_LAUNCH("conv3r2d launch");
  if (nsplit > 1) return avsd_gemm_splitk_reduce(d, s);
  return AVSD_OK;
}

template <int BM, int BN, int WM, int WN, int STAGES, int LW>
int launch_r(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr int NWAVES = LW > 0 ? LW : WM * WN;
  constexpr int PAC = ((BM + HALO_MAX) / 8 + NWAVES - 1) / NWAVES;
  constexpr size_t lds = (size_t)2 * PAC * NWAVES * 1024 + (size_t)STAGES * BN * 128 + 256;
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB of LDS");
  AVSD_REQUIRE(BM % d.ws == 0 && ((d.hs * d.ws) % BM == 0 || BM % (d.hs * d.ws) == 0),
               "gemm/conv3r: a %d-row tile must be whole image rows of one image or whole images (image %d x %d)", BM, d.hs, d.ws);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3r_kernel<BM, BN, WM, WN, STAGES, LW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("conv3r: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const int nsplit = d.split_k > 1 ? d.split_k : 1;
  dim3 grid((unsigned)(ntm * ntn), (unsigned)nsplit, 1);
  hipLaunchKernelGGL((conv3r_kernel<BM, BN, WM, WN, STAGES, LW>), grid, dim3(64 * (WM * WN + LW)), lds, s, d);
  AVSD_CHECK_LAUNCH("conv3r launch");
  if (nsplit > 1) return avsd_gemm_splitk_reduce(d, s);
  return AVSD_OK;
}

}  // namespace

// tile ids AVSD_GEMM_TILE_CONV3R_FIRST + k (include/avsd.h)
int avsd_gemm_dispatch_conv3r(const avsd_gemm_desc& d, hipStream_t s) {
  AVSD_REQUIRE(d.mode == AVSD_GEMM_CONV3 && d.stride == 1 && d.ups == 0 && d.pad == 1 && d.batch == 1 &&
                   !(d.flags & (AVSD_GEMM_X2 | AVSD_GEMM_GEGLU | AVSD_GEMM_LNFUSE)),
               "gemm/conv3r: stride-1 pad-1 3x3 convolutions only (no upsample fold, split precision, GEGLU, LayerNorm fold, batching)");
  if (d.A2) AVSD_REQUIRE(d.k_split > 0 && d.k_split % 64 == 0 && d.k_split < d.cin && d.lda2 % 8 == 0 && (double)d.M * d.lda2 * 2.0 < 2147483648.0,
                         "gemm/conv3r: a two-source input needs k_split %% 64 == 0 inside cin (got %d of %d)", d.k_split, d.cin);
  if (d.tile >= AVSD_GEMM_TILE_CONV3R2D_FIRST && d.tile <= AVSD_GEMM_TILE_CONV3R2D_LAST) {
    AVSD_REQUIRE(d.cin % 64 == 0, "gemm/conv3r2d: cin %% 64 == 0 (got cin %d)", d.cin);
    AVSD_REQUIRE(d.split_k <= 1 || d.split_k <= d.cin / 64, "gemm/conv3r2d: split_k (%d) exceeds the %d channel chunks", d.split_k, d.cin / 64);
    AVSD_REQUIRE((double)d.M * d.lda * 2.0 < 2147483648.0 && (double)d.N * d.ldw * 2.0 < 2147483648.0, "gemm/conv3r2d: operands must be < 2 GiB");
    switch (d.tile - AVSD_GEMM_TILE_CONV3R2D_FIRST) {
      case 0: return launch_r2d<256, 128, 4, 2, 3, 4>(d, s);     // 8 x 32 pixels, 64x64 wave tiles, 4 loader waves
      case 1: return launch_r2d<256, 160, 8, 1, 3, 4>(d, s);     // 8 x 32 pixels, 32x160 wave tiles
      case 2: return launch_r2d<256, 128, 8, 1, 3, 4>(d, s);     // 8 x 32 pixels, 32x128 wave tiles
      case 3: return launch_r2d<128, 128, 2, 2, 4, 2>(d, s);     // 4 x 32 pixels
      default: AVSD_REQUIRE(false, "gemm/conv3r2d: unknown tile %d", d.tile);
    }
  }
  AVSD_REQUIRE(d.cin % 64 == 0 && 2 * d.ws <= HALO_MAX, "gemm/conv3r: cin %% 64 == 0 and image width <= %d (got cin %d, width %d)", HALO_MAX / 2, d.cin, d.ws);
  AVSD_REQUIRE(d.split_k <= 1 || d.split_k <= d.cin / 64, "gemm/conv3r: split_k (%d) exceeds the %d channel chunks", d.split_k, d.cin / 64);
  AVSD_REQUIRE((double)d.M * d.lda * 2.0 < 2147483648.0 && (double)d.N * d.ldw * 2.0 < 2147483648.0, "gemm/conv3r: operands must be < 2 GiB");
  switch (d.tile - AVSD_GEMM_TILE_CONV3R_FIRST) {
    case 0: return launch_r<256, 128, 4, 2, 3, 4>(d, s);     // 128 KB: 64x64 wave tiles, 4 loader waves
    case 2: return launch_r<256, 160, 4, 1, 3, 4>(d, s);     // 140 KB: 64x160 wave tiles (N = 320 in two column tiles)
    case 3: return launch_r<256, 160, 8, 1, 3, 4>(d, s);     // 140 KB: 32x160 wave tiles, 12 waves
    case 4: return launch_r<128, 128, 2, 2, 4, 2>(d, s);     // 112 KB: 64x64 wave tiles
    case 8: return launch_r<256, 256, 4, 2, 2, 0>(d, s);     // 144 KB: 64x128 wave tiles
    default: AVSD_REQUIRE(false, "gemm/conv3r: unknown tile %d", d.tile);
  }
}

// largest BM the image geometry admits for tile id `tile` (0 = unsupported): lets a host enumerate candidates
extern "C" int avsd_gemm_conv3r_supported(int tile, int hs, int ws, int cin) {
  static const int bm[10] = {256, 0, 256, 256, 128, 0, 0, 0, 256, 0};     // 0: measured and dropped (41, 45, 46, 47, 49 were never the tuner's pick)
  const int k = tile - AVSD_GEMM_TILE_CONV3R_FIRST;
  if (k < 0 || k >= 10 || bm[k] == 0 || cin % 64 != 0 || ws <= 0 || hs <= 0 || 2 * ws > HALO_MAX) return 0;
  const int b = bm[k];
  return (b % ws == 0 && ((hs * ws) % b == 0 || b % (hs * ws) == 0)) ? b : 0;
}

// rows per tile of 2-D tile id `tile` (TH image rows x 32 pixels) if the image geometry admits it, else 0
extern "C" int avsd_gemm_conv3r2d_supported(int tile, int hs, int ws, int cin) {
  static const int bm[4] = {256, 256, 256, 128};
  const int k = tile - AVSD_GEMM_TILE_CONV3R2D_FIRST;
  if (k < 0 || k >= 4 || cin % 64 != 0 || ws <= 0 || hs <= 0 || ws % 32 != 0 || hs % (bm[k] / 32) != 0) return 0;
  return bm[k];
}
