// Fused cross-attention block for gfx950: one launch per residual-stream update
//
//     h' = h + to_out( softmax( (LN(h) Wq^T) K^T / sqrt(d) ) V )           (+ bias of to_out)
//
// for the audio and text cross-attentions of BasicTransformerBlock (ff_spatio_audio_temp_transformer_3d.py:315-341;
// diffusers Attention + AttnProcessor2_0).  K and V depend only on the conditioning, so the host caches them once per
// clip — already gathered by the audio segment mask, padded to whole 32-key tiles and with V transposed
// (AudioUNet3DConditionModel.make_cond_block) — and the step-variant work of the block is exactly this kernel.
// It replaces three launches (LayerNorm-folded Q projection, attention, output projection + residual) and two M x C
// round trips of q and o through HBM.
//
// One workgroup = 128 rows of h x all C channels, 8 waves as 4 (rows) x 2 (channel halves):
//   stage 1  Q = LN-fold(h) . Wq'^T        the GEMM main loop of gemm2_kernel<128, C, 4, 2, 2>: LDS-direct K tiles, 2-stage ring
//   stage 2  per head: S^T = K . Q^T  ->  softmax in registers (all keys resident: Lk <= 96)  ->  O^T += V^T . P^T
//            Q never leaves the accumulators: rounded, then the two lanes of a row trade 4-channel quads
//            (v_permlane32_swap) into the MFMA B-operand layout.  Heads are D = C / heads channels wide and need not
//            start on a 16-channel MFMA k-block (D = 40): operands of a k-block / output fragment that straddles two
//            heads are zeroed per lane for the channels of the other head.
//   stage 3  out = O . Wo^T + bias + residual     O (16-bit) goes through LDS in the GEMM tile image, Wo streams through the ring;
//            the shared GEMM epilogue writes the 16-bit stream, the optional f32 master and the LayerNorm row statistics.
// LDS: stage 1 ring 112 KB | stage 2 K + V^T <= 124 KB | stage 3 O tile 80 KB + Wo ring 80 KB — the phases alias.
#include "avsd_common.h"
#include "gemm_common.h"

namespace {

struct XAttnArgs {
  const h16_t* H; int ldh;
  const h16_t* Wq; int ldwq;
  const float* q_colsum; const float* q_bias;
  const float* ln_stats; int ln_nblk; float ln_eps;
  const h16_t* Kc; const h16_t* Vt;       // K [nkv][LKP][C], V^T [nkv][C][LKP]
  int lk, L, q_per_kv;
  float sl2;                              // softmax scale * log2(e)
  const h16_t* Wo; int ldwo;
  avsd_gemm_desc epi;                     // output side: out / ldc / bias / res1 / flags / rowstats / out_master / M / N
};

// N 1-KiB pieces of a K tile, global -> LDS (piece index wave + j * NWAVE; `off` = this lane's element offset per piece)
// (hipcc 7.2, host pass: an array whose bound depends on a template parameter, once captured by a lambda or bound to an
// array-reference parameter inside a kernel template, silently drops the kernel's host stub — the offset arrays below
// therefore have literal bounds and travel as pointers; the loops are fully unrolled, so they stay in registers)
template <int N, int NWAVE>
__device__ __forceinline__ void issue_pieces(const __amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, int wave, const int* off, int add) {
#pragma unroll
  for (int j = 0; j < N; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds_base + (wave + j * NWAVE) * 1024), 16, (off[j] + add) * 2, 0, 0, 0);
}

template <int C, int D, int NT>
__global__ __launch_bounds__(512, 2) void xattn_kernel(const XAttnArgs p) {
  constexpr int BM = 128, WM = 4, WN = 2, NWAVE = WM * WN;
  constexpr int NCOL = C / WN;             // channels per wave
  constexpr int FN = NCOL / 32;            // 32-channel fragments per wave
  constexpr int HPW = NCOL / D;            // heads per wave
  constexpr int GPH = D / 8;               // 8-channel groups per head
  constexpr int LKP = NT * 32;
  constexpr int NKT = C / BK;
  constexpr int A_BYTES = BM * 128, W_BYTES = C * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int PA = (BM / 8) / NWAVE, PW = (C / 8) / NWAVE;
  constexpr int KS = C + 8;                // sK row stride (elements): 16-byte aligned rows, ds_read_b128 spread over the banks
  constexpr int VS = LKP + 4;              // sVt row stride (elements): conflict-free ds_read_b64
  constexpr int SK_BYTES = LKP * KS * 2;
  constexpr int SO_BYTES = NKT * A_BYTES;
  static_assert(NCOL % 32 == 0 && NCOL % D == 0 && D % 8 == 0 && C % BK == 0, "shape");
  static_assert(PA * NWAVE * 8 == BM && PW * NWAVE * 8 == C, "tile rows must split evenly into 1-KiB pieces per wave");
  static_assert(2 * STAGE_BYTES <= 160 * 1024 && SK_BYTES + C * VS * 2 <= 160 * 1024 && SO_BYTES + 2 * W_BYTES <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smx[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5, l31 = lane & 31, hsel = half * 4;
  // XCD-aware tile order: the 8 XCDs each get a contiguous range of row tiles (= whole query batches: one K/V block per L2)
  int tm;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int row0 = tm * BM;
  const int kvb = (row0 / p.L) / p.q_per_kv;

  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)p.H, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsWq = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wq, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsWo = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wo, 0, 0x7fffffff, 0x00020000);

  // this lane's (row, k-chunk) of each 1-KiB piece it loads
  static_assert(PA <= 4 && PW <= 8, "piece-offset arrays");
  int aoff[4], wqoff[8], wooff[8];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    int r, kc;
    piece_row_chunk(wave + j * NWAVE, lane, r, kc);
    aoff[j] = (row0 + r) * p.ldh + kc;
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    int r, kc;
    piece_row_chunk(wave + j * NWAVE, lane, r, kc);
    wqoff[j] = r * p.ldwq + kc;
    wooff[j] = r * p.ldwo + kc;
  }
  // fragment read offsets inside a tile image (16-byte chunk c = 2 * ks + half is XOR-ed in per k-step)
  const int a_row = wm * 32 + l31;
  int w_row[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) w_row[a] = wn * NCOL + a * 32 + l31;

  f32x16 acc[FN][1];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;

  // ================= stage 1: Q = h . Wq'^T =======================================================================
#define XATTN_ISSUE1(kt_)                                                                               \
  do {                                                                                                 \
    unsigned char* sb_ = smx + ((kt_) & 1) * STAGE_BYTES;                                              \
    issue_pieces<PA, NWAVE>(rsH, sb_, wave, aoff, (kt_) * BK);                                         \
    issue_pieces<PW, NWAVE>(rsWq, sb_ + A_BYTES, wave, wqoff, (kt_) * BK);                             \
  } while (0)
  XATTN_ISSUE1(0);
  // K of this tile's conditioning block goes out NOW (round 5): its registers ride through stage 1 and the data has landed when the
  // stage-1 ring is free to take it — the staging phase between the two products was 17 % of the kernel's cycles (profiles/r5_xattn_phases.txt)
  constexpr int KV8 = C / 8, VV8 = LKP / 8;                 // 16-byte vectors per K row / per V^T row
  constexpr int NKV = (LKP * KV8 + 511) / 512, NVV = (C * VV8 + 511) / 512;
  uint4 kreg[NKV];
  {
    const uint4* gk = reinterpret_cast<const uint4*>(p.Kc + (int64_t)kvb * LKP * C);
#pragma unroll
    for (int u = 0; u < NKV; ++u) kreg[u] = gk[min(tid + u * 512, LKP * KV8 - 1)];
  }
  // LayerNorm statistics of this lane's row, fetched while tile 0 is in flight
  float ln_rstd, ln_mr;
  {
    const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (int64_t)(row0 + a_row) * p.ln_nblk;
    float sm = 0.f, sq = 0.f;
    constexpr int NB = C / 32;               // all pairs of the row requested at once: one round trip, not NB
    float2 t[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) t[j] = st[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) { sm += t[j].x; sq += t[j].y; }
    const float inv_k = 1.0f / (float)(p.ln_nblk * 32);
    const float mean = sm * inv_k;
    ln_rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
    ln_mr = mean * ln_rstd;
  }
  for (int kt = 0; kt < NKT; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < NKT) XATTN_ISSUE1(kt + 1);
    const unsigned char* sA = smx + (kt & 1) * STAGE_BYTES;
    const unsigned char* sW = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + half;
      const h16x8 xf = *reinterpret_cast<const h16x8*>(sA + frag_offset(a_row, c));
      h16x8 wf[FN];
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const h16x8*>(sW + frag_offset(w_row[a], c));
#pragma unroll
      for (int a = 0; a < FN; ++a) acc[a][0] = mfma32x32x16(wf[a], xf, acc[a][0], 0, 0, 0);
    }
  }
  // ---- K [LKP][C] and V^T [C][LKP] of this tile's conditioning block (L2-resident, shared by every row tile): every
  // global load is issued before anything waits on one — a load / store loop would pay one L2 round trip per iteration —
  // and the Q conversion below runs while they are in flight
  uint4 vreg[NVV];
  // ---- q = rstd * acc - mean * rstd * colsum + bias, rounded, as MFMA B operands: k-block kb = 16 channels of the wave ----
  h16x8 qf[2 * FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    float v[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = wn * NCOL + a * 32 + 8 * q + hsel;
      const float4 cs = *reinterpret_cast<const float4*>(p.q_colsum + n);
      const float4 bb = *reinterpret_cast<const float4*>(p.q_bias + n);
      v[q][0] = fmaf(acc[a][0][4 * q + 0], ln_rstd, -ln_mr * cs.x) + bb.x;
      v[q][1] = fmaf(acc[a][0][4 * q + 1], ln_rstd, -ln_mr * cs.y) + bb.y;
      v[q][2] = fmaf(acc[a][0][4 * q + 2], ln_rstd, -ln_mr * cs.z) + bb.z;
      v[q][3] = fmaf(acc[a][0][4 * q + 3], ln_rstd, -ln_mr * cs.w) + bb.w;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned e0[2], e1[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const auto e = __builtin_amdgcn_permlane32_swap(pack2h(v[2 * j][2 * d], v[2 * j][2 * d + 1]),
                                                        pack2h(v[2 * j + 1][2 * d], v[2 * j + 1][2 * d + 1]), false, false);
        e0[d] = e[0]; e1[d] = e[1];
      }
      qf[2 * a + j] = __builtin_bit_cast(h16x8, make_uint4(e0[0], e0[1], e1[0], e1[1]));
    }
  }
  {   // the V^T loads take the registers the stage-1 accumulators just left (K went out before the Q conversion)
    const uint4* gv = reinterpret_cast<const uint4*>(p.Vt + (int64_t)kvb * C * LKP);
#pragma unroll
    for (int u = 0; u < NVV; ++u) vreg[u] = gv[min(tid + u * 512, C * VV8 - 1)];
  }
  __syncthreads();   // every wave is done with the stage-1 ring: K / V^T take its place
  unsigned char* sK = smx;
  unsigned char* sVt = smx + SK_BYTES;
#pragma unroll
  for (int u = 0; u < NKV; ++u) {
    const int v = tid + u * 512;
    if (v < LKP * KV8) {
      const int r = v / KV8, c8 = v - r * KV8;
      *reinterpret_cast<uint4*>(sK + r * (KS * 2) + c8 * 16) = kreg[u];
    }
  }
#pragma unroll
  for (int u = 0; u < NVV; ++u) {
    const int v = tid + u * 512;
    if (v < C * VV8) {
      const int r = v / VV8, k8 = v - r * VV8;
      uint2* dst = reinterpret_cast<uint2*>(sVt + r * (VS * 2) + k8 * 16);
      dst[0] = make_uint2(vreg[u].x, vreg[u].y);
      dst[1] = make_uint2(vreg[u].z, vreg[u].w);
    }
  }
  __syncthreads();   // K / V^T staged

  // ================= stage 2: attention, head by head ==============================================================
  f32x16 acc_o[FN];
#pragma unroll
  for (int f = 0; f < FN; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[f][r] = 0.f;
  float inv_l[HPW];
  const h16x8 zero8 = __builtin_bit_cast(h16x8, make_uint4(0, 0, 0, 0));
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
    constexpr int dummy = 0; (void)dummy;
    const int g0 = hh * GPH;                        // first 8-channel group of the head (wave-local)
    const int kb_lo = g0 / 2, kb_hi = (g0 + GPH - 1) / 2;
    // ---- S^T[key][query] = K . Q^T over the head's channels ----
    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2 * FN; ++kb) {
      if (kb < kb_lo || kb > kb_hi) continue;
      const int g = 2 * kb + half;                  // this lane's 8-channel group of the k-block
      const bool mine = g >= g0 && g < g0 + GPH;    // ... belongs to this head
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h16x8 kf = *reinterpret_cast<const h16x8*>(sK + (32 * t + l31) * (KS * 2) + (wn * NCOL + 16 * kb + 8 * half) * 2);
        if (!mine) kf = zero8;
        s[t] = mfma32x32x16(kf, qf[kb], s[t], 0, 0, 0);
      }
    }
    // ---- softmax over all keys: lane owns query l31 and keys (r&3) + 8*(r>>2) + 4*half of each tile ----
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= p.lk) s[t][r] = -1e30f;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * p.sl2;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], p.sl2, nm));
        psum += s[t][r];
      }
    psum += __shfl_xor(psum, 32, 64);
    inv_l[hh] = 1.0f / psum;
    // ---- O^T[channel][query] += V^T . P^T ; k-slot e of MFMA (t, cc) <-> register 8*cc + e (key permutation matched on V) ----
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        uint4 pv;
        pv.x = pack2h(s[t][8 * cc + 0], s[t][8 * cc + 1]);
        pv.y = pack2h(s[t][8 * cc + 2], s[t][8 * cc + 3]);
        pv.z = pack2h(s[t][8 * cc + 4], s[t][8 * cc + 5]);
        pv.w = pack2h(s[t][8 * cc + 6], s[t][8 * cc + 7]);
        const h16x8 pf = __builtin_bit_cast(h16x8, pv);
#pragma unroll
        for (int f = 0; f < FN; ++f) {
          if (32 * f >= D * (hh + 1) || 32 * f + 32 <= D * hh) continue;      // fragment holds no channel of this head
          const int ch = 32 * f + l31;                                        // wave-local output channel of this lane's V^T row
          const unsigned char* vr = sVt + (wn * NCOL + ch) * (VS * 2) + (32 * t + 16 * cc + 4 * half) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(vr);
          const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
          h16x8 vv = __builtin_bit_cast(h16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          const bool whole = 32 * f >= D * hh && 32 * f + 32 <= D * (hh + 1);  // compile-time: every row belongs to the head
          if (!whole && (ch < D * hh || ch >= D * (hh + 1))) vv = zero8;
          acc_o[f] = mfma32x32x16(vv, pf, acc_o[f], 0, 0, 0);
        }
      }
  }
  __syncthreads();   // every wave is done with K / V^T: the O tile and the Wo ring take their place

  // ================= stage 3: out = O . Wo^T (+ epilogue) ==========================================================
  unsigned char* sO = smx;
  unsigned char* sWo = smx + SO_BYTES;
  issue_pieces<PW, NWAVE>(rsWo, sWo, wave, wooff, 0);
  // O / l, rounded, into the GEMM tile image [k tile][row][64 channels]: lanes l / l ^ 32 trade quads -> 16-byte chunks
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    float v[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c0 = 32 * f + 8 * q + i;          // channel of register 4q+i on the low half; +4 on the high half
        v[q][i] = acc_o[f][4 * q + i] * (half ? inv_l[(c0 + 4) / D] : inv_l[c0 / D]);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned e0[2], e1[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const auto e = __builtin_amdgcn_permlane32_swap(pack2h(v[2 * j][2 * d], v[2 * j][2 * d + 1]),
                                                        pack2h(v[2 * j + 1][2 * d], v[2 * j + 1][2 * d + 1]), false, false);
        e0[d] = e[0]; e1[d] = e[1];
      }
      const int ch = wn * NCOL + 32 * f + 16 * j + 8 * half;       // first of this lane's 8 channels
      *reinterpret_cast<uint4*>(sO + (ch / BK) * A_BYTES + frag_offset(a_row, (ch % BK) / 8)) = make_uint4(e0[0], e0[1], e1[0], e1[1]);
    }
  }
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
  for (int kt = 0; kt < NKT; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): this wave's O-tile writes have landed before the barrier
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < NKT) issue_pieces<PW, NWAVE>(rsWo, sWo + ((kt + 1) & 1) * W_BYTES, wave, wooff, (kt + 1) * BK);
    const unsigned char* sA = sO + kt * A_BYTES;
    const unsigned char* sW = sWo + (kt & 1) * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks * 2 + half;
      const h16x8 xf = *reinterpret_cast<const h16x8*>(sA + frag_offset(a_row, c));
      h16x8 wf[FN];
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const h16x8*>(sW + frag_offset(w_row[a], c));
#pragma unroll
      for (int a = 0; a < FN; ++a) acc[a][0] = mfma32x32x16(wf[a], xf, acc[a][0], 0, 0, 0);
    }
  }
  const float no_pre[2] = {0.f, 0.f};
  // term-at-a-time over the wave's five fragments (80 accumulator registers leave room for the batched loads): the fragment-at-a-time form the
  // shared dispatcher picks for five fragments serialises five dependent L2 round trips per wave — 35 % (audio) / 29 % (text) of this kernel's
  // cycles with per-phase cycle stamps (tools/xattn_bench.py, AVSD_XATTN_TIMING; profiles/r5_xattn_phases.txt)
  epilogue_by_term<FN, 1>(p.epi, acc, row0 + wm * 32, wn * NCOL, lane, 0, no_pre, false);
#undef XATTN_ISSUE1
}

template <int C, int D, int NT>
int launch_xattn(const XAttnArgs& a, hipStream_t s) {
  constexpr size_t lds = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel<C, D, NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("cross_attention_block: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((xattn_kernel<C, D, NT>), dim3((unsigned)(a.epi.M / 128)), dim3(512), lds, s, a);
  AVSD_CHECK_LAUNCH("cross_attention_block launch");
  return AVSD_OK;
}

}  // namespace

extern "C" int avsd_cross_attention_block_supported(int C, int heads, int lk_pad) {
  return (C == 320 && heads == 8 && (lk_pad == 32 || lk_pad == 64 || lk_pad == 96)) ? 1 : 0;
}

extern "C" int avsd_cross_attention_block(const avsd_xattn_desc* dp, void* stream) {
  AVSD_REQUIRE(dp != nullptr, "cross_attention_block: null descriptor");
  const avsd_xattn_desc& d = *dp;
  AVSD_REQUIRE(d.h && d.res && d.ln_stats && d.wq && d.q_colsum && d.q_bias && d.k && d.vt && d.wo && d.out,
               "cross_attention_block: null pointer");
  AVSD_REQUIRE(avsd_cross_attention_block_supported(d.C, d.heads, d.lk_pad),
               "cross_attention_block: unsupported shape C=%d heads=%d lk_pad=%d (built: C=320, 8 heads, lk_pad 32/64/96)", d.C, d.heads, d.lk_pad);
  AVSD_REQUIRE(d.M > 0 && d.M % 128 == 0 && d.L > 0 && d.L % 128 == 0 && d.M % d.L == 0,
               "cross_attention_block: M (%d) and L (%d) must be multiples of 128, M a multiple of L", d.M, d.L);
  AVSD_REQUIRE(d.lk > 0 && d.lk <= d.lk_pad && d.q_per_kv > 0, "cross_attention_block: bad key count / q_per_kv");
  AVSD_REQUIRE(d.ldh % 8 == 0 && d.ldwq % 8 == 0 && d.ldwo % 8 == 0 && d.ldo % 8 == 0 && d.ldh >= d.C && d.ldo >= d.C,
               "cross_attention_block: row strides must be multiples of 8 and >= C");
  AVSD_REQUIRE((double)d.M * d.ldh * 2.0 < 2147483648.0, "cross_attention_block: h must be < 2 GiB");
  AVSD_REQUIRE(d.res_f32 ? (d.ldres % 4 == 0) : (d.ldres % 8 == 0), "cross_attention_block: bad residual stride");
  AVSD_REQUIRE(d.scale > 0.f, "cross_attention_block: scale must be positive");
  XAttnArgs a;
  a.H = (const h16_t*)d.h; a.ldh = d.ldh;
  a.Wq = (const h16_t*)d.wq; a.ldwq = d.ldwq;
  a.q_colsum = d.q_colsum; a.q_bias = d.q_bias;
  a.ln_stats = d.ln_stats; a.ln_nblk = d.C / 32; a.ln_eps = d.ln_eps;
  a.Kc = (const h16_t*)d.k; a.Vt = (const h16_t*)d.vt;
  a.lk = d.lk; a.L = d.L; a.q_per_kv = d.q_per_kv;
  a.sl2 = d.scale * 1.4426950408889634f;
  a.Wo = (const h16_t*)d.wo; a.ldwo = d.ldwo;
  avsd_gemm_desc e = {};
  e.out = d.out; e.ldc = d.ldo; e.bias = d.o_bias; e.res1 = d.res; e.ldr1 = d.ldres;
  e.M = d.M; e.N = d.C; e.K = d.C; e.alpha = 1.0f; e.batch = 1;
  e.flags = (d.res_f32 ? AVSD_GEMM_RES1_F32 : 0) | (d.rowstats ? AVSD_GEMM_ROWSTATS : 0);
  e.rowstats = d.rowstats; e.out_master = d.out_master; e.ldm = d.ldm;
  if (d.out_master) AVSD_REQUIRE(d.ldm % 4 == 0 && d.ldm >= d.C, "cross_attention_block: bad ldm");
  if (d.stats_pos) {
    AVSD_REQUIRE(d.rowstats && d.pos_hw > 0 && d.pos_frames > 0, "cross_attention_block: stats_pos needs rowstats, pos_hw > 0 and pos_frames > 0");
    e.stats_pos = d.stats_pos; e.pos_hw = d.pos_hw; e.pos_frames = d.pos_frames;
  }
  a.epi = e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d.lk_pad) {
    case 32: return launch_xattn<320, 40, 1>(a, s);
    case 64: return launch_xattn<320, 40, 2>(a, s);
    default: return launch_xattn<320, 40, 3>(a, s);
  }
}
