// Attention kernels for gfx950.
//
// attn_kernel<D>: flash-style multi-head attention (first-frame spatial attention and the audio /
//   text cross-attentions).  One workgroup = 128 queries (4 waves x 32) of one (query batch, head);
//   K/V tiles of 32 keys staged in LDS (K row-major, V transposed), scores computed "swapped"
//   (S^T = K . Q^T with v_mfma_f32_32x32x16_bf16) so every lane owns one query column: the online
//   softmax is lane-local plus one cross-half shuffle, and P feeds the P.V MFMA straight from
//   registers (the k-index permutation of the C layout is matched on the V side instead of moving P).
//   O^T = V^T . P^T accumulates with the query again on the lane -> the rescale is lane-local and
//   the final store is 8 bytes of consecutive head channels per lane.
//
// tattn_kernel<D, FMAX>: temporal attention across <= FMAX frames for every pixel: tiny sequences,
//   HBM-bound; one workgroup per (clip branch, pixel), K/V rows in LDS, one thread per (head, query
//   frame), f32 VALU math.
//
// Reference: avgen/models/unets/utils.py:133-156 (K/V from frame 0, SDPA),
// ff_spatio_audio_temp_transformer_3d.py:319-341 (audio + text cross-attention, bool mask -> the
// gather list), :352-358 (temporal attention).  diffusers 0.29.2 AttnProcessor2_0 semantics:
// softmax(q k^T * d^-1/2 + mask) v, masked keys = -inf.
#include "avsd_common.h"
#include <stdlib.h>

namespace {

struct AttnArgs {
  const h16_t* Q; const h16_t* K; const h16_t* V; h16_t* O;
  int ldq, ldk, ldv, ldo;
  int Lq, Lk, kv_rows, q_per_kv, frames;
  const int32_t* key_index;
  float scale;
};

// QB = 32-query blocks per wave (the workgroup covers 128*QB queries); QB = 2 halves the LDS traffic per query.
// (min 2 workgroups per CU caps the kernel at 256 registers, which also makes the compiler keep the MFMA
// accumulators in VGPRs where the softmax reads them, instead of shuttling them through AGPRs.)
template <int D, int QB, bool IDX>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnArgs p) {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of Q.K^T, padded to MFMA K
  constexpr int NCK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32;   // output channels, padded to MFMA M
  constexpr int NDB = DV / 32;
  constexpr int KS = DK + 8;               // sK row stride (elements)
  constexpr int VS = 32 + 4;               // sVt row stride (elements): 72 B -> conflict-free b64 reads
  constexpr int KVEC = D / 8;              // 16-byte vectors per key row
  constexpr int KITEMS = 32 * KVEC;        // K staging: one vector per item
  constexpr int VITEMS = 16 * KVEC;        // V staging: one vector of two adjacent keys per item
  constexpr int NKV = (KITEMS + 255) / 256;
  constexpr int NVV = (VITEMS + 255) / 256;
  constexpr int VROT = (256 - (KITEMS & 255)) & 255;  // V items start on the threads the K items left idle
  // With a spare padded output row, row D of V^T is all ones: the P.V MFMA then also accumulates the softmax
  // denominator (of the bf16-rounded probabilities the numerator uses) and no VALU row sum is needed.
  constexpr bool ONES = DV > D;
  constexpr int LB = D / 32, LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3), LH = ((D % 32) >> 2) & 1;

  __shared__ __attribute__((aligned(16))) h16_t sK[2][32 * KS];
  __shared__ __attribute__((aligned(16))) h16_t sVt[2][DV * VS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int head = blockIdx.y;
  const int qb = blockIdx.z;
  const int kb = qb / p.q_per_kv;
  const int frame = qb % p.frames;
  const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB) + l31;

  // zero both stages once: the padding (K columns D..DK, V^T rows D..DV) is never written again
  for (int i = tid; i < (int)(sizeof(sK) / 16); i += 256) reinterpret_cast<uint4*>(&sK[0][0])[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < (int)(sizeof(sVt) / 16); i += 256) reinterpret_cast<uint4*>(&sVt[0][0])[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (MFMA B operand: lane holds Q[q][16c + 8*half + 0..7]) ----------------------
  h16x8 qf[QB][NCK];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int q = q0 + 32 * j;
    const h16_t* qrow = p.Q + ((int64_t)qb * p.Lq + q) * p.ldq + head * D;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int dd = c * 16 + half * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < p.Lq && dd < D) v = *reinterpret_cast<const uint4*>(qrow + dd);
      qf[j][c] = __builtin_bit_cast(h16x8, v);
    }
  }

  f32x16 acc_o[QB][NDB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[j][b][r] = 0.f;
    m_run[j] = -1e30f;
    l_run[j] = 0.f;
  }

  const h16_t* Kb = p.K + (int64_t)kb * p.kv_rows * p.ldk + head * D;
  const h16_t* Vb = p.V + (int64_t)kb * p.kv_rows * p.ldv + head * D;
  const int32_t* kidx = IDX ? p.key_index + (int64_t)frame * p.Lk : nullptr;   // IDX: keys gathered through a list
  const float sl2 = p.scale * 1.4426950408889634f;   // scores are kept in the log2 domain (v_exp_f32 is 2^x)

  const int ntiles = (p.Lk + 31) / 32;
  // Software pipeline, two tiles deep: the global loads of tile t+2 are issued before the MFMA/softmax work of
  // tile t and parked in registers; the registers holding tile t+1 (issued one tile earlier) are written to the
  // other LDS stage after it; one barrier per tile.  K: [32 keys][DK] row-major.  V: transposed [DV][32 keys] so
  // the P.V operand is a contiguous 8-byte read per lane; a thread transposes the same 8 channels of two adjacent
  // keys, so the LDS writes are 4-byte {key 2j, key 2j+1} pairs.
  // Every thread loads on every tile (idle threads and the tiles past the end are clamped onto valid rows): with
  // no branch around the loads the compiler can count them, and waits for tile t+1 with tile t+2 still in flight.
  // Out-of-range keys of the last tile are clamped to the last valid row instead of zero-filled: their scores
  // are masked to -1e30 below, so their probabilities are exactly 0 and any finite K/V row will do.
  struct Regs { uint4 k[NKV]; uint4 v[NVV][2]; };
  Regs ra, rb;
#pragma unroll
  for (int u = 0; u < NKV; ++u) ra.k[u] = rb.k[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NVV; ++u) ra.v[u][0] = ra.v[u][1] = rb.v[u][0] = rb.v[u][1] = make_uint4(0, 0, 0, 0);
  const int vtid = (tid + 256 - VROT) & 255;   // tid - (KITEMS mod 256), wrapped
  auto gload = [&](int t, Regs& r) {
    t = min(t, ntiles - 1);
    int krow[NKV], vrow[NVV][2];
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = min(tid + u * 256, KITEMS - 1);
      const int kk = min(t * 32 + v / KVEC, p.Lk - 1);
      krow[u] = IDX ? kidx[kk] : kk;
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = min(vtid + u * 256, VITEMS - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = min(t * 32 + 2 * (v & 15) + h, p.Lk - 1);
        vrow[u][h] = IDX ? kidx[kk] : kk;
      }
    }
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = min(tid + u * 256, KITEMS - 1);
      r.k[u] = *reinterpret_cast<const uint4*>(Kb + (int64_t)krow[u] * p.ldk + (v % KVEC) * 8);
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = min(vtid + u * 256, VITEMS - 1);
      r.v[u][0] = *reinterpret_cast<const uint4*>(Vb + (int64_t)vrow[u][0] * p.ldv + (v >> 4) * 8);
      r.v[u][1] = *reinterpret_cast<const uint4*>(Vb + (int64_t)vrow[u][1] * p.ldv + (v >> 4) * 8);
    }
  };
  auto lstore = [&](int st, const Regs& r) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      if (v < KITEMS) {
        const int key = v / KVEC;
        const int dv = (v - key * KVEC) * 8;
        *reinterpret_cast<uint4*>(&sK[st][key * KS + dv]) = r.k[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = vtid + u * 256;
      if (v < VITEMS) {
        const int j = v & 15;
        const int dv = (v >> 4) * 8;
        const uint4 a = r.v[u][0], b = r.v[u][1];
        uint32_t* dst = reinterpret_cast<uint32_t*>(&sVt[st][dv * VS + 2 * j]);
        constexpr int RS = VS / 2;   // row stride in dwords
        dst[0 * RS] = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); dst[1 * RS] = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
        dst[2 * RS] = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); dst[3 * RS] = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
        dst[4 * RS] = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u); dst[5 * RS] = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
        dst[6 * RS] = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u); dst[7 * RS] = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
      }
    }
  };
  gload(0, ra);
  __syncthreads();   // zero fill complete
  if (ONES) {
    if (tid < 32) { sVt[0][D * VS + tid] = AVSD_H16_ONE; sVt[1][D * VS + tid] = AVSD_H16_ONE; }
  }
  lstore(0, ra);
  // every prologue load (Q fragments, tile 0) has landed: tells the compiler's wait-count pass that the loop
  // need not wait on them again, or it would drain the prefetch before the first MFMA of each tile
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  gload(1, ra);

  // one tile: `cur` holds tile t+1 (loaded during tile t-1), `nxt` receives tile t+2
  auto tile = [&](const int t, const Regs& cur, Regs& nxt) {
    const int st = t & 1;
    __syncthreads();   // stage st written; every wave is done reading stage st^1 (tile t-1)
    gload(t + 2, nxt);
    const bool tail = (t + 1 == ntiles) && (p.Lk & 31);

    h16x8 pf[QB][2];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
      // ---- S^T[key][query] = K . Q^T -------------------------------------------------------------
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCK; ++c) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(&sK[st][l31 * KS + c * 16 + half * 8]);
        s = mfma32x32x16(kf, qf[j][c], s, 0, 0, 0);
      }

      // ---- online softmax: lane owns query l31, keys (r&3) + 8*(r>>2) + 4*half -------------------
      if (tail) {
        // A REAL branch (the empty asm statement cannot be speculated): hipcc otherwise if-converts this wave-uniform test into 16 x
        // (add, compare, select) executed on EVERY tile — 48 VALU instructions per tile.  (Measured neutral on the kernel's time, round 5:
        // neither the VALU count nor the LDS operand latency bounds it — issuing all operand reads of a tile behind the barrier costs 15
        // registers = one wave per SIMD and ran 72.0 vs 69.3 us at 32 x 32; the kernel lives on its occupancy, profiles/r5_pmc_attention.md)
        asm volatile("; tail tile" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (t * 32 + key >= p.Lk) s[r] = -1e30f;
        }
      }
      // max(a, b) as med3(a, b, +inf): one instruction, no NaN-quieting pre-pass (scores are finite)
      float pmax = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) pmax = __builtin_amdgcn_fmed3f(pmax, s[r], __builtin_inff());
      const float mt = pmax * sl2;   // max of this lane's 16 keys; the other 16 of the tile sit on lane ^ 32
      // The running max is only raised (and the accumulators rescaled) when some query of the wave exceeds it by
      // more than 2^8: a stale max just scales numerator and denominator alike, and both accumulate in f32.
      // The decision covers this tile's probabilities, which are exponentiated after it; tile t-1's P.V is done.
      if (__builtin_amdgcn_ballot_w64(mt > m_run[j] + 8.f) != 0) {
        const float m_new = fmaxf(m_run[j], fmaxf(mt, __shfl_xor(mt, 32, 64)));   // same value on both halves
        const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
        m_run[j] = m_new;
        l_run[j] *= alpha;
#pragma unroll
        for (int b = 0; b < NDB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc_o[j][b][r] *= alpha;
      }
      const hw_f32x2 nm2 = {-m_run[j], -m_run[j]}, sl22 = {sl2, sl2};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const hw_f32x2 x = {s[r], s[r + 1]};
        const hw_f32x2 y = __builtin_elementwise_fma(x, sl22, nm2);   // v_pk_fma_f32
        s[r] = __builtin_amdgcn_exp2f(y[0]);
        s[r + 1] = __builtin_amdgcn_exp2f(y[1]);
      }
      if (!ONES) {
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) psum += s[r];
        l_run[j] += psum;
      }

      // ---- P fragments (MFMA B operand), k-slot e of MFMA c <-> register 8c+e ------------------
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 v;
        v.x = pack2h(s[8 * c + 0], s[8 * c + 1]);
        v.y = pack2h(s[8 * c + 2], s[8 * c + 3]);
        v.z = pack2h(s[8 * c + 4], s[8 * c + 5]);
        v.w = pack2h(s[8 * c + 6], s[8 * c + 7]);
        pf[j][c] = __builtin_bit_cast(h16x8, v);
      }
    }

    // ---- O^T[dcol][query] += V^T . P^T ; V^T k-slots follow the same key permutation -------------
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const h16_t* vrow = &sVt[st][(b * 32 + l31) * VS + 4 * half];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + 16 * c);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16 * c + 8);
        const h16x8 vv = __builtin_bit_cast(h16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
        for (int j = 0; j < QB; ++j)
          acc_o[j][b] = mfma32x32x16(vv, pf[j][c], acc_o[j][b], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) lstore(st ^ 1, cur);
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile(t, ra, rb);
    if (t + 1 < ntiles) tile(t + 1, rb, ra);
  }

#pragma unroll
  for (int j = 0; j < QB; ++j) {
    float l_tot;
    if (ONES) {
      const float mine = acc_o[j][LB][LR];            // row D of O^T: lanes of half LH hold the denominator
      const float other = __shfl_xor(mine, 32, 64);
      l_tot = (half == LH) ? mine : other;
    } else {
      l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
    }
    const float inv = 1.0f / l_tot;
    const int q = q0 + 32 * j;
    if (q < p.Lq) {
      h16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + head * D;
#pragma unroll
      for (int b = 0; b < NDB; ++b) {
        if (b * 32 + 32 <= D && (p.ldo & 7) == 0) {
          // whole 32-channel block: lanes l and l ^ 32 (same query) trade halves so each stores 32 contiguous bytes
          unsigned x[2][2], y[2][2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto e = __builtin_amdgcn_permlane32_swap(pack2h(acc_o[j][b][2 * d] * inv, acc_o[j][b][2 * d + 1] * inv),
                                                            pack2h(acc_o[j][b][8 + 2 * d] * inv, acc_o[j][b][8 + 2 * d + 1] * inv), false, false);
            const auto o = __builtin_amdgcn_permlane32_swap(pack2h(acc_o[j][b][4 + 2 * d] * inv, acc_o[j][b][4 + 2 * d + 1] * inv),
                                                            pack2h(acc_o[j][b][12 + 2 * d] * inv, acc_o[j][b][12 + 2 * d + 1] * inv), false, false);
            x[0][d] = e[0]; x[1][d] = e[1];
            y[0][d] = o[0]; y[1][d] = o[1];
          }
          h16_t* op = orow + b * 32 + 16 * half;
          *reinterpret_cast<uint4*>(op) = make_uint4(x[0][0], x[0][1], x[1][0], x[1][1]);
          *reinterpret_cast<uint4*>(op + 8) = make_uint4(y[0][0], y[0][1], y[1][0], y[1][1]);
        } else {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int dcol = b * 32 + 8 * qd + 4 * half;
            if (dcol < D) {
              uint2 st;
              st.x = pack2h(acc_o[j][b][4 * qd + 0] * inv, acc_o[j][b][4 * qd + 1] * inv);
              st.y = pack2h(acc_o[j][b][4 * qd + 2] * inv, acc_o[j][b][4 * qd + 3] * inv);
              *reinterpret_cast<uint2*>(orow + dcol) = st;
            }
          }
        }
      }
    }
  }
}

template <int D, int QB>
int launch_attn_qb(const AttnArgs& a, int Bq, int heads, hipStream_t s) {
  dim3 grid((unsigned)((a.Lq + 128 * QB - 1) / (128 * QB)), (unsigned)heads, (unsigned)Bq);
  if (a.key_index) hipLaunchKernelGGL((attn_kernel<D, QB, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_kernel<D, QB, false>), grid, dim3(256), 0, s, a);
  AVSD_CHECK_LAUNCH("attention launch");
  return AVSD_OK;
}

template <int D>
int launch_attn(const AttnArgs& a, int Bq, int heads, hipStream_t s) {
  static const int force_qb = [] { const char* e = getenv("AVSD_ATTN_QB"); return e ? atoi(e) : 0; }();
  // (64 queries per wave measured slower on every UNet shape: 83 vs 72 us on the 32x32 spatial attention)
  const bool wide = force_qb == 2;
  if constexpr (D <= 64) {   // beyond that the second accumulator set no longer fits 256 registers
    if (wide) return launch_attn_qb<D, 2>(a, Bq, heads, s);
  }
  return launch_attn_qb<D, 1>(a, Bq, heads, s);
}

// ---------------------------------------------------------------------------------------------------
// attn_wide_kernel<D>: single wide head (D = 512: the VAE mid-block attention, diffusers AutoencoderKL UNetMidBlock2D with
// one head over all channels).  A 32-query block's output accumulator is D x 32 f32 = 256 registers per lane for one
// wave, so the head dimension is split over the 4 waves of the workgroup instead of the queries: wave w owns channels
// [w D/4, (w+1) D/4) for BOTH products.  Per 32-key tile every wave computes the partial scores of its channel slice,
// the four partials meet in LDS and every wave sums them in the same order (identical S in all waves, so the softmax
// decisions agree), then each wave accumulates P . V for its slice.  Flash-style: no L x L score matrix in memory
// (the three-GEMM form wrote n L^2 f32 scores + as many 16-bit probabilities: 1.6 GB + 0.8 GB per 24-frame 512^2 chunk).
template <int D>
__global__ __launch_bounds__(256, 1) void attn_wide_kernel(const AttnArgs p) {
  constexpr int NW = 4;
  constexpr int DS = D / NW;               // channel slice of a wave
  static_assert(DS % 32 == 0, "slice must be whole 32-channel output fragments");
  constexpr int NCK = DS / 16;
  constexpr int NDB = DS / 32;
  constexpr int KS = D + 8;                // sK row stride (elements)
  constexpr int VS = 32 + 4;               // sVt row stride (elements)
  constexpr int KVEC = D / 8;
  constexpr int KITEMS = 32 * KVEC;        // one 16-byte vector per item
  constexpr int VITEMS = 16 * KVEC;        // the same 8 channels of two adjacent keys per item
  constexpr int NKV = KITEMS / 256, NVV = VITEMS / 256;
  static_assert(KITEMS % 256 == 0 && VITEMS % 256 == 0, "staging items must split evenly over 256 threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  h16_t* sK = reinterpret_cast<h16_t*>(smw);                       // [2][32][KS]
  h16_t* sVt = sK + 2 * 32 * KS;                                    // [2][D][VS]
  float* sS = reinterpret_cast<float*>(sVt + 2 * D * VS);           // [NW][64 lanes][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int qb = blockIdx.z;
  const int kb = qb / p.q_per_kv;
  const int q = blockIdx.x * 32 + l31;
  const int c0 = wave * DS;                // first channel of this wave's slice

  h16x8 qf[NCK];
  {
    const h16_t* qrow = p.Q + ((int64_t)qb * p.Lq + min(q, p.Lq - 1)) * p.ldq + c0;
#pragma unroll
    for (int c = 0; c < NCK; ++c) qf[c] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(qrow + c * 16 + half * 8));
  }
  f32x16 acc_o[NDB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[b][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const h16_t* Kb = p.K + (int64_t)kb * p.kv_rows * p.ldk;
  const h16_t* Vb = p.V + (int64_t)kb * p.kv_rows * p.ldv;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int ntiles = (p.Lk + 31) / 32;

  uint4 rk[NKV], rv[NVV][2];
  auto gload = [&](int t) {
    t = min(t, ntiles - 1);
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      const int kk = min(t * 32 + v / KVEC, p.Lk - 1);
      rk[u] = *reinterpret_cast<const uint4*>(Kb + (int64_t)kk * p.ldk + (v % KVEC) * 8);
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = min(t * 32 + 2 * (v & 15) + h, p.Lk - 1);
        rv[u][h] = *reinterpret_cast<const uint4*>(Vb + (int64_t)kk * p.ldv + (v >> 4) * 8);
      }
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      const int key = v / KVEC;
      *reinterpret_cast<uint4*>(sK + (st * 32 + key) * KS + (v - key * KVEC) * 8) = rk[u];
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
      const int j = v & 15;
      const int dv = (v >> 4) * 8;
      const uint4 a = rv[u][0], b = rv[u][1];
      uint32_t* dst = reinterpret_cast<uint32_t*>(sVt + (st * D + dv) * VS + 2 * j);
      constexpr int RS = VS / 2;
      dst[0 * RS] = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); dst[1 * RS] = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
      dst[2 * RS] = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); dst[3 * RS] = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
      dst[4 * RS] = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u); dst[5 * RS] = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
      dst[6 * RS] = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u); dst[7 * RS] = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
    }
  };
  gload(0);
  lstore(0);
  for (int t = 0; t < ntiles; ++t) {
    const int st = t & 1;
    __syncthreads();                     // stage st written; everybody is done with stage st ^ 1 and with sS
    if (t + 1 < ntiles) gload(t + 1);    // in flight during this tile's matrix work
    // ---- partial S^T[key][query] over this wave's channel slice ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const h16x8 kf = *reinterpret_cast<const h16x8*>(sK + (st * 32 + l31) * KS + c0 + c * 16 + half * 8);
      s = mfma32x32x16(kf, qf[c], s, 0, 0, 0);
    }
    float4* my = reinterpret_cast<float4*>(sS + (wave * 64 + lane) * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) my[j] = make_float4(s[4 * j], s[4 * j + 1], s[4 * j + 2], s[4 * j + 3]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {        // fixed order w = 0..3: bit-identical totals in all four waves
      float4 tot = reinterpret_cast<const float4*>(sS + (0 * 64 + lane) * 16)[j];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 o = reinterpret_cast<const float4*>(sS + (w * 64 + lane) * 16)[j];
        tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w;
      }
      s[4 * j] = tot.x; s[4 * j + 1] = tot.y; s[4 * j + 2] = tot.z; s[4 * j + 3] = tot.w;
    }
    if ((t + 1 == ntiles) && (p.Lk & 31)) {
      asm volatile("; tail tile" ::: "memory");   // keeps the wave-uniform test a real branch (attn_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (t * 32 + key >= p.Lk) s[r] = -1e30f;
      }
    }
    // ---- online softmax (as attn_kernel: log2 domain, lazy running max) ----
    float pmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) pmax = fmaxf(pmax, s[r]);
    const float mt = pmax * sl2;
    if (__builtin_amdgcn_ballot_w64(mt > m_run + 8.f) != 0) {
      const float m_new = fmaxf(m_run, fmaxf(mt, __shfl_xor(mt, 32, 64)));
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[b][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -m_run));
      psum += s[r];
    }
    l_run += psum;
    h16x8 pf[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 v;
      v.x = pack2h(s[8 * c + 0], s[8 * c + 1]);
      v.y = pack2h(s[8 * c + 2], s[8 * c + 3]);
      v.z = pack2h(s[8 * c + 4], s[8 * c + 5]);
      v.w = pack2h(s[8 * c + 6], s[8 * c + 7]);
      pf[c] = __builtin_bit_cast(h16x8, v);
    }
    // ---- O^T[channel][query] += V^T . P^T for this wave's channels ----
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const h16_t* vrow = sVt + (st * D + c0 + b * 32 + l31) * VS + 4 * half;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + 16 * c);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16 * c + 8);
        acc_o[b] = mfma32x32x16(__builtin_bit_cast(h16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)), pf[c], acc_o[b], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) lstore(st ^ 1);
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
  if (q < p.Lq) {
    h16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + c0;
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      unsigned x[2][2], y[2][2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const auto e = __builtin_amdgcn_permlane32_swap(pack2h(acc_o[b][2 * d] * inv, acc_o[b][2 * d + 1] * inv),
                                                        pack2h(acc_o[b][8 + 2 * d] * inv, acc_o[b][8 + 2 * d + 1] * inv), false, false);
        const auto o = __builtin_amdgcn_permlane32_swap(pack2h(acc_o[b][4 + 2 * d] * inv, acc_o[b][4 + 2 * d + 1] * inv),
                                                        pack2h(acc_o[b][12 + 2 * d] * inv, acc_o[b][12 + 2 * d + 1] * inv), false, false);
        x[0][d] = e[0]; x[1][d] = e[1];
        y[0][d] = o[0]; y[1][d] = o[1];
      }
      h16_t* op = orow + b * 32 + 16 * half;
      *reinterpret_cast<uint4*>(op) = make_uint4(x[0][0], x[0][1], x[1][0], x[1][1]);
      *reinterpret_cast<uint4*>(op + 8) = make_uint4(y[0][0], y[0][1], y[1][0], y[1][1]);
    }
  }
}

template <int D>
int launch_attn_wide(const AttnArgs& a, int Bq, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * 32 * (D + 8) * 2 + (size_t)2 * D * 36 * 2 + (size_t)4 * 64 * 16 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_wide_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("attention (wide head): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_wide_kernel<D>), dim3((unsigned)((a.Lq + 31) / 32), 1, (unsigned)Bq), dim3(256), lds, s, a);
  AVSD_CHECK_LAUNCH("attention (wide head) launch");
  return AVSD_OK;
}

// ---------------------------------------------------------------------------------------------------
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct TAttnArgs {
  const h16_t* QKV; h16_t* O;
  int64_t qkv_lo, o_lo;   // split-precision planes (avsd_common.h): offsets to the rest planes, 0 = single plane
  int ldqkv, ldo, frames, hw, heads;
  int hpb;    // heads per workgroup (blockIdx.y = head group): low-resolution layers have too few pixels to fill the chip
  float scale;
};

// One thread per (head, query frame, d-slice): the head dimension is cut into DS slices of SL channels held by DS
// adjacent lanes (partial dot products are summed with 1-2 shuffles), so d = 160 runs 4x the threads of d = 40.
// EXACT: frames == FMAX at compile time (the step's 12 frames, cfg 4's 24) and at most 256 threads: no per-key guards, and the
// 256-register budget of a 256-thread bound (the fully unrolled form spills under the 128 registers of a 1024-thread bound).
// Round 5: the K / V staging issues ALL of a thread's loads (<= U per pass) before the first LDS store — the earlier
// `for (v ...) sKV[v] = g[v]` loop compiled to load / s_waitcnt vmcnt(0) / ds_write per iteration, 7-8 dependent HBM round trips per
// workgroup (26 us for 63 MB at the 32 x 32 level: 2.4 TB/s); and the score dot products run as two independent chains.
template <int D, int FMAX, bool X2, bool EXACT>
__global__ __launch_bounds__(EXACT ? 256 : 1024) void tattn_kernel(const TAttnArgs p) {
  constexpr int SL = (D % 40 == 0) ? 40 : 32;   // slice length
  constexpr int DS = D / SL;                    // 1, 2 or 4 lanes per (head, frame)
  constexpr int NV = SL / 8;                    // 16-byte vectors per slice
  constexpr int U = X2 ? 4 : 8;                 // staging vectors in flight per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
  h16_t* sKV = reinterpret_cast<h16_t*>(smem_t);  // [frames][2*C] : k | v
  const int C = p.heads * D;
  const int Cg = p.hpb * D;             // channels of this workgroup's head group
  const int c0g = blockIdx.y * Cg;      // first channel of the group
  const int F = EXACT ? FMAX : p.frames;
  const int b = blockIdx.x / p.hw;
  const int pix = blockIdx.x - b * p.hw;
  const int64_t row0 = ((int64_t)b * F) * p.hw + pix;  // row of frame f = row0 + f*hw
  const int tid = threadIdx.x;

  const int s = tid % DS;
  const int hi = tid / DS;            // local head * F + i
  const bool active = hi < p.hpb * F;
  const int head = active ? hi / F : 0;
  const int i = active ? hi - head * F : 0;
  const int loff = head * D + s * SL; // this lane's channel slice inside the group
  const int coff = c0g + loff;        // ... and in the full row

  // this thread's query slice is requested first, so its round trip overlaps the K/V staging below instead of
  // following the barrier
  uint4 qv[NV], qr[X2 ? NV : 1];
  {
    const h16_t* qrow = p.QKV + (row0 + (int64_t)i * p.hw) * p.ldqkv + coff;
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      qv[d] = *reinterpret_cast<const uint4*>(qrow + d * 8);
      if constexpr (X2) qr[d] = *reinterpret_cast<const uint4*>(qrow + p.qkv_lo + d * 8);
    }
  }

  const int vec_per_row = 2 * Cg / 8;   // [k slice | v slice] of the group
  h16_t* sKVr = sKV + F * 2 * Cg;       // X2: the rest planes of the same rows
  const int total = F * vec_per_row;
  const int nthr = blockDim.x;
  for (int base = tid; base < total; base += U * nthr) {
    u32x4v t[U], tr[X2 ? U : 1];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int v = min(base + u * nthr, total - 1);          // (clamped: the load is unconditional, the store is not)
      const int f = v / vec_per_row;
      const int cv = (v - f * vec_per_row) * 8;
      const int src = cv < Cg ? C + c0g + cv : 2 * C + c0g + (cv - Cg);
      const h16_t* g = p.QKV + (row0 + (int64_t)f * p.hw) * p.ldqkv + src;
      dst[u] = f * 2 * Cg + cv;
      // (volatile: hipcc otherwise sinks each load next to its store — one s_waitcnt vmcnt(0) per vector; the loads must all be
      // issued before the first store)
      t[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(g));
      if constexpr (X2) tr[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(g + p.qkv_lo));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * nthr < total) {
        *reinterpret_cast<u32x4v*>(sKV + dst[u]) = t[u];
        if constexpr (X2) *reinterpret_cast<u32x4v*>(sKVr + dst[u]) = tr[u];
      }
    }
  }
  __syncthreads();

  float a[X2 ? NV : 1][8];
  if constexpr (X2) {
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      float a2[8];
      unpack8(qv[d], a[d]);
      unpack8(qr[d], a2);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[d][e] += a2[e];
    }
  }
  float sc[FMAX];
  float mx = -1e30f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    float dot = 0.f;
    if (EXACT || j < F) {
      const h16_t* krow = sKV + j * 2 * Cg + loff;
      float d0 = 0.f, d1 = 0.f;         // two independent chains
#pragma unroll
      for (int d = 0; d < NV; ++d) {
        const uint4 kv = *reinterpret_cast<const uint4*>(krow + d * 8);
        if constexpr (X2) {
          float k[8], k2[8];
          unpack8(kv, k);
          unpack8(*reinterpret_cast<const uint4*>(krow + (sKVr - sKV) + d * 8), k2);
#pragma unroll
          for (int e = 0; e < 8; ++e) k[e] += k2[e];
#pragma unroll
          for (int e = 0; e < 8; e += 2) { d0 = fmaf(a[d][e], k[e], d0); d1 = fmaf(a[d][e + 1], k[e + 1], d1); }
        } else {
          // the products of 16-bit values are exact in f32 either way; the packed dot product takes the pairs as stored (the
          // per-pixel sequences are tiny — this kernel is bound by its VALU work, not by HBM: 12 x 12 x d MACs per head and pixel
          // with every K element unpacked once per query frame)
          d0 = dot2h(qv[d].x, kv.x, d0); d1 = dot2h(qv[d].y, kv.y, d1);
          d0 = dot2h(qv[d].z, kv.z, d0); d1 = dot2h(qv[d].w, kv.w, d1);
        }
      }
      dot = d0 + d1;
#pragma unroll
      for (int o = DS / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
      dot *= p.scale;
      mx = fmaxf(mx, dot);
    }
    sc[j] = dot;
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    const float e = (EXACT || j < F) ? __expf(sc[j] - mx) : 0.f;
    sc[j] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  if (!active) return;

  h16_t* orow = p.O + (row0 + (int64_t)i * p.hw) * p.ldo + coff;
#pragma unroll
  for (int d = 0; d < NV; ++d) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (EXACT || j < F) {
        float vv[8];
        unpack8(*reinterpret_cast<const uint4*>(sKV + j * 2 * Cg + Cg + loff + d * 8), vv);
        if constexpr (X2) {
          float v2[8];
          unpack8(*reinterpret_cast<const uint4*>(sKVr + j * 2 * Cg + Cg + loff + d * 8), v2);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] += v2[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(sc[j], vv[e], o[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    store8<X2>(orow + d * 8, p.o_lo, o);
  }
}

template <int D, int FMAX, bool X2, bool EXACT>
int launch_tattn_e(const TAttnArgs& a, int B, size_t lds, int threads, hipStream_t s) {
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_kernel<D, FMAX, X2, EXACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("temporal attention: %zu B LDS: %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_lds = lds;
  }
  hipLaunchKernelGGL((tattn_kernel<D, FMAX, X2, EXACT>), dim3((unsigned)(B * a.hw), (unsigned)(a.heads / a.hpb)), dim3(threads), lds, s, a);
  AVSD_CHECK_LAUNCH("temporal attention launch");
  return AVSD_OK;
}

template <int D, int FMAX, bool X2>
int launch_tattn(const TAttnArgs& a0, int B, hipStream_t s) {
  constexpr int DS = D / ((D % 40 == 0) ? 40 : 32);
  TAttnArgs a = a0;
  // one workgroup per (clip branch, pixel, head group): split the heads until ~1024 workgroups exist
  a.hpb = a.heads;
  while (a.hpb > 1 && a.hpb % 2 == 0 && (long)B * a.hw * (a.heads / a.hpb) < 1024) a.hpb /= 2;
  if (X2) {
    while (a.hpb > 1 && a.hpb % 2 == 0 && (size_t)a.frames * 2 * a.hpb * D * sizeof(h16_t) * 2 > 160 * 1024) a.hpb /= 2;
  }
  const size_t lds = (size_t)a.frames * 2 * a.hpb * D * sizeof(h16_t) * (X2 ? 2 : 1);
  AVSD_REQUIRE(lds <= 160 * 1024, "temporal attention: K / V of %d frames x %d heads per workgroup x %d channels need %zu B of LDS (160 KB; the head count per "
               "workgroup only halves while it is even)", a.frames, a.hpb, D, lds);
  int threads = (a.hpb * a.frames * DS + 63) / 64 * 64;
  if (threads > 1024) {
    avsd_set_error("temporal attention: heads*frames*%d = %d threads exceeds 1024", DS, a.hpb * a.frames * DS);
    return AVSD_EINVAL;
  }
  if (a.frames == FMAX && threads <= 256) return launch_tattn_e<D, FMAX, X2, true>(a, B, lds, threads, s);
  return launch_tattn_e<D, FMAX, X2, false>(a, B, lds, threads, s);
}

template <int D>
int dispatch_tattn(const TAttnArgs& a, int B, hipStream_t s) {
  if (a.qkv_lo != 0) {
    if (a.frames <= 12) return launch_tattn<D, 12, true>(a, B, s);
    if (a.frames <= 24) return launch_tattn<D, 24, true>(a, B, s);
    return launch_tattn<D, 32, true>(a, B, s);
  }
  if (a.frames <= 12) return launch_tattn<D, 12, false>(a, B, s);
  if (a.frames <= 24) return launch_tattn<D, 24, false>(a, B, s);
  return launch_tattn<D, 32, false>(a, B, s);
}

}  // namespace

extern "C" int avsd_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                              void* O, int ldo, int Bq, int Lq, int Lk, int kv_rows, int heads, int d,
                              int q_per_kv, const int32_t* key_index, int frames, float scale,
                              void* stream) {
  AVSD_REQUIRE(Q && K && V && O, "attention: null pointer");
  AVSD_REQUIRE(Bq > 0 && Lq > 0 && Lk > 0 && heads > 0, "attention: bad sizes Bq=%d Lq=%d Lk=%d heads=%d", Bq, Lq, Lk, heads);
  AVSD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention: row strides must be multiples of 8 (ldo: 4)");
  AVSD_REQUIRE(q_per_kv > 0 && Bq % q_per_kv == 0, "attention: Bq (%d) must be a multiple of q_per_kv (%d)", Bq, q_per_kv);
  AVSD_REQUIRE(frames > 0, "attention: frames must be positive");
  AVSD_REQUIRE(scale > 0.f, "attention: scale must be positive");
  AVSD_REQUIRE(kv_rows >= Lk || key_index, "attention: kv_rows (%d) < Lk (%d) without a gather list", kv_rows, Lk);
  AttnArgs a;
  a.Q = (const h16_t*)Q; a.K = (const h16_t*)K; a.V = (const h16_t*)V; a.O = (h16_t*)O;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.Lq = Lq; a.Lk = Lk; a.kv_rows = kv_rows; a.q_per_kv = q_per_kv; a.frames = frames;
  a.key_index = key_index; a.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d) {
    case 40: return launch_attn<40>(a, Bq, heads, s);
    case 64: return launch_attn<64>(a, Bq, heads, s);
    case 80: return launch_attn<80>(a, Bq, heads, s);
    case 128: return launch_attn<128>(a, Bq, heads, s);
    case 160: return launch_attn<160>(a, Bq, heads, s);
    case 512:
      AVSD_REQUIRE(heads == 1 && !key_index && (ldo % 8) == 0, "attention: head dim 512 is the single-head form (VAE mid block): heads == 1, no gather list, ldo %% 8 == 0");
      return launch_attn_wide<512>(a, Bq, s);
    default: AVSD_REQUIRE(false, "attention: unsupported head dim %d (40/64/80/128/160, or 512 with one head)", d);
  }
}

extern "C" int avsd_temporal_attention(const void* QKV, int ldqkv, void* O, int ldo, int B, int frames,
                                       int hw, int heads, int d, float scale, void* stream) {
  return avsd_temporal_attention_x2(QKV, ldqkv, 0, O, ldo, 0, B, frames, hw, heads, d, scale, stream);
}

extern "C" int avsd_temporal_attention_x2(const void* QKV, int ldqkv, int64_t qkv_lo, void* O, int ldo, int64_t o_lo, int B, int frames,
                                          int hw, int heads, int d, float scale, void* stream) {
  AVSD_REQUIRE(QKV && O, "temporal attention: null pointer");
  AVSD_REQUIRE(qkv_lo % 8 == 0 && o_lo % 8 == 0 && (qkv_lo != 0) == (o_lo != 0), "temporal attention: plane offsets must be multiples of 8, both tensors split or neither");
  AVSD_REQUIRE(B > 0 && frames > 0 && frames <= 32 && hw > 0 && heads > 0, "temporal attention: bad sizes (frames <= 32)");
  AVSD_REQUIRE(heads * frames <= 256, "temporal attention: heads*frames (%d) must be <= 256", heads * frames);
  AVSD_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0, "temporal attention: strides must be multiples of 8");
  AVSD_REQUIRE((size_t)frames * 2 * heads * d * 2 <= 160 * 1024, "temporal attention: K/V slab exceeds LDS");
  TAttnArgs a;
  a.QKV = (const h16_t*)QKV; a.O = (h16_t*)O; a.ldqkv = ldqkv; a.ldo = ldo; a.qkv_lo = qkv_lo; a.o_lo = o_lo;
  a.frames = frames; a.hw = hw; a.heads = heads; a.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d) {
    case 40: return dispatch_tattn<40>(a, B, s);
    case 64: return dispatch_tattn<64>(a, B, s);
    case 80: return dispatch_tattn<80>(a, B, s);
    case 128: return dispatch_tattn<128>(a, B, s);
    case 160: return dispatch_tattn<160>(a, B, s);
    default: AVSD_REQUIRE(false, "temporal attention: unsupported head dim %d (40/64/80/128/160)", d);
  }
}
