// Split-precision ("x2") attention for gfx950: the flash kernel of attention.hip with every operand a (main, rest) pair
// of 16-bit planes (avsd_common.h) and every matrix product three MFMA passes into one f32 accumulator:
//     S^T = K.Q^T + Kr.Q^T + K.Qr^T          O^T += V^T.P^T + Vr^T.P^T + V^T.Pr^T        (P = main + rest of the f32 softmax)
// — 16 significant bits per operand in bf16 (the rest.rest term is 2^-18 of a product), f32 softmax and accumulation.  It is
// the attention of the mode that meets north_star's 1e-3 against the reference's fp32 pipeline (scripts/animation_gen.py:43-44):
// first-frame spatial attention (avgen/models/unets/utils.py:133-156), audio / text cross-attention
// (ff_spatio_audio_temp_transformer_3d.py:315-341; the bool mask as a key gather list), and — attn_x2_wide_kernel — the single
// 512-channel head of the VAE mid block (diffusers AutoencoderKL).
//
// Structure: 128 queries per workgroup (4 waves x 32), 32-key tiles, two LDS stages, the next tile's global loads parked in
// registers during the current tile's matrix work (one tile ahead), scores computed swapped (S^T = K.Q^T) so that a lane owns
// one query: softmax is lane-local plus one cross-half shuffle and P feeds the P.V MFMA from registers.
#include "avsd_common.h"

namespace {

struct AttnX2Args {
  const h16_t* Q; const h16_t* K; const h16_t* V; h16_t* O;
  int64_t q_lo, k_lo, v_lo, o_lo;
  int ldq, ldk, ldv, ldo;
  int Lq, Lk, kv_rows, q_per_kv, frames;
  const int32_t* key_index;
  float scale;
};

// writes the V tile transposed: one thread holds the same 8 channels of two adjacent keys -> 4-byte {key 2j, key 2j+1} pairs
__device__ __forceinline__ void store_vt_pairs(uint32_t* dst, int rs, const uint4& a, const uint4& b) {
  dst[0 * rs] = __builtin_amdgcn_perm(b.x, a.x, 0x05040100u); dst[1 * rs] = __builtin_amdgcn_perm(b.x, a.x, 0x07060302u);
  dst[2 * rs] = __builtin_amdgcn_perm(b.y, a.y, 0x05040100u); dst[3 * rs] = __builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
  dst[4 * rs] = __builtin_amdgcn_perm(b.z, a.z, 0x05040100u); dst[5 * rs] = __builtin_amdgcn_perm(b.z, a.z, 0x07060302u);
  dst[6 * rs] = __builtin_amdgcn_perm(b.w, a.w, 0x05040100u); dst[7 * rs] = __builtin_amdgcn_perm(b.w, a.w, 0x07060302u);
}

// f32 probabilities of one 32-key tile (lane-local, registers 8c+e <-> k-slot e of MFMA c) -> main and rest operand fragments
__device__ __forceinline__ void split_p(const f32x16& s, h16x8 (&pm)[2], h16x8 (&pr)[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    uint4 m, r;
    split2(s[8 * c + 0], s[8 * c + 1], m.x, r.x);
    split2(s[8 * c + 2], s[8 * c + 3], m.y, r.y);
    split2(s[8 * c + 4], s[8 * c + 5], m.z, r.z);
    split2(s[8 * c + 6], s[8 * c + 7], m.w, r.w);
    pm[c] = __builtin_bit_cast(h16x8, m);
    pr[c] = __builtin_bit_cast(h16x8, r);
  }
}

template <int D, bool IDX>
__global__ __launch_bounds__(256, 1) void attn_x2_kernel(const AttnX2Args p) {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of Q.K^T, padded to the MFMA K
  constexpr int NCK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32;   // output channels, padded to the MFMA M
  constexpr int NDB = DV / 32;
  constexpr int KS = DK + 8;               // sK row stride (elements)
  constexpr int VS = 32 + 4;               // sVt row stride (elements)
  constexpr int KVEC = D / 8;
  constexpr int KITEMS = 32 * KVEC;        // K staging: one 16-byte vector per item
  constexpr int VITEMS = 16 * KVEC;        // V staging: the same 8 channels of two adjacent keys per item
  constexpr int NKV = (KITEMS + 255) / 256;
  constexpr int NVV = (VITEMS + 255) / 256;
  constexpr int KTILE = 32 * KS, VTILE = DV * VS;

  // [stage][plane]: plane 0 = main, plane 1 = rest
  __shared__ __attribute__((aligned(16))) h16_t sK[2][2][KTILE];
  __shared__ __attribute__((aligned(16))) h16_t sVt[2][2][VTILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int head = blockIdx.y;
  const int qb = blockIdx.z;
  const int kb = qb / p.q_per_kv;
  const int frame = qb % p.frames;
  const int q = blockIdx.x * 128 + wave * 32 + l31;

  // zero everything once: the padding (K columns D..DK, V^T rows D..DV) is never written again
  for (int i = tid; i < (int)(sizeof(sK) / 16); i += 256) reinterpret_cast<uint4*>(&sK[0][0][0])[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < (int)(sizeof(sVt) / 16); i += 256) reinterpret_cast<uint4*>(&sVt[0][0][0])[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (MFMA B operand: lane holds Q[q][16c + 8 half + 0..7]), both planes ----------------
  h16x8 qm[NCK], qr[NCK];
  {
    const h16_t* qrow = p.Q + ((int64_t)qb * p.Lq + min(q, p.Lq - 1)) * p.ldq + head * D;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int dd = c * 16 + half * 8;
      uint4 a = make_uint4(0, 0, 0, 0), b = a;
      if (dd < D) {
        a = *reinterpret_cast<const uint4*>(qrow + dd);
        b = *reinterpret_cast<const uint4*>(qrow + p.q_lo + dd);
      }
      qm[c] = __builtin_bit_cast(h16x8, a);
      qr[c] = __builtin_bit_cast(h16x8, b);
    }
  }

  f32x16 acc_o[NDB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[b][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const h16_t* Kb = p.K + (int64_t)kb * p.kv_rows * p.ldk + head * D;
  const h16_t* Vb = p.V + (int64_t)kb * p.kv_rows * p.ldv + head * D;
  const int32_t* kidx = IDX ? p.key_index + (int64_t)frame * p.Lk : nullptr;
  const float sl2 = p.scale * 1.4426950408889634f;   // scores are kept in the log2 domain (v_exp_f32 is 2^x)
  const int ntiles = (p.Lk + 31) / 32;

  // out-of-range keys of the last tile are clamped onto the last valid row: their scores are masked to -1e30 below, so
  // their probabilities are exactly 0 and any finite K / V row will do
  uint4 rkm[NKV], rkr[NKV], rvm[NVV][2], rvr[NVV][2];   // K items (main, rest); V items [item][key of the pair]
  auto gload = [&](int t) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = min(tid + u * 256, KITEMS - 1);
      const int kk = min(t * 32 + v / KVEC, p.Lk - 1);
      const int row = IDX ? kidx[kk] : kk;
      const h16_t* g = Kb + (int64_t)row * p.ldk + (v % KVEC) * 8;
      rkm[u] = *reinterpret_cast<const uint4*>(g);
      rkr[u] = *reinterpret_cast<const uint4*>(g + p.k_lo);
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = min(tid + u * 256, VITEMS - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = min(t * 32 + 2 * (v & 15) + h, p.Lk - 1);
        const int row = IDX ? kidx[kk] : kk;
        const h16_t* g = Vb + (int64_t)row * p.ldv + (v >> 4) * 8;
        rvm[u][h] = *reinterpret_cast<const uint4*>(g);
        rvr[u][h] = *reinterpret_cast<const uint4*>(g + p.v_lo);
      }
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      if (v < KITEMS) {
        const int key = v / KVEC;
        const int dv = (v - key * KVEC) * 8;
        *reinterpret_cast<uint4*>(&sK[st][0][key * KS + dv]) = rkm[u];
        *reinterpret_cast<uint4*>(&sK[st][1][key * KS + dv]) = rkr[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
      if (v < VITEMS) {
        const int j = v & 15;
        const int dv = (v >> 4) * 8;
        store_vt_pairs(reinterpret_cast<uint32_t*>(&sVt[st][0][dv * VS + 2 * j]), VS / 2, rvm[u][0], rvm[u][1]);
        store_vt_pairs(reinterpret_cast<uint32_t*>(&sVt[st][1][dv * VS + 2 * j]), VS / 2, rvr[u][0], rvr[u][1]);
      }
    }
  };

  gload(0);
  __syncthreads();   // zero fill complete
  lstore(0);
  for (int t = 0; t < ntiles; ++t) {
    const int st = t & 1;
    __syncthreads();                       // stage st written; every wave is done reading stage st ^ 1 (tile t - 1)
    gload(min(t + 1, ntiles - 1));         // in flight during this tile's matrix work (no branch around the loads: the
                                           // staging registers stay registers; the last tile re-reads itself)
    // ---- S^T[key][query] = K.Q^T + Kr.Q^T + K.Qr^T ---------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const h16x8 km = *reinterpret_cast<const h16x8*>(&sK[st][0][l31 * KS + c * 16 + half * 8]);
      const h16x8 kr = *reinterpret_cast<const h16x8*>(&sK[st][1][l31 * KS + c * 16 + half * 8]);
      s = mfma32x32x16(kr, qm[c], s, 0, 0, 0);
      s = mfma32x32x16(km, qr[c], s, 0, 0, 0);
      s = mfma32x32x16(km, qm[c], s, 0, 0, 0);
    }
    // ---- online softmax: lane owns query l31, keys (r & 3) + 8 (r >> 2) + 4 half -----------------------
    if ((t + 1 == ntiles) && (p.Lk & 31)) {
      asm volatile("; tail tile" ::: "memory");   // keeps the wave-uniform test a real branch (attention.hip attn_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (t * 32 + key >= p.Lk) s[r] = -1e30f;
      }
    }
    float pmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) pmax = fmaxf(pmax, s[r]);
    const float mt = pmax * sl2;           // max of this lane's 16 keys; the other 16 of the tile sit on lane ^ 32
    // eager running max (the precise tier): both halves agree on m_new, every term is scaled exactly once
    const float m_new = fmaxf(m_run, fmaxf(mt, __shfl_xor(mt, 32, 64)));
    if (m_new != m_run) {
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
    h16x8 pm[2], pr[2];
    split_p(s, pm, pr);
    // ---- O^T[channel][query] += V^T.P^T + Vr^T.P^T + V^T.Pr^T ; V^T k-slots follow the same key permutation --------
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const h16_t* vm_row = &sVt[st][0][(b * 32 + l31) * VS + 4 * half];
      const h16_t* vr_row = &sVt[st][1][(b * 32 + l31) * VS + 4 * half];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(vm_row + 16 * c), a1 = *reinterpret_cast<const uint2*>(vm_row + 16 * c + 8);
        const uint2 b0 = *reinterpret_cast<const uint2*>(vr_row + 16 * c), b1 = *reinterpret_cast<const uint2*>(vr_row + 16 * c + 8);
        const h16x8 vm = __builtin_bit_cast(h16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
        const h16x8 vr = __builtin_bit_cast(h16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
        acc_o[b] = mfma32x32x16(vr, pm[c], acc_o[b], 0, 0, 0);
        acc_o[b] = mfma32x32x16(vm, pr[c], acc_o[b], 0, 0, 0);
        acc_o[b] = mfma32x32x16(vm, pm[c], acc_o[b], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) lstore(st ^ 1);
  }

  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
  if (q < p.Lq) {
    h16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + head * D;
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int dcol = b * 32 + 8 * qd + 4 * half;
        if (dcol < D) {
          uint2 st, sr;
          split2(acc_o[b][4 * qd + 0] * inv, acc_o[b][4 * qd + 1] * inv, st.x, sr.x);
          split2(acc_o[b][4 * qd + 2] * inv, acc_o[b][4 * qd + 3] * inv, st.y, sr.y);
          *reinterpret_cast<uint2*>(orow + dcol) = st;
          *reinterpret_cast<uint2*>(orow + p.o_lo + dcol) = sr;
        }
      }
  }
}

template <int D>
int launch_attn_x2(const AttnX2Args& a, int Bq, int heads, hipStream_t s) {
  dim3 grid((unsigned)((a.Lq + 127) / 128), (unsigned)heads, (unsigned)Bq);
  if (a.key_index) hipLaunchKernelGGL((attn_x2_kernel<D, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_x2_kernel<D, false>), grid, dim3(256), 0, s, a);
  AVSD_CHECK_LAUNCH("attention (x2) launch");
  return AVSD_OK;
}

// ---------------------------------------------------------------------------------------------------
// attn_x2_wide_kernel<D>: the single wide head (D = 512) of the VAE mid block, structured like attn_wide_kernel of
// attention.hip: the head dimension is split over the four waves for both products, partial scores meet in LDS and are summed
// in a fixed order.  32 queries per workgroup.
template <int D>
__global__ __launch_bounds__(256, 1) void attn_x2_wide_kernel(const AttnX2Args p) {
  constexpr int NW = 4;
  constexpr int DS = D / NW;               // channel slice of a wave
  static_assert(DS % 32 == 0, "slice must be whole 32-channel output fragments");
  constexpr int NCK = DS / 16;
  constexpr int NDB = DS / 32;
  constexpr int KS = D + 8;
  constexpr int VS = 32 + 4;
  constexpr int KVEC = D / 8;
  constexpr int KITEMS = 32 * KVEC;
  constexpr int VITEMS = 16 * KVEC;
  constexpr int NKV = KITEMS / 256, NVV = VITEMS / 256;
  static_assert(KITEMS % 256 == 0 && VITEMS % 256 == 0, "staging items must split evenly over 256 threads");
  // one stage (both planes) + the score exchange: 2 * 32 * 520 * 2 + 2 * 512 * 36 * 2 + 16 KB = 156 KB; the next tile's loads
  // wait in registers, so one stage suffices (two barriers per tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  h16_t* sK = reinterpret_cast<h16_t*>(smx);                        // [2 planes][32][KS]
  h16_t* sVt = sK + 2 * 32 * KS;                                     // [2 planes][D][VS]
  float* sS = reinterpret_cast<float*>(sVt + 2 * D * VS);            // [NW][64 lanes][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int qb = blockIdx.z;
  const int kb = qb / p.q_per_kv;
  const int q = blockIdx.x * 32 + l31;
  const int c0 = wave * DS;

  h16x8 qm[NCK], qr[NCK];
  {
    const h16_t* qrow = p.Q + ((int64_t)qb * p.Lq + min(q, p.Lq - 1)) * p.ldq + c0;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      qm[c] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(qrow + c * 16 + half * 8));
      qr[c] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(qrow + p.q_lo + c * 16 + half * 8));
    }
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

  uint4 rkm[NKV], rkr[NKV], rvm[NVV][2], rvr[NVV][2];
  auto gload = [&](int t) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      const int kk = min(t * 32 + v / KVEC, p.Lk - 1);
      const h16_t* g = Kb + (int64_t)kk * p.ldk + (v % KVEC) * 8;
      rkm[u] = *reinterpret_cast<const uint4*>(g);
      rkr[u] = *reinterpret_cast<const uint4*>(g + p.k_lo);
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = min(t * 32 + 2 * (v & 15) + h, p.Lk - 1);
        const h16_t* g = Vb + (int64_t)kk * p.ldv + (v >> 4) * 8;
        rvm[u][h] = *reinterpret_cast<const uint4*>(g);
        rvr[u][h] = *reinterpret_cast<const uint4*>(g + p.v_lo);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      const int key = v / KVEC;
      const int o = key * KS + (v - key * KVEC) * 8;
      *reinterpret_cast<uint4*>(sK + o) = rkm[u];
      *reinterpret_cast<uint4*>(sK + 32 * KS + o) = rkr[u];
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
      const int j = v & 15;
      const int dv = (v >> 4) * 8;
      store_vt_pairs(reinterpret_cast<uint32_t*>(sVt + dv * VS + 2 * j), VS / 2, rvm[u][0], rvm[u][1]);
      store_vt_pairs(reinterpret_cast<uint32_t*>(sVt + D * VS + dv * VS + 2 * j), VS / 2, rvr[u][0], rvr[u][1]);
    }
  };
  gload(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();                     // everybody is done with the previous tile's stage and with sS
    lstore();
    __syncthreads();
    gload(min(t + 1, ntiles - 1));       // in flight during this tile's matrix work (unconditional, see attn_x2_kernel)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const h16x8 km = *reinterpret_cast<const h16x8*>(sK + l31 * KS + c0 + c * 16 + half * 8);
      const h16x8 kr = *reinterpret_cast<const h16x8*>(sK + 32 * KS + l31 * KS + c0 + c * 16 + half * 8);
      s = mfma32x32x16(kr, qm[c], s, 0, 0, 0);
      s = mfma32x32x16(km, qr[c], s, 0, 0, 0);
      s = mfma32x32x16(km, qm[c], s, 0, 0, 0);
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
      asm volatile("; tail tile" ::: "memory");   // keeps the wave-uniform test a real branch (attention.hip attn_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (t * 32 + key >= p.Lk) s[r] = -1e30f;
      }
    }
    float pmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) pmax = fmaxf(pmax, s[r]);
    const float mt = pmax * sl2;
    const float m_new = fmaxf(m_run, fmaxf(mt, __shfl_xor(mt, 32, 64)));
    if (m_new != m_run) {
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
    h16x8 pm[2], pr[2];
    split_p(s, pm, pr);
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const h16_t* vm_row = sVt + (c0 + b * 32 + l31) * VS + 4 * half;
      const h16_t* vr_row = vm_row + D * VS;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(vm_row + 16 * c), a1 = *reinterpret_cast<const uint2*>(vm_row + 16 * c + 8);
        const uint2 b0 = *reinterpret_cast<const uint2*>(vr_row + 16 * c), b1 = *reinterpret_cast<const uint2*>(vr_row + 16 * c + 8);
        const h16x8 vm = __builtin_bit_cast(h16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
        const h16x8 vr = __builtin_bit_cast(h16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
        acc_o[b] = mfma32x32x16(vr, pm[c], acc_o[b], 0, 0, 0);
        acc_o[b] = mfma32x32x16(vm, pr[c], acc_o[b], 0, 0, 0);
        acc_o[b] = mfma32x32x16(vm, pm[c], acc_o[b], 0, 0, 0);
      }
    }
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
  if (q < p.Lq) {
    h16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + c0;
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int dcol = b * 32 + 8 * qd + 4 * half;
        uint2 st, sr;
        split2(acc_o[b][4 * qd + 0] * inv, acc_o[b][4 * qd + 1] * inv, st.x, sr.x);
        split2(acc_o[b][4 * qd + 2] * inv, acc_o[b][4 * qd + 3] * inv, st.y, sr.y);
        *reinterpret_cast<uint2*>(orow + dcol) = st;
        *reinterpret_cast<uint2*>(orow + p.o_lo + dcol) = sr;
      }
  }
}

template <int D>
int launch_attn_x2_wide(const AttnX2Args& a, int Bq, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * 32 * (D + 8) * 2 + (size_t)2 * D * 36 * 2 + (size_t)4 * 64 * 16 * 4;
  static_assert(lds <= 160 * 1024, "stage does not fit LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_x2_wide_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("attention (x2, wide head): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_x2_wide_kernel<D>), dim3((unsigned)((a.Lq + 31) / 32), 1, (unsigned)Bq), dim3(256), lds, s, a);
  AVSD_CHECK_LAUNCH("attention (x2, wide head) launch");
  return AVSD_OK;
}

}  // namespace

extern "C" int avsd_attention_x2(const void* Q, int ldq, int64_t q_lo, const void* K, int ldk, int64_t k_lo, const void* V, int ldv,
                                 int64_t v_lo, void* O, int ldo, int64_t o_lo, int Bq, int Lq, int Lk, int kv_rows, int heads, int d,
                                 int q_per_kv, const int32_t* key_index, int frames, float scale, void* stream) {
  AVSD_REQUIRE(Q && K && V && O, "attention_x2: null pointer");
  AVSD_REQUIRE(q_lo != 0 && k_lo != 0 && v_lo != 0 && o_lo != 0 && ((q_lo | k_lo | v_lo) & 7) == 0 && (o_lo & 3) == 0,
               "attention_x2: all four tensors need rest-plane offsets (multiples of 8; output: 4)");
  AVSD_REQUIRE(Bq > 0 && Lq > 0 && Lk > 0 && heads > 0, "attention_x2: bad sizes Bq=%d Lq=%d Lk=%d heads=%d", Bq, Lq, Lk, heads);
  AVSD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention_x2: row strides must be multiples of 8 (ldo: 4)");
  AVSD_REQUIRE(q_per_kv > 0 && Bq % q_per_kv == 0, "attention_x2: Bq (%d) must be a multiple of q_per_kv (%d)", Bq, q_per_kv);
  AVSD_REQUIRE(frames > 0 && scale > 0.f, "attention_x2: frames and scale must be positive");
  AVSD_REQUIRE(kv_rows >= Lk || key_index, "attention_x2: kv_rows (%d) < Lk (%d) without a gather list", kv_rows, Lk);
  AttnX2Args a;
  a.Q = (const h16_t*)Q; a.K = (const h16_t*)K; a.V = (const h16_t*)V; a.O = (h16_t*)O;
  a.q_lo = q_lo; a.k_lo = k_lo; a.v_lo = v_lo; a.o_lo = o_lo;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.Lq = Lq; a.Lk = Lk; a.kv_rows = kv_rows; a.q_per_kv = q_per_kv; a.frames = frames;
  a.key_index = key_index; a.scale = scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d) {
    case 40: return launch_attn_x2<40>(a, Bq, heads, s);
    case 64: return launch_attn_x2<64>(a, Bq, heads, s);
    case 80: return launch_attn_x2<80>(a, Bq, heads, s);
    case 128: return launch_attn_x2<128>(a, Bq, heads, s);
    case 160: return launch_attn_x2<160>(a, Bq, heads, s);
    case 512:
      AVSD_REQUIRE(heads == 1 && !key_index, "attention_x2: head dim 512 is the single-head form (VAE mid block): heads == 1, no gather list");
      return launch_attn_x2_wide<512>(a, Bq, s);
    default: AVSD_REQUIRE(false, "attention_x2: unsupported head dim %d (40/64/80/128/160, or 512 with one head)", d);
  }
}
