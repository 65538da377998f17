// FP8 (OCP e4m3) Q / K / V attention for gfx950 — BASELINE.json configuration 5 ("fp8 (gfx950 MFMA) attention QKV path").
//
// Same flash structure, tiling and entry contract as attn_kernel (attention.hip): 128 queries per workgroup, 32-key
// tiles through two LDS stages, scores computed swapped (S^T = K . Q^T) so the softmax is lane-local in f32, lazy
// running-max update.  What changes is the operand type of both matrix products: Q, K, V and the probabilities P are
// rounded to e4m3 and fed to v_mfma_f32_32x32x16_fp8_fp8 (8 one-byte elements per lane per operand instead of 8 two-byte
// ones: half the LDS bytes per tile; on gfx950 the non-scaled fp8 MFMA runs at the bf16 rate, so the gain is LDS /
// staging traffic, not matrix throughput).  Softmax, the running statistics and all accumulation stay f32 (north_star).
//   * q, k, v are multiplied by caller-supplied per-tensor scales before rounding (1.0 keeps the raw values: e4m3 holds
//     |x| <= 448 with 3 mantissa bits, exact powers of two are free); 1 / (q_scale k_scale) folds into the softmax
//     scale and 1 / v_scale into the output normalisation.
//   * P in [0, 1] is scaled by 2^8 before rounding so that probabilities down to 2^-14 stay in e4m3's normal range;
//     the factor cancels against the denominator, which is accumulated from the same rounded values (an all-ones
//     row of V^T when the head dimension leaves a spare MFMA row, the f32 sum of the rounded values otherwise).
// Inputs and outputs stay 16-bit tensors: the conversion happens while staging (K, V: once per tile into LDS; Q: once per
// workgroup in registers), so the kernel drops in wherever attn_kernel runs.
// Reference sites it replaces under cfg 5: the SDPA calls of utils.py:151-153 (first-frame attention) and
// ff_spatio_audio_temp_transformer_3d.py:315-341 (audio / text cross-attention).
#include "avsd_common.h"

namespace {

struct AttnF8Args {
  const h16_t* Q; const h16_t* K; const h16_t* V; h16_t* O;
  int ldq, ldk, ldv, ldo;
  int Lq, Lk, kv_rows, q_per_kv, frames;
  const int32_t* key_index;
  float sl2;        // softmax scale * log2(e) / (q_scale * k_scale)
  float q_scale, k_scale, v_scale;
};

constexpr float F8_MAX = 448.0f;

__device__ __forceinline__ float clamp8(float x) { return fminf(fmaxf(x, -F8_MAX), F8_MAX); }
// four f32 -> four e4m3 bytes (round to nearest even, saturating by the clamp)
__device__ __forceinline__ uint32_t pack4_f8(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(a), clamp8(b), v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(c), clamp8(d), v, true);
  return (uint32_t)v;
}
// eight 16-bit values (one uint4) * scale -> eight e4m3 bytes
__device__ __forceinline__ uint2 cvt8_f8(const uint4& x, float scale) {
  float f[8];
  unpack8(x, f);
  return make_uint2(pack4_f8(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale),
                    pack4_f8(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale));
}
__device__ __forceinline__ long as_long(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | (unsigned long)lo); }

template <int D, bool IDX>
__global__ __launch_bounds__(256, 2) void attn_f8_kernel(const AttnF8Args p) {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of Q.K^T, padded to the MFMA K
  constexpr int NCK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32;   // output channels, padded to the MFMA M
  constexpr int NDB = DV / 32;
  constexpr int KS = DK + 8;               // sK row stride (bytes): conflict-free ds_read_b64
  constexpr int VS = 32 + 4;               // sVt row stride (bytes): conflict-free ds_read_b32
  constexpr int KVEC = D / 8;
  constexpr int KITEMS = 32 * KVEC;
  constexpr int VITEMS = 16 * KVEC;
  constexpr int NKV = (KITEMS + 255) / 256;
  constexpr int NVV = (VITEMS + 255) / 256;
  constexpr int VROT = (256 - (KITEMS & 255)) & 255;
  constexpr bool ONES = DV > D;
  constexpr int LB = D / 32, LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3), LH = ((D % 32) >> 2) & 1;

  __shared__ __attribute__((aligned(16))) unsigned char sK[2][32 * KS];
  __shared__ __attribute__((aligned(16))) unsigned char sVt[2][DV * VS];

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

  for (int i = tid; i < (int)(sizeof(sK) / 16); i += 256) reinterpret_cast<uint4*>(&sK[0][0])[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < (int)(sizeof(sVt) / 16); i += 256) reinterpret_cast<uint4*>(&sVt[0][0])[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments, e4m3 (MFMA B operand: lane holds Q[q][16c + 8*half + 0..7]) ----
  long qf[NCK];
  {
    const h16_t* qrow = p.Q + ((int64_t)qb * p.Lq + q) * p.ldq + head * D;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int dd = c * 16 + half * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < p.Lq && dd < D) v = *reinterpret_cast<const uint4*>(qrow + dd);
      const uint2 f8 = cvt8_f8(v, p.q_scale);
      qf[c] = as_long(f8.x, f8.y);
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
  const float sl2 = p.sl2;
  const int ntiles = (p.Lk + 31) / 32;

  struct Regs { uint4 k[NKV]; uint4 v[NVV][2]; };
  Regs ra, rb;
#pragma unroll
  for (int u = 0; u < NKV; ++u) ra.k[u] = rb.k[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NVV; ++u) ra.v[u][0] = ra.v[u][1] = rb.v[u][0] = rb.v[u][1] = make_uint4(0, 0, 0, 0);
  const int vtid = (tid + 256 - VROT) & 255;
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
  // K: [32 keys][DK bytes] row-major; V: transposed [DV][32 keys], a thread converts the same 8 channels of two adjacent
  // keys and writes 2-byte {key 2j, key 2j+1} pairs
  auto lstore = [&](int st, const Regs& r) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      if (v < KITEMS) {
        const int key = v / KVEC;
        const int dv = (v - key * KVEC) * 8;
        *reinterpret_cast<uint2*>(&sK[st][key * KS + dv]) = cvt8_f8(r.k[u], p.k_scale);
      }
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = vtid + u * 256;
      if (v < VITEMS) {
        const int j = v & 15;
        const int dv = (v >> 4) * 8;
        const uint2 a = cvt8_f8(r.v[u][0], p.v_scale), b = cvt8_f8(r.v[u][1], p.v_scale);
        const uint32_t aw[2] = {a.x, a.y}, bw[2] = {b.x, b.y};
        unsigned short* dst = reinterpret_cast<unsigned short*>(&sVt[st][dv * VS + 2 * j]);
        constexpr int RS = VS / 2;   // row stride in 2-byte units
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t ka = (aw[e >> 2] >> (8 * (e & 3))) & 0xffu, kbv = (bw[e >> 2] >> (8 * (e & 3))) & 0xffu;
          dst[e * RS] = (unsigned short)(ka | (kbv << 8));
        }
      }
    }
  };
  gload(0, ra);
  __syncthreads();
  if (ONES) {
    if (tid < 32) { sVt[0][D * VS + tid] = 0x38; sVt[1][D * VS + tid] = 0x38; }     // e4m3 1.0
  }
  lstore(0, ra);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the prologue loads have landed (see attention.hip)
  gload(1, ra);

  auto tile = [&](const int t, const Regs& cur, Regs& nxt) {
    const int st = t & 1;
    __syncthreads();
    gload(t + 2, nxt);
    const bool tail = (t + 1 == ntiles) && (p.Lk & 31);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const uint2 kf = *reinterpret_cast<const uint2*>(&sK[st][l31 * KS + c * 16 + half * 8]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(as_long(kf.x, kf.y), qf[c], s, 0, 0, 0);
    }
    if (tail) {
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
    // unlike the 16-bit kernel the running max is raised whenever it is exceeded: the probabilities must stay <= 1 so that
    // 2^8 p fits e4m3 (a stale max would let p grow to 2^8 and saturate at 448)
    if (__builtin_amdgcn_ballot_w64(mt > m_run) != 0) {
      const float m_new = fmaxf(m_run, fmaxf(mt, __shfl_xor(mt, 32, 64)));
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[b][r] *= alpha;
    }
    // probabilities (<= 1 after the max update above), scaled by 2^8 (exponent shift: exact), rounded to e4m3
    uint32_t pw[4];
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(fmaf(s[4 * w4 + i], sl2, -m_run) + 8.0f);
      pw[w4] = pack4_f8(e[0], e[1], e[2], e[3]);
      if (!ONES) {   // sum of the ROUNDED values (byte select must be an immediate)
        const int w = (int)pw[w4];
        l_run += (__builtin_amdgcn_cvt_f32_fp8(w, 0) + __builtin_amdgcn_cvt_f32_fp8(w, 1)) +
                 (__builtin_amdgcn_cvt_f32_fp8(w, 2) + __builtin_amdgcn_cvt_f32_fp8(w, 3));
      }
    }
    // O^T += V^T . P^T; k-slot e of MFMA c <-> register 8c + e (key permutation matched on the V side, see attention.hip)
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const unsigned char* vrow = &sVt[st][(b * 32 + l31) * VS + 4 * half];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t lo = *reinterpret_cast<const uint32_t*>(vrow + 16 * c);
        const uint32_t hi = *reinterpret_cast<const uint32_t*>(vrow + 16 * c + 8);
        acc_o[b] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(as_long(lo, hi), as_long(pw[2 * c], pw[2 * c + 1]), acc_o[b], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) lstore(st ^ 1, cur);
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile(t, ra, rb);
    if (t + 1 < ntiles) tile(t + 1, rb, ra);
  }

  float l_tot;
  if constexpr (ONES) {
    const float mine = acc_o[LB][LR];
    const float other = __shfl_xor(mine, 32, 64);
    l_tot = (half == LH) ? mine : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / (l_tot * p.v_scale);      // P's 2^8 sits in both the numerator and l_tot
  if (q < p.Lq) {
    h16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + head * D;
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int dcol = b * 32 + 8 * qd + 4 * half;
        if (dcol < D) {
          uint2 st;
          st.x = pack2h(acc_o[b][4 * qd + 0] * inv, acc_o[b][4 * qd + 1] * inv);
          st.y = pack2h(acc_o[b][4 * qd + 2] * inv, acc_o[b][4 * qd + 3] * inv);
          *reinterpret_cast<uint2*>(orow + dcol) = st;
        }
      }
  }
}

template <int D>
int launch_attn_f8(const AttnF8Args& a, int Bq, int heads, hipStream_t s) {
  dim3 grid((unsigned)((a.Lq + 127) / 128), (unsigned)heads, (unsigned)Bq);
  if (a.key_index) hipLaunchKernelGGL((attn_f8_kernel<D, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_f8_kernel<D, false>), grid, dim3(256), 0, s, a);
  AVSD_CHECK_LAUNCH("attention_fp8 launch");
  return AVSD_OK;
}

}  // namespace

extern "C" int avsd_attention_fp8(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                                  int Bq, int Lq, int Lk, int kv_rows, int heads, int d, int q_per_kv,
                                  const int32_t* key_index, int frames, float scale, float q_scale, float k_scale,
                                  float v_scale, void* stream) {
  AVSD_REQUIRE(Q && K && V && O, "attention_fp8: null pointer");
  AVSD_REQUIRE(Bq > 0 && Lq > 0 && Lk > 0 && heads > 0, "attention_fp8: bad sizes Bq=%d Lq=%d Lk=%d heads=%d", Bq, Lq, Lk, heads);
  AVSD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention_fp8: row strides must be multiples of 8 (ldo: 4)");
  AVSD_REQUIRE(q_per_kv > 0 && Bq % q_per_kv == 0, "attention_fp8: Bq (%d) must be a multiple of q_per_kv (%d)", Bq, q_per_kv);
  AVSD_REQUIRE(frames > 0 && scale > 0.f, "attention_fp8: frames and scale must be positive");
  AVSD_REQUIRE(q_scale > 0.f && k_scale > 0.f && v_scale > 0.f, "attention_fp8: tensor scales must be positive");
  AVSD_REQUIRE(kv_rows >= Lk || key_index, "attention_fp8: kv_rows (%d) < Lk (%d) without a gather list", kv_rows, Lk);
  AttnF8Args a;
  a.Q = (const h16_t*)Q; a.K = (const h16_t*)K; a.V = (const h16_t*)V; a.O = (h16_t*)O;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.Lq = Lq; a.Lk = Lk; a.kv_rows = kv_rows; a.q_per_kv = q_per_kv; a.frames = frames;
  a.key_index = key_index;
  a.sl2 = scale * 1.4426950408889634f / (q_scale * k_scale);
  a.q_scale = q_scale; a.k_scale = k_scale; a.v_scale = v_scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d) {
    case 40: return launch_attn_f8<40>(a, Bq, heads, s);
    case 64: return launch_attn_f8<64>(a, Bq, heads, s);
    case 80: return launch_attn_f8<80>(a, Bq, heads, s);
    case 128: return launch_attn_f8<128>(a, Bq, heads, s);
    case 160: return launch_attn_f8<160>(a, Bq, heads, s);
    default: AVSD_REQUIRE(false, "attention_fp8: unsupported head dim %d (40/64/80/128/160)", d);
  }
}
