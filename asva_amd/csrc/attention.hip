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

namespace {

struct AttnArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
  int ldq, ldk, ldv, ldo;
  int Lq, Lk, kv_rows, q_per_kv, frames;
  const int32_t* key_index;
  float scale;
};

template <int D>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p) {
  constexpr int DK = (D + 15) / 16 * 16;   // contraction length of Q.K^T, padded to MFMA K
  constexpr int NCK = DK / 16;
  constexpr int DV = (D + 31) / 32 * 32;   // output channels, padded to MFMA M
  constexpr int NDB = DV / 32;
  constexpr int KS = DK + 8;               // sK row stride (elements)
  constexpr int VS = 32 + 4;               // sVt row stride (elements): 72 B -> conflict-free b64 reads
  constexpr int KVEC = DK / 8;             // 16-byte vectors per key row (K staging)
  constexpr int VVEC = DV / 8;

  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * KS];
  __shared__ __attribute__((aligned(16))) bf16_t sVt[DV * VS];

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
  const bool qvalid = q < p.Lq;

  // ---- Q fragments (MFMA B operand: lane holds Q[q][16c + 8*half + 0..7]) ----------------------
  bf16x8 qf[NCK];
  {
    const bf16_t* qrow = p.Q + ((int64_t)qb * p.Lq + q) * p.ldq + head * D;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int dd = c * 16 + half * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qvalid && dd < D) v = *reinterpret_cast<const uint4*>(qrow + dd);
      qf[c] = __builtin_bit_cast(bf16x8, v);
    }
  }

  f32x16 acc_o[NDB];
#pragma unroll
  for (int b = 0; b < NDB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[b][r] = 0.f;
  float m_run = -1e30f;
  float l_run = 0.f;

  const bf16_t* Kb = p.K + (int64_t)kb * p.kv_rows * p.ldk + head * D;
  const bf16_t* Vb = p.V + (int64_t)kb * p.kv_rows * p.ldv + head * D;
  const int32_t* kidx = p.key_index ? p.key_index + (int64_t)frame * p.Lk : nullptr;

  const int ntiles = (p.Lk + 31) / 32;
  // Software pipeline: the global loads of tile t+1 are issued before the MFMA/softmax work of tile t and parked
  // in registers; they are written to LDS after the tile's trailing barrier.  (K: [32 keys][DK] row-major;
  // V: transposed [DV][32 keys] so the P.V operand is a contiguous 8-byte read per lane.)
  constexpr int NKV = (32 * KVEC + 255) / 256;   // K vectors per thread per tile
  constexpr int NVV = (32 * VVEC + 255) / 256;
  uint4 rk[NKV], rv[NVV];
  auto gload = [&](int t) {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      const int key = v / KVEC;
      const int dv = (v - key * KVEC) * 8;
      const int kk = t * 32 + key;
      rk[u] = make_uint4(0, 0, 0, 0);
      if (v < 32 * KVEC && kk < p.Lk && dv < D) {
        const int row = kidx ? kidx[kk] : kk;
        rk[u] = *reinterpret_cast<const uint4*>(Kb + (int64_t)row * p.ldk + dv);
      }
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
      const int key = v / VVEC;
      const int dv = (v - key * VVEC) * 8;
      const int kk = t * 32 + key;
      rv[u] = make_uint4(0, 0, 0, 0);
      if (v < 32 * VVEC && kk < p.Lk && dv < D) {
        const int row = kidx ? kidx[kk] : kk;
        rv[u] = *reinterpret_cast<const uint4*>(Vb + (int64_t)row * p.ldv + dv);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < NKV; ++u) {
      const int v = tid + u * 256;
      if (v < 32 * KVEC) {
        const int key = v / KVEC;
        const int dv = (v - key * KVEC) * 8;
        *reinterpret_cast<uint4*>(sK + key * KS + dv) = rk[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NVV; ++u) {
      const int v = tid + u * 256;
      if (v < 32 * VVEC) {
        const int key = v / VVEC;
        const int dv = (v - key * VVEC) * 8;
        const uint4 val = rv[u];
        bf16_t* dst = sVt + dv * VS + key;
        dst[0 * VS] = (bf16_t)(val.x & 0xffff); dst[1 * VS] = (bf16_t)(val.x >> 16);
        dst[2 * VS] = (bf16_t)(val.y & 0xffff); dst[3 * VS] = (bf16_t)(val.y >> 16);
        dst[4 * VS] = (bf16_t)(val.z & 0xffff); dst[5 * VS] = (bf16_t)(val.z >> 16);
        dst[6 * VS] = (bf16_t)(val.w & 0xffff); dst[7 * VS] = (bf16_t)(val.w >> 16);
      }
    }
  };
  gload(0);
  for (int t = 0; t < ntiles; ++t) {
    lstore();
    __syncthreads();
    if (t + 1 < ntiles) gload(t + 1);

    // ---- S^T[key][query] = K . Q^T ---------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + l31 * KS + c * 16 + half * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[c], s, 0, 0, 0);
    }

    // ---- online softmax: lane owns query l31, keys (r&3) + 8*(r>>2) + 4*half ---------------------
    float pmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float sv = (t * 32 + key < p.Lk) ? s[r] * p.scale : -1e30f;
      s[r] = sv;
      pmax = fmaxf(pmax, sv);
    }
    pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
    const float m_new = fmaxf(m_run, pmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[b][r] *= alpha;

    // ---- P fragments (MFMA B operand), k-slot e of MFMA c <-> register 8c+e ----------------------
    bf16x8 pf[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 v;
      v.x = pack2bf(s[8 * c + 0], s[8 * c + 1]);
      v.y = pack2bf(s[8 * c + 2], s[8 * c + 3]);
      v.z = pack2bf(s[8 * c + 4], s[8 * c + 5]);
      v.w = pack2bf(s[8 * c + 6], s[8 * c + 7]);
      pf[c] = __builtin_bit_cast(bf16x8, v);
    }

    // ---- O^T[dcol][query] += V^T . P^T ; V^T k-slots follow the same key permutation -------------
#pragma unroll
    for (int b = 0; b < NDB; ++b) {
      const bf16_t* vrow = sVt + (b * 32 + l31) * VS + 4 * half;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + 16 * c);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16 * c + 8);
        const uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
        acc_o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[c], acc_o[b], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    bf16_t* orow = p.O + ((int64_t)qb * p.Lq + q) * p.ldo + head * D;
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int dcol = b * 32 + 8 * qd + 4 * half;
        if (dcol < D) {
          uint2 st;
          st.x = pack2bf(acc_o[b][4 * qd + 0] * inv, acc_o[b][4 * qd + 1] * inv);
          st.y = pack2bf(acc_o[b][4 * qd + 2] * inv, acc_o[b][4 * qd + 3] * inv);
          *reinterpret_cast<uint2*>(orow + dcol) = st;
        }
      }
  }
}

template <int D>
int launch_attn(const AttnArgs& a, int Bq, int heads, hipStream_t s) {
  dim3 grid((unsigned)((a.Lq + 127) / 128), (unsigned)heads, (unsigned)Bq);
  hipLaunchKernelGGL((attn_kernel<D>), grid, dim3(256), 0, s, a);
  AVSD_CHECK_LAUNCH("attention launch");
  return AVSD_OK;
}

// ---------------------------------------------------------------------------------------------------
struct TAttnArgs {
  const bf16_t* QKV; bf16_t* O;
  int ldqkv, ldo, frames, hw, heads;
  float scale;
};

// One thread per (head, query frame, d-slice): the head dimension is cut into DS slices of SL channels held by DS
// adjacent lanes (partial dot products are summed with 1-2 shuffles), so d = 160 runs 4x the threads of d = 40.
template <int D, int FMAX>
__global__ __launch_bounds__(1024) void tattn_kernel(const TAttnArgs p) {
  constexpr int SL = (D % 40 == 0) ? 40 : 32;   // slice length
  constexpr int DS = D / SL;                    // 1, 2 or 4 lanes per (head, frame)
  constexpr int NV = SL / 8;                    // 16-byte vectors per slice
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
  bf16_t* sKV = reinterpret_cast<bf16_t*>(smem_t);  // [frames][2*C] : k | v
  const int C = p.heads * D;
  const int F = p.frames;
  const int b = blockIdx.x / p.hw;
  const int pix = blockIdx.x - b * p.hw;
  const int64_t row0 = ((int64_t)b * F) * p.hw + pix;  // row of frame f = row0 + f*hw
  const int tid = threadIdx.x;

  const int vec_per_row = 2 * C / 8;
  for (int v = tid; v < F * vec_per_row; v += blockDim.x) {
    const int f = v / vec_per_row;
    const int cv = (v - f * vec_per_row) * 8;
    *reinterpret_cast<uint4*>(sKV + f * 2 * C + cv) =
        *reinterpret_cast<const uint4*>(p.QKV + (row0 + (int64_t)f * p.hw) * p.ldqkv + C + cv);
  }
  __syncthreads();

  const int s = tid % DS;
  const int hi = tid / DS;            // head * F + i
  const bool active = hi < p.heads * F;
  const int head = active ? hi / F : 0;
  const int i = active ? hi - head * F : 0;
  const int coff = head * D + s * SL; // this lane's channel slice

  uint4 qv[NV];
  {
    const bf16_t* qrow = p.QKV + (row0 + (int64_t)i * p.hw) * p.ldqkv + coff;
#pragma unroll
    for (int d = 0; d < NV; ++d) qv[d] = *reinterpret_cast<const uint4*>(qrow + d * 8);
  }
  float sc[FMAX];
  float mx = -1e30f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    float dot = 0.f;
    if (j < F) {
      const bf16_t* krow = sKV + j * 2 * C + coff;
#pragma unroll
      for (int d = 0; d < NV; ++d) {
        float a[8], k[8];
        unpack8(qv[d], a);
        unpack8(*reinterpret_cast<const uint4*>(krow + d * 8), k);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot = fmaf(a[e], k[e], dot);
      }
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
    const float e = (j < F) ? __expf(sc[j] - mx) : 0.f;
    sc[j] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  if (!active) return;

  bf16_t* orow = p.O + (row0 + (int64_t)i * p.hw) * p.ldo + coff;
#pragma unroll
  for (int d = 0; d < NV; ++d) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < F) {
        float vv[8];
        unpack8(*reinterpret_cast<const uint4*>(sKV + j * 2 * C + C + coff + d * 8), vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(sc[j], vv[e], o[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    *reinterpret_cast<uint4*>(orow + d * 8) = pack8(o);
  }
}

template <int D, int FMAX>
int launch_tattn(const TAttnArgs& a, int B, hipStream_t s) {
  constexpr int DS = D / ((D % 40 == 0) ? 40 : 32);
  const size_t lds = (size_t)a.frames * 2 * a.heads * D * sizeof(bf16_t);
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tattn_kernel<D, FMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("temporal attention: %zu B LDS: %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_lds = lds;
  }
  int threads = (a.heads * a.frames * DS + 63) / 64 * 64;
  if (threads > 1024) {
    avsd_set_error("temporal attention: heads*frames*%d = %d threads exceeds 1024", DS, a.heads * a.frames * DS);
    return AVSD_EINVAL;
  }
  hipLaunchKernelGGL((tattn_kernel<D, FMAX>), dim3((unsigned)(B * a.hw)), dim3(threads), lds, s, a);
  AVSD_CHECK_LAUNCH("temporal attention launch");
  return AVSD_OK;
}

template <int D>
int dispatch_tattn(const TAttnArgs& a, int B, hipStream_t s) {
  if (a.frames <= 12) return launch_tattn<D, 12>(a, B, s);
  if (a.frames <= 24) return launch_tattn<D, 24>(a, B, s);
  return launch_tattn<D, 32>(a, B, s);
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
  AVSD_REQUIRE(kv_rows >= Lk || key_index, "attention: kv_rows (%d) < Lk (%d) without a gather list", kv_rows, Lk);
  AttnArgs a;
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O;
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
    default: AVSD_REQUIRE(false, "attention: unsupported head dim %d (40/64/80/128/160)", d);
  }
}

extern "C" int avsd_temporal_attention(const void* QKV, int ldqkv, void* O, int ldo, int B, int frames,
                                       int hw, int heads, int d, float scale, void* stream) {
  AVSD_REQUIRE(QKV && O, "temporal attention: null pointer");
  AVSD_REQUIRE(B > 0 && frames > 0 && frames <= 32 && hw > 0 && heads > 0, "temporal attention: bad sizes (frames <= 32)");
  AVSD_REQUIRE(heads * frames <= 256, "temporal attention: heads*frames (%d) must be <= 256", heads * frames);
  AVSD_REQUIRE(ldqkv % 8 == 0 && ldo % 8 == 0, "temporal attention: strides must be multiples of 8");
  AVSD_REQUIRE((size_t)frames * 2 * heads * d * 2 <= 160 * 1024, "temporal attention: K/V slab exceeds LDS");
  TAttnArgs a;
  a.QKV = (const bf16_t*)QKV; a.O = (bf16_t*)O; a.ldqkv = ldqkv; a.ldo = ldo;
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
