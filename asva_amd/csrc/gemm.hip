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

namespace {

constexpr int BK = 64;
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
    r.hb = oh * p.stride - 1;
    r.wb = ow * p.stride - 1;
  }
  return r;
}

template <int MODE>
__device__ __forceinline__ uint4 load_a(const avsd_gemm_desc& p, const bf16_t* A, const bf16_t* A2,
                                        const RowInfo& r, int k0) {
  uint4 z = make_uint4(0, 0, 0, 0);
  if (!r.valid || k0 >= p.K) return z;
  const bf16_t* ptr;
  if (MODE == AVSD_GEMM_PLAIN) {
    ptr = (k0 < p.k_split) ? (A + r.o0 + k0) : (A2 + r.o1 + (k0 - p.k_split));
  } else if (MODE == AVSD_GEMM_TMIX) {
    const int seg = k0 / p.cseg;
    const int kk = k0 - seg * p.cseg;
    const int64_t o = seg == 0 ? r.o0 : (seg == 1 ? r.o1 : r.o2);
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

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const avsd_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);      // [2][BM][LDS_STRIDE]
  bf16_t* sW = sA + 2 * BM * LDS_STRIDE;             // [2][BN][LDS_STRIDE]

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
  const int tn = wg % ntn;
  const int tm = wg / ntn;

  const int64_t bz = blockIdx.z;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + bz * p.batch_stride_a;
  const bf16_t* A2 = p.A2 ? reinterpret_cast<const bf16_t*>(p.A2) + bz * p.batch_stride_a : nullptr;
  const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W) + bz * p.batch_stride_w;

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
    const bf16_t* bA = sA + (buf * BM + wm * (BM / 2) + frow) * LDS_STRIDE + fk;
    const bf16_t* bW = sW + (buf * BN + wn * (BN / 2) + frow) * LDS_STRIDE + fk;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 xf[FM], wf[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xf[b] = *reinterpret_cast<const bf16x8*>(bA + b * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a) wf[a] = *reinterpret_cast<const bf16x8*>(bW + a * 32 * LDS_STRIDE + ks * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds row m = ..+(lane&31), columns n = ..+8q+4*(lane>>5)+{0..3} ----------
  const bool geglu = (p.flags & AVSD_GEMM_GEGLU) != 0;
  const bool out_f32 = (p.flags & AVSD_GEMM_OUT_F32) != 0;
  const bf16_t* R1 = reinterpret_cast<const bf16_t*>(p.res1);
  const bf16_t* R2 = reinterpret_cast<const bf16_t*>(p.res2);
  const int hsel = (lane >> 5) * 4;
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = tm * BM + wm * (BM / 2) + b * 32 + frow;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.ldv : nullptr;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int nb = tn * BN + wn * (BN / 2) + a * 32;  // first packed column of this fragment
      if (nb >= p.N) continue;
      if (!geglu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + 8 * q + hsel;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = p.alpha * acc[a][b][4 * q + i];
          if (p.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
          }
          if (rv) {
            const float4 bb = *reinterpret_cast<const float4*>(rv + n);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
          }
          if (R1) {
            const uint2 rr = *reinterpret_cast<const uint2*>(R1 + bz * p.batch_stride_out + (int64_t)m * p.ldr1 + n);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          }
          if (R2) {
            const uint2 rr = *reinterpret_cast<const uint2*>(R2 + bz * p.batch_stride_out + (int64_t)m * p.ldr2 + n);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          }
          const int64_t o = bz * p.batch_stride_out + (int64_t)m * p.ldc + n;
          if (out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 st;
            st.x = pack2bf(v[0], v[1]);
            st.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = st;
          }
        }
      } else {
        // packed 32-row block = [16 value rows | 16 gate rows]; quads 0,1 hold values, 2,3 gates
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int nval = nb + 8 * q + hsel;   // packed column of the value
          const int ngate = nval + 16;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float val = p.alpha * acc[a][b][4 * q + i];
            float gate = p.alpha * acc[a][b][4 * (q + 2) + i];
            if (p.bias) {
              val += p.bias[nval + i];
              gate += p.bias[ngate + i];
            }
            v[i] = val * gelu_erf_f(gate);
          }
          const int no = (nb >> 1) + 8 * q + hsel;  // output feature
          const int64_t o = bz * p.batch_stride_out + (int64_t)m * p.ldc + no;
          if (out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 st;
            st.x = pack2bf(v[0], v[1]);
            st.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = st;
          }
        }
      }
    }
  }
}

template <int BM, int BN, int MODE>
int launch(const avsd_gemm_desc& d, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("gemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  dim3 grid((unsigned)(ntm * ntn), 1, (unsigned)d.batch);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE>), grid, dim3(256), lds, s, d);
  AVSD_CHECK_LAUNCH("gemm launch");
  return AVSD_OK;
}

template <int MODE>
int dispatch_tile(const avsd_gemm_desc& d, int tile, hipStream_t s) {
  switch (tile) {
    case 1: return launch<128, 128, MODE>(d, s);
    case 2: return launch<128, 64, MODE>(d, s);
    default: return launch<64, 64, MODE>(d, s);
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

extern "C" int avsd_gemm_bf16(const avsd_gemm_desc* dp, void* stream) {
  AVSD_REQUIRE(dp != nullptr, "gemm: null descriptor");
  avsd_gemm_desc d = *dp;
  AVSD_REQUIRE(d.A && d.W && d.out, "gemm: A, W and out must be non-null");
  AVSD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", d.M, d.N, d.K);
  AVSD_REQUIRE(d.K % 8 == 0 && d.ldw % 8 == 0 && d.ldw >= d.K, "gemm: K (%d) and ldw (%d) must be multiples of 8, ldw >= K", d.K, d.ldw);
  AVSD_REQUIRE(d.N % 4 == 0 && d.ldc % 4 == 0, "gemm: N (%d) and ldc (%d) must be multiples of 4", d.N, d.ldc);
  AVSD_REQUIRE(d.lda % 8 == 0, "gemm: lda (%d) must be a multiple of 8", d.lda);
  if (d.batch <= 0) d.batch = 1;
  if (d.flags & AVSD_GEMM_GEGLU) AVSD_REQUIRE(d.N % 32 == 0, "gemm: GEGLU needs N %% 32 == 0 (got %d)", d.N);
  if (d.res1) AVSD_REQUIRE(d.ldr1 % 4 == 0, "gemm: ldr1 must be a multiple of 4");
  if (d.res2) AVSD_REQUIRE(d.ldr2 % 4 == 0, "gemm: ldr2 must be a multiple of 4");
  if (d.rowvec) AVSD_REQUIRE(d.rows_per_vec > 0 && d.ldv % 4 == 0, "gemm: rowvec needs rows_per_vec > 0 and ldv %% 4 == 0");
  if (d.mode == AVSD_GEMM_PLAIN) {
    if (!d.A2) d.k_split = d.K;
    AVSD_REQUIRE(d.k_split % 8 == 0 && d.k_split <= d.K, "gemm: k_split (%d) must be a multiple of 8 and <= K", d.k_split);
    if (d.A2) AVSD_REQUIRE(d.lda2 % 8 == 0, "gemm: lda2 must be a multiple of 8");
  } else if (d.mode == AVSD_GEMM_TMIX) {
    AVSD_REQUIRE(d.cseg > 0 && d.cseg % 8 == 0 && d.K == 3 * d.cseg, "gemm/tmix: K (%d) must equal 3*cseg (%d)", d.K, d.cseg);
    AVSD_REQUIRE(d.hw > 0 && d.frames > 0 && d.M % (d.hw * d.frames) == 0, "gemm/tmix: M (%d) must be a multiple of frames*hw (%d*%d)", d.M, d.frames, d.hw);
  } else if (d.mode == AVSD_GEMM_CONV3) {
    AVSD_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.K == 9 * d.cin, "gemm/conv3: K (%d) must equal 9*cin (%d), cin %% 8 == 0", d.K, d.cin);
    AVSD_REQUIRE(d.stride == 1 || d.stride == 2, "gemm/conv3: stride must be 1 or 2");
    AVSD_REQUIRE(d.ups == 0 || d.ups == 1, "gemm/conv3: ups must be 0 or 1");
    AVSD_REQUIRE(d.hs > 0 && d.ws > 0 && d.ho > 0 && d.wo > 0 && d.M % (d.ho * d.wo) == 0, "gemm/conv3: bad image geometry");
    const int hin = d.hs << d.ups, win = d.ws << d.ups;
    AVSD_REQUIRE(d.ho == (hin + 2 - 3) / d.stride + 1 && d.wo == (win + 2 - 3) / d.stride + 1, "gemm/conv3: (ho,wo)=(%d,%d) inconsistent with input (%d,%d) stride %d", d.ho, d.wo, hin, win, d.stride);
  } else {
    AVSD_REQUIRE(false, "gemm: unknown mode %d", d.mode);
  }
  int tile = d.tile;
  if (tile < 1 || tile > 3) tile = pick_tile(d.M, (d.flags & AVSD_GEMM_GEGLU) ? d.N : d.N, d.batch);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d.mode) {
    case AVSD_GEMM_PLAIN: return dispatch_tile<AVSD_GEMM_PLAIN>(d, tile, s);
    case AVSD_GEMM_TMIX: return dispatch_tile<AVSD_GEMM_TMIX>(d, tile, s);
    default: return dispatch_tile<AVSD_GEMM_CONV3>(d, tile, s);
  }
}
