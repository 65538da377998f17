// Fused GEGLU feed-forward block for gfx950: one launch for the last residual update of BasicTransformerBlock
//
//     h' = h + W2 . ( value * gelu_erf(gate) ) + b2 ,   [value | gate] = LayerNorm3(h) . W1^T + b1
//
// (ff_spatio_audio_temp_transformer_3d.py:361-371; diffusers FeedForward with GEGLU).  It replaces two launches — the
// LayerNorm-folded GEGLU projection C -> 2 x 4C and the output projection 4C -> C — and the round trip of the M x 4C hidden
// tensor through memory (63 MB written and read back per block at the 32 x 32 level), which never exists here.
//
// One workgroup = 96 rows of h (24576 rows per clip = 256 workgroups = one per CU) walks the hidden dimension in chunks of 16
// features; the work of a chunk is split over three kinds of waves that run as a pipeline, ONE barrier per chunk:
//   3 A waves       wave w owns rows 32w..32w+31.  Its X fragments (one row per lane, all of K) live in REGISTERS for the whole
//                   kernel, so stage A  S[32 x 32] = X . W1c^T  (W1c = 32 packed rows [16 value | 16 gate] x C) reads one LDS
//                   fragment per MFMA; beside the MFMAs of chunk j the wave finishes chunk j-1 on the vector pipe: LayerNorm fold
//                   (rstd, mean from the producer's row statistics), bias, value * gelu(gate), rounding, and the half-row trade
//                   (v_permlane32_swap) into MFMA operand order -> P[32 x 16] in LDS
//   3 B waves       wave w, on the same SIMD, owns the same rows:  acc[32 x C] += P . W2c^T  (W2c = C x 16, the host packs W2
//                   chunk-major; K = 16, 10 accumulators), two chunks behind
//   2 loader waves  stream the weight chunks (LDS-direct loads, 30 KB per chunk) TWO chunks ahead into rings (W1c x 3, W2c x 3,
//                   fold terms x 4) and wait for them with counted vmcnt; each piece costs the issuing wave 60-180 cycles
//                   (MI355X_MICROARCH.md), which the MFMA waves never pay
// History of the design (tools/ffn_bench.py, 24576-row layer; the two GEMMs: 127 us): all three jobs in the same four waves with
// the activation tile in LDS 172 us, + loader waves 133 us, S handed to separate VALU waves through LDS 154 us (stage A then
// needs two LDS fragments per MFMA from one wave per SIMD: LDS-latency bound) — hence X in registers and the GELU beside stage A.
// The f32 epilogue (bias, residual, 16-bit stream + optional f32 master) is the shared one of the GEMM family.
#include "avsd_common.h"
#include "gemm_common.h"

#ifndef FFN_ABL
#define FFN_ABL 0      // probe builds: 1 no GELU, 2 no stage-A MFMAs, 3 no stage-B MFMAs, 4 no weight streaming
#endif

namespace {

struct FfnArgs {
  const h16_t* H; int ldh;
  const h16_t* W1; int ldw1;             // [2 nh][C] packed, LayerNorm gain folded in
  const float* cb1;                      // [nh / 16][2][32]: colsum | bias of the 32 packed columns of every chunk
  const float* ln_stats; int ln_nblk; float ln_eps;
  const h16_t* W2c;                      // [nh / 16][C][16]
  int nh;
  avsd_gemm_desc epi;                    // output side: out / ldc / bias / res1 / flags / out_master / M / N
};

template <int C>
__global__ __launch_bounds__(512) void ffn_kernel(const FfnArgs p) {
  constexpr int BM = 96, NLW = 2;
  constexpr int NKT = C / BK;                       // 64-wide K tiles of stage A
  constexpr int NKS = C / 16;                       // MFMA k-steps of stage A
  constexpr int W1_BYTES = NKT * 32 * 128;          // one chunk of W1: NKT images of [32][64]
  constexpr int W1_PIECES = NKT * 4;
  constexpr int W2_PIECES = C * 32 / 1024;          // one chunk of W2c: [C][16] = C * 32 bytes, linear
  constexpr int W2_BYTES = W2_PIECES * 1024;
  constexpr int CB_SLOT = 512;                      // fold terms of a chunk: [colsum 32 | bias 32] f32 (+ the second loader's copy)
  constexpr int P_BYTES = BM * 32;                  // P[96][16] 16-bit
  constexpr int FNB = C / 32;                       // stage-B column fragments (a B wave owns all C channels of its rows)
  constexpr int L1 = W1_PIECES / NLW, L2 = W2_PIECES / NLW;   // loads per loader per chunk
  static_assert(C % 64 == 0 && (C * 32) % 1024 == 0 && W1_PIECES % NLW == 0 && W2_PIECES % NLW == 0 && NKS % 5 == 0, "shape");
  static_assert(3 * W1_BYTES + 3 * W2_BYTES + 2 * P_BYTES + 4 * CB_SLOT <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smf[];
  unsigned char* sW1 = smf;                         // [3][W1_BYTES]
  unsigned char* sW2 = sW1 + 3 * W1_BYTES;          // [3][W2_BYTES]
  unsigned char* sP = sW2 + 3 * W2_BYTES;           // [2][P_BYTES]
  unsigned char* sCB = sP + 2 * P_BYTES;            // [4][CB_SLOT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves of a workgroup go to the SIMDs cyclically, so waves w and w + 4 share one: A wave rb and B wave rb (the same 32 rows)
  // sit on the same SIMD, the loaders 3 and 7 on the fourth
  const bool is_loader = (wave & 3) == 3;
  const bool is_a = wave < 3;
  const int lw = wave >> 2;                         // loader index (waves 3, 7)
  const int rb = wave & 3;                          // row block of an A / B wave
  const int half = lane >> 5, l31 = lane & 31, hsel = half * 4;
  int tm;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int row0 = tm * BM;
  const int n = p.nh / 16;                          // chunks

  if (is_loader) {
    // ======================= loader waves ================================================================================
    // iteration j issues W1c(j+2) + fold terms(j+2) (needed by A at j+2 / j+3) and W2c(j) (needed by B at j+2), then waits for
    // everything issued BEFORE this iteration: two chunks are always in flight
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2c, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsCB = __builtin_amdgcn_make_buffer_rsrc((void*)p.cb1, 0, 0x7fffffff, 0x00020000);
    // (literal bound: hipcc 7.2 drops the kernel's host stub when a lambda captures an array whose bound depends on a template
    // parameter — see the note in xattn.hip)
    static_assert(L1 <= 16, "w1off");
    int w1off[16];
#pragma unroll
    for (int u = 0; u < L1; ++u) {
      const int pc = lw + u * NLW, kt = pc >> 2, q = pc & 3;
      int r, kc;
      piece_row_chunk(q, lane, r, kc);
      w1off[u] = r * p.ldw1 + kt * BK + kc;
    }
    auto issue_w1 = [&](int j) {             // L1 + 1 loads
      unsigned char* d1 = sW1 + (j % 3) * W1_BYTES;
#pragma unroll
      for (int u = 0; u < L1; ++u) {
        const int pc = lw + u * NLW, kt = pc >> 2, q = pc & 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr_t)(d1 + kt * (32 * 128) + q * 1024), 16, (j * 32 * p.ldw1 + w1off[u]) * 2, 0, 0, 0);
      }
      // the fold terms of the chunk: 64 floats, one per lane (both loaders issue it so their load counts match; loader 0's copy is read)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsCB, (lds_ptr_t)(sCB + (j & 3) * CB_SLOT + lw * 256), 4, (j * 64 + lane) * 4, 0, 0, 0);
    };
    auto issue_w2 = [&](int j) {             // L2 loads
      unsigned char* d2 = sW2 + (j % 3) * W2_BYTES;
#pragma unroll
      for (int u = 0; u < L2; ++u) {
        const int piece = lw + u * NLW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (lds_ptr_t)(d2 + piece * 1024), 16, j * (C * 32) + piece * 1024 + lane * 16, 0, 0, 0);
      }
    };
    issue_w1(0);
    if (n > 1) issue_w1(1);
    if (n > 1) wait_vmcnt<L1 + 1>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();            // W1c(0) is in LDS
    for (int j = 0; j < n + 2; ++j) {
      const bool a = j + 2 < n, b = j < n;
#if FFN_ABL != 4
      if (a) issue_w1(j + 2);
      if (b) issue_w2(j);
#endif
#if FFN_ABL != 4
      if (a) wait_vmcnt<L1 + 1 + L2>();      // (a implies b)
      else if (b) wait_vmcnt<L2>();
      else wait_vmcnt<0>();
#else
      wait_vmcnt<0>();
#endif
      __builtin_amdgcn_s_barrier();          // end of iteration j: W1c(j+1), fold terms(j+1), W2c(j-1) have landed
    }
    return;
  }

  if (is_a) {
    // ======================= A waves: S(j) = X . W1c(j)^T on the matrix cores, beside it the GELU of chunk j-1 =================
    const int grow = row0 + rb * 32 + l31;
    // X fragments of this lane's row, for all of K: resident in registers (the tile is read once, never staged)
    h16x8 xf[NKS];
    {
      const h16_t* xrow = p.H + (int64_t)grow * p.ldh + half * 8;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) xf[ks] = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(xrow + ks * 16));
    }
    float ln_rstd, ln_mr;
    {
      const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (int64_t)grow * p.ln_nblk;
      constexpr int NB = C / 32;
      float2 t[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) t[j] = st[j];
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int j = 0; j < NB; ++j) { sm += t[j].x; sq += t[j].y; }
      const float inv_k = 1.0f / (float)(p.ln_nblk * 32);
      const float mean = sm * inv_k;
      ln_rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
      ln_mr = mean * ln_rstd;
    }
    unsigned char* prow = sP + (rb * 32 + l31) * 32 + half * 16;
    f32x16 s_old;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_old[r] = 0.f;
    __builtin_amdgcn_s_barrier();            // W1c(0) is in LDS
    for (int j = 0; j < n + 2; ++j) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      if (j < n) {
        // fragments of W1c(j): lane reads packed row l31, k = 16 ks + 8 half; five k-steps' reads in flight ahead of their MFMAs
        const unsigned char* w1 = sW1 + (j % 3) * W1_BYTES;
        h16x8 wa[5], wb[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) wa[u] = *reinterpret_cast<const h16x8*>(w1 + (u >> 2) * (32 * 128) + frag_offset(l31, (u & 3) * 2 + half));
#pragma unroll
        for (int g = 0; g < NKS / 5; g += 2) {
          if (g + 1 < NKS / 5) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int ks = (g + 1) * 5 + u;
              wb[u] = *reinterpret_cast<const h16x8*>(w1 + (ks >> 2) * (32 * 128) + frag_offset(l31, (ks & 3) * 2 + half));
            }
          }
#pragma unroll
#if FFN_ABL != 2
          for (int u = 0; u < 5; ++u) s = mfma32x32x16(wa[u], xf[g * 5 + u], s, 0, 0, 0);
#else
          for (int u = 0; u < 5; ++u) asm volatile("" :: "v"(wa[u]));
#endif
          if (g + 2 < NKS / 5) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int ks = (g + 2) * 5 + u;
              wa[u] = *reinterpret_cast<const h16x8*>(w1 + (ks >> 2) * (32 * 128) + frag_offset(l31, (ks & 3) * 2 + half));
            }
          }
          if (g + 1 < NKS / 5) {
#pragma unroll
#if FFN_ABL != 2
            for (int u = 0; u < 5; ++u) s = mfma32x32x16(wb[u], xf[(g + 1) * 5 + u], s, 0, 0, 0);
#else
            for (int u = 0; u < 5; ++u) asm volatile("" :: "v"(wb[u]));
#endif
          }
        }
      }
      if (j >= 1 && j <= n) {
        // ---- chunk j-1: LayerNorm fold + bias, value * gelu(gate) (quads 0, 1: values of features 8q + 4 half + i; quads 2, 3: gates)
        const float* cbp = reinterpret_cast<const float*>(sCB + ((j - 1) & 3) * CB_SLOT);
        unsigned wq[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 cv = *reinterpret_cast<const float4*>(cbp + 8 * q + hsel), cg = *reinterpret_cast<const float4*>(cbp + 8 * (q + 2) + hsel);
          const float4 bv = *reinterpret_cast<const float4*>(cbp + 32 + 8 * q + hsel), bg = *reinterpret_cast<const float4*>(cbp + 32 + 8 * (q + 2) + hsel);
          const float c4[4] = {cv.x, cv.y, cv.z, cv.w}, g4[4] = {cg.x, cg.y, cg.z, cg.w};
          const float b4[4] = {bv.x, bv.y, bv.z, bv.w}, h4[4] = {bg.x, bg.y, bg.z, bg.w};
          float g[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float val = fmaf(s_old[4 * q + i], ln_rstd, -ln_mr * c4[i]) + b4[i];
            const float gate = fmaf(s_old[4 * (q + 2) + i], ln_rstd, -ln_mr * g4[i]) + h4[i];
#if FFN_ABL != 1
            g[i] = val * gelu_erf_f(gate);
#else
            g[i] = val * gate;
#endif
          }
          wq[q][0] = pack2h(g[0], g[1]);
          wq[q][1] = pack2h(g[2], g[3]);
        }
        // the two lanes of a row hold features {4h..4h+3} u {8+4h..8+4h+3}: trade so that lane h holds 8h..8h+7 (MFMA operand order)
        unsigned o[4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto e = __builtin_amdgcn_permlane32_swap(wq[0][d], wq[1][d], false, false);
          o[d] = e[0];
          o[2 + d] = e[1];
        }
        *reinterpret_cast<uint4*>(prow + ((j - 1) & 1) * P_BYTES) = make_uint4(o[0], o[1], o[2], o[3]);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): P(j-1) is in LDS
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s_old[r] = s[r];
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ========================= B waves: acc += P(j-2) . W2c(j-2)^T ============================================================
  const int a_row = rb * 32 + l31;
  f32x16 acc[FNB][1];
#pragma unroll
  for (int a = 0; a < FNB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
  __builtin_amdgcn_s_barrier();              // (prologue barrier)
  for (int j = 0; j < n + 2; ++j) {
    if (j >= 2) {
      const unsigned char* w2 = sW2 + ((j - 2) % 3) * W2_BYTES;
      const h16x8 pf = *reinterpret_cast<const h16x8*>(sP + ((j - 2) & 1) * P_BYTES + a_row * 32 + half * 16);
      h16x8 w2f[FNB];
#pragma unroll
      for (int a = 0; a < FNB; ++a) w2f[a] = *reinterpret_cast<const h16x8*>(w2 + (a * 32 + l31) * 32 + half * 16);
#pragma unroll
#if FFN_ABL != 3
      for (int a = 0; a < FNB; ++a) acc[a][0] = mfma32x32x16(w2f[a], pf, acc[a][0], 0, 0, 0);
#else
      for (int a = 0; a < FNB; ++a) asm volatile("" :: "v"(w2f[a]), "v"(pf));
#endif
    }
    __builtin_amdgcn_s_barrier();
  }
  const float no_pre[2] = {0.f, 0.f};
  epilogue<FNB, 1>(p.epi, acc, row0 + rb * 32, 0, lane, 0, no_pre, false);
}

template <int C>
int launch_ffn(const FfnArgs& a, hipStream_t s) {
  constexpr int NKT = C / 64;
  constexpr size_t lds = 3 * ((size_t)NKT * 32 * 128) + 3 * ((size_t)C * 32) + 2 * 96 * 32 + 4 * 512;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      avsd_set_error("ffn_block: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
      return AVSD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((ffn_kernel<C>), dim3((unsigned)(a.epi.M / 96)), dim3(512), lds, s, a);
  AVSD_CHECK_LAUNCH("ffn_block launch");
  return AVSD_OK;
}

}  // namespace

extern "C" int avsd_ffn_block_supported(int C, int nh) { return (C == 320 && nh % 16 == 0 && nh > 0) ? 1 : 0; }

extern "C" int avsd_sizeof_ffn_desc(void) { return (int)sizeof(avsd_ffn_desc); }

extern "C" int avsd_ffn_block(const avsd_ffn_desc* dp, void* stream) {
  AVSD_REQUIRE(dp != nullptr, "ffn_block: null descriptor");
  const avsd_ffn_desc& d = *dp;
  AVSD_REQUIRE(d.h && d.res && d.ln_stats && d.w1 && d.cb1 && d.w2c && d.bias2 && d.out, "ffn_block: null pointer");
  AVSD_REQUIRE(avsd_ffn_block_supported(d.C, d.nh), "ffn_block: unsupported shape C=%d nh=%d (built: C = 320, nh %% 16 == 0)", d.C, d.nh);
  AVSD_REQUIRE(d.M > 0 && d.M % 96 == 0, "ffn_block: M (%d) must be a multiple of 96", d.M);
  AVSD_REQUIRE(d.ldh % 8 == 0 && d.ldw1 % 8 == 0 && d.ldo % 8 == 0 && d.ldh >= d.C && d.ldw1 >= d.C && d.ldo >= d.C,
               "ffn_block: row strides must be multiples of 8 and >= C");
  AVSD_REQUIRE((double)d.M * d.ldh * 2.0 < 2147483648.0 && (double)d.nh * 2 * d.ldw1 * 2.0 < 2147483648.0, "ffn_block: operands must be < 2 GiB");
  AVSD_REQUIRE(d.res_f32 ? (d.ldres % 4 == 0) : (d.ldres % 8 == 0), "ffn_block: bad residual stride");
  FfnArgs a;
  a.H = (const h16_t*)d.h; a.ldh = d.ldh;
  a.W1 = (const h16_t*)d.w1; a.ldw1 = d.ldw1;
  a.cb1 = d.cb1;
  a.ln_stats = d.ln_stats; a.ln_nblk = d.C / 32; a.ln_eps = d.ln_eps;
  a.W2c = (const h16_t*)d.w2c; a.nh = d.nh;
  avsd_gemm_desc e = {};
  e.out = d.out; e.ldc = d.ldo; e.bias = d.bias2; e.res1 = d.res; e.ldr1 = d.ldres;
  e.M = d.M; e.N = d.C; e.K = d.nh; e.alpha = 1.0f; e.batch = 1;
  e.flags = (d.res_f32 ? AVSD_GEMM_RES1_F32 : 0) | (d.rowstats ? AVSD_GEMM_ROWSTATS : 0);
  e.rowstats = d.rowstats; e.out_master = d.out_master; e.ldm = d.ldm;
  if (d.out_master) AVSD_REQUIRE(d.ldm % 4 == 0 && d.ldm >= d.C, "ffn_block: bad ldm");
  a.epi = e;
  return launch_ffn<320>(a, reinterpret_cast<hipStream_t>(stream));
}
