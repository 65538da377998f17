// One-launch GroupNorm (+ SiLU, + channel concat) for SMALL normalisation batches (gfx950).
//
// The stats + apply pair of norm.hip costs two kernel boundaries and a second read of the input.  At the bottom of the
// UNet the batches are tiny: a ResBlock norm at 4x4 pools 192 rows (12 frames x 16 pixels) of 40-80 channels per group, a
// Transformer3D norm pools the <= 256 pixels of ONE frame.  Here a workgroup owns whole groups —
// `gpw` consecutive groups of one normalisation batch, chosen so that its channel span is a multiple of 8 (one 16-byte
// vector never crosses the span) — and keeps its slab (rows x span) in registers between the statistics and the apply:
//     load (all vectors of a thread in flight at once)  ->  per-thread (sum, sum of squares) of the <= 2 groups a vector
//     touches  ->  LDS  ->  one wave per group folds the threads in a fixed order, double accumulation  ->  mean, rstd
//     ->  y = act(x * rstd * gamma + (beta - mean * rstd * gamma)) from the registers.
// No atomics, no cross-workgroup hand-off: results do not depend on scheduling.  The thread -> (row, vector) map keeps a
// thread on ONE vector column (blockDim % vectors-per-row == 0), so gamma / beta and the group ids are loaded once.
//
// Where it pays (tools/gn_bench.py, profiles/r3_gn_probe.txt): 5.1-7.4 us against 9.5-12.2 us for the pair up to ~1900
// vectors per workgroup.  Above that it loses — a ResBlock norm has only 2 x 32 / gpw = 16-64 workgroups, and the
// per-element work (unpack, fma, SiLU's exp + rcp, pack: ~25 VALU ops) then runs on 16-64 CUs instead of 256: 31 us vs 13 us
// at 3072 rows x 640 channels with the whole slab in registers (16 vectors per thread, measured before the limit below
// was set).  The geometry function therefore admits only <= 4 vectors per thread on <= 480 threads.
//
// Reference: torch.nn.GroupNorm at ff_spatio_temp_resnet_3d.py:130,146 (pooled over frames, H, W),
// ff_spatio_audio_temp_transformer_3d.py:62 (per frame, eps 1e-6), audio_cond_unet_3d_condition.py:445; biased variance,
// eps inside the sqrt; SiLU at ff_spatio_temp_resnet_3d.py:165,175.
#include "avsd_common.h"

namespace {

template <int NV, bool X2, int TMAX>
__global__ __launch_bounds__(TMAX) void gn_fused_kernel(const h16_t* x1, int ld1, int c1, int64_t lo1, const h16_t* x2, int ld2, int c2,
                                                      int64_t lo2, int rows_per_batch, int cg, int gpw, float eps,
                                                      const float* gamma, const float* beta, int act, h16_t* y, int ldy, int64_t loy) {
  __shared__ float4 red[TMAX];         // per thread: (sum, sumsq) of its first group, (sum, sumsq) of its second
  __shared__ float smean[8], srstd[8];
  const int tid = threadIdx.x, T = blockDim.x;
  const int span = gpw * cg, vpr = span / 8;
  const int cv = tid % vpr, rr = tid / vpr, rpi = T / vpr;      // this thread's vector column, first row, row stride
  const int cl = cv * 8;                                          // first channel of the vector inside the span
  const int c = blockIdx.x * span + cl;                           // ... inside the concat
  const int b = blockIdx.y;
  const int ga = cl / cg, gb = (cl + 7) / cg;                     // local groups of element 0 and element 7 (gb - ga <= 1)
  const int nb_first = min(8, (ga + 1) * cg - cl);                // elements of the vector that belong to group ga

  const h16_t* src;
  int ld;
  int64_t lo;
  if (c < c1) { src = x1 + c; ld = ld1; lo = lo1; } else { src = x2 + (c - c1); ld = ld2; lo = lo2; }
  src += (int64_t)b * rows_per_batch * ld;

  uint4 v[NV], w[X2 ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int r = rr + i * rpi;
    if (r < rows_per_batch) {
      v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld);
      if constexpr (X2) w[i] = *reinterpret_cast<const uint4*>(src + lo + (int64_t)r * ld);
    } else {
      v[i] = make_uint4(0, 0, 0, 0);
      if constexpr (X2) w[i] = make_uint4(0, 0, 0, 0);
    }
  }
  float gm[8], bt[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
    gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
    bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
  }
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {                // rows past the end hold zeros: they add nothing
    float f[8];
    unpack8(v[i], f);
    if constexpr (X2) {
      float g[8];
      unpack8(w[i], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += g[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
  }
  float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (e < nb_first) { mine.x += s[e]; mine.y += q[e]; } else { mine.z += s[e]; mine.w += q[e]; }
  }
  red[tid] = mine;
  // the slab stays PACKED across the barrier (4 registers per vector, not 8 unpacked floats): make the registers opaque so
  // that the apply phase below re-derives the floats instead of keeping the ones of the statistics phase alive
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
    if constexpr (X2) asm volatile("" : "+v"(w[i].x), "+v"(w[i].y), "+v"(w[i].z), "+v"(w[i].w));
  }
  __syncthreads();
  {
    const int wave = tid >> 6, lane = tid & 63, nfull = T >> 6;      // only full waves fold (T need not be a multiple of 64)
    if (wave < nfull) {
      for (int g = wave; g < gpw; g += nfull) {     // threads in index order, lanes strided, then a butterfly: a fixed order
        double a = 0.0, qq = 0.0;
        for (int t = lane; t < T; t += 64) {
          const int tcl = (t % vpr) * 8;
          const int tga = tcl / cg, tgb = (tcl + 7) / cg;
          const float4 u = red[t];
          if (tga == g) { a += (double)u.x; qq += (double)u.y; }
          if (tgb == g && tgb != tga) { a += (double)u.z; qq += (double)u.w; }
        }
        for (int off = 32; off > 0; off >>= 1) {
          a += __shfl_xor(a, off, 64);
          qq += __shfl_xor(qq, off, 64);
        }
        if (lane == 0) {
          const double n = (double)rows_per_batch * cg;
          const double mean = a / n;
          double var = qq / n - mean * mean;
          if (var < 0.0) var = 0.0;
          smean[g] = (float)mean;
          srstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
      }
    }
  }
  __syncthreads();
  float sc[8], sh[8];
  {
    const float ma = smean[ga], ra = srstd[ga], mb = smean[gb], rb = srstd[gb];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool first = e < nb_first;
      sc[e] = (first ? ra : rb) * gm[e];
      sh[e] = bt[e] - (first ? ma : mb) * sc[e];
    }
  }
  h16_t* dst = y + (int64_t)b * rows_per_batch * ldy + c;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int r = rr + i * rpi;
    if (r >= rows_per_batch) break;
    float f[8];
    unpack8(v[i], f);
    if constexpr (X2) {
      float g[8];
      unpack8(w[i], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += g[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = fmaf(f[e], sc[e], sh[e]);
      f[e] = act ? silu_f(t) : t;
    }
    store8<X2>(dst + (int64_t)r * ldy, loy, f);
  }
}

// Geometry of the one-launch form, or false when the batch is too large for it to pay: gpw groups per workgroup (span a multiple of 8
// channels, >= 40 channels wide when the group count allows: 80-byte row segments keep the line fetches useful),
// T threads (a multiple of the vectors per row), nv vectors per thread.
struct gn_fused_geo { int gpw, threads, nv; };

bool gn_fused_geometry(int nb, int rows_per_batch, int groups, int C, bool x2, gn_fused_geo* out) {
  if (groups <= 0 || C % groups) return false;
  const int cg = C / groups;
  if (cg < 8 && cg != 4) return false;                 // a vector may touch at most two groups
  int gpw = 1;
  while (gpw <= 8 && gpw <= groups && ((gpw * cg) % 8 != 0 || (gpw * cg < 40 && gpw * 2 <= groups && gpw * 2 <= 8))) gpw *= 2;
  if (gpw > 8 || gpw > groups || groups % gpw || (gpw * cg) % 8) return false;
  const int vpr = gpw * cg / 8;
  for (int T : {240, 480}) {
    if (T % vpr) continue;
    const int rpi = T / vpr;
    const int nv = (rows_per_batch + rpi - 1) / rpi;
    if (nv <= 4) {
      out->gpw = gpw;
      out->threads = T;
      out->nv = nv;
      return true;
    }
  }
  // Round 5: MANY batches (the per-frame norm of a Transformer3D block at 32 x 32: 24 frames x 8 group-quads = 192 workgroups) keep
  // the chip busy even with a bigger slab per workgroup — 960 threads, up to 8 vectors each.  The limit above exists for the pooled
  // ResBlock norms, whose 2 x 32 / gpw workgroups cannot (see the header).  Not in split precision (two planes per vector).
  if (!x2 && (int64_t)nb * (groups / gpw) >= 128 && 960 % vpr == 0) {
    const int rpi = 960 / vpr;
    const int nv = (rows_per_batch + rpi - 1) / rpi;
    if (nv <= 8) {
      out->gpw = gpw;
      out->threads = 960;
      out->nv = nv;
      return true;
    }
  }
  return false;
}

template <int NV, bool X2, int TMAX = 480>
void launch_gn_fused(dim3 grid, int threads, hipStream_t s, const h16_t* x1, int ld1, int c1, int64_t lo1, const h16_t* x2, int ld2, int c2,
                     int64_t lo2, int rows_per_batch, int cg, int gpw, float eps, const float* gamma, const float* beta, int act, h16_t* y,
                     int ldy, int64_t loy) {
  hipLaunchKernelGGL((gn_fused_kernel<NV, X2, TMAX>), grid, dim3((unsigned)threads), 0, s, x1, ld1, c1, lo1, x2, ld2, c2, lo2, rows_per_batch, cg, gpw,
                     eps, gamma, beta, act, y, ldy, loy);
}

}  // namespace

extern "C" int avsd_groupnorm_fused_supported(int nb, int rows_per_batch, int groups, int c1, int c2, int split) {
  if (nb <= 0 || rows_per_batch <= 0 || c1 <= 0 || c1 % 8 || c2 < 0 || c2 % 8) return 0;
  if (groups < 4 || groups > 64 || (groups & (groups - 1))) return 0;
  gn_fused_geo g;
  return gn_fused_geometry(nb, rows_per_batch, groups, c1 + c2, split != 0, &g) ? 1 : 0;
}

extern "C" int avsd_groupnorm_fused_x2(const void* x1, int ld1, int c1, int64_t lo1, const void* x2, int ld2, int c2, int64_t lo2, int nb,
                                       int rows_per_batch, int groups, const float* gamma, const float* beta, float eps, int act, void* y,
                                       int ldy, int64_t loy, void* stream) {
  AVSD_REQUIRE(x1 && c1 > 0 && c1 % 8 == 0 && ld1 % 8 == 0 && ld1 >= c1, "groupnorm_fused: bad first source (c1=%d ld1=%d)", c1, ld1);
  AVSD_REQUIRE(c2 >= 0 && c2 % 8 == 0 && (c2 == 0 || (x2 && ld2 % 8 == 0 && ld2 >= c2)), "groupnorm_fused: bad second source (c2=%d ld2=%d)", c2, ld2);
  const int C = c1 + c2;
  AVSD_REQUIRE(groups >= 4 && groups <= 64 && (groups & (groups - 1)) == 0 && C % groups == 0,
               "groupnorm_fused: groups (%d) must be a power of two in 4..64 dividing the channels (%d)", groups, C);
  AVSD_REQUIRE(nb > 0 && rows_per_batch > 0, "groupnorm_fused: bad batch geometry nb=%d rows=%d", nb, rows_per_batch);
  AVSD_REQUIRE(lo1 % 8 == 0 && lo2 % 8 == 0 && loy % 8 == 0 && (lo1 != 0) == (loy != 0) && (c2 == 0 || (lo1 != 0) == (lo2 != 0)),
               "groupnorm_fused: plane offsets must be multiples of 8; sources and output all split or none");
  AVSD_REQUIRE(y && gamma && beta && ldy % 8 == 0 && ldy >= C, "groupnorm_fused: bad output (ldy=%d)", ldy);
  const bool split = lo1 != 0;
  gn_fused_geo g;
  AVSD_REQUIRE(gn_fused_geometry(nb, rows_per_batch, groups, C, split, &g),
               "groupnorm_fused: a (%d rows x %d channels / %d groups) batch is outside the one-launch geometry (avsd_groupnorm_fused_supported)",
               rows_per_batch, C, groups);
  const dim3 grid((unsigned)(groups / g.gpw), (unsigned)nb);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const h16_t *a = (const h16_t*)x1, *bsrc = (const h16_t*)x2;
  h16_t* o = (h16_t*)y;
  const int cg = C / groups;
#define AVSD_GNF(NVV)                                                                                                                      \
  (split ? launch_gn_fused<NVV, true>(grid, g.threads, s, a, ld1, c1, lo1, bsrc, ld2, c2, lo2, rows_per_batch, cg, g.gpw, eps, gamma, beta, act, \
                                      o, ldy, loy)                                                                                       \
         : launch_gn_fused<NVV, false>(grid, g.threads, s, a, ld1, c1, lo1, bsrc, ld2, c2, lo2, rows_per_batch, cg, g.gpw, eps, gamma, beta, act, \
                                       o, ldy, loy))
  if (g.threads > 480)
    launch_gn_fused<8, false, 960>(grid, g.threads, s, a, ld1, c1, lo1, bsrc, ld2, c2, lo2, rows_per_batch, cg, g.gpw, eps, gamma, beta, act, o, ldy, loy);
  else if (g.nv <= 2) AVSD_GNF(2);
  else AVSD_GNF(4);
#undef AVSD_GNF
  AVSD_CHECK_LAUNCH("groupnorm_fused launch");
  return AVSD_OK;
}

extern "C" int avsd_groupnorm_fused(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb, int rows_per_batch, int groups,
                                    const float* gamma, const float* beta, float eps, int act, void* y, int ldy, void* stream) {
  return avsd_groupnorm_fused_x2(x1, ld1, c1, 0, x2, ld2, c2, 0, nb, rows_per_batch, groups, gamma, beta, eps, act, y, ldy, 0, stream);
}

