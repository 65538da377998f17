// Audio conditioning front-end kernels: everything between a 16 kHz waveform and the token
// matrix the ImageBind audio ViT consumes.  They run once per clip; none is on the denoising loop.
//
// fbank_kernel: Kaldi-compatible log-mel filterbank, the algorithm torchaudio.compliance.kaldi.fbank runs for
//   ImageBind's waveform2melspec(htk_compat, hanning window, 25 ms / 10 ms frames, 128 bins, dither 0) followed
//   by the (mel, time) transpose, zero padding / cropping to `t_out` frames and the (x - mean) / std normalisation
//   of avgen/data/utils.py:26-55 (waveform_to_melspectrogram).  One workgroup per output frame: DC removal,
//   pre-emphasis, window, a direct 512-point DFT in f32 (400 x 257 complex MACs - 40 MFLOP per clip), power
//   spectrum, triangular mel filters, log.
// patchify_kernel: the non-overlapping-free 16x16 stride-10 patch gather in front of ImageBind's audio stem
//   (Conv2d(1, 768, 16, stride 10, bias=False) == im2col + GEMM), f32 image -> bf16 rows [B*ph*pw][C*kh*kw].
// tokens_kernel: token matrix = [cls | patch embeddings] + learned position table, in sequences padded by
//   `tail` extra rows (the add_bias_kv key/value slot of torch.nn.MultiheadAttention lives in the first one).
#include "avsd_common.h"

namespace {

constexpr int FB_THREADS = 256;

struct FbankArgs {
  const float* wave; int64_t wave_stride; int n_samples;
  const float* window; const float* mel_fb;
  int win, shift, nfft, n_mel, n_frames;
  float preemph; int remove_dc;
  float* out; int t_out; float mean, inv_std;
};

__global__ __launch_bounds__(FB_THREADS) void fbank_kernel(const FbankArgs p) {
  extern __shared__ float fb_smem[];
  const int nbin = p.nfft / 2 + 1;
  float* x = fb_smem;                 // [nfft]   framed samples
  float* twc = x + p.nfft;            // [nfft]   cos(2 pi j / nfft)
  float* tws = twc + p.nfft;          // [nfft]   sin(2 pi j / nfft)
  float* pw = tws + p.nfft;           // [nbin]   power spectrum
  __shared__ float red[FB_THREADS / 64];

  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* out = p.out + ((int64_t)b * p.n_mel) * p.t_out + t;
  if (t >= p.n_frames) {              // zero padding of the (un-normalised) log-mel matrix
    for (int m = tid; m < p.n_mel; m += FB_THREADS) out[(int64_t)m * p.t_out] = (0.f - p.mean) * p.inv_std;
    return;
  }
  const float* w = p.wave + (int64_t)b * p.wave_stride + (int64_t)t * p.shift;
  float part = 0.f;
  for (int i = tid; i < p.nfft; i += FB_THREADS) {
    const float v = i < p.win ? w[i] : 0.f;
    x[i] = v;
    part += v;
    float s, c;
    sincospif(2.0f * (float)i / (float)p.nfft, &s, &c);
    twc[i] = c;
    tws[i] = s;
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  float mean = 0.f;
  if (p.remove_dc) {
    for (int i = 0; i < FB_THREADS / 64; ++i) mean += red[i];
    mean /= (float)p.win;
  }
  // pre-emphasis on the DC-free frame (Kaldi: x[i] -= c * x[i-1] from the back, x[0] -= c * x[0]), then window
  float y[2];
  int ny = 0;
  for (int i = tid; i < p.win; i += FB_THREADS) {
    const float cur = x[i] - mean;
    const float prev = x[i > 0 ? i - 1 : 0] - mean;
    y[ny++] = (cur - p.preemph * prev) * p.window[i];
  }
  __syncthreads();
  ny = 0;
  for (int i = tid; i < p.win; i += FB_THREADS) x[i] = y[ny++];
  __syncthreads();

  // direct DFT: bin k = sum_n x[n] * exp(-2 pi i k n / nfft)
  const int mask = p.nfft - 1;        // nfft is a power of two (host-checked)
  for (int k = tid; k < nbin; k += FB_THREADS) {
    float re = 0.f, im = 0.f;
    int j = 0;
    for (int n = 0; n < p.win; ++n) {
      const float v = x[n];
      re = fmaf(v, twc[j], re);
      im = fmaf(-v, tws[j], im);
      j = (j + k) & mask;
    }
    pw[k] = re * re + im * im;
  }
  __syncthreads();
  for (int m = tid; m < p.n_mel; m += FB_THREADS) {
    const float* f = p.mel_fb + (int64_t)m * nbin;
    float e = 0.f;
    for (int k = 0; k < nbin; ++k) e = fmaf(f[k], pw[k], e);
    e = logf(fmaxf(e, 1.1920928955078125e-07f));      // torch.finfo(float32).eps floor, as torchaudio
    out[(int64_t)m * p.t_out] = (e - p.mean) * p.inv_std;
  }
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ src, h16_t* __restrict__ dst, int B, int C,
                                                       int H, int W, int kh, int kw, int stride, int ph, int pw_) {
  const int kk = C * kh * kw;
  const int64_t total = (int64_t)B * ph * pw_ * kk;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % kk);
    const int64_t r = i / kk;
    const int px = (int)(r % pw_);
    const int py = (int)((r / pw_) % ph);
    const int b = (int)(r / ((int64_t)pw_ * ph));
    const int dx = e % kw, dy = (e / kw) % kh, c = e / (kw * kh);
    dst[i] = f2h(src[(((int64_t)b * C + c) * H + py * stride + dy) * W + px * stride + dx]);
  }
}

__global__ __launch_bounds__(256) void tokens_kernel(const h16_t* __restrict__ patches, const float* __restrict__ cls,
                                                     const float* __restrict__ pos, h16_t* __restrict__ out, int B, int np,
                                                     int C, int tail) {
  const int L = 1 + np + tail;
  const int64_t total = (int64_t)B * L * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int l = (int)((i / C) % L);
    const int b = (int)(i / ((int64_t)C * L));
    float v = 0.f;
    if (l == 0) v = cls[c] + pos[c];
    else if (l <= np) v = h2f(patches[((int64_t)b * np + (l - 1)) * C + c]) + pos[(int64_t)l * C + c];
    out[i] = f2h(v);
  }
}

}  // namespace

extern "C" int avsd_kaldi_fbank(const float* wave, int batch, int n_samples, int64_t wave_stride, const float* window,
                                const float* mel_fb, int win, int shift, int nfft, int n_mel, float preemph, int remove_dc,
                                float* out, int t_out, float mean, float std, void* stream) {
  AVSD_REQUIRE(wave && window && mel_fb && out, "fbank: null pointer");
  AVSD_REQUIRE(batch > 0 && win > 0 && shift > 0 && n_mel > 0 && t_out > 0, "fbank: bad sizes");
  AVSD_REQUIRE(nfft >= win && (nfft & (nfft - 1)) == 0 && nfft <= 4096, "fbank: nfft (%d) must be a power of two >= win (%d)", nfft, win);
  AVSD_REQUIRE(win <= 2 * FB_THREADS, "fbank: window of %d samples exceeds %d", win, 2 * FB_THREADS);
  AVSD_REQUIRE(std > 0.f, "fbank: std must be positive");
  FbankArgs a;
  a.wave = wave; a.wave_stride = wave_stride; a.n_samples = n_samples; a.window = window; a.mel_fb = mel_fb;
  a.win = win; a.shift = shift; a.nfft = nfft; a.n_mel = n_mel;
  a.n_frames = n_samples < win ? 0 : 1 + (n_samples - win) / shift;      // snip_edges = True
  a.preemph = preemph; a.remove_dc = remove_dc; a.out = out; a.t_out = t_out; a.mean = mean; a.inv_std = 1.0f / std;
  const size_t lds = (size_t)(3 * nfft + nfft / 2 + 1) * sizeof(float);
  hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)t_out, (unsigned)batch), dim3(FB_THREADS), lds,
                     reinterpret_cast<hipStream_t>(stream), a);
  AVSD_CHECK_LAUNCH("fbank launch");
  return AVSD_OK;
}

extern "C" int avsd_patchify(const float* src, void* dst, int B, int C, int H, int W, int kh, int kw, int stride, void* stream) {
  AVSD_REQUIRE(src && dst, "patchify: null pointer");
  AVSD_REQUIRE(B > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && H >= kh && W >= kw, "patchify: bad sizes");
  const int ph = (H - kh) / stride + 1, pw = (W - kw) / stride + 1;
  const int64_t total = (int64_t)B * ph * pw * C * kh * kw;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     reinterpret_cast<h16_t*>(dst), B, C, H, W, kh, kw, stride, ph, pw);
  AVSD_CHECK_LAUNCH("patchify launch");
  return AVSD_OK;
}

extern "C" int avsd_vit_tokens(const void* patches, const float* cls, const float* pos, void* out, int B, int n_patches, int C,
                               int tail_rows, void* stream) {
  AVSD_REQUIRE(patches && cls && pos && out, "vit_tokens: null pointer");
  AVSD_REQUIRE(B > 0 && n_patches > 0 && C > 0 && tail_rows >= 0, "vit_tokens: bad sizes");
  const int64_t total = (int64_t)B * (1 + n_patches + tail_rows) * C;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(tokens_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16_t*>(patches), cls, pos, reinterpret_cast<h16_t*>(out), B, n_patches, C, tail_rows);
  AVSD_CHECK_LAUNCH("vit_tokens launch");
  return AVSD_OK;
}
