// Batched MFCC for gfx950: framing (edge reflection) → DC removal → pre-emphasis → Povey window → 512-point real FFT
// (256-point complex Stockham radix-4 in LDS, one wavefront per frame) → power spectrum → 23 mel bins → log → DCT-II →
// lifter.  Replaces kalpy MfccComputer.compute_mfccs (MFA/corpus/features.py:235; Kaldi feat/feature-mfcc.cc,
// SURVEY Appendix A.1).  HBM-bound stage: reads 2 B/sample, writes 52 B/frame; everything else lives in LDS/registers.
//
// Layout: grid (frame tiles, utterances); block = 4 wavefronts; each wavefront owns FRAMES_PER_WAVE consecutive frames
// and a private 6 KiB LDS slice (frame staging + two complex ping-pong buffers).  Compiled with -ffp-contract=off: the
// only fused multiply-adds are the explicit fmaf()s, matching the oracle's arithmetic.
#include <cmath>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int kNfft = 512;
constexpr int kHalf = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerWave = 8;
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;
constexpr int kMaxBins = 32;
constexpr int kMaxCeps = 32;

struct MfccParams {
  int win, shift, nbins, nceps, snip_edges, remove_dc;
  float preemph;
  const float *window;      // [win]
  const float *tw256;       // [256][2]: cos(2*pi*m/256), -sin(2*pi*m/256)
  const float *tw512;       // [256][2]: cos(2*pi*k/512), -sin(2*pi*k/512)
  const float *melw;        // concatenated triangle weights
  const int32_t *melidx;    // [nbins][3]: first fft bin, length, offset into melw
  const float *dct;         // [nceps][nbins]
  const float *lifter;      // [nceps]
};

// Every wavefront works in its own LDS slice, so no workgroup barrier is needed: LDS operations of one wavefront execute
// in program order; the fences only stop the compiler from moving accesses across the hand-over points.
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(256) void mfcc_kernel(MfccParams p, const int16_t *__restrict__ pcm,
                                                   const int64_t *__restrict__ sample_off,
                                                   const int64_t *__restrict__ frame_off, float *__restrict__ out) {
  __shared__ float lds[kWavesPerBlock][kNfft * 3];
  const int utt = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s0 = sample_off[utt], n = sample_off[utt + 1] - s0;
  const int64_t f0 = frame_off[utt];
  const int T = (int)(frame_off[utt + 1] - f0);
  if ((int)blockIdx.x * kFramesPerBlock + wave * kFramesPerWave >= T) return;  // wavefronts are independent
  float *frame = lds[wave];             // [512] staging / power spectrum
  float2 *bufA = (float2 *)(lds[wave] + kNfft);      // [256]
  float2 *bufB = (float2 *)(lds[wave] + 2 * kNfft);  // [256]
  const int16_t *x = pcm + s0;

  for (int it = 0; it < kFramesPerWave; it++) {
    const int f = blockIdx.x * kFramesPerBlock + wave * kFramesPerWave + it;
    const bool valid = f < T;
    // ---- ExtractWindow (reflection at the edges) + DC removal
    const int64_t start = p.snip_edges ? (int64_t)f * p.shift : (int64_t)p.shift * f + p.shift / 2 - p.win / 2;
    float v[8];
    float sum = 0.0f;  // int16-valued samples: the sum is an exact integer < 2^24 in any order
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int s = lane + 64 * j;
      float val = 0.0f;
      if (valid && s < p.win) {
        int64_t si = start + s;
        while (si < 0 || si >= n) si = (si < 0) ? (-si - 1) : (2 * n - 1 - si);
        val = (float)x[si];
      }
      v[j] = val;
      sum += val;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float off = p.remove_dc ? (-sum / (float)p.win) : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int s = lane + 64 * j;
      frame[s] = (s < p.win) ? v[j] + off : 0.0f;
    }
    WAVE_SYNC();
    // ---- pre-emphasis + window, packed as complex z[m] = x[2m] + i x[2m+1]
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int s = lane + 64 * j;
      float val = 0.0f;
      if (s < p.win) {
        float cur = frame[s], prev = frame[s > 0 ? s - 1 : 0];
        val = (cur - p.preemph * prev) * p.window[s];
      }
      v[j] = val;
    }
    WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < 8; j++) ((float *)bufA)[lane + 64 * j] = v[j];
    WAVE_SYNC();
    // ---- 256-point complex FFT, Stockham radix-4, 4 stages, one butterfly per lane per stage
    float2 *src = bufA, *dst = bufB;
#pragma unroll
    for (int stage = 0; stage < 4; stage++) {
      const int Ns = 1 << (2 * stage);
      const int k = lane & (Ns - 1);
      float2 a[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        a[r] = src[lane + 64 * r];
        if (stage > 0 && r > 0) {
          int m = k * r * (64 / Ns);  // angle = -2*pi*k*r/(4*Ns) in units of 2*pi/256
          float2 w = make_float2(p.tw256[2 * m], p.tw256[2 * m + 1]);
          a[r] = cmul(a[r], w);
        }
      }
      float2 s02 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), d02 = make_float2(a[0].x - a[2].x, a[0].y - a[2].y);
      float2 s13 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y);
      float2 d13 = make_float2(a[1].y - a[3].y, -(a[1].x - a[3].x));  // (a1 - a3) * (-i)
      const int idxD = ((lane >> (2 * stage)) << (2 * stage + 2)) + k;
      dst[idxD] = make_float2(s02.x + s13.x, s02.y + s13.y);
      dst[idxD + Ns] = make_float2(d02.x + d13.x, d02.y + d13.y);
      dst[idxD + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
      dst[idxD + 3 * Ns] = make_float2(d02.x - d13.x, d02.y - d13.y);
      WAVE_SYNC();
      float2 *t = src; src = dst; dst = t;
    }
    // src now holds Z[0..255] in natural order.  Real-FFT post-processing → power spectrum P[0..256] into frame[].
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int k = lane + 64 * j;
      float2 zk = src[k], zn = src[(kHalf - k) & (kHalf - 1)];
      // E = (Z[k] + conj(Z[N-k]))/2, O = (Z[k] - conj(Z[N-k]))/(2i); X[k] = E + w^k O
      float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
      float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
      float2 w = make_float2(p.tw512[2 * k], p.tw512[2 * k + 1]);
      float2 wo = cmul(w, o);
      float re = e.x + wo.x, im = e.y + wo.y;
      frame[k] = (k == 0) ? re * re : re * re + im * im;  // Kaldi ComputePowerSpectrum: bin 0 = DC^2
      if (k == 0) { float ny = zk.x - zk.y; frame[kHalf] = ny * ny; }
    }
    WAVE_SYNC();
    // ---- mel filterbank: lane = (bin, half); each half sums a contiguous part of the triangle in ascending order
    float *mel = (float *)dst;  // reuse the idle ping-pong buffer: mel[0..nbins)
    {
      int bin = lane >> 1, half = lane & 1;
      float acc = 0.0f;
      if (bin < p.nbins) {
        int first = p.melidx[3 * bin], len = p.melidx[3 * bin + 1], woff = p.melidx[3 * bin + 2];
        int mid = len >> 1;
        int a0 = half ? mid : 0, a1 = half ? len : mid;
        for (int i = a0; i < a1; i++) acc = fmaf(p.melw[woff + i], frame[first + i], acc);
      }
      float other = __shfl_xor(acc, 1);
      if (bin < p.nbins && half == 0) {
        float e = acc + other;
        e = fmaxf(e, 1.1920928955078125e-07f);  // floor at FLT_EPSILON
        mel[bin] = logf(e);
      }
    }
    WAVE_SYNC();
    // ---- DCT-II rows 0..nceps-1 + lifter
    if (valid && lane < p.nceps) {
      float acc = 0.0f;
      for (int b = 0; b < p.nbins; b++) acc = fmaf(p.dct[lane * p.nbins + b], mel[b], acc);
      out[(f0 + f) * p.nceps + lane] = acc * p.lifter[lane];
    }
    WAVE_SYNC();
  }
}

float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

}  // namespace

extern "C" {

MFA_API int mfa_mfcc_configure(mfa_ctx *c, const mfa_mfcc_opts *o) {
  hipSetDevice(c->device);
  int win = (int)(o->sample_frequency * 0.001f * o->frame_length_ms);
  int shift = (int)(o->sample_frequency * 0.001f * o->frame_shift_ms);
  int nfft = 1;
  while (nfft < win) nfft <<= 1;
  if (nfft != kNfft) return c->fail("MFCC kernel supports a 512-point FFT (window %d samples → %d)", win, nfft);
  if (o->use_energy) return c->fail("use_energy=true is not supported by the MFCC kernel (MFA default is false)");
  if (o->num_mel_bins > kMaxBins || o->num_coefficients > kMaxCeps || o->num_coefficients > o->num_mel_bins)
    return c->fail("unsupported num_mel_bins/num_coefficients %d/%d", o->num_mel_bins, o->num_coefficients);
  int nb = o->num_mel_bins, nc = o->num_coefficients;
  // window ("povey"), Kaldi FeatureWindowFunction
  std::vector<float> window(win);
  double a = 2.0 * M_PI / (win - 1);
  for (int i = 0; i < win; i++) window[i] = (float)pow(0.5 - 0.5 * cos(a * (double)i), 0.85);
  std::vector<float> tw(2 * 256 * 2);
  for (int m = 0; m < 256; m++) {
    tw[2 * m] = (float)cos(2.0 * M_PI * m / 256.0);
    tw[2 * m + 1] = (float)(-sin(2.0 * M_PI * m / 256.0));
    tw[512 + 2 * m] = (float)cos(2.0 * M_PI * m / 512.0);
    tw[512 + 2 * m + 1] = (float)(-sin(2.0 * M_PI * m / 512.0));
  }
  // Kaldi MelBanks (float arithmetic as in mel-computations.cc)
  float nyquist = 0.5f * o->sample_frequency;
  float low = o->low_frequency, high = o->high_frequency > 0.0f ? o->high_frequency : nyquist + o->high_frequency;
  if (low < 0.0f || low >= nyquist || high <= 0.0f || high > nyquist || high <= low)
    return c->fail("bad mel frequency range [%f, %f]", low, high);
  float fft_bin_width = o->sample_frequency / nfft;
  float mel_low = mel_scale(low), mel_high = mel_scale(high);
  float mel_delta = (mel_high - mel_low) / (nb + 1);
  std::vector<float> melw;
  std::vector<int32_t> melidx(3 * nb);
  for (int bin = 0; bin < nb; bin++) {
    float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta, right = mel_low + (bin + 2) * mel_delta;
    int first = -1, last = -1;
    std::vector<float> w(nfft / 2, 0.0f);
    for (int i = 0; i < nfft / 2; i++) {
      float mel = mel_scale(fft_bin_width * i);
      if (mel > left && mel < right) {
        w[i] = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
        if (first == -1) first = i;
        last = i;
      }
    }
    if (first == -1) return c->fail("mel bin %d is empty (num_mel_bins too large)", bin);
    melidx[3 * bin] = first;
    melidx[3 * bin + 1] = last + 1 - first;
    melidx[3 * bin + 2] = (int32_t)melw.size();
    melw.insert(melw.end(), w.begin() + first, w.begin() + last + 1);
  }
  std::vector<float> dct((size_t)nc * nb), lifter(nc);
  float norm0 = std::sqrt(1.0f / (float)nb), norm = std::sqrt(2.0f / (float)nb);
  for (int k = 0; k < nc; k++)
    for (int n = 0; n < nb; n++)
      dct[(size_t)k * nb + n] = (k == 0) ? norm0 : (float)(norm * std::cos((double)M_PI / nb * (n + 0.5) * k));
  for (int i = 0; i < nc; i++)
    lifter[i] = o->cepstral_lifter != 0.0f ? (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * i / o->cepstral_lifter)) : 1.0f;

  auto upload = [&](void **dptr, const void *h, size_t bytes) -> int {
    if (*dptr) { hipFree(*dptr); *dptr = nullptr; }
    MFA_HIP_CHECK(c, hipMalloc(dptr, bytes));
    MFA_HIP_CHECK(c, hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (upload((void **)&c->d_window, window.data(), window.size() * 4)) return -1;
  if (upload((void **)&c->d_twiddle, tw.data(), tw.size() * 4)) return -1;
  if (upload((void **)&c->d_melw, melw.data(), melw.size() * 4)) return -1;
  if (upload((void **)&c->d_melidx, melidx.data(), melidx.size() * 4)) return -1;
  if (upload((void **)&c->d_dct, dct.data(), dct.size() * 4)) return -1;
  if (upload((void **)&c->d_lifter, lifter.data(), lifter.size() * 4)) return -1;
  c->mfcc = *o;
  c->win = win; c->shift = shift; c->nfft = nfft;
  c->mfcc_ready = true;
  return 0;
}

MFA_API int32_t mfa_mfcc_num_frames(mfa_ctx *c, int64_t n) {
  if (!c->mfcc_ready) return -1;
  if (c->mfcc.snip_edges) return n < c->win ? 0 : (int32_t)(1 + (n - c->win) / c->shift);
  return (int32_t)((n + c->shift / 2) / c->shift);
}

MFA_API int mfa_mfcc_batch(mfa_ctx *c, const int16_t *d_pcm, const int64_t *d_sample_off, const int64_t *d_frame_off,
                           int32_t n_utt, int32_t max_frames, float *d_mfcc) {
  if (!c->mfcc_ready) return c->fail("mfa_mfcc_configure has not been called");
  if (n_utt <= 0 || max_frames <= 0) return 0;
  if (n_utt > 65535) return c->fail("at most 65535 utterances per MFCC launch (got %d)", n_utt);
  MfccParams p;
  p.win = c->win; p.shift = c->shift; p.nbins = c->mfcc.num_mel_bins; p.nceps = c->mfcc.num_coefficients;
  p.snip_edges = c->mfcc.snip_edges; p.remove_dc = c->mfcc.remove_dc_offset; p.preemph = c->mfcc.preemphasis;
  p.window = c->d_window; p.tw256 = c->d_twiddle; p.tw512 = c->d_twiddle + 512;
  p.melw = c->d_melw; p.melidx = c->d_melidx; p.dct = c->d_dct; p.lifter = c->d_lifter;
  dim3 grid((max_frames + kFramesPerBlock - 1) / kFramesPerBlock, n_utt);
  KernelTimer kt(c, MFA_K_MFCC);
  hipLaunchKernelGGL(mfcc_kernel, grid, dim3(256), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
