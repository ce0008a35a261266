// Batched MFCC for gfx950: framing (edge reflection) → DC removal → pre-emphasis → Povey window → 512-point real FFT
// (256-point complex Stockham radix-4 in LDS, one wavefront per frame) → power spectrum → 23 mel bins → log → DCT-II →
// lifter.  Replaces kalpy MfccComputer.compute_mfccs (MFA/corpus/features.py:235; Kaldi feat/feature-mfcc.cc,
// SURVEY Appendix A.1).  HBM-bound stage: reads 2 B/sample, writes 52 B/frame; everything else lives in LDS/registers.
//
// Layout: grid (frame tiles, utterances); block = 4 wavefronts; each wavefront owns kFramesPerWave consecutive frames and
// a private 4 KiB LDS slice (two complex ping-pong buffers, reused for the power spectrum and the mel energies).
// Everything a frame needs besides its samples is loaded ONCE per wavefront: the window and every FFT twiddle a lane uses
// sit in registers for the wavefront's lifetime, the mel triangles / DCT matrix sit in a workgroup LDS table.  The samples
// of frame i+1 are requested before frame i is processed, so no global-memory latency is exposed inside the frame loop
// (round-1 profile of the first version: 6.1 ms per 2M frames, almost all of it waiting on table loads from global
// memory inside the serial mel / DCT loops).
// Compiled with -ffp-contract=off: the only fused multiply-adds are the explicit fmaf()s.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int kNfft = 512;
constexpr int kHalf = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerWave = 16;
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;
constexpr int kMaxBins = 32;
constexpr int kMaxCeps = 16;    // four lanes per cepstral coefficient in the DCT
constexpr int kMaxMelW = 768;   // Σ triangle lengths (≈ 2·256 for any bin count)

struct MfccParams {
  int win, shift, nbins, nceps, snip_edges, remove_dc, n_melw;
  float preemph;
  const float *window;      // [win]
  const float *tw256;       // [256][2]: cos(2*pi*m/256), -sin(2*pi*m/256)
  const float *tw512;       // [256][2]: cos(2*pi*k/512), -sin(2*pi*k/512)
  const float *melw;        // concatenated triangle weights
  const int32_t *melseg;    // [64][4]: per lane {mel bin or -1, first fft bin, weight offset, taps | parts<<16 | first<<24}
  const float *dct;         // [nceps][nbins]
  const float *lifter;      // [nceps]
};

// Every wavefront works in its own LDS slice, so no workgroup barrier is needed inside the frame loop: LDS operations of
// one wavefront execute in program order; the fences only stop the compiler from moving accesses across the hand-over.
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// Σ over the wavefront (returned in every lane).  The addends here are integer-valued and the sum stays below 2^24, so
// the reduction order does not matter.
__device__ __forceinline__ float wave_sum_exact(float v) {
  v += dpp_f32<0xB1>(0.0f, v);          // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(0.0f, v);          // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(0.0f, v);         // row_half_mirror
  v += dpp_f32<0x140>(0.0f, v);         // row_mirror: every lane of a 16-lane row holds the row sum
  v += dpp_f32<0x142, 0xA>(0.0f, v);    // row_bcast15 into rows 1 and 3
  v += dpp_f32<0x143, 0xC>(0.0f, v);    // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__global__ __launch_bounds__(256, 4) void mfcc_kernel(MfccParams p, const int16_t *__restrict__ pcm,
                                                   const int64_t *__restrict__ sample_off,
                                                   const int64_t *__restrict__ frame_off, float *__restrict__ out) {
  __shared__ float lds[kWavesPerBlock][2 * kNfft];
  __shared__ float s_melw[kMaxMelW];
  __shared__ float s_dct[kMaxCeps * kMaxBins];
  const int utt = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t s0 = sample_off[utt], n = sample_off[utt + 1] - s0;
  const int64_t f0 = frame_off[utt];
  const int T = (int)(frame_off[utt + 1] - f0);
  if ((int)blockIdx.x * kFramesPerBlock >= T) return;  // whole workgroup: nothing to do
  for (int i = threadIdx.x; i < p.n_melw; i += 256) s_melw[i] = p.melw[i];
  for (int i = threadIdx.x; i < p.nceps * p.nbins; i += 256) s_dct[i] = p.dct[i];
  __syncthreads();
  const int fbase = blockIdx.x * kFramesPerBlock + wave * kFramesPerWave;
  const int nfr = min(kFramesPerWave, T - fbase);
  if (nfr <= 0) return;  // wavefronts are independent from here on
  float2 *bufA = (float2 *)lds[wave];              // [256]
  float2 *bufB = (float2 *)(lds[wave] + kNfft);    // [256]
  const int16_t *x = pcm + s0;

  // ---- per-lane constants
  float wnd[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { const int s = lane + 64 * j; wnd[j] = s < p.win ? p.window[s] : 0.0f; }
  float2 tw[3][3];  // stage 1..3, r = 1..3
#pragma unroll
  for (int stage = 1; stage < 4; stage++) {
    const int Ns = 1 << (2 * stage);
    const int k = lane & (Ns - 1);
#pragma unroll
    for (int r = 1; r < 4; r++) {
      const int m = k * r * (64 / Ns);  // angle = -2*pi*k*r/(4*Ns) in units of 2*pi/256
      tw[stage - 1][r - 1] = make_float2(p.tw256[2 * m], p.tw256[2 * m + 1]);
    }
  }
  float2 w512[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { const int k = lane + 64 * j; w512[j] = make_float2(p.tw512[2 * k], p.tw512[2 * k + 1]); }
  const int seg_bin = p.melseg[4 * lane], seg_first = p.melseg[4 * lane + 1], seg_woff = p.melseg[4 * lane + 2];
  const int seg_info = p.melseg[4 * lane + 3];
  const int seg_taps = seg_info & 0xFFFF, seg_parts = (seg_info >> 16) & 0xFF, seg_is_first = seg_info >> 24;
  const float lift_k = (lane >> 2) < p.nceps ? p.lifter[lane >> 2] : 0.0f;   // lane quad k owns cepstral coefficient k

  auto load_frame = [&](int f, float (&v)[8]) {
    const int64_t start = p.snip_edges ? (int64_t)f * p.shift : (int64_t)p.shift * f + p.shift / 2 - p.win / 2;
    if (start >= 0 && start + p.win <= n) {   // interior frame (all but the first and last one or two): no reflection
      const int16_t *xs = x + start;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int s = lane + 64 * j;
        v[j] = s < p.win ? (float)xs[s] : 0.0f;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int s = lane + 64 * j;
      float val = 0.0f;
      if (s < p.win) {
        int64_t si = start + s;
        while (si < 0 || si >= n) si = (si < 0) ? (-si - 1) : (2 * n - 1 - si);  // reflection at the edges
        val = (float)x[si];
      }
      v[j] = val;
    }
  };

  float nxt[8];
  load_frame(fbase, nxt);
  for (int it = 0; it < nfr; it++) {
    const int f = fbase + it;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = nxt[j];
    if (it + 1 < nfr) load_frame(f + 1, nxt);  // in flight while this frame is processed
    // ---- DC removal (int16-valued samples: the sum is an exact integer < 2^24 in any order)
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) sum += v[j];
    sum = wave_sum_exact(sum);
    const float off = p.remove_dc ? (-sum / (float)p.win) : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (lane + 64 * j < p.win) ? v[j] + off : 0.0f;
    // ---- pre-emphasis + window; sample s-1 lives in the previous lane (lane 0: lane 63 of the previous register)
    float z[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float prev = dpp_f32<0x138>(0.0f, v[j]);  // wave_shr:1
      const float wrap = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j > 0 ? v[j - 1] : v[0]), j > 0 ? 63 : 0));
      if (lane == 0) prev = wrap;               // s = 0 uses itself (Kaldi: frame[0] -= preemph * frame[0])
      z[j] = (v[j] - p.preemph * prev) * wnd[j];
    }
    // packed as complex z[m] = x[2m] + i x[2m+1]
#pragma unroll
    for (int j = 0; j < 8; j++) ((float *)bufA)[lane + 64 * j] = z[j];
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
        if (stage > 0 && r > 0) a[r] = cmul(a[r], tw[stage > 0 ? stage - 1 : 0][r > 0 ? r - 1 : 0]);
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
    // src (= bufA) holds Z[0..255] in natural order.  Real-FFT post-processing → power spectrum P[0..255] into bufB
    // (Kaldi's mel banks never touch the Nyquist bin).
    float *ps = (float *)bufB;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = lane + 64 * j;
      float2 zk = src[k], zn = src[(kHalf - k) & (kHalf - 1)];
      // E = (Z[k] + conj(Z[N-k]))/2, O = (Z[k] - conj(Z[N-k]))/(2i); X[k] = E + w^k O
      float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
      float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
      float2 wo = cmul(w512[j], o);
      float re = e.x + wo.x, im = e.y + wo.y;
      ps[k] = (k == 0) ? re * re : re * re + im * im;  // Kaldi ComputePowerSpectrum: bin 0 = DC^2
    }
    WAVE_SYNC();
    // ---- mel filterbank: every lane sums one contiguous piece of one triangle (ascending FFT bin); the first lane of a
    // triangle then adds the pieces in ascending order.  The split is fixed by the host table, so results are
    // reproducible run to run.
    float *mel = ps + kHalf;  // mel[0..nbins)
    {
      float acc = 0.0f;
      const float *wv = s_melw + seg_woff;
      const float *pv = ps + seg_first;
#pragma unroll 4
      for (int i = 0; i < seg_taps; i++) acc = fmaf(wv[i], pv[i], acc);
      float e = acc;
#pragma unroll
      for (int q = 1; q < 4; q++) {
        const float other = __shfl_down(acc, q);
        if (q < seg_parts) e += other;
      }
      // floor at FLT_EPSILON, then log (hardware log2: ≲1e-6 absolute on values of 10–25, far inside the 2e-3 the FFT leaves)
      if (seg_is_first) mel[seg_bin] = __builtin_amdgcn_logf(fmaxf(e, 1.1920928955078125e-07f)) * 0.693147180559945309f;
    }
    WAVE_SYNC();
    // ---- DCT-II rows 0..nceps-1 + lifter
    {
      // four lanes per coefficient, each a quarter of the mel bins (ascending), added pairwise inside the quad
      const int k = lane >> 2, part = lane & 3;
      const int per = (p.nbins + 3) >> 2;
      const int b0 = part * per, b1 = min(p.nbins, b0 + per);
      float acc = 0.0f;
      if (k < p.nceps) {
        const float *d = s_dct + k * p.nbins;
        for (int b = b0; b < b1; b++) acc = fmaf(d[b], mel[b], acc);
      }
      acc += dpp_f32<0xB1>(0.0f, acc);   // quad_perm [1,0,3,2]
      acc += dpp_f32<0x4E>(0.0f, acc);   // quad_perm [2,3,0,1]
      if (part == 0 && k < p.nceps) out[(f0 + f) * p.nceps + k] = acc * lift_k;
    }
    WAVE_SYNC();
  }
}

float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

}  // namespace

extern "C" {

MFA_API int mfa_mfcc_configure(mfa_ctx *c, const mfa_mfcc_opts *o) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
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
  std::vector<int32_t> first_of(nb), len_of(nb), woff_of(nb);
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
    first_of[bin] = first;
    len_of[bin] = last + 1 - first;
    woff_of[bin] = (int32_t)melw.size();
    melw.insert(melw.end(), w.begin() + first, w.begin() + last + 1);
  }
  if ((int)melw.size() > kMaxMelW) return c->fail("mel filterbank has %zu taps (kernel table holds %d)", melw.size(), kMaxMelW);
  // Lane plan for the filterbank: a triangle of `len` taps is cut into 1..4 contiguous pieces (one lane each) so that no
  // lane walks more than ≈ target taps; pieces of one triangle sit on consecutive lanes.
  std::vector<int32_t> melseg(64 * 4, 0);
  {
    int target = 8;
    std::vector<int> parts(nb);
    for (;; target++) {
      int total = 0;
      for (int b = 0; b < nb; b++) { parts[b] = std::min(4, (len_of[b] + target - 1) / target); total += parts[b]; }
      if (total <= 64) break;
      if (target > 512) return c->fail("cannot lay %d mel bins out on one wavefront", nb);
    }
    int lane = 0;
    for (int b = 0; b < nb; b++) {
      int per = (len_of[b] + parts[b] - 1) / parts[b], done = 0;
      for (int q = 0; q < parts[b]; q++, lane++) {
        int taps = std::min(per, len_of[b] - done);
        if (taps < 0) taps = 0;
        melseg[4 * lane] = b;
        melseg[4 * lane + 1] = first_of[b] + done;
        melseg[4 * lane + 2] = woff_of[b] + done;
        melseg[4 * lane + 3] = taps | (parts[b] << 16) | ((q == 0 ? 1 : 0) << 24);
        done += taps;
      }
    }
    for (; lane < 64; lane++) { melseg[4 * lane] = 0; melseg[4 * lane + 3] = 0; }  // idle lanes: no taps, not first
  }
  std::vector<float> dct((size_t)nc * nb), lifter(nc);
  float norm0 = std::sqrt(1.0f / (float)nb), norm = std::sqrt(2.0f / (float)nb);
  for (int k = 0; k < nc; k++)
    for (int n = 0; n < nb; n++)
      dct[(size_t)k * nb + n] = (k == 0) ? norm0 : (float)(norm * std::cos((double)M_PI / nb * (n + 0.5) * k));
  for (int i = 0; i < nc; i++)
    lifter[i] = o->cepstral_lifter != 0.0f ? (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * i / o->cepstral_lifter)) : 1.0f;

  auto upload = [&](void **dptr, const void *h, size_t bytes) -> int {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    MFA_HIP_CHECK(c, hipMalloc(dptr, bytes));
    MFA_HIP_CHECK(c, hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (upload((void **)&c->d_window, window.data(), window.size() * 4)) return -1;
  if (upload((void **)&c->d_twiddle, tw.data(), tw.size() * 4)) return -1;
  if (upload((void **)&c->d_melw, melw.data(), melw.size() * 4)) return -1;
  if (upload((void **)&c->d_melidx, melseg.data(), melseg.size() * 4)) return -1;
  c->n_melw = (int)melw.size();
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
  p.melw = c->d_melw; p.melseg = c->d_melidx; p.n_melw = c->n_melw; p.dct = c->d_dct; p.lifter = c->d_lifter;
  dim3 grid((max_frames + kFramesPerBlock - 1) / kFramesPerBlock, n_utt);
  KernelTimer kt(c, MFA_K_MFCC);
  hipLaunchKernelGGL(mfcc_kernel, grid, dim3(256), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
