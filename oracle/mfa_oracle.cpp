// mfa_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A plain single-threaded restatement of the arithmetic MFA's alignment hot path
// runs inside kalpy/Kaldi (un-vendored dependency: kalpy 0.6.7 over conda-forge
// kaldi CPU build + openfst 1.8.3).  Every function cites
// the reference call site it stands behind (paths relative to /root/reference/,
// MFA/ = montreal_forced_aligner/) and names the Kaldi routine whose published
// algorithm it restates.
//
// PARITY STATUS: "parity unpinned" — the reference repo holds no golden vectors at
// the kalpy boundary and kalpy/Kaldi cannot be imported or built in this
// container.  This file is pinned only by (i) an independent
// numpy restatement (oracle/np_oracle.py), (ii) the structural fixtures the
// reference ships (model dims, topology, TextGrids) and (iii) its own output on
// those fixtures, frozen as tests/golden/oracle_vectors.npz (generator:
// tests/golden/make_golden.py) so that it cannot drift unnoticed.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file.  The product path never does.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off: no implicit FMA fusion,
// so every fused multiply-add below is an explicit fmaf()).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// A.1  MFCC  — reference: MfccFunction._run → MfccComputer.compute_mfccs_for_export
// (MFA/corpus/features.py:193-251, :235); options FeatureConfigMixin.mfcc_options
// (MFA/corpus/features.py:780-820).  Restates Kaldi feat/feature-window.cc
// (NumFrames, FirstSampleOfFrame, ExtractWindow, ProcessWindow),
// feat/mel-computations.cc (MelBanks), feat/feature-mfcc.cc (MfccComputer::Compute),
// matrix/matrix-functions.cc (ComputeDctMatrix).
// ---------------------------------------------------------------------------

struct orc_mfcc_opts {
  float samp_freq;        // 16000
  float frame_length_ms;  // 25
  float frame_shift_ms;   // 10
  float preemph;          // 0.97
  float low_freq;         // 20
  float high_freq;        // 7800 (<=0: offset from nyquist)
  float cepstral_lifter;  // 22
  float energy_floor;     // 0
  int32_t num_mel_bins;   // 23
  int32_t num_ceps;       // 13
  int32_t snip_edges;     // MFA default 0, model meta default 1
  int32_t remove_dc_offset;  // 1
  int32_t use_energy;     // 0
  int32_t raw_energy;     // 1
};

static int window_shift(const orc_mfcc_opts *o) { return (int)(o->samp_freq * 0.001f * o->frame_shift_ms); }
static int window_size(const orc_mfcc_opts *o) { return (int)(o->samp_freq * 0.001f * o->frame_length_ms); }
static int padded_window_size(const orc_mfcc_opts *o) {
  int n = window_size(o), p = 1;
  while (p < n) p <<= 1;  // round_to_power_of_two = true
  return p;
}

ORC_API int32_t orc_mfcc_num_frames(int64_t num_samples, const orc_mfcc_opts *o) {
  int64_t shift = window_shift(o), len = window_size(o);
  if (o->snip_edges) {
    if (num_samples < len) return 0;
    return (int32_t)(1 + (num_samples - len) / shift);
  }
  return (int32_t)((num_samples + shift / 2) / shift);
}

static inline float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

// In-place complex radix-2 DIT FFT on n complex points (float arithmetic, double twiddles
// rounded to float).  Kaldi uses a split-radix real FFT (srfft); the transform computed is
// the same, rounding differs at the 1e-7 level.
static void cfft(float *re, float *im, int n) {
  for (int i = 1, j = 0; i < n; i++) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (int len = 2; len <= n; len <<= 1) {
    double ang = -2.0 * M_PI / len;
    for (int i = 0; i < n; i += len) {
      for (int k = 0; k < len / 2; k++) {
        float wr = (float)cos(ang * k), wi = (float)sin(ang * k);
        float ur = re[i + k], ui = im[i + k];
        float xr = re[i + k + len / 2], xi = im[i + k + len / 2];
        float vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
        re[i + k] = ur + vr; im[i + k] = ui + vi;
        re[i + k + len / 2] = ur - vr; im[i + k + len / 2] = ui - vi;
      }
    }
  }
}

// Power spectrum of a real frame of length n (power of two): out[0..n/2].
// Kaldi ComputePowerSpectrum: bin0 = DC^2, bin n/2 = Nyquist^2, else re^2+im^2.
static void power_spectrum(const float *x, int n, float *out) {
  std::vector<float> re(n), im(n, 0.0f);
  for (int i = 0; i < n; i++) re[i] = x[i];
  cfft(re.data(), im.data(), n);
  out[0] = re[0] * re[0];
  for (int i = 1; i < n / 2; i++) out[i] = re[i] * re[i] + im[i] * im[i];
  out[n / 2] = re[n / 2] * re[n / 2];
}

struct MelBank { int first; std::vector<float> w; };

static void make_mel_banks(const orc_mfcc_opts *o, std::vector<MelBank> *banks) {
  int num_bins = o->num_mel_bins;
  float sample_freq = o->samp_freq;
  int padded = padded_window_size(o);
  int num_fft_bins = padded / 2;
  float nyquist = 0.5f * sample_freq;
  float low = o->low_freq, high = (o->high_freq > 0.0f) ? o->high_freq : nyquist + o->high_freq;
  float fft_bin_width = sample_freq / padded;
  float mel_low = mel_scale(low), mel_high = mel_scale(high);
  float mel_delta = (mel_high - mel_low) / (num_bins + 1);
  banks->resize(num_bins);
  for (int bin = 0; bin < num_bins; bin++) {
    float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta,
          right = mel_low + (bin + 2) * mel_delta;
    std::vector<float> this_bin(num_fft_bins, 0.0f);
    int first = -1, last = -1;
    for (int i = 0; i < num_fft_bins; i++) {
      float freq = fft_bin_width * i;
      float mel = mel_scale(freq);
      if (mel > left && mel < right) {
        float weight;
        if (mel <= center) weight = (mel - left) / (center - left);
        else weight = (right - mel) / (right - center);
        this_bin[i] = weight;
        if (first == -1) first = i;
        last = i;
      }
    }
    (*banks)[bin].first = first;
    (*banks)[bin].w.assign(this_bin.begin() + first, this_bin.begin() + last + 1);
  }
}

// Exposes the mel filterbank / DCT / lifter / window tables so the product's host-side table
// builder can be checked against the oracle's (tests only).
ORC_API int32_t orc_mfcc_tables(const orc_mfcc_opts *o, float *window /*[win]*/, float *mel_dense /*[bins][padded/2]*/,
                                float *dct /*[num_ceps][bins]*/, float *lifter /*[num_ceps]*/) {
  int win = window_size(o), padded = padded_window_size(o), nb = o->num_mel_bins;
  double a = 2.0 * M_PI / (win - 1);
  for (int i = 0; i < win; i++) window[i] = (float)pow(0.5 - 0.5 * cos(a * (double)i), 0.85);  // "povey"
  std::vector<MelBank> banks; make_mel_banks(o, &banks);
  memset(mel_dense, 0, sizeof(float) * nb * (padded / 2));
  for (int b = 0; b < nb; b++)
    for (size_t k = 0; k < banks[b].w.size(); k++) mel_dense[b * (padded / 2) + banks[b].first + k] = banks[b].w[k];
  // ComputeDctMatrix (Real = float): row 0 = sqrt(1/N); row k = sqrt(2/N) cos(pi/N (n+0.5) k)
  float norm0 = std::sqrt(1.0f / (float)nb), norm = std::sqrt(2.0f / (float)nb);
  for (int k = 0; k < o->num_ceps; k++)
    for (int n = 0; n < nb; n++)
      dct[k * nb + n] = (k == 0) ? norm0 : (float)(norm * std::cos((double)M_PI / nb * (n + 0.5) * k));
  for (int i = 0; i < o->num_ceps; i++)
    lifter[i] = (o->cepstral_lifter != 0.0f) ? (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * i / o->cepstral_lifter)) : 1.0f;
  return 0;
}

// wave: samples on the int16 scale (Kaldi convention), dither must be 0 (parity domain).
ORC_API int32_t orc_mfcc(const float *wave, int64_t n, const orc_mfcc_opts *o, float *out /*[T][num_ceps]*/) {
  int T = orc_mfcc_num_frames(n, o);
  int win = window_size(o), shift = window_shift(o), padded = padded_window_size(o);
  int nb = o->num_mel_bins, nc = o->num_ceps;
  std::vector<float> window(win), mel_dense((size_t)nb * (padded / 2)), dct((size_t)nc * nb), lifter(nc);
  orc_mfcc_tables(o, window.data(), mel_dense.data(), dct.data(), lifter.data());
  std::vector<MelBank> banks; make_mel_banks(o, &banks);
  std::vector<float> frame(padded), ps(padded / 2 + 1), mel(nb);
  float log_energy_floor = (o->energy_floor > 0.0f) ? logf(o->energy_floor) : -std::numeric_limits<float>::infinity();
  for (int f = 0; f < T; f++) {
    int64_t start = o->snip_edges ? (int64_t)f * shift : (int64_t)shift * f + shift / 2 - win / 2;
    // ExtractWindow with reflection at the edges
    for (int s = 0; s < win; s++) {
      int64_t si = s + start;
      while (si < 0 || si >= n) { if (si < 0) si = -si - 1; else si = 2 * n - 1 - si; }
      frame[s] = wave[si];
    }
    for (int s = win; s < padded; s++) frame[s] = 0.0f;
    // ProcessWindow: (dither=0) → remove DC (Sum() accumulates in double) → raw log energy → preemphasis → window
    if (o->remove_dc_offset) {
      double sum = 0.0; for (int s = 0; s < win; s++) sum += frame[s];
      float off = -((float)sum) / win;
      for (int s = 0; s < win; s++) frame[s] += off;
    }
    float raw_log_energy = 0.0f;
    if (o->use_energy && o->raw_energy) {
      float e = 0.0f; for (int s = 0; s < win; s++) e = fmaf(frame[s], frame[s], e);
      raw_log_energy = logf(std::max(e, std::numeric_limits<float>::epsilon()));
    }
    if (o->preemph != 0.0f) {
      for (int s = win - 1; s > 0; s--) frame[s] -= o->preemph * frame[s - 1];
      frame[0] -= o->preemph * frame[0];
    }
    for (int s = 0; s < win; s++) frame[s] *= window[s];
    if (o->use_energy && !o->raw_energy) {
      float e = 0.0f; for (int s = 0; s < win; s++) e = fmaf(frame[s], frame[s], e);
      raw_log_energy = logf(std::max(e, std::numeric_limits<float>::epsilon()));
    }
    power_spectrum(frame.data(), padded, ps.data());
    for (int b = 0; b < nb; b++) {
      float e = 0.0f;  // VecVec (sdot); order here: ascending FFT bin, fused multiply-add
      for (size_t k = 0; k < banks[b].w.size(); k++) e = fmaf(banks[b].w[k], ps[banks[b].first + k], e);
      e = std::max(e, std::numeric_limits<float>::epsilon());  // ApplyFloor
      mel[b] = logf(e);                                        // ApplyLog
    }
    for (int k = 0; k < nc; k++) {
      float acc = 0.0f;  // AddMatVec (sgemv); order: ascending mel bin
      for (int b = 0; b < nb; b++) acc = fmaf(dct[k * nb + b], mel[b], acc);
      out[(size_t)f * nc + k] = acc * lifter[k];
    }
    if (o->use_energy) {
      if (o->energy_floor > 0.0f && raw_log_energy < log_energy_floor) raw_log_energy = log_energy_floor;
      out[(size_t)f * nc] = raw_log_energy;
    }
  }
  return T;
}

// ---------------------------------------------------------------------------
// A.2  CMVN — reference: AcousticCorpusMixin.calc_cmvn → CmvnComputer.export_cmvn
// (MFA/corpus/acoustic_corpus.py:1315-1367); online compute_cmvn_from_features
// (MFA/online/alignment.py:86-88).  Restates Kaldi transform/cmvn.cc AccCmvnStats / ApplyCmvn
// (norm_vars = false).
// ---------------------------------------------------------------------------
ORC_API void orc_cmvn_acc(const float *feats, int32_t T, int32_t dim, double *stats /*[2][dim+1], accumulated into*/) {
  for (int t = 0; t < T; t++) {
    const float *x = feats + (size_t)t * dim;
    stats[dim] += 1.0f;
    for (int d = 0; d < dim; d++) {
      stats[d] += (double)(x[d] * 1.0f);
      stats[(dim + 1) + d] += (double)(x[d] * x[d] * 1.0f);
    }
  }
}

ORC_API void orc_cmvn_apply(const double *stats, float *feats, int32_t T, int32_t dim) {
  double count = stats[dim];
  for (int d = 0; d < dim; d++) {
    double mean = stats[d] / count;
    float offset = (float)(-mean);
    for (int t = 0; t < T; t++) feats[(size_t)t * dim + d] += offset;
  }
}

// ---------------------------------------------------------------------------
// A.3  Deltas / splice / affine transforms — reference: Job.construct_feature_archive
// (MFA/db.py:2101-2136), FineTuneFunction (MFA/alignment/multiprocessing.py:1287-1304).
// Restates Kaldi feat/feature-functions.cc (DeltaFeatures, SpliceFrames) and
// transform/transform-common.cc (ApplyAffineTransform).
// ---------------------------------------------------------------------------
static void delta_scales(int order, int window, std::vector<std::vector<float> > *scales) {
  scales->assign(order + 1, std::vector<float>());
  (*scales)[0].assign(1, 1.0f);
  for (int i = 1; i <= order; i++) {
    std::vector<float> &prev = (*scales)[i - 1], &cur = (*scales)[i];
    int prev_offset = ((int)prev.size() - 1) / 2, cur_offset = prev_offset + window;
    cur.assign(prev.size() + 2 * window, 0.0f);
    float normalizer = 0.0f;
    for (int j = -window; j <= window; j++) {
      normalizer += j * j;
      for (int k = -prev_offset; k <= prev_offset; k++)
        cur[j + k + cur_offset] += (float)j * prev[k + prev_offset];
    }
    float s = (float)(1.0 / normalizer);
    for (size_t k = 0; k < cur.size(); k++) cur[k] *= s;
  }
}

ORC_API void orc_delta_scales(int32_t order, int32_t window, float *out /*[order+1][2*order*window+1] zero padded, centred*/) {
  std::vector<std::vector<float> > sc; delta_scales(order, window, &sc);
  int width = 2 * order * window + 1, c = order * window;
  for (int i = 0; i <= order; i++) {
    for (int k = 0; k < width; k++) out[i * width + k] = 0.0f;
    int mo = ((int)sc[i].size() - 1) / 2;
    for (int j = -mo; j <= mo; j++) out[i * width + c + j] = sc[i][j + mo];
  }
}

ORC_API void orc_deltas(const float *in, int32_t T, int32_t dim, int32_t order, int32_t window, float *out /*[T][(order+1)*dim]*/) {
  std::vector<std::vector<float> > sc; delta_scales(order, window, &sc);
  int od = (order + 1) * dim;
  for (int t = 0; t < T; t++) {
    float *o = out + (size_t)t * od;
    for (int k = 0; k < od; k++) o[k] = 0.0f;
    for (int i = 0; i <= order; i++) {
      int mo = ((int)sc[i].size() - 1) / 2;
      for (int j = -mo; j <= mo; j++) {
        int tf = t + j; if (tf < 0) tf = 0; else if (tf >= T) tf = T - 1;
        float s = sc[i][j + mo];
        if (s != 0.0f)
          for (int d = 0; d < dim; d++) o[i * dim + d] = fmaf(s, in[(size_t)tf * dim + d], o[i * dim + d]);  // AddVec (saxpy)
      }
    }
  }
}

ORC_API void orc_splice(const float *in, int32_t T, int32_t dim, int32_t left, int32_t right, float *out /*[T][dim*(left+right+1)]*/) {
  int n = left + right + 1;
  for (int t = 0; t < T; t++)
    for (int j = 0; j < n; j++) {
      int tf = t + j - left; if (tf < 0) tf = 0; if (tf >= T) tf = T - 1;
      memcpy(out + ((size_t)t * n + j) * dim, in + (size_t)tf * dim, sizeof(float) * dim);
    }
}

// y = M x (cols == dim) or y = M [x;1] (cols == dim+1).  Order: ascending input index, fmaf chain from 0,
// offset added last.
ORC_API void orc_affine(const float *in, int32_t T, int32_t dim, const float *M, int32_t rows, int32_t cols, float *out /*[T][rows]*/) {
  for (int t = 0; t < T; t++)
    for (int r = 0; r < rows; r++) {
      float acc = 0.0f;
      for (int d = 0; d < dim; d++) acc = fmaf(M[(size_t)r * cols + d], in[(size_t)t * dim + d], acc);
      if (cols == dim + 1) acc += M[(size_t)r * cols + dim];
      out[(size_t)t * rows + r] = acc;
    }
}

// ---------------------------------------------------------------------------
// A.6  Diagonal-GMM log-likelihoods — reference: acoustic scoring inside
// GmmAligner.align_utterance (MFA/alignment/multiprocessing.py:846-853) and
// gmm_compute_likes (:1415).  Restates Kaldi gmm/decodable-am-diag-gmm.cc
// DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased and
// VectorBase<float>::LogSumExp(prune=-1).
//   loglikes = gconsts; loglikes += means_invvars * x; loglikes += -0.5 * inv_vars * x^2   (two sgemv)
// The order of the float sums inside BLAS sgemv is unspecified; the oracle fixes it as one fmaf chain:
//   acc = gconst; for d: acc = fmaf(mi[d], x[d], acc); for d: acc = fmaf(-0.5*iv[d], x[d]*x[d], acc)
// (the same k-ordered chain an f32 MFMA accumulates, so the device can match it bit for bit).
// ---------------------------------------------------------------------------
static inline float gauss_ll(const float *x, const float *x2, int D, float gconst, const float *mi, const float *iv) {
  float acc = gconst;
  for (int d = 0; d < D; d++) acc = fmaf(mi[d], x[d], acc);
  for (int d = 0; d < D; d++) acc = fmaf(-0.5f * iv[d], x2[d], acc);
  return acc;
}

static inline float log_sum_exp(const float *v, int n) {
  float mx = v[0];
  for (int i = 1; i < n; i++) mx = std::max(mx, v[i]);
  float cutoff = mx + logf(std::numeric_limits<float>::epsilon());  // kMinLogDiffFloat
  double sum = 0.0;
  for (int i = 0; i < n; i++) if (v[i] >= cutoff) sum += expf(v[i] - mx);
  return (float)((double)mx + log(sum));
}

ORC_API void orc_gmm_loglikes(const float *feats, int32_t T, int32_t D, const float *gconsts, const float *means_invvars,
                              const float *inv_vars, const int32_t *pdf_offsets /*[P+1] gaussian offsets*/,
                              const int32_t *pdf_list, int32_t n_pdf, float *out /*[T][n_pdf]*/) {
  std::vector<float> x2(D), ll;
  for (int t = 0; t < T; t++) {
    const float *x = feats + (size_t)t * D;
    for (int d = 0; d < D; d++) x2[d] = x[d] * x[d];
    for (int j = 0; j < n_pdf; j++) {
      int p = pdf_list[j], g0 = pdf_offsets[p], g1 = pdf_offsets[p + 1];
      ll.resize(g1 - g0);
      for (int g = g0; g < g1; g++)
        ll[g - g0] = gauss_ll(x, x2.data(), D, gconsts[g], means_invvars + (size_t)g * D, inv_vars + (size_t)g * D);
      out[(size_t)t * n_pdf + j] = log_sum_exp(ll.data(), g1 - g0);
    }
  }
}

// ---------------------------------------------------------------------------
// A.5  TransitionModel derived tables — reference: kalpy.gmm.utils.read_transition_model as used at
// MFA/alignment/base.py:339, MFA/models.py:481-491.  Restates Kaldi hmm/transition-model.cc
// (ComputeDerived, IsSelfLoop, IsFinal, GetNonSelfLoopLogProb) and hmm/hmm-utils.cc
// GetScaledTransitionLogProb (used by AddTransitionProbs).
//
// Topology is passed flattened: phone2entry[phone] (-1 = none); entry e covers topo states
// entry_off[e]..entry_off[e+1]; topo state s has transitions trans_off[s]..trans_off[s+1] of (dst, prob).
// tuples: [n][4] = (phone, hmm_state, forward_pdf, self_loop_pdf).  log_probs[0] unused.
// ---------------------------------------------------------------------------
ORC_API int32_t orc_tm_derive(const int32_t *phone2entry, const int32_t *entry_off, const int32_t *trans_off,
                              const int32_t *trans_dst, const int32_t *tuples, int32_t n_tuples,
                              int32_t *state2id /*[n_tuples+2]*/, int32_t *id2state, int32_t *id2pdf,
                              int32_t *is_self_loop, int32_t *is_final, int32_t cap_ids) {
  int cur = 1;
  for (int ts = 1; ts <= n_tuples + 1; ts++) {
    state2id[ts] = cur;
    if (ts <= n_tuples) {
      int phone = tuples[(ts - 1) * 4], hs = tuples[(ts - 1) * 4 + 1];
      int s = entry_off[phone2entry[phone]] + hs;
      cur += trans_off[s + 1] - trans_off[s];
    }
  }
  state2id[0] = 0;
  if (cur > cap_ids) return -cur;
  for (int ts = 1; ts <= n_tuples; ts++) {
    int phone = tuples[(ts - 1) * 4], hs = tuples[(ts - 1) * 4 + 1];
    int e = phone2entry[phone], s = entry_off[e] + hs;
    for (int tid = state2id[ts]; tid < state2id[ts + 1]; tid++) {
      int idx = tid - state2id[ts];
      int dst = trans_dst[trans_off[s] + idx];
      id2state[tid] = ts;
      is_self_loop[tid] = (dst == hs);
      int ds = entry_off[e] + dst;
      is_final[tid] = (trans_off[ds + 1] - trans_off[ds] == 0);
      id2pdf[tid] = is_self_loop[tid] ? tuples[(ts - 1) * 4 + 3] : tuples[(ts - 1) * 4 + 2];
    }
  }
  id2state[0] = 0; id2pdf[0] = -1; is_self_loop[0] = 0; is_final[0] = 0;
  return cur - 1;  // number of transition-ids
}

// scaled[tid] = GetScaledTransitionLogProb(tid, transition_scale, self_loop_scale)
ORC_API void orc_tm_scaled_logprobs(const int32_t *state2id, const int32_t *id2state, const int32_t *is_self_loop,
                                    const float *log_probs, int32_t n_ids, float transition_scale, float self_loop_scale,
                                    float *scaled /*[n_ids+1]*/) {
  scaled[0] = 0.0f;
  for (int tid = 1; tid <= n_ids; tid++) {
    if (transition_scale == self_loop_scale) { scaled[tid] = log_probs[tid] * transition_scale; continue; }
    if (is_self_loop[tid]) { scaled[tid] = self_loop_scale * log_probs[tid]; continue; }
    int ts = id2state[tid];
    int self_loop = 0;
    for (int t2 = state2id[ts]; t2 < state2id[ts + 1]; t2++) if (is_self_loop[t2]) { self_loop = t2; break; }
    float non_self_lp;
    if (self_loop == 0) non_self_lp = 0.0f;
    else {
      float slp = expf(log_probs[self_loop]), nslp = 1.0f - slp;
      if (nslp <= 0.0f) nslp = 1.0e-10f;
      non_self_lp = logf(nslp);
    }
    float ignoring = log_probs[tid] - non_self_lp;  // GetTransitionLogProbIgnoringSelfLoops
    scaled[tid] = self_loop_scale * non_self_lp + transition_scale * ignoring;
  }
}

// ---------------------------------------------------------------------------
// A.8/A.9  Alignment = AddTransitionProbs + AlignUtteranceWrapper + FasterDecoder — reference:
// GmmAligner.align_utterance(fst, feats) / .export_alignments (MFA/alignment/multiprocessing.py:846-853,
// :1311-1315; MFA/online/alignment.py:107).  Restates Kaldi decoder/faster-decoder.cc (FasterDecoder with
// FasterDecoderOptions defaults: max_active INT_MAX, min_active 20, beam_delta 0.5, hash_ratio 2.0),
// util/hash-list-inl.h (HashList: list order = buckets by first occupancy, within bucket by insertion),
// decoder/decoder-wrappers.cc (AlignUtteranceWrapper: beam, then retry_beam) and
// fstext GetLinearSymbolSequence.
// ---------------------------------------------------------------------------
struct orc_arc { int32_t ilabel, olabel; float weight; int32_t nextstate; };

namespace {

struct Token {
  orc_arc arc; int prev; double cost;  // prev: index into token pool (-1 none)
};

struct Elem { int32_t key; int val; int tail; };  // val: token index, tail: elem index (-1 end)

// HashList restated with indices instead of pointers.
struct HashList {
  struct Bucket { size_t prev_bucket; int last_elem; };
  std::vector<Elem> elems;   // pool (per frame generation; never freed inside a decode)
  std::vector<Bucket> buckets;
  int list_head = -1; size_t bucket_list_tail = (size_t)-1; size_t hash_size = 0;
  void SetSize(size_t sz) { hash_size = sz; if (sz > buckets.size()) buckets.resize(sz, Bucket{0, -1}); }
  size_t Size() const { return hash_size; }
  int GetList() const { return list_head; }
  int Clear() {
    for (size_t b = bucket_list_tail; b != (size_t)-1; b = buckets[b].prev_bucket) buckets[b].last_elem = -1;
    bucket_list_tail = (size_t)-1;
    int ans = list_head; list_head = -1; return ans;
  }
  // returns elem index; *inserted tells whether val was stored
  int Insert(int32_t key, int val, bool *inserted) {
    size_t index = (size_t)key % hash_size;
    Bucket &bucket = buckets[index];
    if (bucket.last_elem != -1) {
      int head = (bucket.prev_bucket == (size_t)-1) ? list_head : elems[buckets[bucket.prev_bucket].last_elem].tail;
      int tail = elems[bucket.last_elem].tail;
      for (int e = head; e != tail; e = elems[e].tail) if (elems[e].key == key) { *inserted = false; return e; }
    }
    int ei = (int)elems.size();
    elems.push_back(Elem{key, val, -1});
    *inserted = true;
    if (bucket.last_elem == -1) {
      if (bucket_list_tail == (size_t)-1) list_head = ei;
      else elems[buckets[bucket_list_tail].last_elem].tail = ei;
      elems[ei].tail = -1;
      bucket.last_elem = ei;
      bucket.prev_bucket = bucket_list_tail;
      bucket_list_tail = index;
    } else {
      elems[ei].tail = elems[bucket.last_elem].tail;
      elems[bucket.last_elem].tail = ei;
      bucket.last_elem = ei;
    }
    return ei;
  }
};

struct Graph {
  int32_t num_states, start; const int64_t *arc_off; const orc_arc *arcs; const float *final_w;
};

// Two forms of the decodable.  Dense: a precomputed [T][ncols] matrix (gmm_compute_likes-style callers, and what most tests
// feed).  Lazy: Kaldi's own, DecodableAmDiagGmmScaled over DecodableAmDiagGmmUnmapped (gmm/decodable-am-diag-gmm.{h,cc}):
// LogLikelihood(frame, tid) = scale * LogLikelihoodZeroBased(frame, TransitionIdToPdf(tid)), and LogLikelihoodZeroBased
// keeps one cached value per pdf, valid while its hit_time equals the frame asked for — so a (frame, pdf) score is computed
// only when a live token's arc asks for it, once.  Same gauss_ll / log_sum_exp as orc_gmm_loglikes: identical floats.
struct Decodable {
  const float *loglikes; int32_t T, ncols; const int32_t *tid2col; float scale;
  // lazy form (feats != nullptr): tid2col then maps transition-id -> pdf id
  const float *feats = nullptr; int32_t D = 0; const float *gconsts = nullptr, *means_invvars = nullptr, *inv_vars = nullptr;
  const int32_t *pdf_offsets = nullptr;
  mutable std::vector<float> cache, x2, ll; mutable std::vector<int32_t> hit; mutable int x2_frame = -1; mutable int64_t evals = 0;
  void InitLazy(int32_t num_pdfs) { cache.assign(num_pdfs, 0.0f); hit.assign(num_pdfs, -1); x2.resize(D); }
  inline float LogLikelihood(int frame, int tid) const {
    if (!feats) return scale * loglikes[(size_t)frame * ncols + tid2col[tid]];
    int p = tid2col[tid];
    if (hit[p] == frame) return scale * cache[p];
    const float *x = feats + (size_t)frame * D;
    if (x2_frame != frame) { for (int d = 0; d < D; d++) x2[d] = x[d] * x[d]; x2_frame = frame; }
    int g0 = pdf_offsets[p], g1 = pdf_offsets[p + 1];
    ll.resize(g1 - g0);
    for (int g = g0; g < g1; g++)
      ll[g - g0] = gauss_ll(x, x2.data(), D, gconsts[g], means_invvars + (size_t)g * D, inv_vars + (size_t)g * D);
    float v = log_sum_exp(ll.data(), g1 - g0);
    cache[p] = v; hit[p] = frame; evals++;
    return scale * v;
  }
};

struct FasterDecoder {
  const Graph &g; float beam; int32_t max_active, min_active; float beam_delta, hash_ratio;
  HashList toks; std::vector<Token> pool; std::vector<int> queue; std::vector<double> tmp; int num_frames_decoded = -1;
  // statistics for tests / design (max tokens alive, total candidates)
  int64_t stat_max_toks = 0, stat_sum_toks = 0;
  // optional (design studies of the device's lazy-scoring bands): per frame, over the tokens ENTERING the frame,
  // {min of depth[s][1], max of depth[s][0], number of tokens, depth[.][0] of the best token}
  const int32_t *stat_depth = nullptr; int32_t *stat_band = nullptr;

  FasterDecoder(const Graph &gr, float bm, int32_t maxa, int32_t mina, float bd, float hr)
      : g(gr), beam(bm), max_active(maxa), min_active(mina), beam_delta(bd), hash_ratio(hr) { toks.SetSize(1000); }

  int NewToken(const orc_arc &arc, float ac_cost, int prev, bool with_ac) {
    Token t; t.arc = arc; t.prev = prev;
    if (prev >= 0) t.cost = with_ac ? pool[prev].cost + arc.weight + ac_cost : pool[prev].cost + arc.weight;
    else t.cost = with_ac ? (double)(arc.weight + ac_cost) : (double)arc.weight;
    pool.push_back(t); return (int)pool.size() - 1;
  }

  void InitDecoding() {
    toks.Clear(); toks.elems.clear(); pool.clear();
    orc_arc dummy{0, 0, 0.0f, g.start};
    bool ins; toks.Insert(g.start, NewToken(dummy, 0.0f, -1, false), &ins);
    ProcessNonemitting(std::numeric_limits<float>::max());
    num_frames_decoded = 0;
  }

  double GetCutoff(int list_head, size_t *tok_count, float *adaptive_beam, int *best_elem) {
    double best_cost = std::numeric_limits<double>::infinity();
    size_t count = 0;
    if (max_active == std::numeric_limits<int32_t>::max() && min_active == 0) {
      for (int e = list_head; e != -1; e = toks.elems[e].tail, count++) {
        double w = pool[toks.elems[e].val].cost;
        if (w < best_cost) { best_cost = w; *best_elem = e; }
      }
      *tok_count = count; *adaptive_beam = beam; return best_cost + beam;
    }
    tmp.clear();
    for (int e = list_head; e != -1; e = toks.elems[e].tail, count++) {
      double w = pool[toks.elems[e].val].cost; tmp.push_back(w);
      if (w < best_cost) { best_cost = w; *best_elem = e; }
    }
    *tok_count = count;
    double beam_cutoff = best_cost + beam, min_active_cutoff = std::numeric_limits<double>::infinity(),
           max_active_cutoff = std::numeric_limits<double>::infinity();
    if (tmp.size() > (size_t)max_active) {
      std::nth_element(tmp.begin(), tmp.begin() + max_active, tmp.end());
      max_active_cutoff = tmp[max_active];
    }
    if (max_active_cutoff < beam_cutoff) { *adaptive_beam = (float)(max_active_cutoff - best_cost + beam_delta); return max_active_cutoff; }
    if (tmp.size() > (size_t)min_active) {
      if (min_active == 0) min_active_cutoff = best_cost;
      else {
        std::nth_element(tmp.begin(), tmp.begin() + min_active,
                         tmp.size() > (size_t)max_active ? tmp.begin() + max_active : tmp.end());
        min_active_cutoff = tmp[min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) { *adaptive_beam = (float)(min_active_cutoff - best_cost + beam_delta); return min_active_cutoff; }
    *adaptive_beam = beam; return beam_cutoff;
  }

  double ProcessEmitting(const Decodable &dec) {
    int frame = num_frames_decoded;
    int last_toks = toks.Clear();
    size_t tok_cnt; float adaptive_beam; int best_elem = -1;
    double weight_cutoff = GetCutoff(last_toks, &tok_cnt, &adaptive_beam, &best_elem);
    if (stat_depth && stat_band) {
      int32_t lo = std::numeric_limits<int32_t>::max(), hi = -1;
      for (int e = last_toks; e != -1; e = toks.elems[e].tail) {
        int32_t st = toks.elems[e].key;
        lo = std::min(lo, stat_depth[2 * st + 1]); hi = std::max(hi, stat_depth[2 * st]);
      }
      stat_band[4 * frame] = lo; stat_band[4 * frame + 1] = hi; stat_band[4 * frame + 2] = (int32_t)tok_cnt;
      stat_band[4 * frame + 3] = best_elem >= 0 ? stat_depth[2 * toks.elems[best_elem].key] : -1;
    }
    stat_max_toks = std::max<int64_t>(stat_max_toks, (int64_t)tok_cnt); stat_sum_toks += (int64_t)tok_cnt;
    size_t new_sz = (size_t)((float)tok_cnt * hash_ratio);  // PossiblyResizeHash
    if (new_sz > toks.Size()) toks.SetSize(new_sz);
    double next_weight_cutoff = std::numeric_limits<double>::infinity();
    if (best_elem != -1) {
      int32_t state = toks.elems[best_elem].key; int tok = toks.elems[best_elem].val;
      for (int64_t a = g.arc_off[state]; a < g.arc_off[state + 1]; a++) {
        const orc_arc &arc = g.arcs[a];
        if (arc.ilabel != 0) {
          float ac_cost = -dec.LogLikelihood(frame, arc.ilabel);
          double new_weight = arc.weight + pool[tok].cost + ac_cost;
          if (new_weight + adaptive_beam < next_weight_cutoff) next_weight_cutoff = new_weight + adaptive_beam;
        }
      }
    }
    for (int e = last_toks; e != -1; e = toks.elems[e].tail) {
      int32_t state = toks.elems[e].key; int tok = toks.elems[e].val;
      if (pool[tok].cost < weight_cutoff) {
        for (int64_t a = g.arc_off[state]; a < g.arc_off[state + 1]; a++) {
          orc_arc arc = g.arcs[a];
          if (arc.ilabel != 0) {
            float ac_cost = -dec.LogLikelihood(frame, arc.ilabel);
            double new_weight = arc.weight + pool[tok].cost + ac_cost;
            if (new_weight < next_weight_cutoff) {
              int new_tok = NewToken(arc, ac_cost, tok, true);
              bool ins; int e_found = toks.Insert(arc.nextstate, new_tok, &ins);
              if (new_weight + adaptive_beam < next_weight_cutoff) next_weight_cutoff = new_weight + adaptive_beam;
              if (!ins) {
                if (pool[toks.elems[e_found].val].cost > pool[new_tok].cost) toks.elems[e_found].val = new_tok;
              }
            }
          }
        }
      }
    }
    num_frames_decoded++;
    return next_weight_cutoff;
  }

  void ProcessNonemitting(double cutoff) {
    for (int e = toks.GetList(); e != -1; e = toks.elems[e].tail) queue.push_back(e);
    while (!queue.empty()) {
      int e = queue.back(); queue.pop_back();
      int32_t state = toks.elems[e].key; int tok = toks.elems[e].val;
      if (pool[tok].cost > cutoff) continue;
      for (int64_t a = g.arc_off[state]; a < g.arc_off[state + 1]; a++) {
        const orc_arc &arc = g.arcs[a];
        if (arc.ilabel == 0) {
          int new_tok = NewToken(arc, 0.0f, tok, false);
          if (pool[new_tok].cost > cutoff) { pool.pop_back(); continue; }
          bool ins; int e_found = toks.Insert(arc.nextstate, new_tok, &ins);
          if (ins) queue.push_back(e_found);
          else if (pool[toks.elems[e_found].val].cost > pool[new_tok].cost) { toks.elems[e_found].val = new_tok; queue.push_back(e_found); }
        }
      }
    }
  }

  void Decode(const Decodable &dec) {
    InitDecoding();
    while (num_frames_decoded < dec.T) { double c = ProcessEmitting(dec); ProcessNonemitting(c); }
  }

  bool ReachedFinal() const {
    for (int e = toks.GetList(); e != -1; e = toks.elems[e].tail)
      if (pool[toks.elems[e].val].cost != std::numeric_limits<double>::infinity() &&
          g.final_w[toks.elems[e].key] != std::numeric_limits<float>::infinity()) return true;
    return false;
  }

  // GetBestPath + GetLinearSymbolSequence. Returns false if no output.
  bool BestPath(std::vector<int32_t> *ali, std::vector<int32_t> *words, float *graph_cost, float *ac_cost,
                std::vector<float> *per_frame_ac) const {
    int best_tok = -1; bool is_final = ReachedFinal();
    if (!is_final) {
      for (int e = toks.GetList(); e != -1; e = toks.elems[e].tail)
        if (best_tok == -1 || pool[best_tok].cost > pool[toks.elems[e].val].cost) best_tok = toks.elems[e].val;
    } else {
      double infinity = std::numeric_limits<double>::infinity(), best_cost = infinity;
      for (int e = toks.GetList(); e != -1; e = toks.elems[e].tail) {
        double this_cost = pool[toks.elems[e].val].cost + (double)g.final_w[toks.elems[e].key];
        if (this_cost < best_cost && this_cost != infinity) { best_cost = this_cost; best_tok = toks.elems[e].val; }
      }
    }
    if (best_tok == -1) return false;
    struct LArc { int32_t il, ol; float g, a; };
    std::vector<LArc> rev;
    for (int tok = best_tok; tok != -1; tok = pool[tok].prev) {
      float tot_cost = (float)(pool[tok].cost - (pool[tok].prev >= 0 ? pool[pool[tok].prev].cost : 0.0));
      float gc = pool[tok].arc.weight, ac = tot_cost - gc;
      rev.push_back(LArc{pool[tok].arc.ilabel, pool[tok].arc.olabel, gc, ac});
    }
    rev.pop_back();  // the fake start token
    float w1 = 0.0f, w2 = 0.0f;
    ali->clear(); words->clear(); if (per_frame_ac) per_frame_ac->clear();
    for (int i = (int)rev.size() - 1; i >= 0; i--) {
      w1 += rev[i].g; w2 += rev[i].a;
      if (rev[i].il != 0) { ali->push_back(rev[i].il); if (per_frame_ac) per_frame_ac->push_back(rev[i].a); }
      if (rev[i].ol != 0) words->push_back(rev[i].ol);
    }
    if (is_final) w1 += g.final_w[pool[best_tok].arc.nextstate];
    *graph_cost = w1; *ac_cost = w2;
    return true;
  }
};

}  // namespace

// arcs' weights must already include the scaled transition log-probs (orc_add_transition_probs).
// status: 0 ok first beam, 1 ok after retry, 2 failed.  like = -(graph+ac)/acoustic_scale.
static int32_t align_with(const Graph &g, const Decodable &dec, int32_t T, float acoustic_scale, float beam, float retry_beam,
                          int32_t *ali, int32_t *words, int32_t cap_words, int32_t *n_words, float *like,
                          float *per_frame_loglike, int64_t *stats, const int32_t *stat_depth = nullptr,
                          int32_t *stat_band = nullptr) {
  FasterDecoder d(g, beam, std::numeric_limits<int32_t>::max(), 20, 0.5f, 2.0f);
  d.stat_depth = stat_depth; d.stat_band = stat_band;
  d.Decode(dec);
  bool ans = d.ReachedFinal(); int status = 0;
  if (!ans && retry_beam != 0.0f) { status = 1; d.beam = retry_beam; d.Decode(dec); ans = d.ReachedFinal(); }
  if (stats) { stats[0] = d.stat_max_toks; stats[1] = d.stat_sum_toks; }
  if (!ans) return 2;
  std::vector<int32_t> a, w; std::vector<float> pf; float gc, ac;
  if (!d.BestPath(&a, &w, &gc, &ac, &pf)) return 2;
  if ((int)a.size() != T) return 3;
  memcpy(ali, a.data(), sizeof(int32_t) * T);
  *n_words = (int32_t)w.size();
  for (int i = 0; i < (int)w.size() && i < cap_words; i++) words[i] = w[i];
  *like = -(gc + ac) / acoustic_scale;
  if (per_frame_loglike) for (int t = 0; t < T; t++) per_frame_loglike[t] = pf[t] * (-1.0f / acoustic_scale);
  return status;
}

ORC_API int32_t orc_align(int32_t num_states, int32_t start, const int64_t *arc_off, const orc_arc *arcs, const float *final_w,
                          const float *loglikes, int32_t T, int32_t ncols, const int32_t *tid2col, float acoustic_scale,
                          float beam, float retry_beam, int32_t *ali /*[T]*/, int32_t *words /*[cap_words]*/, int32_t cap_words,
                          int32_t *n_words, float *like, float *per_frame_loglike /*[T] or NULL*/, int64_t *stats /*[2] or NULL*/) {
  if (start < 0 || num_states == 0) return 2;
  Graph g{num_states, start, arc_off, arcs, final_w};
  Decodable dec{loglikes, T, ncols, tid2col, acoustic_scale};
  return align_with(g, dec, T, acoustic_scale, beam, retry_beam, ali, words, cap_words, n_words, like, per_frame_loglike, stats);
}

// The same alignment with Kaldi's LAZY decodable: features and the acoustic model in, a (frame, pdf) log-likelihood
// computed when a live token's arc first asks for it — GmmAligner.align_utterance(fst, feats) as the reference runs it
// (MFA/alignment/multiprocessing.py:846-853).  tid2pdf[tid] = pdf id of a transition-id.  stats[2] (when given) receives
// the number of (frame, pdf) cells evaluated, both beams included.  Results equal orc_gmm_loglikes + orc_align bit for bit.
ORC_API int32_t orc_align_feats(int32_t num_states, int32_t start, const int64_t *arc_off, const orc_arc *arcs,
                                const float *final_w, const float *feats, int32_t T, int32_t D, const float *gconsts,
                                const float *means_invvars, const float *inv_vars, const int32_t *pdf_offsets, int32_t num_pdfs,
                                const int32_t *tid2pdf, float acoustic_scale, float beam, float retry_beam, int32_t *ali,
                                int32_t *words, int32_t cap_words, int32_t *n_words, float *like, float *per_frame_loglike,
                                int64_t *stats /*[3] or NULL*/, const int32_t *state_depth /*[S][2] or NULL*/,
                                int32_t *frame_band /*[T][4] or NULL: see FasterDecoder::stat_band (first beam's pass)*/) {
  if (start < 0 || num_states == 0) return 2;
  Graph g{num_states, start, arc_off, arcs, final_w};
  Decodable dec{nullptr, T, 0, tid2pdf, acoustic_scale};
  dec.feats = feats; dec.D = D; dec.gconsts = gconsts; dec.means_invvars = means_invvars; dec.inv_vars = inv_vars;
  dec.pdf_offsets = pdf_offsets; dec.InitLazy(num_pdfs);
  int32_t st = align_with(g, dec, T, acoustic_scale, beam, retry_beam, ali, words, cap_words, n_words, like, per_frame_loglike, stats,
                          state_depth, frame_band);
  if (stats) stats[2] = dec.evals;
  return st;
}

// AddTransitionProbs: arc.weight = Times(arc.weight, -scaled[tid]) for ilabel in [1, n_ids].
ORC_API int32_t orc_add_transition_probs(orc_arc *arcs, int64_t n_arcs, const float *scaled, int32_t n_ids) {
  for (int64_t a = 0; a < n_arcs; a++) {
    int32_t l = arcs[a].ilabel;
    if (l >= 1 && l <= n_ids) arcs[a].weight = arcs[a].weight + (-scaled[l]);
    else if (l != 0) return -1;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// A.10  SplitToPhones (reordered) → phone intervals — reference: Alignment.generate_ctm
// (MFA/alignment/multiprocessing.py:1734; MFA/online/alignment.py:113-117).  Restates Kaldi
// hmm/hmm-utils.cc SplitToPhonesInternal(reordered=true).  Output: per phone (begin_frame, n_frames, phone).
// ---------------------------------------------------------------------------
ORC_API int32_t orc_split_to_phones(const int32_t *ali, int32_t T, const int32_t *id2state, const int32_t *is_self_loop,
                                    const int32_t *is_final, const int32_t *tuples, int32_t *out /*[cap][3]*/, int32_t cap,
                                    int32_t *was_ok) {
  std::vector<size_t> end_points; *was_ok = 1;
  for (size_t i = 0; i < (size_t)T; i++) {
    int tid = ali[i];
    if (is_final[tid]) {
      while (i + 1 < (size_t)T && is_self_loop[ali[i + 1]]) {
        if (id2state[ali[i]] != id2state[ali[i + 1]]) { *was_ok = 0; break; }
        i++;
      }
      end_points.push_back(i + 1);
    } else if (i + 1 == (size_t)T) {
      *was_ok = 0; end_points.push_back(i + 1);
    } else {
      int ts = id2state[ali[i]], ns = id2state[ali[i + 1]];
      if (ts == ns) continue;
      int tp = tuples[(ts - 1) * 4], np = tuples[(ns - 1) * 4];
      if (tp != np) { *was_ok = 0; end_points.push_back(i + 1); }
    }
  }
  size_t cur = 0; int n = 0;
  for (size_t k = 0; k < end_points.size(); k++) {
    if (n < cap) {
      int ts = id2state[ali[cur]];
      out[n * 3] = (int32_t)cur; out[n * 3 + 1] = (int32_t)(end_points[k] - cur); out[n * 3 + 2] = tuples[(ts - 1) * 4];
    }
    n++; cur = end_points[k];
  }
  return n;
}

ORC_API int32_t orc_version() { return 1; }

// ---------------------------------------------------------------------------
// N3  fMLLR estimation — reference: CalcFmllrFunction (MFA/corpus/features.py:460-548; options :759-766: update type
// "full", silence weight 0) between the two alignment passes (MFA/alignment/base.py:510-539).  Restates Kaldi
// transform/fmllr-diag-gmm.cc: FmllrDiagGmmAccs::AccumulateForGmm / AccumulateFromPosteriors / CommitSingleFrameStats
// and ComputeFmllrMatrixDiagGmmFull + FmllrInnerUpdate + FmllrAuxFuncDiagGmm (min_count 500, num_iters 40).
// Statistics: beta (count), K [D][D+1], G [D][(D+1)x(D+1)] (full symmetric storage here), all double.
// ---------------------------------------------------------------------------
// One utterance: frame t is aligned to pdf ali_pdf[t] with weight w[t] (0 for silence frames when silence_weight = 0).
// Two-model form (the one the reference runs for every model that ships final.alimdl — MFA/corpus/features.py:503-511:
// FmllrComputer(ali_model_path, model_path, ...)): Gaussian posteriors come from the ALIGNMENT model (gconsts,
// means_invvars, inv_vars) evaluated on the speaker-independent features, the statistics a, b are formed with the FINAL
// model's means and variances (stat_means_invvars, stat_inv_vars) — Kaldi gmm-post-to-gpost with the alignment model
// followed by FmllrDiagGmmAccs::AccumulateFromPosteriors on the final one; the two models share their Gaussian layout.
// stat_* == NULL is the single-model form.
ORC_API void orc_fmllr_acc2(const float *feats, int32_t T, int32_t D, const int32_t *ali_pdf, const float *weight,
                            const float *gconsts, const float *means_invvars, const float *inv_vars,
                            const float *stat_means_invvars, const float *stat_inv_vars,
                            const int32_t *pdf_offsets, double *beta, double *K /*[D][D+1]*/, double *G /*[D][D+1][D+1]*/);
ORC_API void orc_fmllr_acc(const float *feats, int32_t T, int32_t D, const int32_t *ali_pdf, const float *weight,
                           const float *gconsts, const float *means_invvars, const float *inv_vars,
                           const int32_t *pdf_offsets, double *beta, double *K /*[D][D+1]*/, double *G /*[D][D+1][D+1]*/) {
  orc_fmllr_acc2(feats, T, D, ali_pdf, weight, gconsts, means_invvars, inv_vars, nullptr, nullptr, pdf_offsets, beta, K, G);
}
ORC_API void orc_fmllr_acc2(const float *feats, int32_t T, int32_t D, const int32_t *ali_pdf, const float *weight,
                            const float *gconsts, const float *means_invvars, const float *inv_vars,
                            const float *stat_means_invvars, const float *stat_inv_vars,
                            const int32_t *pdf_offsets, double *beta, double *K /*[D][D+1]*/, double *G /*[D][D+1][D+1]*/) {
  const float *smi = stat_means_invvars ? stat_means_invvars : means_invvars;
  const float *siv = stat_inv_vars ? stat_inv_vars : inv_vars;
  const int D1 = D + 1;
  std::vector<float> x2(D), ll, a(D), b(D);
  std::vector<double> xplus(D1);
  for (int t = 0; t < T; t++) {
    if (weight[t] == 0.0f) continue;
    const float *x = feats + (size_t)t * D;
    for (int d = 0; d < D; d++) x2[d] = x[d] * x[d];
    int p = ali_pdf[t], g0 = pdf_offsets[p], g1 = pdf_offsets[p + 1], n = g1 - g0;
    ll.resize(n);
    for (int g = 0; g < n; g++)
      ll[g] = gauss_ll(x, x2.data(), D, gconsts[g0 + g], means_invvars + (size_t)(g0 + g) * D, inv_vars + (size_t)(g0 + g) * D);
    // ComponentPosteriors: ApplySoftMax in float, then scale by the frame weight
    float mx = ll[0];
    for (int g = 1; g < n; g++) mx = std::max(mx, ll[g]);
    float sum = 0.0f;
    for (int g = 0; g < n; g++) { ll[g] = expf(ll[g] - mx); sum += ll[g]; }
    float inv = 1.0f / sum;
    float count = 0.0f;
    for (int d = 0; d < D; d++) { a[d] = 0.0f; b[d] = 0.0f; }
    for (int g = 0; g < n; g++) {
      float post = ll[g] * inv * weight[t];
      count += post;
      const float *mi = smi + (size_t)(g0 + g) * D, *iv = siv + (size_t)(g0 + g) * D;
      for (int d = 0; d < D; d++) { a[d] = fmaf(mi[d], post, a[d]); b[d] = fmaf(iv[d], post, b[d]); }
    }
    if (count == 0.0f) continue;
    for (int d = 0; d < D; d++) xplus[d] = x[d];
    xplus[D] = 1.0;
    *beta += count;
    for (int d = 0; d < D; d++)
      for (int e = 0; e < D1; e++) K[(size_t)d * D1 + e] += (double)a[d] * xplus[e];
    for (int d = 0; d < D; d++) {
      double bd = b[d];
      double *Gd = G + (size_t)d * D1 * D1;
      for (int e = 0; e < D1; e++)
        for (int f = 0; f < D1; f++) Gd[(size_t)e * D1 + f] += bd * xplus[e] * xplus[f];
    }
  }
}

static bool invert_spd(std::vector<double> &m, int n) {  // Gauss-Jordan with partial pivoting, in place
  std::vector<double> inv((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++) if (std::fabs(m[(size_t)r * n + c]) > std::fabs(m[(size_t)piv * n + c])) piv = r;
    if (m[(size_t)piv * n + c] == 0.0) return false;
    if (piv != c) for (int k = 0; k < n; k++) { std::swap(m[(size_t)c * n + k], m[(size_t)piv * n + k]); std::swap(inv[(size_t)c * n + k], inv[(size_t)piv * n + k]); }
    double d = 1.0 / m[(size_t)c * n + c];
    for (int k = 0; k < n; k++) { m[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; r++) if (r != c) {
      double f = m[(size_t)r * n + c];
      if (f != 0.0) for (int k = 0; k < n; k++) { m[(size_t)r * n + k] -= f * m[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
    }
  }
  m = inv;
  return true;
}

static double fmllr_auxf(const std::vector<double> &W, int D, double beta, const double *K, const double *G) {
  const int D1 = D + 1;
  std::vector<double> A((size_t)D * D);
  for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) A[(size_t)i * D + j] = W[(size_t)i * D1 + j];
  // log |det A| by elimination
  double logdet = 0.0;
  for (int c = 0; c < D; c++) {
    int piv = c;
    for (int r = c + 1; r < D; r++) if (std::fabs(A[(size_t)r * D + c]) > std::fabs(A[(size_t)piv * D + c])) piv = r;
    if (piv != c) for (int k = 0; k < D; k++) std::swap(A[(size_t)c * D + k], A[(size_t)piv * D + k]);
    double p = A[(size_t)c * D + c];
    logdet += std::log(std::fabs(p));
    for (int r = c + 1; r < D; r++) { double f = A[(size_t)r * D + c] / p; for (int k = c; k < D; k++) A[(size_t)r * D + k] -= f * A[(size_t)c * D + k]; }
  }
  double obj = beta * logdet;
  for (int i = 0; i < D; i++) for (int e = 0; e < D1; e++) obj += W[(size_t)i * D1 + e] * K[(size_t)i * D1 + e];
  for (int d = 0; d < D; d++) {
    const double *Gd = G + (size_t)d * D1 * D1;
    double q = 0.0;
    for (int e = 0; e < D1; e++) { double r = 0.0; for (int f = 0; f < D1; f++) r += Gd[(size_t)e * D1 + f] * W[(size_t)d * D1 + f]; q += r * W[(size_t)d * D1 + e]; }
    obj -= 0.5 * q;
  }
  return obj;
}

// Returns the auxiliary-function improvement (0 and identity transform when beta < min_count or no improvement).
ORC_API double orc_fmllr_solve(int32_t D, double beta, const double *K, const double *G, int32_t num_iters, double min_count,
                               float *out /*[D][D+1]*/) {
  const int D1 = D + 1;
  std::vector<double> W((size_t)D * D1, 0.0);
  for (int i = 0; i < D; i++) W[(size_t)i * D1 + i] = 1.0;
  auto store = [&](const std::vector<double> &M) { for (size_t i = 0; i < M.size(); i++) out[i] = (float)M[i]; };
  store(W);
  if (beta < min_count) return 0.0;
  std::vector<std::vector<double> > invG(D);
  for (int d = 0; d < D; d++) {
    invG[d].assign(G + (size_t)d * D1 * D1, G + (size_t)(d + 1) * D1 * D1);
    if (!invert_spd(invG[d], D1)) return 0.0;
  }
  std::vector<double> Wn(W);
  double old_obj = fmllr_auxf(W, D, beta, K, G);
  std::vector<double> At((size_t)D * D), cof(D1), cg(D1);
  for (int it = 0; it < num_iters; it++) {
    for (int row = 0; row < D; row++) {
      // cofactor row = row of inverse(A^T)
      for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) At[(size_t)i * D + j] = Wn[(size_t)j * D1 + i];
      std::vector<double> inv(At);
      if (!invert_spd(inv, D)) return 0.0;
      for (int j = 0; j < D; j++) cof[j] = inv[(size_t)row * D + j];
      cof[D] = 0.0;
      const std::vector<double> &iG = invG[row];
      const double *k = K + (size_t)row * D1;
      for (int e = 0; e < D1; e++) { double r = 0.0; for (int f = 0; f < D1; f++) r += iG[(size_t)e * D1 + f] * cof[f]; cg[e] = r; }
      double e1 = 0.0, e2 = 0.0;
      for (int e = 0; e < D1; e++) { e1 += cg[e] * cof[e]; e2 += cg[e] * k[e]; }
      double discr = std::sqrt(e2 * e2 + 4 * e1 * beta);
      double alpha1 = (-e2 + discr) / (2 * e1), alpha2 = (-e2 - discr) / (2 * e1);
      double auxf1 = beta * std::log(std::fabs(alpha1 * e1 + e2)) - 0.5 * alpha1 * alpha1 * e1;
      double auxf2 = beta * std::log(std::fabs(alpha2 * e1 + e2)) - 0.5 * alpha2 * alpha2 * e1;
      double alpha = auxf1 > auxf2 ? alpha1 : alpha2;
      for (int e = 0; e < D1; e++) cof[e] = alpha * cof[e] + k[e];
      for (int e = 0; e < D1; e++) { double r = 0.0; for (int f = 0; f < D1; f++) r += iG[(size_t)e * D1 + f] * cof[f]; Wn[(size_t)row * D1 + e] = r; }
    }
  }
  double new_obj = fmllr_auxf(Wn, D, beta, K, G);
  double impr = new_obj - old_obj;
  if (impr < 0.0 && !(std::fabs(new_obj - old_obj) <= 0.001 * (std::fabs(new_obj) + std::fabs(old_obj)))) return 0.0;
  store(Wn);
  return impr;
}
