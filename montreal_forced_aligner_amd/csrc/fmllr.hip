// fMLLR sufficient statistics on gfx950 (SURVEY "next" row N3).
// Replaces the accumulation half of CalcFmllrFunction (MFA/corpus/features.py:506-527; Kaldi
// FmllrDiagGmmAccs::AccumulateForGmm / CommitSingleFrameStats): per frame the posteriors of the aligned pdf's Gaussians
// give a = Σ_m γ_m·means_invvars_m and b = Σ_m γ_m·inv_vars_m (float32, as Kaldi's single-frame stats); per speaker
//   β = Σ count,  K = Σ_t a_t ξ_tᵀ,  G_d = Σ_t b_t[d] ξ_t ξ_tᵀ   (ξ = [x; 1], float64, fixed summation order).
// The per-speaker solve is host-side (fmllr.py).
//
// Kernel 1 (one wavefront per frame): lane = Gaussian for the log-likelihood fmaf chain and the softmax, then lane =
// dimension for a/b (ascending Gaussian order — the oracle's order).  Kernel 2 (block per speaker × group of 8 output
// rows d): streams the speaker's frames through LDS in chunks of 64 and keeps 7 (e,f) pairs × 8 rows of G in registers
// (float64 FMA); deterministic, no atomics.  Bound: f64 vector FMA (≈72 k DFMA per frame).
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int kMaxD = 48;
constexpr int kChunk = 64;

struct FmllrFrameParams {
  int D, kpad;
  const float *w; const float *gc; const int32_t *row0; const int32_t *nrows;  // packed model (gmm.hip layout)
  const float *ws;   // packed rows of the model the statistics are formed with (two-model form) — same layout; == w otherwise
  const float *feats; const int32_t *ali_pdf; const float *weight; int64_t total_frames;
  float *a; float *b; float *cnt;
};

// logical k of a packed row is stored at 8m + 4(o&1) + (o>>1) with k = 8m + o

__global__ __launch_bounds__(256) void fmllr_frame_kernel(FmllrFrameParams p) {
  __shared__ float post_s[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (t >= p.total_frames) return;
  float *post = post_s[wave];
  const float wgt = p.weight[t];
  const int pdf = p.ali_pdf[t];
  float *a_out = p.a + t * p.D, *b_out = p.b + t * p.D;
  if (wgt == 0.0f || pdf < 0) {
    if (lane < p.D) { a_out[lane] = 0.0f; b_out[lane] = 0.0f; }
    if (lane == 0) p.cnt[t] = 0.0f;
    return;
  }
  const float *x = p.feats + t * p.D;
  const int r0 = p.row0[pdf], n = p.nrows[pdf];
  // log-likelihood of every Gaussian (pad rows: gconst -1e30 → posterior 0); n ≤ 128 handled in two rounds of 64
  float mx = -INFINITY;
  float ll[2];
#pragma unroll
  for (int rnd = 0; rnd < 2; rnd++) {
    const int g = lane + 64 * rnd;
    float acc = -INFINITY;
    if (g < n) {
      acc = p.gc[r0 + g];
      for (int k = 0; k < p.D; k++) acc = fmaf(p.w[mfa_packed_offset(r0 + g, k, p.kpad)], x[k], acc);
      for (int k = 0; k < p.D; k++) { float xv = x[k]; acc = fmaf(p.w[mfa_packed_offset(r0 + g, p.D + k, p.kpad)], xv * xv, acc); }
    }
    ll[rnd] = acc;
    mx = fmaxf(mx, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e0 = lane < n ? expf(ll[0] - mx) : 0.0f, e1 = lane + 64 < n ? expf(ll[1] - mx) : 0.0f;
  float sum = e0 + e1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  post[lane] = e0 * inv * wgt;
  post[lane + 64] = e1 * inv * wgt;
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float count = 0.0f;
  for (int g = 0; g < n; g++) count += ((volatile float *)post)[g];
  if (lane < p.D) {
    float a = 0.0f, b = 0.0f;
    for (int g = 0; g < n; g++) {
      const float pg = ((volatile float *)post)[g];
      a = fmaf(p.ws[mfa_packed_offset(r0 + g, lane, p.kpad)], pg, a);
      b = fmaf(-2.0f * p.ws[mfa_packed_offset(r0 + g, p.D + lane, p.kpad)], pg, b);  // stored −½·inv_var
    }
    a_out[lane] = a; b_out[lane] = b;
  }
  if (lane == 0) p.cnt[t] = count;
}

struct FmllrSpkParams {
  int D;
  const float *feats; const float *a; const float *b; const float *cnt; const int64_t *frame_off;
  const int32_t *spk_utt_off; const int32_t *spk_utt;
  double *beta; double *K; double *G;
};

// grid (speaker, row group of 8); 256 threads.  Thread owns (e,f) pairs p = tid + 256 i (i < 7) of the (D+1)² matrix.
__global__ __launch_bounds__(256) void fmllr_spk_kernel(FmllrSpkParams p) {
  __shared__ float xi[kChunk][kMaxD + 1];
  __shared__ float bs[kChunk][8], as[kChunk][8];
  const int spk = blockIdx.x, d0 = blockIdx.y * 8;
  const int D = p.D, D1 = D + 1, npairs = D1 * D1;
  double g[7][8];
  int pe[7], pf[7], ke[2], kr[2];  // (e,f) of this thread's pairs and (row, column) of its K entries: hoisted divisions
#pragma unroll
  for (int i = 0; i < 7; i++) { const int pr = threadIdx.x + 256 * i; pe[i] = pr < npairs ? pr / D1 : 0; pf[i] = pr < npairs ? pr % D1 : 0; }
#pragma unroll
  for (int i = 0; i < 2; i++) { const int q = threadIdx.x + 256 * i; kr[i] = q < 8 * D1 ? q / D1 : 0; ke[i] = q < 8 * D1 ? q % D1 : 0; }
  double kacc[2] = {0.0, 0.0};  // K entries: thread tid < 8*D1 → (row d0 + tid / D1, column tid % D1); second for tid+256
  double beta = 0.0;
#pragma unroll
  for (int i = 0; i < 7; i++)
#pragma unroll
    for (int r = 0; r < 8; r++) g[i][r] = 0.0;
  for (int u = p.spk_utt_off[spk]; u < p.spk_utt_off[spk + 1]; u++) {
    const int utt = p.spk_utt[u];
    const int64_t f0 = p.frame_off[utt];
    const int T = (int)(p.frame_off[utt + 1] - f0);
    for (int c0 = 0; c0 < T; c0 += kChunk) {
      const int nc = min(kChunk, T - c0);
      __syncthreads();
      for (int i = threadIdx.x; i < nc * D1; i += 256) {
        int t = i / D1, e = i % D1;
        xi[t][e] = e < D ? p.feats[(f0 + c0 + t) * D + e] : 1.0f;
      }
      for (int i = threadIdx.x; i < nc * 8; i += 256) {
        int t = i >> 3, r = i & 7;
        bool ok = d0 + r < D;
        bs[t][r] = ok ? p.b[(f0 + c0 + t) * D + d0 + r] : 0.0f;
        as[t][r] = ok ? p.a[(f0 + c0 + t) * D + d0 + r] : 0.0f;
      }
      __syncthreads();
      for (int t = 0; t < nc; t++) {
        if (blockIdx.y == 0 && threadIdx.x == 0) beta += (double)p.cnt[f0 + c0 + t];
#pragma unroll
        for (int i = 0; i < 7; i++) {
          const int pr = threadIdx.x + 256 * i;
          if (pr < npairs) {
            const double ef = (double)xi[t][pe[i]] * (double)xi[t][pf[i]];
#pragma unroll
            for (int r = 0; r < 8; r++) g[i][r] = fma((double)bs[t][r], ef, g[i][r]);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const int q = threadIdx.x + 256 * i;
          if (q < 8 * D1) kacc[i] = fma((double)as[t][kr[i]], (double)xi[t][ke[i]], kacc[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const int pr = threadIdx.x + 256 * i;
    if (pr < npairs)
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (d0 + r < D) p.G[((size_t)spk * D + d0 + r) * npairs + pr] = g[i][r];
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int q = threadIdx.x + 256 * i;
    if (q < 8 * D1 && d0 + q / D1 < D) p.K[((size_t)spk * D + d0 + q / D1) * D1 + q % D1] = kacc[i];
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) p.beta[spk] = beta;
}

}  // namespace

extern "C" {

MFA_API int mfa_fmllr_acc_batch(mfa_ctx *c, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                int64_t total_frames, const int32_t *d_ali_pdf, const float *d_weight,
                                const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_beta,
                                double *d_K, double *d_G) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  const int D = c->dim;
  if (D > kMaxD) return c->fail("fMLLR statistics: feature dim %d > %d", D, kMaxD);
  if ((D + 1) * (D + 1) > 7 * 256 || 8 * (D + 1) > 512) return c->fail("fMLLR statistics: dim %d too large for the tiling", D);
  if (n_utt <= 0 || n_spk <= 0 || total_frames <= 0) return 0;
  size_t need = (size_t)total_frames * (2 * D + 1) * sizeof(float);
  if (c->ws_bytes < need) {
    if (c->d_ws) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_ws); c->d_ws = nullptr; c->ws_bytes = 0; }
    MFA_HIP_CHECK(c, hipMalloc(&c->d_ws, need));
    c->ws_bytes = need;
  }
  float *a = (float *)c->d_ws, *b = a + (size_t)total_frames * D, *cnt = b + (size_t)total_frames * D;
  // rows per pdf (slot, or 32·nblk): derive on the fly from the slot/nblk tables kept by mfa_load_gmm
  if (!c->d_nrows) {
    std::vector<int32_t> nrows(c->num_pdfs);
    for (int p = 0; p < c->num_pdfs; p++) nrows[p] = c->h_slot[p] == 32 ? 32 * c->h_nblk[p] : c->h_slot[p];
    for (int p = 0; p < c->num_pdfs; p++)
      if (nrows[p] > 128) return c->fail("fMLLR statistics: pdf %d has more than 128 Gaussians", p);
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_nrows, nrows.size() * 4));
    MFA_HIP_CHECK(c, hipMemcpy(c->d_nrows, nrows.data(), nrows.size() * 4, hipMemcpyHostToDevice));
  }
  FmllrFrameParams fp{D, c->kpad, c->d_w, c->d_gc, c->d_row0, c->d_nrows, c->d_w_stats ? c->d_w_stats : c->d_w,
                      d_feats, d_ali_pdf, d_weight, total_frames, a, b, cnt};
  hipLaunchKernelGGL(fmllr_frame_kernel, dim3((unsigned)((total_frames + 3) / 4)), dim3(256), 0, c->stream, fp);
  FmllrSpkParams sp{D, d_feats, a, b, cnt, d_frame_off, d_spk_utt_off, d_spk_utt, d_beta, d_K, d_G};
  hipLaunchKernelGGL(fmllr_spk_kernel, dim3(n_spk, (D + 7) / 8), dim3(256), 0, c->stream, sp);
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

// alignment (transition-ids) → per-frame pdf and weight through the host-built tables (0 / unknown ids: weight 0)
__global__ void fmllr_tid_lookup_kernel(const int32_t *ali, int64_t n, const int32_t *tid2pdf, const float *tid_weight,
                                        int32_t n_tids, int32_t *pdf, float *weight) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int tid = ali[t];
  const bool ok = tid > 0 && tid < n_tids;
  pdf[t] = ok ? tid2pdf[tid] : -1;
  weight[t] = ok ? tid_weight[tid] : 0.0f;
}

MFA_API int mfa_fmllr_acc_ali_batch(mfa_ctx *c, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                    int64_t total_frames, const int32_t *d_ali, const int32_t *d_tid2pdf,
                                    const float *d_tid_weight, int32_t n_tids, int32_t *d_pdf_scratch, float *d_weight_scratch,
                                    const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_beta,
                                    double *d_K, double *d_G) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  if (n_utt <= 0 || n_spk <= 0 || total_frames <= 0) return 0;
  hipLaunchKernelGGL(fmllr_tid_lookup_kernel, dim3((unsigned)((total_frames + 255) / 256)), dim3(256), 0, c->stream, d_ali,
                     total_frames, d_tid2pdf, d_tid_weight, n_tids, d_pdf_scratch, d_weight_scratch);
  MFA_HIP_CHECK(c, hipGetLastError());
  return mfa_fmllr_acc_batch(c, d_feats, d_frame_off, n_utt, total_frames, d_pdf_scratch, d_weight_scratch, d_spk_utt_off,
                             d_spk_utt, n_spk, d_beta, d_K, d_G);
}

MFA_API int mfa_fmllr_stats_model(mfa_ctx *c, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets,
                                  const float *h_means_invvars, const float *h_inv_vars) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  if (c->d_w_stats) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_w_stats); c->d_w_stats = nullptr; }
  if (!h_means_invvars) return 0;   // back to the single-model form
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  if (dim != c->dim || num_pdfs != c->num_pdfs) return c->fail("fMLLR statistics model: %d pdfs of dim %d, loaded model has %d of dim %d", num_pdfs, dim, c->num_pdfs, c->dim);
  for (int p = 0; p < num_pdfs; p++)
    if (h_pdf_offsets[p + 1] - h_pdf_offsets[p] != c->h_ngauss[p])
      return c->fail("fMLLR statistics model: pdf %d has %d Gaussians, the loaded (alignment) model %d — the two models must share their Gaussian layout", p, h_pdf_offsets[p + 1] - h_pdf_offsets[p], c->h_ngauss[p]);
  const int blocks = (c->num_rows + 1 + 31) / 32;
  std::vector<float> w((size_t)blocks * 32 * c->kpad, 0.0f);
  for (int p = 0; p < num_pdfs; p++) {
    const int g0 = h_pdf_offsets[p], g = h_pdf_offsets[p + 1] - g0;
    for (int i = 0; i < g; i++) {
      const float *mi = h_means_invvars + (size_t)(g0 + i) * dim, *iv = h_inv_vars + (size_t)(g0 + i) * dim;
      for (int k = 0; k < 2 * dim; k++)
        w[mfa_packed_offset(c->h_row0[p] + i, k, c->kpad)] = k < dim ? mi[k] : -0.5f * iv[k - dim];
    }
  }
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_w_stats, w.size() * 4));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_w_stats, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

}  // extern "C"
