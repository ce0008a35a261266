// CMVN statistics and final-feature kernels for gfx950.
//   mfa_cmvn_stats : per-speaker Σx, Σx², count in float64 (Kaldi AccCmvnStats), deterministic order.
//   mfa_feats_batch: ApplyCmvn → Δ+ΔΔ (Kaldi DeltaFeatures order 2 window 2) or splice(±ctx)+LDA(+fMLLR)
//                    (Kaldi SpliceFrames / ApplyAffineTransform; Appendix A.3) — the chain of
//                    MFA/alignment/multiprocessing.py:1287-1304 / MFA/db.py:2101-2136.
// HBM-bound streaming stages: each input row is read once per tile (+halo) and each output row written once.
// -ffp-contract=off: the fmaf() chains below are exactly the oracle's.
#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"

namespace {

constexpr int kTile = 64;       // frames per block
constexpr int kMaxDim = 16;     // base feature dim (13 MFCC; 16 with pitch)
constexpr int kMaxOut = 64;     // LDA rows

// ---- CMVN statistics -------------------------------------------------------------------------------------------
// One block per utterance: thread (stripe s = tid/16, dim d = tid%16) sums frames s, s+16, … in double; the 16
// stripes are then added in stripe order.  Output: per-utterance partial [2][dim+1].
__global__ __launch_bounds__(256) void cmvn_utt_kernel(const float *__restrict__ feats, const int64_t *__restrict__ frame_off,
                                                       int dim, double *__restrict__ partial) {
  __shared__ double sx[16][kMaxDim], sxx[16][kMaxDim];
  const int utt = blockIdx.x, d = threadIdx.x & 15, s = threadIdx.x >> 4;
  const int64_t f0 = frame_off[utt];
  const int T = (int)(frame_off[utt + 1] - f0);
  double a = 0.0, b = 0.0;
  if (d < dim)
    for (int t = s; t < T; t += 16) {
      float x = feats[(f0 + t) * dim + d];
      a += (double)x;
      b += (double)(x * x);
    }
  sx[s][d] = a; sxx[s][d] = b;
  __syncthreads();
  if (s == 0 && d < dim) {
    double ta = 0.0, tb = 0.0;
    for (int k = 0; k < 16; k++) { ta += sx[k][d]; tb += sxx[k][d]; }
    partial[(size_t)utt * 2 * (dim + 1) + d] = ta;
    partial[(size_t)utt * 2 * (dim + 1) + (dim + 1) + d] = tb;
  }
  if (threadIdx.x == 0) {
    partial[(size_t)utt * 2 * (dim + 1) + dim] = (double)T;
    partial[(size_t)utt * 2 * (dim + 1) + (dim + 1) + dim] = 0.0;
  }
}

// One thread per (speaker, stat entry): adds the speaker's utterance partials in list order.
__global__ void cmvn_spk_kernel(const double *__restrict__ partial, const int32_t *__restrict__ spk_utt_off,
                                const int32_t *__restrict__ spk_utt, int n_spk, int dim, double *__restrict__ stats) {
  const int width = 2 * (dim + 1);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_spk * width) return;
  const int spk = idx / width, e = idx % width;
  double acc = 0.0;
  for (int k = spk_utt_off[spk]; k < spk_utt_off[spk + 1]; k++) acc += partial[(size_t)spk_utt[k] * width + e];
  stats[idx] = acc;
}

// ---- final features --------------------------------------------------------------------------------------------
struct FeatParams {
  int dim, mode, ctx, lda_rows, lda_cols;
  const float *mfcc; const int64_t *frame_off; const int32_t *utt2spk; const double *cmvn;
  const float *lda; const float *fmllr; float *out;
};

// Kaldi DeltaFeatures scales for order 2, window 2 (float arithmetic of feature-functions.cc: each level is
// Σ_j j·prev(k) scaled by 1/Σj²).
__constant__ float kDelta1[5];
__constant__ float kDelta2[9];

__global__ __launch_bounds__(256) void feats_kernel(FeatParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int utt = blockIdx.y;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int t0 = blockIdx.x * kTile;
  if (t0 >= T) return;
  const int halo = (p.mode == 0) ? 4 : p.ctx;
  const int rows = kTile + 2 * halo;
  float *x = smem;                       // [rows][dim] CMVN-applied base features (frame t0-halo+r, clamped)
  float *mat = x + rows * p.dim;         // mode 1: LDA [lda_rows][lda_cols] then fMLLR [lda_rows][lda_rows+1]
  float *y = mat + (p.mode == 1 ? p.lda_rows * p.lda_cols + p.lda_rows * (p.lda_rows + 1) : 0);  // [kTile][lda_rows]
  // CMVN offsets (Kaldi ApplyCmvn without variance normalisation): offset = (float)(-mean)
  const int spk = p.utt2spk ? p.utt2spk[utt] : 0;
  for (int i = threadIdx.x; i < rows * p.dim; i += blockDim.x) {
    int r = i / p.dim, d = i % p.dim;
    int t = t0 - halo + r;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    float v = p.mfcc[(f0 + t) * p.dim + d];
    if (p.cmvn) {
      const double *st = p.cmvn + (size_t)spk * 2 * (p.dim + 1);
      float offset = (float)(-(st[d] / st[p.dim]));
      v += offset;
    }
    x[i] = v;
  }
  if (p.mode == 1) {
    for (int i = threadIdx.x; i < p.lda_rows * p.lda_cols; i += blockDim.x) mat[i] = p.lda[i];
    if (p.fmllr) {
      const float *fm = p.fmllr + (size_t)spk * p.lda_rows * (p.lda_rows + 1);
      float *fdst = mat + p.lda_rows * p.lda_cols;
      for (int i = threadIdx.x; i < p.lda_rows * (p.lda_rows + 1); i += blockDim.x) fdst[i] = fm[i];
    }
  }
  __syncthreads();
  if (p.mode == 0) {
    const int od = 3 * p.dim;
    for (int i = threadIdx.x; i < kTile * od; i += blockDim.x) {
      int r = i / od, c = i % od, t = t0 + r;
      if (t >= T) continue;
      int level = c / p.dim, d = c % p.dim;
      float acc = 0.0f;
      const float *xc = x + (r + halo) * p.dim + d;  // row of frame t
      if (level == 0) acc = fmaf(1.0f, xc[0], acc);
      else if (level == 1) {
#pragma unroll
        for (int j = -2; j <= 2; j++) {
          int tf = t + j; tf = tf < 0 ? 0 : (tf >= T ? T - 1 : tf);
          float s = kDelta1[j + 2];
          if (s != 0.0f) acc = fmaf(s, xc[(tf - t) * p.dim], acc);
        }
      } else {
#pragma unroll
        for (int j = -4; j <= 4; j++) {
          int tf = t + j; tf = tf < 0 ? 0 : (tf >= T ? T - 1 : tf);
          float s = kDelta2[j + 4];
          if (s != 0.0f) acc = fmaf(s, xc[(tf - t) * p.dim], acc);
        }
      }
      p.out[(f0 + t) * od + c] = acc;
    }
    return;
  }
  // mode 1: splice ±ctx → LDA (fmaf chain over the spliced vector, offset column last) → optional fMLLR
  const int nsp = 2 * p.ctx + 1, sdim = nsp * p.dim;
  const bool lda_offset = (p.lda_cols == sdim + 1);
  for (int i = threadIdx.x; i < kTile * p.lda_rows; i += blockDim.x) {
    int r = i / p.lda_rows, o = i % p.lda_rows, t = t0 + r;
    if (t >= T) continue;
    const float *m = mat + o * p.lda_cols;
    float acc = 0.0f;
    for (int j = 0; j < nsp; j++) {
      int tf = t + j - p.ctx; tf = tf < 0 ? 0 : (tf >= T ? T - 1 : tf);
      const float *xr = x + (tf - t0 + halo) * p.dim;
      for (int d = 0; d < p.dim; d++) acc = fmaf(m[j * p.dim + d], xr[d], acc);
    }
    if (lda_offset) acc += m[sdim];
    if (p.fmllr) y[r * p.lda_rows + o] = acc;
    else p.out[(f0 + t) * p.lda_rows + o] = acc;
  }
  if (!p.fmllr) return;
  __syncthreads();
  const float *fm = mat + p.lda_rows * p.lda_cols;
  for (int i = threadIdx.x; i < kTile * p.lda_rows; i += blockDim.x) {
    int r = i / p.lda_rows, o = i % p.lda_rows, t = t0 + r;
    if (t >= T) continue;
    const float *m = fm + o * (p.lda_rows + 1);
    float acc = 0.0f;
    for (int d = 0; d < p.lda_rows; d++) acc = fmaf(m[d], y[r * p.lda_rows + d], acc);
    acc += m[p.lda_rows];
    p.out[(f0 + t) * p.lda_rows + o] = acc;
  }
}

// splice ±ctx → LDA → optional fMLLR with the matrix rows in REGISTERS.
// Thread (o, g): output row o, frame group g; it keeps row o of the LDA matrix (kSdim spliced inputs) and of the speaker's
// fMLLR matrix (kR inputs) in VGPRs for the whole tile, so a multiply-add costs one LDS read (the frame's input, a
// broadcast to all rows) instead of the generic kernel's two.  The clamped halo rows make the spliced vector of frame r
// the contiguous floats x[r·dim .. r·dim + sdim).  Same fmaf chains, same order, as feats_kernel and the oracle.
// Round-1 measurement: 2.05 ms → 0.6 ms per 2M frames (13 → 40 dims, ±3 splice).
constexpr int kTileLda = 128;   // frames per block of the register-row kernel
// Round 2: one LDS read per multiply-add made the kernel LDS-issue-bound (91 ds_read_b32 per output and frame, the LDS pipe
// shared by the CU's four SIMDs).  The base rows are now padded to 16 floats and the LDA outputs to 16-byte alignment, so a
// spliced vector is read as 7 × (3 × 16 bytes + 4) and the fMLLR input as 10 × 16 bytes: 28 + 10 reads instead of 91 + 40,
// the multiply-add chains in the same order (results unchanged).
template <int kDim, int kCtx, int kR>   // exact base dimension, splice context and output rows: no bounds tests inside the unrolled chains
__global__ __launch_bounds__(256) void feats_lda_kernel(FeatParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kSdim = (2 * kCtx + 1) * kDim;
  constexpr int kXS = (kDim + 3) & ~3;               // row stride of the base-feature tile (16-byte aligned rows)
  static_assert(kR % 4 == 0, "LDA output rows are read 16 bytes at a time");
  const int utt = blockIdx.y;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int t0 = blockIdx.x * kTileLda;
  if (t0 >= T) return;
  constexpr int halo = kCtx, rows = kTileLda + 2 * halo;
  constexpr int R = kR;
  // X: [rows][kXS] CMVN-applied base features (frame t0-halo+r, clamped); later the staging area of the fMLLR matrix
  // Y: staging area of the LDA matrix, then [kTileLda][R] LDA outputs (with fMLLR)
  float *X = smem;
  constexpr int x_floats = rows * kXS > R * (R + 1) ? rows * kXS : R * (R + 1);
  float *Y = X + x_floats;
  const int spk = p.utt2spk ? p.utt2spk[utt] : 0;
  for (int i = threadIdx.x; i < rows * kDim; i += blockDim.x) {
    int r = i / kDim, d = i % kDim;
    int t = t0 - halo + r;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    float v = p.mfcc[(f0 + t) * kDim + d];
    if (p.cmvn) {
      const double *st = p.cmvn + (size_t)spk * 2 * (kDim + 1);
      v += (float)(-(st[d] / st[kDim]));
    }
    X[r * kXS + d] = v;
  }
  for (int i = threadIdx.x; i < R * p.lda_cols; i += blockDim.x) Y[i] = p.lda[i];   // coalesced; rows go to registers below
  __syncthreads();
  constexpr int G = 256 / R;                         // frame groups
  const int o = threadIdx.x % R, g = threadIdx.x / R;
  const bool active = g < G;
  const bool lda_offset = (p.lda_cols == kSdim + 1);
  float m[kSdim];
#pragma unroll
  for (int k = 0; k < kSdim; k++) m[k] = Y[o * p.lda_cols + k];   // spare threads (g == G) read a valid row too
  const float m_off = lda_offset ? Y[o * p.lda_cols + kSdim] : 0.0f;
  __syncthreads();                                   // Y is free for the outputs
  if (active) {
    // two frames per trip (r and r + G): two independent multiply-add chains in flight per thread — each chain is the
    // same 91 dependent fmaf's in the same order as before (results unchanged), but the VALU no longer waits on one
    for (int r = g; r < kTileLda; r += 2 * G) {
      const int t = t0 + r;
      if (t >= T) break;
      const int r2 = r + G;
      const bool two = r2 < kTileLda && t0 + r2 < T;
      const float *xa = X + r * kXS;                 // frames t-ctx .. t+ctx: rows r .. r+2·ctx of the tile
      const float *xb = X + (two ? r2 : r) * kXS;
      float acc = 0.0f, acc2 = 0.0f;
#pragma unroll
      for (int j = 0; j < 2 * kCtx + 1; j++) {
        const float *xr = xa + j * kXS, *xr2 = xb + j * kXS;
#pragma unroll
        for (int d4 = 0; d4 + 4 <= kDim; d4 += 4) {
          const float4 x4 = *reinterpret_cast<const float4 *>(xr + d4);
          const float4 y4 = *reinterpret_cast<const float4 *>(xr2 + d4);
          acc = fmaf(m[j * kDim + d4], x4.x, acc);      acc2 = fmaf(m[j * kDim + d4], y4.x, acc2);
          acc = fmaf(m[j * kDim + d4 + 1], x4.y, acc);  acc2 = fmaf(m[j * kDim + d4 + 1], y4.y, acc2);
          acc = fmaf(m[j * kDim + d4 + 2], x4.z, acc);  acc2 = fmaf(m[j * kDim + d4 + 2], y4.z, acc2);
          acc = fmaf(m[j * kDim + d4 + 3], x4.w, acc);  acc2 = fmaf(m[j * kDim + d4 + 3], y4.w, acc2);
        }
#pragma unroll
        for (int d = kDim & ~3; d < kDim; d++) { acc = fmaf(m[j * kDim + d], xr[d], acc); acc2 = fmaf(m[j * kDim + d], xr2[d], acc2); }
      }
      if (lda_offset) { acc += m_off; acc2 += m_off; }
      if (p.fmllr) { Y[r * R + o] = acc; if (two) Y[r2 * R + o] = acc2; }
      else { p.out[(f0 + t) * R + o] = acc; if (two) p.out[(f0 + t0 + r2) * R + o] = acc2; }
    }
  }
  if (!p.fmllr) return;
  __syncthreads();                                   // X is dead, Y complete
  const float *fm = p.fmllr + (size_t)spk * R * (R + 1);
  for (int i = threadIdx.x; i < R * (R + 1); i += blockDim.x) X[i] = fm[i];
  __syncthreads();
  if (!active) return;
  float f[kR];
#pragma unroll
  for (int k = 0; k < kR; k++) f[k] = X[o * (R + 1) + k];
  const float f_off = X[o * (R + 1) + R];
  for (int r = g; r < kTileLda; r += 2 * G) {        // two frames per trip, as above
    const int t = t0 + r;
    if (t >= T) break;
    const int r2 = r + G;
    const bool two = r2 < kTileLda && t0 + r2 < T;
    const float *ys = Y + r * R, *ys2 = Y + (two ? r2 : r) * R;
    float acc = 0.0f, acc2 = 0.0f;
#pragma unroll
    for (int k = 0; k < kR; k += 4) {
      const float4 y4 = *reinterpret_cast<const float4 *>(ys + k);
      const float4 z4 = *reinterpret_cast<const float4 *>(ys2 + k);
      acc = fmaf(f[k], y4.x, acc);          acc2 = fmaf(f[k], z4.x, acc2);
      acc = fmaf(f[k + 1], y4.y, acc);      acc2 = fmaf(f[k + 1], z4.y, acc2);
      acc = fmaf(f[k + 2], y4.z, acc);      acc2 = fmaf(f[k + 2], z4.z, acc2);
      acc = fmaf(f[k + 3], y4.w, acc);      acc2 = fmaf(f[k + 3], z4.w, acc2);
    }
    acc += f_off; acc2 += f_off;
    p.out[(f0 + t) * R + o] = acc;
    if (two) p.out[(f0 + t0 + r2) * R + o] = acc2;
  }
}

bool g_delta_uploaded = false;

int upload_delta_scales(mfa_ctx *c) {
  // Kaldi DeltaFeatures::DeltaFeatures, order 2, window 2, float arithmetic
  float s0[1] = {1.0f}, s1[5] = {0}, s2[9] = {0};
  {
    float normalizer = 0.0f;
    for (int j = -2; j <= 2; j++) { normalizer += j * j; s1[j + 2] += (float)j * s0[0]; }
    float sc = (float)(1.0 / normalizer);
    for (float &v : s1) v *= sc;
  }
  {
    float normalizer = 0.0f;
    for (int j = -2; j <= 2; j++) {
      normalizer += j * j;
      for (int k = -2; k <= 2; k++) s2[j + k + 4] += (float)j * s1[k + 2];
    }
    float sc = (float)(1.0 / normalizer);
    for (float &v : s2) v *= sc;
  }
  MFA_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(kDelta1), s1, sizeof(s1)));
  MFA_HIP_CHECK(c, hipMemcpyToSymbol(HIP_SYMBOL(kDelta2), s2, sizeof(s2)));
  return 0;
}

}  // namespace

extern "C" {

MFA_API int mfa_cmvn_stats(mfa_ctx *c, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt, int32_t dim,
                           const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_stats) {
  if (dim > kMaxDim) return c->fail("CMVN: feature dim %d > %d", dim, kMaxDim);
  if (n_utt <= 0 || n_spk <= 0) return 0;
  size_t need = (size_t)n_utt * 2 * (dim + 1) * sizeof(double);
  if (c->ws_bytes < need) {
    if (c->d_ws) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_ws); c->d_ws = nullptr; c->ws_bytes = 0; }
    MFA_HIP_CHECK(c, hipMalloc(&c->d_ws, need));
    c->ws_bytes = need;
  }
  KernelTimer kt(c, MFA_K_CMVN);
  hipLaunchKernelGGL(cmvn_utt_kernel, dim3(n_utt), dim3(256), 0, c->stream, d_feats, d_frame_off, dim, (double *)c->d_ws);
  int total = n_spk * 2 * (dim + 1);
  hipLaunchKernelGGL(cmvn_spk_kernel, dim3((total + 255) / 256), dim3(256), 0, c->stream, (const double *)c->d_ws,
                     d_spk_utt_off, d_spk_utt, n_spk, dim, d_stats);
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

MFA_API int mfa_feats_batch(mfa_ctx *c, const float *d_mfcc, const int64_t *d_frame_off, int32_t n_utt, int32_t max_frames,
                            int32_t dim, const int32_t *d_utt2spk, const double *d_cmvn, int32_t mode, int32_t splice_ctx,
                            const float *d_lda, int32_t lda_rows, int32_t lda_cols, const float *d_fmllr, float *d_out) {
  if (dim > kMaxDim) return c->fail("feats: base dim %d > %d", dim, kMaxDim);
  if (n_utt <= 0 || max_frames <= 0) return 0;
  if (n_utt > 65535) return c->fail("at most 65535 utterances per feature launch (got %d)", n_utt);
  if (mode != 0 && mode != 1) return c->fail("feats: bad mode %d", mode);
  if (d_cmvn && !d_utt2spk) return c->fail("feats: CMVN given without utt2spk");
  if (!g_delta_uploaded) { if (upload_delta_scales(c)) return -1; g_delta_uploaded = true; }
  FeatParams p;
  p.dim = dim; p.mode = mode; p.ctx = splice_ctx; p.lda_rows = lda_rows; p.lda_cols = lda_cols;
  p.mfcc = d_mfcc; p.frame_off = d_frame_off; p.utt2spk = d_utt2spk; p.cmvn = d_cmvn; p.lda = d_lda; p.fmllr = d_fmllr;
  p.out = d_out;
  size_t lds = 0;
  if (mode == 0) {
    lds = (size_t)(kTile + 8) * dim * 4;
  } else {
    int sdim = (2 * splice_ctx + 1) * dim;
    if (!d_lda || lda_rows <= 0 || lda_rows > kMaxOut || (lda_cols != sdim && lda_cols != sdim + 1))
      return c->fail("feats: LDA %dx%d does not match spliced dim %d", lda_rows, lda_cols, sdim);
    lds = ((size_t)(kTile + 2 * splice_ctx) * dim + (size_t)lda_rows * lda_cols + (size_t)lda_rows * (lda_rows + 1) +
           (size_t)kTile * lda_rows) * 4;
  }
  dim3 grid((max_frames + kTile - 1) / kTile, n_utt);
  KernelTimer kt(c, MFA_K_FEATS);
  // register-row kernel for MFA's standard shape (13 MFCCs spliced ±3 → 91, LDA to 40); other shapes take the generic kernel
  constexpr int kDim = 13, kCtx = 3, kR = 40;
  const char *generic = getenv("MFA_FEATS_GENERIC");
  if (mode == 1 && dim == kDim && splice_ctx == kCtx && lda_rows == kR && !(generic && generic[0] == '1')) {
    const int rows = kTileLda + 2 * splice_ctx;
    size_t x_floats = std::max((size_t)rows * ((kDim + 3) & ~3), (size_t)lda_rows * (lda_rows + 1));
    size_t y_floats = std::max((size_t)kTileLda * lda_rows, (size_t)lda_rows * lda_cols);
    dim3 grid2((max_frames + kTileLda - 1) / kTileLda, n_utt);
    hipLaunchKernelGGL((feats_lda_kernel<kDim, kCtx, kR>), grid2, dim3(256), (x_floats + y_floats) * 4, c->stream, p);
  } else {
    hipLaunchKernelGGL(feats_kernel, grid, dim3(256), lds, c->stream, p);
  }
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
