// Diagonal-GMM acoustic scoring on gfx950 with exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
// Replaces DecodableAmDiagGmmScaled::LogLikelihood / gmm_compute_likes (MFA/alignment/multiprocessing.py:846-853, :1415;
// Kaldi gmm/decodable-am-diag-gmm.cc, VectorBase<float>::LogSumExp).
//
// Per Gaussian:  ll = gconst + Σ_d means_invvars[d]·x[d] + Σ_d (−½ inv_vars[d])·x[d]²   — a [rows × 2D]·[2D × frames]
// contraction.  The MFMA accumulates a k-ordered fmaf chain starting from C = gconst, i.e. bit for bit the oracle's chain.
// Per pdf:       LL = max + log Σ_{ll ≥ max+ln ε} exp(ll − max)  (exp in f32, sum and log in f64, as Kaldi).
//
// Packed model (built once in mfa_load_gmm): every pdf owns `slot` consecutive rows of W[rows][kpad]
// (slot ∈ {1,4,8,16,32·n}; pad rows have zero weights and gconst −1e30 so they fall under the cutoff).  Within each group
// of 8 k-values a row is stored permuted, stored[8m+4h+c] = logical[8m+2c+h], so that one 16-byte load per lane yields the
// A operands of four consecutive MFMA steps (lane l feeds A[row l&31][k = 2s + (l>>5)]).
// One 32-row MFMA block then serves 32/slot pdfs of the utterance's (slot-sorted) pdf list; rows ↔ accumulator registers:
// row = (r&3) + 8(r>>2) + 4(l>>5), so 4-row slots reduce inside a lane and 8/16/32-row slots add one cross-half shuffle.
//
// Work decomposition: grid (frame tiles of 256, utterances); a wavefront owns NT×32 frames, keeps their x̃ = [x, x²]
// operands in registers (the B side) and streams the utterance's model rows (the A side, L2/MALL-resident).
// Output: [T][P_u] row-major, the layout the Viterbi kernel gathers from.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ctx.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a wavefront owns NT 32-frame tiles; a workgroup is kWaves wavefronts (template parameter)
constexpr float kPadGconst = -1.0e30f;

struct GmmParams {
  int dim, kpad, num_rows;  // num_rows = index of the dummy row
  const float *w; const float *gc; const int32_t *row0; const int32_t *nblk; const int32_t *slot;
  const float *feats; const int64_t *frame_off;
  const int32_t *pdf_list; const int64_t *pdf_off; const int32_t *class_counts; const int64_t *ll_off;
  float *out;
  float min_log_diff;  // logf(FLT_EPSILON), computed on the host so device and oracle use the same constant
  int skip_single;     // >0: that many leading single-block 32-row pdfs are handled by gmm_sp_kernel
  int n_utt, tiles;    // generic kernel: 1-D grid of 8·ceil(n_utt/8)·tiles workgroups
  int prio;            // MFA_GMM_PRIO: s_setprio level for the second half of a 512-thread workgroup (0 = none)
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask); hi = __shfl_xor(hi, mask);
  return __hiloint2double(hi, lo);
}

// row index (within a 32-row MFMA block) held by accumulator register r of a lane in half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// partner lane's value across the two 32-lane halves: one v_permlane32_swap instead of a round trip through the LDS
// crossbar (ds_bpermute)
__device__ __forceinline__ float swap32(float v, int h) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(h ? r[0] : r[1]);
#else
  return __shfl_xor(v, 32);
#endif
}
__device__ __forceinline__ double swap32_f64(double v, int h) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)(h ? b[0] : b[1]), (int)(h ? a[0] : a[1]));
#else
  return shfl_xor_f64(v, 32);
#endif
}

template <int M8, int kNT>
struct Tile {
  // One wavefront: B operands for kNT frame tiles, generic block evaluation.
  float b[kNT][4 * M8];

  __device__ __forceinline__ void load_b(const GmmParams &p, int64_t f0, int T, int t_base, int lane) {
    const int col = lane & 31, h = lane >> 5;
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      int t = t_base + 32 * n + col;
      t = t < T ? t : T - 1;
      const float *x = p.feats + (f0 + t) * p.dim;
      // branch-free (independent loads, one round trip): clamp the index, then select x, x² or the zero pad
#pragma unroll
      for (int s = 0; s < 4 * M8; s++) {
        const int k = 2 * s + h;
        const int idx = k < p.dim ? k : (k < 2 * p.dim ? k - p.dim : 0);
        const float xv = x[idx];
        b[n][s] = k < p.dim ? xv : (k < 2 * p.dim ? xv * xv : 0.0f);
      }
    }
  }

  // acc[n] = gconst(rows) + W(block rows) · x̃(tile n).  arow: this lane's A row (already offset by 4h floats);
  // gcv: gconst of the row this lane (lane&31) addresses.
  __device__ __forceinline__ static void load_a(const float *arow, f32x4 (&a)[M8]) {
#pragma unroll
    for (int m = 0; m < M8; m++) a[m] = *reinterpret_cast<const f32x4 *>(arow + 8 * m);
  }
  __device__ __forceinline__ void block(const float *arow, float gcv, int lane, f32x16 (&acc)[kNT]) const {
    f32x4 a[M8];
    load_a(arow, a);
    run(a, gcv, lane, acc);
  }
  // gconst of the accumulator rows of a contiguous, 4-row-aligned 32-row block: four 16-byte loads (rows 8q+4h..+3)
  __device__ __forceinline__ static void load_gc32(const float *gc_block, int h, f32x4 (&g)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = *reinterpret_cast<const f32x4 *>(gc_block + 8 * q + 4 * h);
  }
  __device__ __forceinline__ void run32(const f32x4 (&a)[M8], const f32x4 (&g)[4], f32x16 (&acc)[kNT]) const {
    f32x16 init;
#pragma unroll
    for (int r = 0; r < 16; r++) init[r] = g[r >> 2][r & 3];
    mfma(a, init, acc);
  }
  __device__ __forceinline__ void run(const f32x4 (&a)[M8], float gcv, int lane, f32x16 (&acc)[kNT]) const {
    const int h = lane >> 5;
    f32x16 init;
#pragma unroll
    for (int r = 0; r < 16; r++) init[r] = __shfl(gcv, acc_row(r, h));
    mfma(a, init, acc);
  }
  __device__ __forceinline__ void mfma(const f32x4 (&a)[M8], const f32x16 &init, f32x16 (&acc)[kNT]) const {
#pragma unroll
    for (int n = 0; n < kNT; n++) acc[n] = init;
#pragma unroll
    for (int m = 0; m < M8; m++) {
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
#pragma unroll
        for (int n = 0; n < kNT; n++)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][cc], b[n][4 * m + cc], acc[n], 0, 0, 0);
      }
    }
  }
};

// log-sum-exp pieces (Kaldi LogSumExp semantics)
template <int R0, int R1>
__device__ __forceinline__ float reg_max(const f32x16 &v) {
  float m = v[R0];
#pragma unroll
  for (int r = R0 + 1; r < R1; r++) m = fmaxf(m, v[r]);
  return m;
}
// Σ_r exp(v[r] − mx) over the rows r ∈ [R0, R1) that pass Kaldi's cutoff (v[r] ≥ max + ln ε).
// exp(x) = 2^(x·log2 e) on the hardware exp2 (≈1 ulp); the rounding of the product x·log2 e adds |x|·6e-8 relative
// error to a term, which only matters for terms that are themselves ≤ e^x of the sum — below 1e-7 of the total.
// The ≤16 terms per lane are added in float32 as a balanced tree (error ≲ 4 ulp of the sum, i.e. ≲ 2.5e-7 absolute on
// the log-likelihood — an order of magnitude below half an ulp of a float32 score of magnitude ≥ 16).  Round 1 summed
// in float64 after a 6-instruction exponential: measured, that epilogue cost as much VALU time as the MFMAs it follows.
template <int R0, int R1>
__device__ __forceinline__ float reg_expsum(const f32x16 &v, float mx, float cutoff) {
  constexpr int n = R1 - R0;
  float e[n];
#pragma unroll
  for (int r = 0; r < n; r++) {
    const float t = __builtin_amdgcn_exp2f((v[R0 + r] - mx) * 1.44269504088896341f);
    e[r] = v[R0 + r] >= cutoff ? t : 0.0f;
  }
#pragma unroll
  for (int w = 1; w < n; w <<= 1)
#pragma unroll
    for (int r = 0; r + w < n; r += 2 * w) e[r] += e[r + w];
  return e[0];
}
// LL = max + ln(sum) with the hardware log2 (1 ulp on a value ≤ 7, i.e. ≲4e-7 absolute).
__device__ __forceinline__ float finish(float mx, float sum) {
  return fmaf(__builtin_amdgcn_logf(sum), 0.693147180559945309f, mx);
}

template <int M8, int kNT, int kMinWaves, int kWaves>
__global__ __launch_bounds__(64 * kWaves, kMinWaves) void gmm_kernel(GmmParams p) {
  constexpr int kFramesPerWave = 32 * kNT;
  // 512-thread workgroups put two wavefronts on every SIMD.  Left alone they fall into lockstep (both in their MFMA
  // phase, then both in their log-sum-exp epilogue, matrix pipe idle).  A static priority for the second half lets that
  // wavefront own the matrix pipe while the other fills the pipe during the first one's epilogue
  // (cdna_hip_programming.md T5, static form).
  if (kWaves == 8 && p.prio && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(2);
  // XCD-aware block→(utterance, frame tile) map (cdna_hip_programming.md T1): workgroups are dealt round-robin over the
  // 8 XCDs, each with a private L2.  All frame tiles of one utterance stream the same model rows, so they are given
  // consecutive slots on ONE XCD: the rows are then fetched from HBM/Infinity Cache once per utterance, not once per tile.
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int utt = (seq / p.tiles) * 8 + xcd;
  const int tile_x = seq % p.tiles;
  if (utt >= p.n_utt) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int t_base = (tile_x * kWaves + wave) * kFramesPerWave;
  if (t_base >= T) return;  // wavefronts are independent (no barriers in this kernel)
  const int col = lane & 31, h = lane >> 5;
  const int64_t l0 = p.pdf_off[utt];
  const int P = (int)(p.pdf_off[utt + 1] - l0);
  const int32_t *list = p.pdf_list + l0;
  const int32_t *cc6 = p.class_counts + (size_t)utt * 6;
  // class_counts[u] = {32-row single-block, 32-row multi-block, 16, 8, 4, 1}; with p.skip_single the single-block pdfs
  // are scored by gmm_sp_kernel and this kernel starts after them
  const int first32 = p.skip_single ? min(cc6[0], p.skip_single) : 0;
  const int32_t cc[5] = {cc6[0] + cc6[1], cc6[2], cc6[3], cc6[4], cc6[5]};
  if (first32 == cc[0] && cc[1] + cc[2] + cc[3] + cc[4] == 0) return;  // nothing left for this kernel
  float *out = p.out + p.ll_off[utt];

  Tile<M8, kNT> tile;
  tile.load_b(p, f0, T, t_base, lane);
  f32x16 acc[kNT];
  // per-wavefront output staging tile (volatile: written lane-per-frame, read row-wise by the same wavefront)
  __shared__ float stage_all[kWaves][64 * 33];
  float *stage = stage_all[wave];
  const int n_single = cc6[0];

  // ---- single-block 32-row pdfs (the bulk of a context-dependent model): one pdf per MFMA block.
  // Software pipeline, no extra registers: as soon as the MFMAs that read operand group a[m] of block j have been issued,
  // the same registers are re-loaded with block j+1's rows, so every load has a whole block period (≈5k cycles of MFMA
  // issue plus the epilogue) to come back from L2 / Infinity Cache.  The packed-row lookups (pdf id → first row) run
  // two and three blocks ahead, so they never sit on the critical path.
  // (Round-1 measurements, tools/mfma_f32_microbench2.hip: operands requested just in time 116 TFLOP/s, one block
  // ahead 143 TFLOP/s.)
  if (n_single > first32) {
    const int last = n_single - 1;
    auto pdf_at = [&](int jj) { return __builtin_amdgcn_readfirstlane(list[min(jj, last)]); };
    auto row_of = [&](int pdf) { return __builtin_amdgcn_readfirstlane(p.row0[pdf]); };
    const float *wl = p.w + (size_t)col * p.kpad + 4 * h;  // this lane's row within a block, k offset of its half
    f32x4 a[M8], g[4];
    int r1 = row_of(pdf_at(first32 + 1));
    int pdf2 = pdf_at(first32 + 2);
    {
      const int r0 = row_of(pdf_at(first32));
      // same issue order as inside the loop (gconst rows, then operand groups): the compiler's vmcnt bookkeeping at the
      // loop head is the merge of both paths, and a different order here makes it wait for every outstanding load
      Tile<M8, kNT>::load_gc32(p.gc + r0, h, g);
      __builtin_amdgcn_sched_barrier(0);
      Tile<M8, kNT>::load_a(wl + (size_t)r0 * p.kpad, a);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = first32; j < n_single; j++) {
      {
        f32x16 init;
#pragma unroll
        for (int r = 0; r < 16; r++) init[r] = g[r >> 2][r & 3];
#pragma unroll
        for (int n = 0; n < kNT; n++) acc[n] = init;
      }
      // The lookups are vector loads (the compiler cannot prove the lists are not aliased by `out`), issued first so that
      // they are the oldest entries of the in-order vmcnt queue: reading them back after the MFMA phase then waits for
      // nothing younger.
      const int x_r2 = p.row0[pdf2];
      const int x_pdf3 = list[min(j + 3, last)];
      __builtin_amdgcn_sched_barrier(0);
      const float *wn = wl + (size_t)r1 * p.kpad;
      Tile<M8, kNT>::load_gc32(p.gc + r1, h, g);
#pragma unroll
      for (int m = 0; m < M8; m++) {
#pragma unroll
        for (int cc4 = 0; cc4 < 4; cc4++) {
#pragma unroll
          for (int n = 0; n < kNT; n++)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][cc4], tile.b[n][4 * m + cc4], acc[n], 0, 0, 0);
        }
        a[m] = *reinterpret_cast<const f32x4 *>(wn + 8 * m);
        __builtin_amdgcn_sched_barrier(0);
      }
      r1 = __builtin_amdgcn_readfirstlane(x_r2);
      pdf2 = __builtin_amdgcn_readfirstlane(x_pdf3);
      float mx[kNT], sum[kNT];
#pragma unroll
      for (int n = 0; n < kNT; n++) {
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, swap32(m, h));
        float sv = reg_expsum<0, 16>(acc[n], m, m + p.min_log_diff);
        sv += swap32(sv, h);
        mx[n] = m; sum[n] = sv;
      }
      if constexpr (kNT == 2) {
        // both halves hold every tile's (max, sum): half h finishes tile h (one log per lane).
        // Stage [64 frames][32 pdfs] in LDS and flush whole 128-byte row segments: a lane-per-frame store would touch 64
        // different lines per instruction and (measured, round 1) inflate HBM write traffic 11x with partial lines.
        const float v = finish(h ? mx[1] : mx[0], h ? sum[1] : sum[0]);
        const int jj = (j - first32) & 31;
        stage[(32 * h + col) * 33 + jj] = v;
        if (jj == 31 || j == last) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int j0 = j - jj, cnt = jj + 1;
          const int c = lane & 31;
#pragma unroll 4
          for (int i = 0; i < 32; i++) {
            const int r = (lane >> 5) + 2 * i, t = t_base + r;
            // streaming store: the scores are written once and read once by the decoder; keeping them out of the
            // Infinity Cache leaves room for the 51 MB of model rows every workgroup keeps re-reading
            if (c < cnt && t < T) __builtin_nontemporal_store(stage[r * 33 + c], &out[(size_t)t * P + j0 + c]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      } else {
#pragma unroll
        for (int n = 0; n < kNT; n++) {
          const int t = t_base + 32 * n + col;
          if (h == (n & 1) && t < T) out[(size_t)t * P + j] = finish(mx[n], sum[n]);
        }
      }
    }
  }

  // ---- 32-row pdfs with more than 32 Gaussians: several blocks, two passes (max, then the sum against that max)
  const int n32 = cc[0];
  for (int j = max(first32, n_single); j < n32; j++) {
    const int pdf = list[j];
    const int r0 = p.row0[pdf], nb = p.nblk[pdf];
    float mx[kNT], sum[kNT];
#pragma unroll
    for (int n = 0; n < kNT; n++) { mx[n] = -INFINITY; sum[n] = 0.0f; }
    for (int blk = 0; blk < nb; blk++) {
      const int rr = r0 + 32 * blk + col;
      tile.block(p.w + (size_t)rr * p.kpad + 4 * h, p.gc[rr], lane, acc);
#pragma unroll
      for (int n = 0; n < kNT; n++) {
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, __shfl_xor(m, 32));
        mx[n] = fmaxf(mx[n], m);
      }
    }
    for (int blk = 0; blk < nb; blk++) {
      const int rr = r0 + 32 * blk + col;
      tile.block(p.w + (size_t)rr * p.kpad + 4 * h, p.gc[rr], lane, acc);
#pragma unroll
      for (int n = 0; n < kNT; n++) {
        float sv = reg_expsum<0, 16>(acc[n], mx[n], mx[n] + p.min_log_diff);
        sv += swap32(sv, h);
        sum[n] += sv;
      }
    }
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      const int t = t_base + 32 * n + col;
      if (h == (n & 1) && t < T) out[(size_t)t * P + j] = finish(mx[n], sum[n]);
    }
  }

  // ---- smaller slots: 32/slot pdfs share one MFMA block
  int base = n32;
  // slot 16
  for (int j = 0; j < cc[1]; j += 2) {
    const int which = col >> 4, within = col & 15;
    const int idx = j + which;
    const int row = idx < cc[1] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(p.w + (size_t)row * p.kpad + 4 * h, p.gc[row], lane, acc);
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      int t = t_base + 32 * n + col;
      float m0 = reg_max<0, 8>(acc[n]), m1 = reg_max<8, 16>(acc[n]);
      m0 = fmaxf(m0, __shfl_xor(m0, 32)); m1 = fmaxf(m1, __shfl_xor(m1, 32));
      float s0 = reg_expsum<0, 8>(acc[n], m0, m0 + p.min_log_diff), s1 = reg_expsum<8, 16>(acc[n], m1, m1 + p.min_log_diff);
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
      if (h == 0 && t < T) {
        out[(size_t)t * P + base + j] = finish(m0, s0);
        if (j + 1 < cc[1]) out[(size_t)t * P + base + j + 1] = finish(m1, s1);
      }
    }
  }
  base += cc[1];
  // slot 8
  for (int j = 0; j < cc[2]; j += 4) {
    const int which = col >> 3, within = col & 7;
    const int idx = j + which;
    const int row = idx < cc[2] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(p.w + (size_t)row * p.kpad + 4 * h, p.gc[row], lane, acc);
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      int t = t_base + 32 * n + col;
      float m[4]; float s[4];
      m[0] = reg_max<0, 4>(acc[n]); m[1] = reg_max<4, 8>(acc[n]); m[2] = reg_max<8, 12>(acc[n]); m[3] = reg_max<12, 16>(acc[n]);
#pragma unroll
      for (int q = 0; q < 4; q++) m[q] = fmaxf(m[q], __shfl_xor(m[q], 32));
      s[0] = reg_expsum<0, 4>(acc[n], m[0], m[0] + p.min_log_diff); s[1] = reg_expsum<4, 8>(acc[n], m[1], m[1] + p.min_log_diff);
      s[2] = reg_expsum<8, 12>(acc[n], m[2], m[2] + p.min_log_diff); s[3] = reg_expsum<12, 16>(acc[n], m[3], m[3] + p.min_log_diff);
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] += __shfl_xor(s[q], 32);
      if (h == 0 && t < T) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (j + q < cc[2]) out[(size_t)t * P + base + j + q] = finish(m[q], s[q]);
      }
    }
  }
  base += cc[2];
  // slot 4: rows 8q+4h..8q+4h+3 live in registers 4q..4q+3 of one lane → pdf index 2q+h, no shuffle
  for (int j = 0; j < cc[3]; j += 8) {
    const int which = col >> 2, within = col & 3;
    const int idx = j + which;
    const int row = idx < cc[3] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(p.w + (size_t)row * p.kpad + 4 * h, p.gc[row], lane, acc);
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      int t = t_base + 32 * n + col;
      float m[4]; float s[4];
      m[0] = reg_max<0, 4>(acc[n]); m[1] = reg_max<4, 8>(acc[n]); m[2] = reg_max<8, 12>(acc[n]); m[3] = reg_max<12, 16>(acc[n]);
      s[0] = reg_expsum<0, 4>(acc[n], m[0], m[0] + p.min_log_diff); s[1] = reg_expsum<4, 8>(acc[n], m[1], m[1] + p.min_log_diff);
      s[2] = reg_expsum<8, 12>(acc[n], m[2], m[2] + p.min_log_diff); s[3] = reg_expsum<12, 16>(acc[n], m[3], m[3] + p.min_log_diff);
      if (t < T) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int pi = j + 2 * q + h;
          if (pi < cc[3]) out[(size_t)t * P + base + pi] = finish(m[q], s[q]);
        }
      }
    }
  }
  base += cc[3];
  // slot 1: every row is its own single-Gaussian pdf: LL = ll (max + log(1) exactly)
  for (int j = 0; j < cc[4]; j += 32) {
    const int idx = j + col;
    const int row = idx < cc[4] ? p.row0[list[base + idx]] : p.num_rows;
    tile.block(p.w + (size_t)row * p.kpad + 4 * h, p.gc[row], lane, acc);
#pragma unroll
    for (int n = 0; n < kNT; n++) {
      int t = t_base + 32 * n + col;
      if (t < T) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int pi = j + acc_row(r, h);
          if (pi < cc[4]) out[(size_t)t * P + base + pi] = acc[n][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined scoring of the single-block 32-row pdfs (the bulk of a context-dependent model).
// One wavefront per SIMD (whole 512-register file): it owns 4 frame tiles (128 frames, x̃ in 160 VGPRs) and two
// accumulator sets.  While the matrix pipe works through block j's 160 MFMAs, the same wavefront's VALU stream runs the
// log-sum-exp epilogue of block j-1 from the other accumulator set, and block j+1's model rows are already in flight —
// the pipe never waits for an epilogue phase (two co-resident wavefronts running [MFMA phase | epilogue phase] fall
// into lockstep and leave it idle ≈40 % of the time).
constexpr int kSpMaxBlocks = 2048;
template <int M8>
__global__ __launch_bounds__(256, 1) void gmm_sp_kernel(GmmParams p) {
  constexpr int NT = 4;
  const int utt = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int t_base = (blockIdx.x * 4 + wave) * (32 * NT);
  const int32_t *cc = p.class_counts + (size_t)utt * 6;
  const int n1 = min(cc[0], kSpMaxBlocks);  // the generic kernel takes whatever exceeds the row cache
  const int64_t l0 = p.pdf_off[utt];
  const int P = (int)(p.pdf_off[utt + 1] - l0);
  const int32_t *list = p.pdf_list + l0;
  // first packed row of every block, staged in LDS once per workgroup: the per-block lookups then ride on lgkmcnt and
  // never serialise with the model-row prefetch on vmcnt
  __shared__ int rows_lds[kSpMaxBlocks];
  for (int i = threadIdx.x; i < n1; i += blockDim.x) rows_lds[i] = p.row0[list[i]];
  __syncthreads();
  if (t_base >= T || n1 == 0) return;
  const int col = lane & 31, h = lane >> 5;
  float *out = p.out + p.ll_off[utt];

  Tile<M8, NT> tile;
  tile.load_b(p, f0, T, t_base, lane);

  f32x4 aA[M8], aB[M8], gA[4], gB[4];
  f32x16 accA[NT], accB[NT];
  // first packed row of block jj; fetched two blocks ahead of its use (two dependent loads that must not stall the pipe)
  auto row_of = [&](int jj) { return jj < n1 ? rows_lds[jj] : 0; };
  auto request = [&](int r0, f32x4 (&a)[M8], f32x4 (&g)[4]) {
    r0 = __builtin_amdgcn_readfirstlane(r0);
    Tile<M8, NT>::load_a(p.w + (size_t)(r0 + col) * p.kpad + 4 * h, a);
    Tile<M8, NT>::load_gc32(p.gc + r0, h, g);
  };
  auto epilogue = [&](const f32x16 (&acc)[NT], int j) {
    float mx[NT]; float sum[NT];
#pragma unroll
    for (int n = 0; n < NT; n++) {
      float m = reg_max<0, 16>(acc[n]);
      m = fmaxf(m, swap32(m, h));
      float s = reg_expsum<0, 16>(acc[n], m, m + p.min_log_diff);
      s += swap32(s, h);
      mx[n] = m; sum[n] = s;
    }
    // half h finishes tiles 2i+h
#pragma unroll
    for (int i = 0; i < NT / 2; i++) {
      const float mxs = h ? mx[2 * i + 1] : mx[2 * i];
      const float sums = h ? sum[2 * i + 1] : sum[2 * i];
      const int t = t_base + 32 * (2 * i + h) + col;
      if (t < T) out[(size_t)t * P + j] = finish(mxs, sums);
    }
  };

  // Operand sets are requested one whole block (160 MFMAs) before their first use; the sched_barrier keeps the compiler
  // from sinking the loads down to that use.
  request(row_of(0), aA, gA);
  if (n1 > 1) request(row_of(1), aB, gB);
  int r_next = row_of(2);   // row of block j+1 at the top of each half-trip below
  __builtin_amdgcn_sched_barrier(0);
  tile.run32(aA, gA, accA);  // block 0
  int j = 1;
  // two blocks per trip so the accumulator/operand sets alternate by name (no register copies)
  for (; j + 1 < n1; j += 2) {
    request(r_next, aA, gA);    // block j+1, for the second half of this trip (aA is free: block j-1's MFMAs are issued)
    r_next = row_of(j + 2);
    __builtin_amdgcn_sched_barrier(0);
    tile.run32(aB, gB, accB);   // block j      → accB   ┐ independent MFMA and VALU streams:
    epilogue(accA, j - 1);      // block j-1 from accA   ┘ the scheduler interleaves them
    if (j + 2 < n1) request(r_next, aB, gB);
    r_next = row_of(j + 3);
    __builtin_amdgcn_sched_barrier(0);
    tile.run32(aA, gA, accA);   // block j+1    → accA
    epilogue(accB, j);
  }
  if (j < n1) {
    tile.run32(aB, gB, accB);
    epilogue(accA, j - 1);
    epilogue(accB, j);
  } else {
    epilogue(accA, j - 1);
  }
}

// Straightforward one-thread-per-(frame,pdf) kernel: used for feature dims the MFMA kernel is not instantiated for and,
// with MFA_GMM_NAIVE=1, as an on-device cross-check of the MFMA path.  Same fmaf chain, same log-sum-exp.
__global__ void gmm_naive_kernel(GmmParams p) {
  const int utt = blockIdx.y;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int64_t l0 = p.pdf_off[utt];
  const int P = (int)(p.pdf_off[utt + 1] - l0);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * P) return;
  const int t = (int)(idx / P), j = (int)(idx % P);
  const int pdf = p.pdf_list[l0 + j];
  const int r0 = p.row0[pdf];
  const int rows = p.slot[pdf] == 32 ? 32 * p.nblk[pdf] : p.slot[pdf];  // pad rows carry gconst -1e30
  const float *x = p.feats + (f0 + t) * p.dim;
  float mx = -INFINITY;
  double sum = 0.0;
  for (int pass = 0; pass < 2; pass++) {
    for (int r = 0; r < rows; r++) {
      const float *w = p.w + (size_t)(r0 + r) * p.kpad;
      float acc = p.gc[r0 + r];
      for (int k = 0; k < 2 * p.dim; k++) {
        int m = k >> 3, o = k & 7;  // logical k = 8m + 2c + h is stored at 8m + 4h + c
        float xv = k < p.dim ? x[k] : x[k - p.dim] * x[k - p.dim];
        acc = fmaf(w[8 * m + 4 * (o & 1) + (o >> 1)], xv, acc);
      }
      if (pass == 0) mx = fmaxf(mx, acc);
      else if (acc >= mx + p.min_log_diff) sum += (double)expf(acc - mx);
    }
  }
  p.out[p.ll_off[utt] + (size_t)t * P + j] = (float)((double)mx + log(sum));
}

int slot_of(int g) { return g <= 1 ? 1 : g <= 4 ? 4 : g <= 8 ? 8 : g <= 16 ? 16 : 32; }
int class_index(int slot) { return slot == 32 ? 0 : slot == 16 ? 1 : slot == 8 ? 2 : slot == 4 ? 3 : 4; }

}  // namespace

extern "C" {

MFA_API int mfa_load_gmm(mfa_ctx *c, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets, const float *h_gconsts,
                         const float *h_means_invvars, const float *h_inv_vars) {
  hipSetDevice(c->device);
  if (dim <= 0 || num_pdfs <= 0) return c->fail("mfa_load_gmm: bad dim/num_pdfs %d/%d", dim, num_pdfs);
  // the MFMA kernel is instantiated for rows of exactly 80 or 96 floats; wider models use the naive kernel
  const int kpad = 2 * dim <= 80 ? 80 : (2 * dim <= 96 ? 96 : ((2 * dim + 7) / 8) * 8);
  std::vector<int32_t> row0(num_pdfs + 1), nblk(num_pdfs), slot(num_pdfs);
  for (int p = 0; p < num_pdfs; p++) {
    int g = h_pdf_offsets[p + 1] - h_pdf_offsets[p];
    if (g <= 0) return c->fail("mfa_load_gmm: pdf %d has no Gaussians", p);
    slot[p] = slot_of(g);
    nblk[p] = slot[p] == 32 ? (g + 31) / 32 : 1;
  }
  // rows are handed out class by class (32, 16, 8, 4, 1): every pdf then starts at a multiple of its slot size, so the
  // gconst rows of a 32-row block can be fetched with aligned 16-byte loads
  int rows = 0;
  for (int cls : {32, 16, 8, 4, 1})
    for (int p = 0; p < num_pdfs; p++)
      if (slot[p] == cls) { row0[p] = rows; rows += cls == 32 ? 32 * nblk[p] : cls; }
  rows = (rows + 3) & ~3;
  row0[num_pdfs] = rows;
  std::vector<float> w((size_t)(rows + 1) * kpad, 0.0f), gc(rows + 1, kPadGconst);
  for (int p = 0; p < num_pdfs; p++) {
    int g0 = h_pdf_offsets[p], g = h_pdf_offsets[p + 1] - g0;
    for (int i = 0; i < g; i++) {
      float *dst = w.data() + (size_t)(row0[p] + i) * kpad;
      const float *mi = h_means_invvars + (size_t)(g0 + i) * dim, *iv = h_inv_vars + (size_t)(g0 + i) * dim;
      for (int k = 0; k < 2 * dim; k++) {
        float v = k < dim ? mi[k] : -0.5f * iv[k - dim];
        int m = k >> 3, o = k & 7;
        dst[8 * m + 4 * (o & 1) + (o >> 1)] = v;  // stored[8m+4h+c] = logical[8m+2c+h]
      }
      gc[row0[p] + i] = h_gconsts[g0 + i];
    }
  }
  void *old[] = {c->d_w, c->d_gc, c->d_row0, c->d_nblk, c->d_slot, c->d_nrows};
  for (void *q : old) if (q) hipFree(q);
  c->d_w = nullptr; c->d_gc = nullptr; c->d_row0 = nullptr; c->d_nblk = nullptr; c->d_slot = nullptr; c->d_nrows = nullptr;
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_w, w.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gc, gc.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_row0, row0.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_nblk, nblk.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_slot, slot.size() * 4));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_gc, gc.data(), gc.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_row0, row0.data(), row0.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_nblk, nblk.data(), nblk.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice));
  c->dim = dim; c->kpad = kpad; c->num_pdfs = num_pdfs; c->num_rows = rows;
  c->h_slot = slot;
  c->h_nblk = nblk;
  c->gmm_ready = true;
  return 0;
}

MFA_API int32_t mfa_gmm_slot(mfa_ctx *c, int32_t pdf) {
  if (!c->gmm_ready || pdf < 0 || pdf >= c->num_pdfs) return -1;
  return c->h_slot[pdf];
}

MFA_API int mfa_gmm_sort_pdf_list(mfa_ctx *c, int32_t *h_pdfs, int32_t n, int32_t *h_class_counts) {
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  std::vector<int32_t> bucket[6];
  for (int i = 0; i < n; i++) {
    int p = h_pdfs[i];
    if (p < 0 || p >= c->num_pdfs) return c->fail("pdf id %d out of range [0,%d)", p, c->num_pdfs);
    int ci = class_index(c->h_slot[p]);
    bucket[ci == 0 ? (c->h_nblk[p] == 1 ? 0 : 1) : ci + 1].push_back(p);
  }
  int k = 0;
  for (int b = 0; b < 6; b++) {
    h_class_counts[b] = (int32_t)bucket[b].size();
    for (int p : bucket[b]) h_pdfs[k++] = p;
  }
  return 0;
}

MFA_API int mfa_gmm_score_batch(mfa_ctx *c, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                int32_t max_frames, const int32_t *d_pdf_list, const int64_t *d_pdf_off,
                                const int32_t *d_class_counts, const int64_t *d_ll_off, float *d_loglikes) {
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  if (n_utt <= 0 || max_frames <= 0) return 0;
  if (n_utt > 65535) return c->fail("at most 65535 utterances per scoring launch (got %d)", n_utt);
  GmmParams p;
  p.dim = c->dim; p.kpad = c->kpad; p.num_rows = c->num_rows;
  p.w = c->d_w; p.gc = c->d_gc; p.row0 = c->d_row0; p.nblk = c->d_nblk; p.slot = c->d_slot;
  p.feats = d_feats; p.frame_off = d_frame_off; p.pdf_list = d_pdf_list; p.pdf_off = d_pdf_off;
  p.class_counts = d_class_counts; p.ll_off = d_ll_off; p.out = d_loglikes;
  p.min_log_diff = logf(1.1920928955078125e-07f);
  p.skip_single = 0;
  { const char *pr = getenv("MFA_GMM_PRIO"); p.prio = pr ? atoi(pr) : 0; }
  const char *naive = getenv("MFA_GMM_NAIVE");
  const int m8 = c->kpad / 8;
  KernelTimer kt(c, MFA_K_GMM);
  if ((naive && naive[0] == '1') || m8 > 12) {
    // worst-case P is not known on the host side of this call: cover max_frames * num_pdfs threads per utterance
    int64_t per_utt = (int64_t)max_frames * c->num_pdfs;
    dim3 grid((unsigned)((per_utt + 255) / 256), n_utt);
    hipLaunchKernelGGL(gmm_naive_kernel, grid, dim3(256), 0, c->stream, p);
  } else {
    // frame tiles per wavefront: 1 → more resident wavefronts (≤128 VGPRs, 4 per SIMD) to interleave MFMA and epilogue
    // phases; 2 → half the model-row traffic.  MFA_GMM_NT overrides for experiments.
    const char *nt_env = getenv("MFA_GMM_NT");
    const int nt = nt_env ? atoi(nt_env) : 2;
    // single-block 32-row pdfs can go through the software-pipelined one-wavefront-per-SIMD kernel (MFA_GMM_SP=1).
    // Measured on MI355X (round 1): 88.6 TFLOP/s vs 94.8 for the generic two-wavefront kernel, so it is off by default.
    const char *sp_env = getenv("MFA_GMM_SP");
    const bool use_sp = sp_env && sp_env[0] == '1';
    p.skip_single = use_sp ? kSpMaxBlocks : 0;
    if (use_sp) {
      dim3 gsp((max_frames + 511) / 512, n_utt);
      if (m8 <= 10) hipLaunchKernelGGL((gmm_sp_kernel<10>), gsp, dim3(256), 0, c->stream, p);
      else hipLaunchKernelGGL((gmm_sp_kernel<12>), gsp, dim3(256), 0, c->stream, p);
    }
    const char *wg_env = getenv("MFA_GMM_WG");
    const int wg = wg_env ? atoi(wg_env) : 256;
    const int fpw = 32 * (nt == 2 ? 2 : 1);
    const int waves_per_wg = (nt == 2 && wg == 512) ? 8 : 4;
    p.n_utt = n_utt;
    p.tiles = (max_frames + waves_per_wg * fpw - 1) / (waves_per_wg * fpw);
    dim3 grid((unsigned)(((n_utt + 7) / 8) * 8 * p.tiles));
    if (nt == 2 && wg == 512) {
      if (m8 <= 10) hipLaunchKernelGGL((gmm_kernel<10, 2, 2, 8>), grid, dim3(512), 0, c->stream, p);
      else hipLaunchKernelGGL((gmm_kernel<12, 2, 2, 8>), grid, dim3(512), 0, c->stream, p);
    } else if (nt == 2) {
      if (m8 <= 10) hipLaunchKernelGGL((gmm_kernel<10, 2, 2, 4>), grid, dim3(256), 0, c->stream, p);
      else hipLaunchKernelGGL((gmm_kernel<12, 2, 2, 4>), grid, dim3(256), 0, c->stream, p);
    } else {
      if (m8 <= 10) hipLaunchKernelGGL((gmm_kernel<10, 1, 3, 4>), grid, dim3(256), 0, c->stream, p);
      else hipLaunchKernelGGL((gmm_kernel<12, 1, 3, 4>), grid, dim3(256), 0, c->stream, p);
    }
  }
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
