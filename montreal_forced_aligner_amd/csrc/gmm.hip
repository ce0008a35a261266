// Diagonal-GMM acoustic scoring on gfx950's matrix pipe.  Four kernel families in this file:
//   gmm_kernel               exact-f32 MFMA (v_mfma_f32_32x32x2_f32): single-Gaussian pdfs, and everything under
//                            MFA_GMM_BF16=0 — the description below is this kernel's;
//   gmm_split_single_kernel  pdfs that are one 32-row block (17–32 Gaussians): float32 products from two f16 (or three
//                            bf16) operand pieces on v_mfma_f32_32x32x16_{f16,bf16}, blocks shared through LDS;
//   gmm_split_small_kernel   the 16-, 8- and 4-row slot classes as gathered virtual 32-row blocks on the same pipe;
//   gmm_bf16_kernel          pdfs of more than 32 Gaussians: runs of blocks merged by an online log-sum-exp.
// mfa_gmm_score_batch (end of file) decides which launches a model and the environment call for.
// Replaces DecodableAmDiagGmmScaled::LogLikelihood / gmm_compute_likes (MFA/alignment/multiprocessing.py:846-853, :1415;
// Kaldi gmm/decodable-am-diag-gmm.cc, VectorBase<float>::LogSumExp).
//
// Per Gaussian:  ll = gconst + Σ_d means_invvars[d]·x[d] + Σ_d (−½ inv_vars[d])·x[d]²   — a [rows × 2D]·[2D × frames]
// contraction.  The MFMA accumulates a k-ordered fmaf chain starting from C = gconst, i.e. bit for bit the oracle's chain.
// Per pdf:       LL = max + log Σ_{ll ≥ max+ln ε} exp(ll − max)  (Kaldi: expf, double sum, log; here: hardware exp2/log2
//                and a float32 tree sum — within 1 ulp of the Kaldi value at score magnitudes ≥ 16, see reg_expsum).
//
// Packed model (built once in mfa_load_gmm): every pdf owns `slot` consecutive rows (slot ∈ {1,4,8,16,32·n}; pad rows have
// zero weights and gconst −1e30 so they fall under the cutoff).  Rows are stored in blocks of 32, operand-major
// (mfa_packed_offset in ctx.hpp): one 16-byte load per lane yields the A operands of four consecutive MFMA steps (lane l
// feeds A[row l&31][k = 2s + (l>>5)]), and the 32 lanes of a half-wavefront read 512 contiguous bytes.
// One 32-row MFMA block then serves 32/slot pdfs of the utterance's (slot-sorted) pdf list; rows ↔ accumulator registers:
// row = (r&3) + 8(r>>2) + 4(l>>5), so 4-row slots reduce inside a lane and 8/16/32-row slots add one cross-half shuffle.
//
// Work decomposition: 1-D grid of (utterance, 256-frame tile) workgroups dealt XCD-aware; a wavefront owns NT×32 frames,
// keeps their x̃ = [x, x²] operands in registers (the B side) and streams the utterance's model rows (the A side,
// L2/MALL-resident).  With reachability information (first_frame) a wavefront only walks the prefix of each class that
// its frames can be asked for.
// Output: [T][P_u] row-major, the layout the Viterbi kernel gathers from.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "ctx.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// a wavefront owns NT 32-frame tiles; a workgroup is kWaves wavefronts (template parameter)
constexpr float kPadGconst = -1.0e30f;

struct GmmParams {
  int dim, kpad, num_rows;  // num_rows = index of the dummy row
  const float *w; const float *gc; const int32_t *row0; const int32_t *nblk; const int32_t *slot;
  const uint4 *wb;   // bf16×3 split of the packed rows, 32-row blocks of [step][split][half][row] 16-byte units (or NULL)
  const uint4 *wh;   // f16×2 split of the column-scaled rows, same block layout with two pieces (or NULL)
  const float *gch;  // gconsts × S for the f16 kernel
  const float *fscale;   // [kpad] feature column scales S·2^-e_k for the f16 kernel
  float acc_scale_inv;   // 1 / S
  int *redo;         // [n_utt × tiles] tiles the f16 kernel declined (scaled feature outside the f16 range)
  int redo_mode;     // 0: score everything; 2: score only the tiles flagged in redo
  int *redo_count;   // number of flagged tiles (device scalar, zeroed per launch): the redo sweep returns at once when 0
  const float *feats; const int64_t *frame_off;
  const int32_t *pdf_list; const int64_t *pdf_off; const int32_t *class_counts; const int64_t *ll_off;
  unsigned long long *trace;   // debug (mfa_debug_gmm_trace): per workgroup {start, end, hw id, blocks} or NULL
  int skip_cc0;                // gmm_bf16_kernel: 1 = the single-block 32-row class was scored by gmm_split_single_kernel
  int skip_single;             // 1: the 32-row classes (0 and 1) are left to the split kernels; 2: the 16/8/4-row classes too
  int ff_bias;                 // debug (MFA_GMM_FF_BIAS): added to the tile's last frame before the reachability test
  const int32_t *first_frame;  // parallel to pdf_list (ascending inside each class) or NULL: see mfa_gmm_score_batch
  float *out;
  float min_log_diff;  // logf(FLT_EPSILON), computed on the host so device and oracle use the same constant
  int n_utt, tiles;    // tiles = 256-frame tiles per utterance (ceil(max_frames / 256)); items = (utterance, tile)
  int *queue;          // [0..8) phase-1 and [8..16) phase-2 per-XCD item counters; zeroed per launch
  const int *max_ff;   // largest first_frame of the batch (device scalar)
  // ---- lazy (windowed) scoring, mfa_gmm_score_window: one wavefront scores the 64 frames [b_t_begin + 64 r, +64) of one
  // utterance for the pdfs inside the band the decoder published for this window
  int b_mode;                  // 1: band mode
  int b_t_begin, b_sub;        // window start; 64-frame sub-tiles per window
  const int32_t *b_band;       // [n_utt][2] {min longest-path depth of a live token, max BFS depth reachable in the window}
  const int32_t *b_utt_list; const int32_t *b_n_list;
  const int32_t *b_done; int b_done_stride, b_done_word;
  // utterances one window behind the launch (their speculative window failed and is scored again, with the proven band, by the
  // next launch): lag word of the decoder's per-utterance state, frames per window
  const int32_t *b_lag; int b_lag_stride, b_lag_word, b_lag_frames;
  const int32_t *last_depth;   // parallel to pdf_list: running max (inside a class) of the longest-path depth of the pdf's sources
  int b_skip0;                 // f32 band kernel: classes 0..4 were scored by gmm_band_kernel (it keeps 5: single Gaussians, f32-exact)
  int b_chunk, b_nchunk;       // gmm_band_kernel: columns per wavefront (0 = the whole band) and chunks per sub-tile
  const uint4 *xsplit;         // band kernel: pre-split f16 operands [tile][2][kSteps][2][64 lanes] (gmm_presplit_kernel) or NULL
  const int *xsplit_bad;       // [tile]: 1 = a scaled feature of the tile left the f16 range (the bf16×3 pass takes it)
  const int32_t *col_row0;     // band kernel: row0[pdf_list[j]] of every column of the batch (gmm_col_rows_kernel) or NULL
  int32_t *ranges;             // [n_utt][kRangeSlots][2] band index ranges of the window (gmm_band_ranges_kernel) or NULL
  // Grouped plans (mfa_build_score_plan_grouped): class 0 of every utterance is laid out in `groups` runs (pdf id mod groups),
  // each ordered by first depth.  gmm_band_kernel then runs `groups` wavefronts per sub-tile, wavefront x — in a workgroup
  // with blockIdx % groups == x, i.e. (groups = 8) always on the same XCD — scoring run x: that XCD's L2 only ever sees
  // an eighth of the model.
  int groups;                  // 0/1: ungrouped
  const int32_t *group_counts; // [n_utt][groups]
  int b_split;                 // 1: this launch's grid holds `groups` workgroups per four sub-tiles (gmm_band_kernel)
  int b_hi_slack;              // band mode: arcs taken off the band's upper depth bound (speculative look-ahead), 0 = none
  int col_nb_packed;           // 1: col_row0 of a pdf of several blocks carries (blocks − 1) in its five low bits (rows of
                               //    the 32-row classes are multiples of 32; models whose largest pdf has ≤ 1 024 Gaussians)
};

// Band of one (utterance, window): pdf j of a class is needed iff first_frame[j] <= hi and last_depth[j] >= lo; both keys
// are non-decreasing along a class, so the needed pdfs are the index range [count(last_depth < lo), count(first_frame <= hi)).
struct Band { int lo, hi; };
__device__ __forceinline__ int band_lag(const GmmParams &p, int utt) {
  return (p.b_lag && p.b_t_begin > 0) ? (p.b_lag[(size_t)utt * p.b_lag_stride + p.b_lag_word] != 0 ? 1 : 0) : 0;
}
// first frame of the utterance's window in this launch
__device__ __forceinline__ int band_t_begin(const GmmParams &p, int utt) { return p.b_t_begin - band_lag(p, utt) * p.b_lag_frames; }
__device__ __forceinline__ Band band_of(const GmmParams &p, int utt) {
  Band b;
  const int lag = band_lag(p, utt);
  if (p.b_t_begin - lag * p.b_lag_frames <= 0) { b.lo = 0; b.hi = 64 * p.b_sub - 1; }   // only the start state is live: BFS depth 0
  else { b.lo = p.b_band[2 * utt]; b.hi = p.b_band[2 * utt + 1]; }
  // speculative look-ahead (the decoder checks what it reads) — not for a window that is being redone: the proven band
  if (p.b_hi_slack > 0 && !lag && b.hi != INT32_MAX) b.hi -= p.b_hi_slack;
  return b;
}
// gmm_band_kernel launches over a grouped plan: workgroup → (index of its four sub-tiles, run of class 0).  Consecutive
// workgroups go to consecutive XCDs, so the run's XCD is blockIdx % 8.  With 16 runs an XCD serves two of them — x and
// x + 8 — one after the other: the first half of the grid is runs 0..7, the second half runs 8..15, so that at any time an
// XCD's L2 is asked for one sixteenth of the model.  (Measured on the 51 MB model of BASELINE configs[2]: 12.8 ms per step
// against 11.2 with eight runs — sixteen wavefronts per sub-tile pay sixteen start-up chains; eight is the default.)
__device__ __forceinline__ int2 band_split_block(const GmmParams &p) {
  if (p.groups <= 8) return make_int2((int)(blockIdx.x / (unsigned)p.groups), (int)(blockIdx.x % (unsigned)p.groups));
  const unsigned half = gridDim.x >> 1, phase = blockIdx.x >= half ? 1u : 0u, rem = blockIdx.x - phase * half;
  return make_int2((int)(rem >> 3), (int)(phase * 8u + (rem & 7u)));
}
// wavefront → (utterance, 64-frame sub-tile) of a band-mode launch; false: nothing to do
__device__ __forceinline__ bool band_item(const GmmParams &p, int wave, int &utt, int &r, int *chunk = nullptr) {
  int witem = (p.b_split ? band_split_block(p).x : (int)blockIdx.x) * 4 + wave;
  if (chunk) { const int q = witem / p.b_nchunk; *chunk = witem - q * p.b_nchunk; witem = q; }
  const int item = witem / p.b_sub;
  r = witem - item * p.b_sub;
  const int n_items = p.b_n_list ? *p.b_n_list : p.n_utt;
  if (item >= n_items) return false;
  utt = p.b_utt_list ? p.b_utt_list[item] : item;
  if (p.b_t_begin > 0 && p.b_done && p.b_done[(size_t)utt * p.b_done_stride + p.b_done_word] != 0) return false;
  return true;
}

// one past the last index (i0 + lane) whose bit is set in a 64-lane ballot, 0 if none
__device__ __forceinline__ int prefix_end(unsigned long long mask, int i0) { return mask ? i0 + 64 - __clzll((long long)mask) : 0; }

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
// address of the 4-float piece (operand group 0, half h) of a packed row: see mfa_packed_offset in ctx.hpp
__device__ __forceinline__ const float *row_ptr(const float *w, int kpad, int row, int h) {
  return w + (size_t)(row >> 5) * 32 * kpad + (h * 32 + (row & 31)) * 4;
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
  // arow: this lane's piece of operand group 0 (row_ptr below); group m lies 2·32·4 floats further on
  __device__ __forceinline__ static void load_a(const float *arow, f32x4 (&a)[M8]) {
#pragma unroll
    for (int m = 0; m < M8; m++) a[m] = *reinterpret_cast<const f32x4 *>(arow + 256 * m);
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
// split-operand paths: Σ_r exp(v[r] − mx) without Kaldi's cutoff (terms below max + ln ε add < 4e-6 to the sum in total —
// inside those paths' tolerance, and mathematically the exact log-sum-exp), two terms per packed instruction: 16 exp2,
// 8 v_pk_add_f32, 8 v_pk_mul_f32, 8 packed adds per tile instead of ≈110 instructions.  The difference is formed BEFORE the
// multiplication by log2 e: fma(v, log2e, −mx·log2e) would carry the rounding of mx·log2e (2^-24·|mx|) into every term —
// 5e-6 on the result at |mx| = 100 and an overflow to inf for an outlier frame with |mx| ≳ 1e9.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float reg_expsum_fast(const f32x16 &v, float mx, float l2e = 1.44269504088896341f) {
  const f32x2 lv = {l2e, l2e};
  const f32x2 mv = {mx, mx};
  f32x2 e[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const f32x2 x = {v[2 * r], v[2 * r + 1]};
    const f32x2 arg = (x - mv) * lv;
    e[r].x = __builtin_amdgcn_exp2f(arg.x);
    e[r].y = __builtin_amdgcn_exp2f(arg.y);
  }
#pragma unroll
  for (int w = 1; w < 8; w <<= 1)
#pragma unroll
    for (int r = 0; r + w < 8; r += 2 * w) e[r] += e[r + w];
  return e[0].x + e[0].y;
}

// LL = max + ln(sum) with the hardware log2 (1 ulp on a value ≤ 7, i.e. ≲4e-7 absolute).
__device__ __forceinline__ float finish(float mx, float sum) {
  return fmaf(__builtin_amdgcn_logf(sum), 0.693147180559945309f, mx);
}

// One work item = (utterance, 64-frame tile): score_tile walks the utterance's pdf list for those frames.
template <int M8, int kNT>
__device__ __forceinline__ void score_tile(const GmmParams &p, int utt, int t_base, int lane, float *stage, int rec_index) {
  constexpr int kFramesPerWave = 32 * kNT;
  unsigned long long t_start = 0;
  if (p.trace) t_start = wall_clock64();
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  if (t_base >= T) return;
  const int col = lane & 31, h = lane >> 5;
  const int64_t l0 = p.pdf_off[utt];
  const int P = (int)(p.pdf_off[utt + 1] - l0);
  const int32_t *list = p.pdf_list + l0;
  const int32_t *cc6 = p.class_counts + (size_t)utt * 6;
  // class_counts[u] = {32-row single-block, 32-row multi-block, 16, 8, 4, 1}
  const int32_t cc[5] = {cc6[0] + cc6[1], cc6[2], cc6[3], cc6[4], cc6[5]};
  // need[c]: how many pdfs of class c this wavefront's frames can be asked for.  Without reachability information that
  // is all of them; with it, the pdfs whose first possible frame lies at or before the tile's last frame — a prefix of
  // the class, because the host ordered each class by that frame.
  int need[6], lo_[6];
  {
    int t_last = min(T, t_base + kFramesPerWave) - 1 + p.ff_bias;
    int d_lo = 0;
    if (p.b_mode) { const Band bd = band_of(p, utt); t_last = bd.hi; d_lo = bd.lo; }
    int off = 0;
#pragma unroll
    for (int cls = 0; cls < 6; cls++) {
      const int cnt = cc6[cls];
      int nd = cnt, lw = 0;
      if (cls == 0 && p.groups > 1) { need[0] = 0; lo_[0] = 0; off += cnt; continue; }   // grouped plan: searched run by run below
      if (p.first_frame) {
        nd = 0;
        for (int i0 = 0; i0 < cnt; i0 += 64) {
          const int i = i0 + lane;
          const bool ok = i < cnt && p.first_frame[l0 + off + i] <= t_last;
          nd = max(nd, prefix_end(__ballot(ok), i0));   // = the count for a class ordered by first frame; a superset prefix when
                                                        // the caller passes a grouped plan's lists without its run counts
          if (p.b_mode) lw += __popcll(__ballot(i < cnt && p.last_depth[l0 + off + i] < d_lo));
        }
      }
      need[cls] = nd;
      lo_[cls] = min(lw, nd);
      off += cnt;
    }
  }
  float *out = p.out + p.ll_off[utt];
  // skip_single: the 32-row pdfs (single- and multi-block) are scored by gmm_bf16_kernel; only the small-slot classes are
  // left for this launch
  if (p.skip_single >= 2) { need[2] = 0; need[3] = 0; need[4] = 0; }   // slots 16 / 8 / 4 went to gmm_split_small_kernel
  if (p.skip_single && need[2] + need[3] + need[4] + need[5] == 0) return;
  if (p.b_skip0) {   // band mode after gmm_band_kernel: single Gaussians are left
    need[1] = 0; need[2] = 0; need[3] = 0; need[4] = 0; lo_[1] = 0; lo_[2] = 0; lo_[3] = 0; lo_[4] = 0;
    if (need[5] - lo_[5] == 0) return;
  }

  Tile<M8, kNT> tile;
  tile.load_b(p, f0, T, t_base, lane);
  f32x16 acc[kNT];
  // Class 0 is one run ordered by first depth, or (grouped plan) `groups` runs — searched and walked one after the other.
  const bool skip0 = p.skip_single || p.b_skip0;
  const int nruns = p.groups > 1 ? p.groups : 1;
  int run_off = 0;
  for (int run = 0; run < nruns; run++) {
  int n_single = skip0 ? 0 : need[0];
  int first32 = lo_[0];         // band mode: the class-0 range starts here (0 otherwise)
  if (p.groups > 1 && !skip0) {
    const int cnt = p.group_counts[(size_t)utt * p.groups + run];
    int nd = cnt, lw = 0;
    if (p.first_frame) {
      int t_last = min(T, t_base + kFramesPerWave) - 1 + p.ff_bias, d_lo = 0;
      if (p.b_mode) { const Band bd = band_of(p, utt); t_last = bd.hi; d_lo = bd.lo; }
      nd = 0;
      for (int i0 = 0; i0 < cnt; i0 += 64) {
        const int i = i0 + lane;
        nd += __popcll(__ballot(i < cnt && p.first_frame[l0 + run_off + i] <= t_last));
        if (p.b_mode) lw += __popcll(__ballot(i < cnt && p.last_depth[l0 + run_off + i] < d_lo));
      }
    }
    first32 = run_off + min(lw, nd); n_single = run_off + nd;
    run_off += cnt;
  }

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
    const float *wl = p.w + (h * 32 + col) * 4;  // this lane's piece inside a block (blocks start at multiples of 32 rows)
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
#ifndef GMM_DIAG_NO_LOADS
      Tile<M8, kNT>::load_gc32(p.gc + r1, h, g);
#endif
#pragma unroll
      for (int m = 0; m < M8; m++) {
#pragma unroll
        for (int cc4 = 0; cc4 < 4; cc4++) {
#pragma unroll
          for (int n = 0; n < kNT; n++)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][cc4], tile.b[n][4 * m + cc4], acc[n], 0, 0, 0);
        }
#ifndef GMM_DIAG_NO_LOADS   // timing-only builds (tools/: -DGMM_DIAG_*): results are wrong by construction
        a[m] = *reinterpret_cast<const f32x4 *>(wn + 256 * m);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      r1 = __builtin_amdgcn_readfirstlane(x_r2);
      pdf2 = __builtin_amdgcn_readfirstlane(x_pdf3);
      float mx[kNT], sum[kNT];
#pragma unroll
      for (int n = 0; n < kNT; n++) {
#ifdef GMM_DIAG_NO_EPILOGUE
        mx[n] = acc[n][0] + acc[n][15]; sum[n] = 1.0f;
#else
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, swap32(m, h));
        float sv = reg_expsum<0, 16>(acc[n], m, m + p.min_log_diff);
        sv += swap32(sv, h);
        mx[n] = m; sum[n] = sv;
#endif
      }
      if constexpr (kNT == 2) {
        // both halves hold every tile's (max, sum): half h finishes tile h (one log per lane).
        // Stage [64 frames][32 pdfs] in LDS and flush whole 128-byte row segments: a lane-per-frame store would touch 64
        // different lines per instruction and (measured, round 1) inflate HBM write traffic 11x with partial lines.
        const float v = finish(h ? mx[1] : mx[0], h ? sum[1] : sum[0]);
        const int jj = (j - first32) & 31;
        stage[(32 * h + col) * 33 + jj] = v;
#ifdef GMM_DIAG_NO_FLUSH
        if (v == 12345.678f) {   // never true, but keeps the scores alive
#else
        if (jj == 31 || j == last) {
#endif
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
  }   // runs of class 0

  // ---- 32-row pdfs with more than 32 Gaussians: several blocks, two passes (max, then the sum against that max)
  const int n32 = cc[0];
  for (int j = cc6[0] + lo_[1]; j < cc6[0] + (p.skip_single ? 0 : need[1]); j++) {
    const int pdf = list[j];
    const int r0 = p.row0[pdf], nb = p.nblk[pdf];
    float mx[kNT], sum[kNT];
#pragma unroll
    for (int n = 0; n < kNT; n++) { mx[n] = -INFINITY; sum[n] = 0.0f; }
    for (int blk = 0; blk < nb; blk++) {
      const int rr = r0 + 32 * blk + col;
      tile.block(row_ptr(p.w, p.kpad, rr, h), p.gc[rr], lane, acc);
#pragma unroll
      for (int n = 0; n < kNT; n++) {
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, __shfl_xor(m, 32));
        mx[n] = fmaxf(mx[n], m);
      }
    }
    for (int blk = 0; blk < nb; blk++) {
      const int rr = r0 + 32 * blk + col;
      tile.block(row_ptr(p.w, p.kpad, rr, h), p.gc[rr], lane, acc);
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
  for (int j = lo_[2] & ~1; j < need[2]; j += 2) {
    const int which = col >> 4, within = col & 15;
    const int idx = j + which;
    const int row = idx < cc[1] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(row_ptr(p.w, p.kpad, row, h), p.gc[row], lane, acc);
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
  for (int j = lo_[3] & ~3; j < need[3]; j += 4) {
    const int which = col >> 3, within = col & 7;
    const int idx = j + which;
    const int row = idx < cc[2] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(row_ptr(p.w, p.kpad, row, h), p.gc[row], lane, acc);
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
  for (int j = lo_[4] & ~7; j < need[4]; j += 8) {
    const int which = col >> 2, within = col & 3;
    const int idx = j + which;
    const int row = idx < cc[3] ? p.row0[list[base + idx]] + within : p.num_rows;
    tile.block(row_ptr(p.w, p.kpad, row, h), p.gc[row], lane, acc);
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
  for (int j = lo_[5] & ~31; j < need[5]; j += 32) {
    const int idx = j + col;
    const int row = idx < cc[4] ? p.row0[list[base + idx]] : p.num_rows;
    tile.block(row_ptr(p.w, p.kpad, row, h), p.gc[row], lane, acc);
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
  if (p.trace && lane == 0) {
    unsigned long long *rec = p.trace + (size_t)rec_index * 4;
    rec[0] = t_start; rec[1] = wall_clock64();
    rec[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32);
    rec[3] = (unsigned long long)(need[0] + need[1]);
  }
}

// Persistent scoring kernel.  Round-1 timeline of the one-workgroup-per-tile version (tools/gmm_timeline.py): the hardware
// deals workgroups to CUs in a fixed round-robin order — every CU received exactly 32 of the 8192 workgroups and, the
// tile index being periodic in the grid, always the SAME tile type — so CUs with cheap tiles idled (slot occupancy 82 %)
// and skipping unreachable cells bought no time at all.  Here the grid is just enough workgroups to fill the chip
// (2 per CU) and work is pulled from queues (atomic counters) until they run dry, in two phases:
//   phase 1, workgroup items (utterance, 256-frame tile) for the tiles whose four 64-frame sub-tiles all need the whole
//     pdf list: the four wavefronts take one sub-tile each and walk the list at the same pace, so the model rows they
//     stream come through the CU's L1 once, not four times (measured: 5.3 µs per 32-row block against 5.9 µs when every
//     wavefront streams its own rows);
//   phase 2, wavefront items (utterance, 64-frame tile) for the leading tiles, where reachability makes the sub-tiles
//     unequal (a workgroup item would idle three wavefronts behind the fourth); being short, they also fill the tail.
// One queue per XCD and phase, holding the utterances u ≡ xcd (mod 8): all tiles of an utterance stream the same rows
// through that XCD's private L2 (cdna_hip_programming.md T1); the XCD a workgroup runs on is read from XCC_ID.  Items go
// utterance by utterance, last frames first.  A workgroup whose queue is empty takes items from the other XCDs' queues,
// so a phase ends within one item.  Every wavefront leaves a loop once all eight counters have passed their item counts:
// the grid always drains.
template <int M8, int kNT, int kMinWaves, int kWaves>
__global__ __launch_bounds__(64 * kWaves, kMinWaves) void gmm_kernel(GmmParams p) {
  constexpr int kFramesPerWave = 32 * kNT, kFramesPerTile = kFramesPerWave * kWaves;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-wavefront output staging tile (written lane-per-frame, read row-wise by the same wavefront)
  __shared__ float stage_all[kWaves][64 * 33];
  __shared__ int s_item;
  const int my_xcd = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
  // leading tiles whose first sub-tile cannot yet see every pdf (first possible frame beyond that sub-tile's last frame)
  int light = 0;
  if (p.first_frame) {
    const int mff = __builtin_amdgcn_readfirstlane(*p.max_ff);
    light = mff >= kFramesPerWave ? min(p.tiles, (mff - (kFramesPerWave - 1) + kFramesPerTile - 1) / kFramesPerTile) : 0;
  }
  const int heavy = p.tiles - light;
  // Opaque copies inside the loops: without them the compiler hoists every lane-dependent address out of the item loop
  // and keeps it in registers for the kernel's lifetime (measured: 256 VGPRs + 240 bytes of scratch instead of 217 VGPRs).
  if (heavy > 0) {
    for (int hop = 0; hop < 8; hop++) {
      const int q = (my_xcd + hop) & 7;
      const int n_items = ((p.n_utt - q + 7) >> 3) * heavy;   // utterances q, q+8, q+16, ...
      for (;;) {
        __syncthreads();                             // every wavefront is done with the previous item (and has read s_item)
        if (threadIdx.x == 0) s_item = atomicAdd(&p.queue[q], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= n_items) break;                  // uniform over the workgroup
        int lane_i = lane, wave_i = wave;
        asm volatile("" : "+v"(lane_i), "+v"(wave_i));
        wave_i = __builtin_amdgcn_readfirstlane(wave_i);
        const int v = item / heavy, tl = p.tiles - 1 - item % heavy;
        score_tile<M8, kNT>(p, v * 8 + q, (tl * kWaves + wave_i) * kFramesPerWave, lane_i, stage_all[wave_i],
                            ((v * 8 + q) * p.tiles + tl) * kWaves + wave_i);
      }
    }
  }
  if (light > 0) {
    const int per_utt = light * kWaves;
    for (int hop = 0; hop < 8; hop++) {
      const int q = (my_xcd + hop) & 7;
      const int n_items = ((p.n_utt - q + 7) >> 3) * per_utt;
      for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&p.queue[8 + q], 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;                  // uniform over the wavefront
        int lane_i = lane, wave_i = wave;
        asm volatile("" : "+v"(lane_i), "+v"(wave_i));
        wave_i = __builtin_amdgcn_readfirstlane(wave_i);
        const int v = item / per_utt, r = per_utt - 1 - item % per_utt;
        score_tile<M8, kNT>(p, v * 8 + q, r * kFramesPerWave, lane_i, stage_all[wave_i], (v * 8 + q) * p.tiles * kWaves + r);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16×3 scoring of the single-block 32-row pdfs on v_mfma_f32_32x32x16_bf16 (the default for that slot class;
// MFA_GMM_BF16=0 sends it back to the bit-exact f32 kernel).
// A float32 value is the exact sum of three bf16 pieces (8 + 8 + 8 mantissa bits), x = x1 + x2 + x3, and a product of two
// bf16 values is exact in float32, so   x·w ≈ x1w1 + (x1w2 + x2w1) + (x1w3 + x2w2 + x3w1)   with a relative error of
// ≈2^-24 per term — the error of ONE float32 rounding (tools/mfma_bf16_layout_test.hip: 5.0e-8 of Σ|terms| against
// float64).  Six bf16 MFMAs of 32 cycles cover 16 k-values that cost eight f32 MFMAs of 64 cycles: 2.7× the f32 rate.
// What changes is the order of the accumulation, so scores agree with the fmaf-chain oracle to float32 rounding noise
// (≲2e-4 absolute on |score| ≈ 100; north_star's bar is 1e-3), not bit for bit like the f32 path.
//
// At this MFMA rate a wavefront cannot stream its own copy of the model rows (4× the L1/L2 traffic of the f32 kernel per
// unit time), so the kernel is organised like a GEMM: the workgroup's four wavefronts (64 frames each, x̃ split once into
// registers: 120 VGPRs) share every 32-row block through LDS, double-buffered — while block j is multiplied out of one
// buffer, block j+1 travels global → registers → the other buffer; one barrier per block.
// General form (models that contain multi-block pdfs); gmm_split_single_kernel below is the lean form for single-block pdfs.
// x̃ = [x, x²] of a wavefront's two 32-frame tiles (frames t_base + 32 n + col, clamped into the utterance), split into
// the MFMA's B operands: b[tile][step][piece], lane = (frame col, k-half h).  kPieces = 3: bf16 triples (v = v1 + v2 + v3,
// round to nearest even each).  kPieces = 2: f16 pairs of the column-scaled value; returns true when a scaled value
// leaves the f16 range (or is NaN) — the caller then hands the whole tile to the bf16×3 pass.
template <int kSteps, int kPieces, typename Op8>
__device__ __forceinline__ bool split_features(const GmmParams &p, int64_t f0, int T, int t_base, int col, int h,
                                               Op8 (&b)[2][kSteps][kPieces]) {
  bool bad = false;
#pragma unroll
  for (int n = 0; n < 2; n++) {
    int t = t_base + 32 * n + col;
    t = t < T ? t : T - 1;
    t = t < 0 ? 0 : t;
    const float *x = p.feats + (f0 + t) * p.dim;
    const bool vec8 = (p.dim & 7) == 0;   // every group of 8 operand columns then lies wholly in x, in x² or in the padding
#pragma unroll
    for (int s = 0; s < kSteps; s++) {
      float xv8[8], fs8[8];
      {
        const int k0 = 16 * s + 8 * h;
        if (vec8) {   // two 16-byte loads per group instead of eight 4-byte ones (same values)
          const int i0 = k0 < p.dim ? k0 : (k0 < 2 * p.dim ? k0 - p.dim : 0);
          const float4 lo4 = *reinterpret_cast<const float4 *>(x + i0), hi4 = *reinterpret_cast<const float4 *>(x + i0 + 4);
          xv8[0] = lo4.x; xv8[1] = lo4.y; xv8[2] = lo4.z; xv8[3] = lo4.w; xv8[4] = hi4.x; xv8[5] = hi4.y; xv8[6] = hi4.z; xv8[7] = hi4.w;
          if constexpr (kPieces == 2) {
            const float4 f0_ = *reinterpret_cast<const float4 *>(p.fscale + k0), f1_ = *reinterpret_cast<const float4 *>(p.fscale + k0 + 4);
            fs8[0] = f0_.x; fs8[1] = f0_.y; fs8[2] = f0_.z; fs8[3] = f0_.w; fs8[4] = f1_.x; fs8[5] = f1_.y; fs8[6] = f1_.z; fs8[7] = f1_.w;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int k = k0 + e;
            xv8[e] = x[k < p.dim ? k : (k < 2 * p.dim ? k - p.dim : 0)];
            if constexpr (kPieces == 2) fs8[e] = p.fscale[k];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int k = 16 * s + 8 * h + e;
        const float xv = xv8[e];
        const float v = k < p.dim ? xv : (k < 2 * p.dim ? xv * xv : 0.0f);
        if constexpr (kPieces == 2) {
          const float sv = v * fs8[e];
          bad |= !(fabsf(sv) <= 65000.0f);
          const _Float16 v1 = (_Float16)sv;
          b[n][s][0][e] = v1; b[n][s][1][e] = (_Float16)(sv - (float)v1);
        } else {
          const __bf16 v1 = (__bf16)v;
          const float r1 = v - (float)v1;
          const __bf16 v2 = (__bf16)r1;
          const float r2 = r1 - (float)v2;
          b[n][s][0][e] = v1; b[n][s][1][e] = v2; b[n][s][2][e] = (__bf16)r2;
        }
      }
    }
  }
  return bad;
}

// One 32-row model block (split operands in LDS: [step][piece][half][row] 16-byte units; its 32 gconsts) times a
// wavefront's two frame tiles → acc.  Operand pieces of step s+1 are read from LDS while step s is multiplied; six (three)
// products per 16 k-values, smallest terms first; the two tiles alternate so that consecutive MFMAs never wait on each
// other's accumulator; the gconsts enter as the first MFMA's addend.
template <int kSteps, int kPieces, typename Op8>
__device__ __forceinline__ void multiply_block(const uint4 *a_blk, const float *gc_blk, const Op8 (&b)[2][kSteps][kPieces],
                                               f32x16 (&acc)[2], int col, int h) {
  constexpr bool kHalf = kPieces == 2;
  f32x16 init;
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    const float4 gq = *reinterpret_cast<const float4 *>(&gc_blk[8 * qq + 4 * h]);
    init[4 * qq] = gq.x; init[4 * qq + 1] = gq.y; init[4 * qq + 2] = gq.z; init[4 * qq + 3] = gq.w;
  }
  auto read_a = [&](int s, Op8 (&a)[kPieces]) {
#pragma unroll
    for (int qq = 0; qq < kPieces; qq++) a[qq] = __builtin_bit_cast(Op8, a_blk[((s * kPieces + qq) * 2 + h) * 32 + col]);
  };
  Op8 a_cur[kPieces], a_nxt[kPieces];
  read_a(0, a_cur);
  constexpr int kProd = kHalf ? 3 : 6;
  constexpr int pa[6] = {kHalf ? 1 : 2, kHalf ? 0 : 1, 0, 1, 0, 0}, pb[6] = {0, 1, kHalf ? 0 : 2, 0, 1, 0};
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    if (s + 1 < kSteps) read_a(s + 1, a_nxt);
#pragma unroll
    for (int t6 = 0; t6 < kProd; t6++)
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const f32x16 &cin = (s == 0 && t6 == 0) ? init : acc[n];
        if constexpr (kHalf) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[pa[t6]], b[n][s][pb[t6]], cin, 0, 0, 0);
        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[pa[t6]], b[n][s][pb[t6]], cin, 0, 0, 0);
      }
#pragma unroll
    for (int qq = 0; qq < kPieces; qq++) a_cur[qq] = a_nxt[qq];
  }
}

// kPieces = 3: bf16 triples; kPieces = 2: scaled f16 pairs with the per-tile range fallback (see gmm_split_single_kernel).
template <int kSteps, int kPieces>   // 16-k steps per row: 5 for D ≤ 40, 6 for D ≤ 48
__global__ __launch_bounds__(256, 2) void gmm_bf16_kernel(GmmParams p) {
  constexpr bool kMulti = true;
  constexpr bool kHalf = kPieces == 2;
  using op8 = std::conditional_t<kHalf, f16x8, bf16x8>;
  constexpr int kNT = 2, kWaves = 4, kFramesPerWave = 64, kFramesPerTile = 256;
  constexpr int kUnits = kSteps * kPieces * 2 * 32;    // 16-byte units per block
  constexpr int kLoads = (kUnits + 255) / 256;         // units each thread moves per block
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float stage_all[kWaves][64 * 33];
  __shared__ uint4 a_lds[2][kUnits];
  __shared__ __attribute__((aligned(16))) float gc_lds[2][32];
  // Entry table of the item, staged in chunks (two dependent global loads per pdf must not sit in the block loop).  An
  // entry is one 32-row block: a single-block pdf is one entry; a pdf with more than 32 Gaussians is a run of entries
  // whose (max, sum) pairs are merged on the fly (online log-sum-exp) and emitted with its last block.
  constexpr int kBlkCache = 1024;
  constexpr int kFirst = 1 << 30, kLast = 1 << 31;
  __shared__ int blk_lds[kBlkCache];                  // 32-row block index
  __shared__ int col_lds[kBlkCache];                  // output column | kFirst | kLast
  __shared__ int s_item;
  float *stage = stage_all[wave];
  if (!kHalf && p.redo_mode == 2 && *p.redo_count == 0) return;   // uniform: the f16 pass declined nothing
  const int my_xcd = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
  for (int hop = 0; hop < 8; hop++) {
    const int q = (my_xcd + hop) & 7;
    const int n_items = ((p.n_utt - q + 7) >> 3) * p.tiles;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_item = atomicAdd(&p.queue[q], 1);
      __syncthreads();
      const int item = s_item;
      if (item >= n_items) break;
      int lane = lane0;                                // opaque per item: keeps lane-dependent addresses out of long-lived registers
      asm volatile("" : "+v"(lane));
      const int col = lane & 31, h = lane >> 5;
      const int utt = (item / p.tiles) * 8 + q, tl = p.tiles - 1 - item % p.tiles;
      const int64_t f0 = p.frame_off[utt];
      const int T = (int)(p.frame_off[utt + 1] - f0);
      if (tl * kFramesPerTile >= T) continue;          // uniform over the workgroup
      if (!kHalf && p.redo_mode == 2 && p.redo[(size_t)utt * p.tiles + tl] == 0) continue;   // only what the f16 pass left
      const int t_base = (tl * kWaves + wave) * kFramesPerWave;
      const bool active = t_base < T;                  // a wavefront past the end still helps move blocks and joins barriers
      const int64_t l0 = p.pdf_off[utt];
      const int P = (int)(p.pdf_off[utt + 1] - l0);
      const int32_t *list = p.pdf_list + l0;
      const int cc0 = p.class_counts[(size_t)utt * 6], cc1 = kMulti ? p.class_counts[(size_t)utt * 6 + 1] : 0;
      // n0 / n1: single-block / multi-block pdfs the tile's LAST frame can be asked for — the prefixes the workgroup walks
      // together (block copies and barriers are collective).  n0_mine / n1_mine: the shorter prefixes this wavefront's own
      // 64 frames can be asked for; beyond them the wavefront only helps with the copies.
      int n0 = cc0, n1 = cc1, n0_mine = cc0, n1_mine = cc1;
      if (p.first_frame) {
        const int t_last = min(T, (tl + 1) * kFramesPerTile) - 1 + p.ff_bias;
        const int t_mine = min(T, t_base + kFramesPerWave) - 1 + p.ff_bias;
        n0 = n1 = n0_mine = n1_mine = 0;
        for (int i0 = 0; i0 < cc0 + cc1; i0 += 64) {
          const int i = i0 + lane;
          const int ff = i < cc0 + cc1 ? p.first_frame[l0 + i] : 0x7fffffff;
          const unsigned long long all = __ballot(ff <= t_last), mine = __ballot(ff <= t_mine);
          const unsigned long long c0m = __ballot(i < cc0);
          // class 0: the prefix up to the LAST pdf that can be asked for (= the count when the class is ordered by first
          // frame; a superset of what is needed when a grouped plan lays it out in several ordered runs)
          n0 = max(n0, prefix_end(all & c0m, i0)); n1 += __popcll(all & ~c0m);
          n0_mine = max(n0_mine, prefix_end(mine & c0m, i0)); n1_mine += __popcll(mine & ~c0m);
        }
      }
      if (p.skip_cc0) { n0 = 0; n0_mine = 0; }         // columns keep their places: multi-block pdfs start at column cc0
      // total entries: one per single-block pdf, nblk per multi-block pdf
      int e_multi = 0;
      for (int i0 = 0; i0 < n1; i0 += 64) {
        const int i = i0 + lane;
        int nb = i < n1 ? p.nblk[list[cc0 + i]] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nb += __shfl_xor(nb, o);
        e_multi += nb;
      }
      const int n_entries = n0 + e_multi;
      float *out = p.out + p.ll_off[utt];
      if (n_entries > 0) {
        // ---- x̃ = [x, x²] of this wavefront's 64 frames, split into bf16 triples: b[tile][step][piece], lane (frame, half)
        op8 b[kNT][kSteps][kPieces];
        // kHalf: `bad` = a scaled feature outside the f16 range (or NaN)
        const bool bad = split_features<kSteps, kPieces>(p, f0, T, t_base, col, h, b);
        if constexpr (kHalf) {
          if (__syncthreads_or(bad)) {                     // uniform: the whole tile goes to the bf16×3 pass
            if (threadIdx.x == 0) { p.redo[(size_t)utt * p.tiles + tl] = 1; atomicAdd(p.redo_count, 1); }
            continue;
          }
        }
        const uint4 *wsrc = kHalf ? p.wh : p.wb;
        const float *gsrc = kHalf ? p.gch : p.gc;
        const float inv_s = kHalf ? p.acc_scale_inv : 1.0f;
        const float l2e_s = 1.44269504088896341f * inv_s;  // inv_s is a power of two: scaling commutes with the rounding
        // Block copy global → registers (requested before block j is multiplied) → LDS (written after it).  The LDS-DMA form
        // (global_load_lds) measured the same when it overlapped and much worse when it did not: the compiler cannot tell the
        // two LDS buffers apart and drains vmcnt before every LDS read while a DMA write is in flight.
        uint4 mv[kLoads];
        float4 gmv = make_float4(0.f, 0.f, 0.f, 0.f);
        auto fetch = [&](int blk) {
          const uint4 *src = wsrc + (size_t)blk * kUnits;
#pragma unroll
          for (int i = 0; i < kLoads; i++) {
            const int u = threadIdx.x + 256 * i;
            mv[i] = u < kUnits ? src[u] : make_uint4(0, 0, 0, 0);
          }
          if (threadIdx.x < 8) gmv = *reinterpret_cast<const float4 *>(gsrc + (size_t)blk * 32 + 4 * threadIdx.x);
        };
        auto deposit = [&](int buf) {
#pragma unroll
          for (int i = 0; i < kLoads; i++) {
            const int u = threadIdx.x + 256 * i;
            if (u < kUnits) a_lds[buf][u] = mv[i];
          }
          if (threadIdx.x < 8) *reinterpret_cast<float4 *>(&gc_lds[buf][4 * threadIdx.x]) = gmv;
        };
        int multi_pdf = 0, multi_blk = 0;                  // thread 0's cursor into the multi-block pdfs
        int staged = 0, stage_col0 = 0;                    // columns waiting in the staging tile: stage_col0 .. +staged-1
        float mx_run[kNT], sum_run[kNT];
#pragma unroll
        for (int n = 0; n < kNT; n++) { mx_run[n] = -INFINITY; sum_run[n] = 0.0f; }
        auto flush = [&]() {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
          for (int i = 0; i < 32; i++) {
            const int r = h + 2 * i, t = t_base + r;
            if (col < staged && t < T) __builtin_nontemporal_store(stage[r * 33 + col], &out[(size_t)t * P + stage_col0 + col]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          staged = 0;
        };
        for (int c0 = 0; c0 < n_entries; c0 += kBlkCache) {
        const int c1 = min(n_entries, c0 + kBlkCache);
        __syncthreads();                                   // previous chunk's table is no longer read
        for (int i = c0 + threadIdx.x; i < min(c1, n0); i += 256) {
          blk_lds[i - c0] = p.row0[list[i]] >> 5;
          if (kMulti) col_lds[i - c0] = i | kFirst | kLast;
        }
        if (kMulti && threadIdx.x == 0) {
          for (int e = max(c0, n0); e < c1; e++) {
            const int pdf = list[cc0 + multi_pdf], nb = p.nblk[pdf];
            blk_lds[e - c0] = (p.row0[pdf] >> 5) + multi_blk;
            col_lds[e - c0] = (cc0 + multi_pdf) | (multi_blk == 0 ? kFirst : 0) | (multi_blk == nb - 1 ? kLast : 0);
            if (++multi_blk == nb) { multi_blk = 0; multi_pdf++; }
          }
        }
        __syncthreads();
        auto block_of = [&](int jj) { return blk_lds[min(jj, c1 - 1) - c0]; };
        fetch(block_of(c0));
        deposit(0);
        __syncthreads();
        for (int j = c0; j < c1; j++) {
          const int buf = (j - c0) & 1;
#ifndef BF16_DIAG_NO_FETCH   // timing-only builds (tools/gmm_ablation.sh): results are wrong by construction
          fetch(block_of(j + 1));                          // block j+1 (the chunk's last trip re-fetches its last block: harmless)
#endif
          const int ecol = kMulti ? col_lds[j - c0] : (j | kFirst | kLast);
          const int out_col = ecol & ~(kFirst | kLast);
          const bool mine = out_col < cc0 ? out_col < n0_mine : out_col - cc0 < n1_mine;
          if (active && mine) {
            f32x16 acc[kNT];
            multiply_block<kSteps, kPieces>(a_lds[buf], gc_lds[buf], b, acc, col, h);
            // ---- log-sum-exp epilogue and LDS-staged, coalesced score stores: as in score_tile
            float mx[kNT], sum[kNT];
#pragma unroll
            for (int n = 0; n < kNT; n++) {
#ifdef BF16_DIAG_NO_EPILOGUE
              mx[n] = acc[n][0] + acc[n][15]; sum[n] = 1.0f;
#else
              float m = reg_max<0, 16>(acc[n]);
              m = fmaxf(m, swap32(m, h));
              float sv = reg_expsum_fast(acc[n], m, l2e_s);
              sv += swap32(sv, h);
              mx[n] = m; sum[n] = sv;                        // mx stays in accumulator units (× S) until the pdf's last block
#endif
            }
            if (kMulti && !(ecol & kFirst)) {
              // online log-sum-exp: fold this block's (max, sum) into the pdf's running pair
#pragma unroll
              for (int n = 0; n < kNT; n++) {
                const float M = fmaxf(mx_run[n], mx[n]);
                sum[n] = sum_run[n] * __builtin_amdgcn_exp2f((mx_run[n] - M) * l2e_s) +
                         sum[n] * __builtin_amdgcn_exp2f((mx[n] - M) * l2e_s);
                mx[n] = M;
              }
            }
            if (kMulti) {
#pragma unroll
              for (int n = 0; n < kNT; n++) { mx_run[n] = mx[n]; sum_run[n] = sum[n]; }
            }
#ifdef BF16_DIAG_NO_FLUSH
            if (mx[0] == 12345.678f) {
#else
            if (ecol & kLast) {
#endif
              const float v = finish((h ? mx[1] : mx[0]) * inv_s, h ? sum[1] : sum[0]);
              if (staged > 0 && out_col != stage_col0 + staged) flush();   // a jump in the column sequence (class change)
              if (staged == 0) stage_col0 = out_col;
              stage[(32 * h + col) * 33 + staged] = v;
              if (++staged == 32) flush();
            }
          }
#ifndef BF16_DIAG_NO_FETCH
          deposit(buf ^ 1);
#endif
#ifndef BF16_DIAG_NO_BARRIER
          __syncthreads();                               // block j+1 is in place; everybody is done with block j
#endif
        }
        }
        if (staged > 0) flush();
      }
    }
  }
}

// Lean instantiation for models WITHOUT multi-block pdfs (the headline configuration): every entry is a whole pdf, so there
// is no entry table beyond the block indices, no merge state, fixed 32-column staging phases, and the block copies go
// global → LDS directly (global_load_lds_dwordx4; here the compiler lets them overlap).  3 % faster than the general kernel
// on configs[2]; same arithmetic, same results.
//
// kPieces = 3: operands are bf16 triples, six products per 16 k-values (2^-24 per term, any exponent range).
// kPieces = 2: operands are f16 pairs, three products (a2·b1, a1·b2, a1·b1: 3·2^-22 per term worst case, half the matrix
//   work).  f16 has 5 exponent bits, so the operands are scaled by powers of two chosen from the model at load time
//   (mfa_load_gmm: weight column k × 2^e_k, feature column k × S·2^-e_k, accumulators therefore × S; all exact) and a tile
//   whose scaled features leave the f16 range is not scored here: it is flagged in p.redo and scored by the kPieces = 3
//   kernel, launched next with redo_mode 2.
template <int kSteps, int kPieces>   // 16-k steps per row: 5 for D ≤ 40, 6 for D ≤ 48
__global__ __launch_bounds__(256, 2) void gmm_split_single_kernel(GmmParams p) {
  constexpr int kNT = 2, kWaves = 4, kFramesPerWave = 64, kFramesPerTile = 256;
  constexpr bool kHalf = kPieces == 2;
  using op8 = std::conditional_t<kHalf, f16x8, bf16x8>;
  constexpr int kUnits = kSteps * kPieces * 2 * 32;    // 16-byte units per block
  constexpr int kLoads = (kUnits + 255) / 256;         // units each thread moves per block
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float stage_all[kWaves][64 * 33];
  __shared__ uint4 a_lds[2][kUnits];
  __shared__ __attribute__((aligned(16))) float gc_lds[2][32];
  constexpr int kBlkCache = 1024;                     // pdf → 32-row block index, staged per item (two dependent global
  __shared__ int blk_lds[kBlkCache];                  // loads per pdf must not sit in the block loop)
  __shared__ int s_item;
  float *stage = stage_all[wave];
  if (!kHalf && p.redo_mode == 2 && *p.redo_count == 0) return;   // uniform: the f16 pass declined nothing
  const int my_xcd = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
  for (int hop = 0; hop < 8; hop++) {
    const int q = (my_xcd + hop) & 7;
    const int n_items = ((p.n_utt - q + 7) >> 3) * p.tiles;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_item = atomicAdd(&p.queue[q], 1);
      __syncthreads();
      const int item = s_item;
      if (item >= n_items) break;
      int lane = lane0;                                // opaque per item: keeps lane-dependent addresses out of long-lived registers
      asm volatile("" : "+v"(lane));
      const int col = lane & 31, h = lane >> 5;
      const int utt = (item / p.tiles) * 8 + q, tl = p.tiles - 1 - item % p.tiles;
      const int64_t f0 = p.frame_off[utt];
      const int T = (int)(p.frame_off[utt + 1] - f0);
      if (tl * kFramesPerTile >= T) continue;          // uniform over the workgroup
      if (!kHalf && p.redo_mode == 2 && p.redo[(size_t)utt * p.tiles + tl] == 0) continue;   // only what the f16 pass left
      const int t_base = (tl * kWaves + wave) * kFramesPerWave;
      const bool active = t_base < T;                  // a wavefront past the end still helps move blocks and joins barriers
      const int64_t l0 = p.pdf_off[utt];
      const int P = (int)(p.pdf_off[utt + 1] - l0);
      const int32_t *list = p.pdf_list + l0;
      const int n_all = p.class_counts[(size_t)utt * 6];
      // n_single: pdfs the tile's LAST frame can be asked for — the prefix the workgroup walks together (block copies and
      // barriers are collective).  n_mine: the shorter prefix this wavefront's own 64 frames can be asked for; beyond it
      // the wavefront only helps with the copies.
      int n_single = n_all, n_mine = n_all;
      if (p.first_frame) {
        const int t_last = min(T, (tl + 1) * kFramesPerTile) - 1 + p.ff_bias;
        const int t_mine = min(T, t_base + kFramesPerWave) - 1 + p.ff_bias;
        n_single = 0; n_mine = 0;
        for (int i0 = 0; i0 < n_all; i0 += 64) {
          const int i = i0 + lane;
          const int ff = i < n_all ? p.first_frame[l0 + i] : 0x7fffffff;
          n_single = max(n_single, prefix_end(__ballot(ff <= t_last), i0));   // (see gmm_bf16_kernel: superset for grouped plans)
          n_mine = max(n_mine, prefix_end(__ballot(ff <= t_mine), i0));
        }
      }
      float *out = p.out + p.ll_off[utt];
      if (n_single > 0) {
        // ---- x̃ = [x, x²] of this wavefront's 64 frames, split into bf16 triples: b[tile][step][piece], lane (frame, half)
        op8 b[kNT][kSteps][kPieces];
        // kHalf: `bad` = a scaled feature outside the f16 range (or NaN)
        const bool bad = split_features<kSteps, kPieces>(p, f0, T, t_base, col, h, b);
        if constexpr (kHalf) {
          if (__syncthreads_or(bad)) {                     // uniform: the whole tile goes to the bf16×3 pass
            if (threadIdx.x == 0) { p.redo[(size_t)utt * p.tiles + tl] = 1; atomicAdd(p.redo_count, 1); }
            continue;
          }
        }
        const uint4 *wsrc = kHalf ? p.wh : p.wb;
        const float *gsrc = kHalf ? p.gch : p.gc;
        const float inv_s = kHalf ? p.acc_scale_inv : 1.0f;
        const float l2e_s = 1.44269504088896341f * inv_s;  // inv_s is a power of two: (x·inv_s)·log2e == x·(log2e·inv_s)
        // Block copy global → LDS without a register stop (global_load_lds_dwordx4: every lane's 16 bytes land at a
        // wavefront-uniform LDS base + 16·lane, which is exactly the linear unit order of a block).
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        auto fetch = [&](int blk, int buf) {
          const uint4 *src = wsrc + (size_t)blk * kUnits;
#pragma unroll
          for (int i = 0; i < kLoads; i++) {
            const int u0 = 64 * wave_u + 256 * i;        // first unit this wavefront moves in round i (uniform)
            if (u0 < kUnits)
              __builtin_amdgcn_global_load_lds((gptr_t)(src + u0 + lane), (lptr_t)&a_lds[buf][u0], 16, 0, 0);
          }
          if (wave_u == 0 && lane < 8)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)blk * 32 + 4 * lane), (lptr_t)&gc_lds[buf][0], 16, 0, 0);
        };
        auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // ---- block loop, software-pipelined inside the wavefront.  An 8-pass MFMA holds the matrix pipe for 32 cycles but the
        // issue port for 4; a wavefront that issues its MFMAs back to back and its log-sum-exp afterwards leaves one of the
        // two idle in turn, and the two wavefronts of a SIMD fall into step (whoever leads is slowed by sharing, whoever lags
        // runs alone and catches up), so nothing overlaps.  Here the epilogue of block j-1 is cut into ≤ 7-instruction
        // chunks and one chunk follows each MFMA of block j in program order (sched_barrier pins it): every stretch of the
        // instruction stream keeps both the matrix pipe and the VALU busy.  Two accumulator sets alternate by block parity.
        f32x16 acc2[2][kNT];
#pragma unroll
        for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
          for (int n = 0; n < kNT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[q2][n][r] = 0.0f;
        float mxv[kNT] = {0.0f, 0.0f}, smv[kNT] = {1.0f, 1.0f}, tm[8];
        f32x2 ex[8];
        constexpr int kChunks = 27;
        // chunk c of the epilogue of the block held in pv; results are bit-identical to reg_max / reg_expsum_fast / finish
        auto epi = [&](int c, const f32x16 (&pv)[kNT], int column) {
#ifdef BF16_DIAG_NO_EPILOGUE
          if (c == 26) { smv[0] = pv[0][0] + pv[1][15]; stage[(32 * h + col) * 33 + column] = smv[0]; }
#else
          const int n = (c < 3 || (c >= 6 && c < 16)) ? 0 : 1;           // tile the chunk works on
          if (c == 0 || c == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) tm[r] = fmaxf(pv[n][r], pv[n][r + 8]);
          } else if (c == 1 || c == 4) {
#pragma unroll
            for (int r = 0; r < 4; r++) tm[r] = fmaxf(tm[r], tm[r + 4]);
            tm[0] = fmaxf(tm[0], tm[2]); tm[1] = fmaxf(tm[1], tm[3]);
            tm[0] = fmaxf(tm[0], tm[1]);
          } else if (c == 2 || c == 5) {
            mxv[n] = fmaxf(tm[0], swap32(tm[0], h));
          } else if ((c >= 6 && c < 14) || (c >= 16 && c < 24)) {
            const int g = c < 14 ? c - 6 : c - 16;
            const f32x2 x = {pv[n][2 * g], pv[n][2 * g + 1]};
            const f32x2 mv2 = {mxv[n], mxv[n]};
            const f32x2 lv = {l2e_s, l2e_s};
            const f32x2 arg = (x - mv2) * lv;
            ex[g].x = __builtin_amdgcn_exp2f(arg.x);
            ex[g].y = __builtin_amdgcn_exp2f(arg.y);
          } else if (c == 14 || c == 24) {
#pragma unroll
            for (int w = 1; w < 8; w <<= 1)
#pragma unroll
              for (int r = 0; r + w < 8; r += 2 * w) ex[r] += ex[r + w];
          } else if (c == 15 || c == 25) {
            const float sv = ex[0].x + ex[0].y;
            smv[n] = sv + swap32(sv, h);
          } else if (c == 26) {
            stage[(32 * h + col) * 33 + column] = finish((h ? mxv[1] : mxv[0]) * inv_s, h ? smv[1] : smv[0]);
          }
#endif
        };
        auto flush = [&](int jdone) {                        // columns [jdone − jdone%32, jdone] of the staged scores → HBM
          const int jj = jdone & 31;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int j0 = jdone - jj, cnt = jj + 1;
#pragma unroll 4
          for (int i = 0; i < 32; i++) {
            const int r = h + 2 * i, t = t_base + r;
            if (col < cnt && t < T) __builtin_nontemporal_store(stage[r * 33 + col], &out[(size_t)t * P + j0 + col]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        };
        for (int c0 = 0; c0 < n_single; c0 += kBlkCache) {
        const int c1 = min(n_single, c0 + kBlkCache);
        __syncthreads();                                   // previous chunk's table is no longer read
        for (int i = c0 + threadIdx.x; i < c1; i += 256) blk_lds[i - c0] = p.row0[list[i]] >> 5;
        __syncthreads();
        auto block_of = [&](int jj) { return blk_lds[min(jj, c1 - 1) - c0]; };
        fetch(block_of(c0), 0);
        landed();
        __syncthreads();
        // one trip: block j (parity par: c0 is even, so par is also the LDS buffer) is multiplied into acc2[par] while the
        // epilogue of block j-1 runs out of acc2[par ^ 1]
        auto trip = [&](auto par_c, int j) {
          constexpr int par = decltype(par_c)::value;
          constexpr int buf = par;
#if !defined(BF16_DIAG_NO_FLUSH) && !defined(GMM_FLUSH_AT_END)
          // A full window of 32 staged columns (its last one, block j-2's, was written during the previous trip) goes to
          // HBM at the START of a trip: stores share vmcnt with the block copy, and this way they have a whole trip to be
          // acknowledged before landed() waits on the counter — issued at the end of a trip they were waited for at once.
          if (active && j < n_mine && j > 1 && ((j - 2) & 31) == 31) flush(j - 2);
#endif
#ifndef BF16_DIAG_NO_FETCH   // timing-only builds (tools/gmm_ablation.sh): results are wrong by construction
          fetch(block_of(j + 1), buf ^ 1);                 // block j+1 (the chunk's last trip re-fetches its last block: harmless)
#endif
          if (active && j < n_mine) {
            f32x16 (&cur)[kNT] = acc2[par];
            const f32x16 (&prev)[kNT] = acc2[par ^ 1];
            f32x16 init;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
              const float4 gq = *reinterpret_cast<const float4 *>(&gc_lds[buf][8 * qq + 4 * h]);
              init[4 * qq] = gq.x; init[4 * qq + 1] = gq.y; init[4 * qq + 2] = gq.z; init[4 * qq + 3] = gq.w;
            }
            const int column = j == 0 ? 32 : ((j - 1) & 31); // the first block of an item has no predecessor: padding column
            // operand pieces of step s+1 are read from LDS while step s is multiplied
            auto read_a = [&](int s, op8 (&a)[kPieces]) {
#pragma unroll
              for (int qq = 0; qq < kPieces; qq++)
                a[qq] = __builtin_bit_cast(op8, a_lds[buf][((s * kPieces + qq) * 2 + h) * 32 + col]);
            };
            op8 a_cur[kPieces], a_nxt[kPieces];
            read_a(0, a_cur);
            // six (three) products per 16 k-values, smallest terms first; the two tiles alternate so that consecutive MFMAs
            // never wait on each other's accumulator
            constexpr int kProd = kHalf ? 3 : 6;
            constexpr int kStride = (kSteps * kProd * kNT) / 30;   // MFMA slots per epilogue chunk
            constexpr int pa[6] = {kHalf ? 1 : 2, kHalf ? 0 : 1, 0, 1, 0, 0}, pb[6] = {0, 1, kHalf ? 0 : 2, 0, 1, 0};
#pragma unroll
            for (int s = 0; s < kSteps; s++) {
              if (s + 1 < kSteps) read_a(s + 1, a_nxt);
#pragma unroll
              for (int t6 = 0; t6 < kProd; t6++)
#pragma unroll
                for (int n = 0; n < kNT; n++) {
                  const f32x16 &cin = (s == 0 && t6 == 0) ? init : cur[n];
                  if constexpr (kHalf)
                    cur[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[pa[t6]], b[n][s][pb[t6]], cin, 0, 0, 0);
                  else
                    cur[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[pa[t6]], b[n][s][pb[t6]], cin, 0, 0, 0);
                  const int slot = (s * kProd + t6) * kNT + n;
                  if (slot % kStride == 0 && slot / kStride < kChunks) epi(slot / kStride, prev, column);
#ifndef BF16_DIAG_NO_INTERLEAVE
                  __builtin_amdgcn_sched_barrier(0);
#endif
                }
#pragma unroll
              for (int qq = 0; qq < kPieces; qq++) a_cur[qq] = a_nxt[qq];
            }
#ifdef BF16_DIAG_NO_FLUSH
            if (smv[0] == 12345.678f) flush(j - 1);          // timing-only build: keeps the staged values alive
#else
#ifdef GMM_FLUSH_AT_END
            if (j > 0 && ((j - 1) & 31) == 31) flush(j - 1);
#endif
#endif
          }
          landed();
#ifndef BF16_DIAG_NO_BARRIER
          __syncthreads();                                 // block j+1 is in place; everybody is done with block j
#endif
        };
        for (int j = c0; j < c1; j += 2) {
          trip(std::integral_constant<int, 0>{}, j);
          if (j + 1 < c1) trip(std::integral_constant<int, 1>{}, j + 1);
        }
        }
        if (active && n_mine > 0) {                          // drain: the last block's epilogue and the open columns
          const int jp = n_mine - 1;
#if !defined(BF16_DIAG_NO_FLUSH) && !defined(GMM_FLUSH_AT_END)
          if (jp > 0 && ((jp - 1) & 31) == 31) flush(jp - 1);   // a window completed by the last trip is still staged
#endif
          if (jp & 1) {
#pragma unroll
            for (int c = 0; c < kChunks; c++) epi(c, acc2[1], jp & 31);
          } else {
#pragma unroll
            for (int c = 0; c < kChunks; c++) epi(c, acc2[0], jp & 31);
          }
#ifdef BF16_DIAG_NO_FLUSH
          if (smv[0] == 12345.678f)
#endif
          flush(jp);
        }
      }
    }
  }
}

// The same kernel for the small-slot classes: pdfs of at most kSlot ∈ {16, 8, 4} Gaussians occupy kSlot consecutive model
// rows (pad rows: zero weights, gconst −1e30), and 32 / kSlot of them — whichever the utterance's list puts next to each
// other — are gathered into one virtual 32-row block: global_load_lds takes a per-lane source address, so the copy costs
// what the contiguous one does.  The MFMAs are those of the 32-row class; the log-sum-exp runs over the kSlot rows of each
// pdf (accumulator registers [8k, 8k+8) of both half-waves for kSlot = 16, [4k, 4k+4) for 8, [4i, 4i+4) of ONE half-wave
// for 4) and a block yields 32 / kSlot score columns.  Not software-pipelined (the epilogues differ per class and these
// classes are a minority of the rows of a 32-Gaussian model; for MFA's released models they are the majority — next step).
template <int kSteps, int kPieces, int kSlot>
__global__ __launch_bounds__(256, 2) void gmm_split_small_kernel(GmmParams p) {
  constexpr int kNT = 2, kWaves = 4, kFramesPerWave = 64, kFramesPerTile = 256;
  constexpr bool kHalf = kPieces == 2;
  using op8 = std::conditional_t<kHalf, f16x8, bf16x8>;
  constexpr int kUnits = kSteps * kPieces * 2 * 32;    // 16-byte units per block
  constexpr int kLoads = (kUnits + 255) / 256;         // units each thread moves per block
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float stage_all[kWaves][64 * 33];
  __shared__ uint4 a_lds[2][kUnits];
  __shared__ __attribute__((aligned(16))) float gc_lds[2][64];
  constexpr int kPdfs = 32 / kSlot;                   // pdfs per virtual block = score columns per block
  constexpr int kCls = kSlot == 16 ? 2 : kSlot == 8 ? 3 : 4;   // position of this class in class_counts
  constexpr int kBlkCache = 1024;                     // pdf → first model row, staged per item (two dependent global loads
  __shared__ int blk_lds[kBlkCache];                  // per pdf must not sit in the block loop); a multiple of 32 pdfs
  __shared__ int s_item;
  float *stage = stage_all[wave];
  if (!kHalf && p.redo_mode == 2 && *p.redo_count == 0) return;   // uniform: the f16 pass declined nothing
  const int my_xcd = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);
  for (int hop = 0; hop < 8; hop++) {
    const int q = (my_xcd + hop) & 7;
    const int n_items = ((p.n_utt - q + 7) >> 3) * p.tiles;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_item = atomicAdd(&p.queue[q], 1);
      __syncthreads();
      const int item = s_item;
      if (item >= n_items) break;
      int lane = lane0;                                // opaque per item: keeps lane-dependent addresses out of long-lived registers
      asm volatile("" : "+v"(lane));
      const int col = lane & 31, h = lane >> 5;
      const int utt = (item / p.tiles) * 8 + q, tl = p.tiles - 1 - item % p.tiles;
      const int64_t f0 = p.frame_off[utt];
      const int T = (int)(p.frame_off[utt + 1] - f0);
      if (tl * kFramesPerTile >= T) continue;          // uniform over the workgroup
      if (!kHalf && p.redo_mode == 2 && p.redo[(size_t)utt * p.tiles + tl] == 0) continue;   // only what the f16 pass left
      const int t_base = (tl * kWaves + wave) * kFramesPerWave;
      const bool active = t_base < T;                  // a wavefront past the end still helps move blocks and joins barriers
      const int64_t l0 = p.pdf_off[utt];
      const int P = (int)(p.pdf_off[utt + 1] - l0);
      const int32_t *list = p.pdf_list + l0;
      const int32_t *cc6 = p.class_counts + (size_t)utt * 6;
      int base = cc6[0] + cc6[1];                      // columns of the classes in front of this one
#pragma unroll
      for (int q3 = 2; q3 < kCls; q3++) base += cc6[q3];
      const int n_all = cc6[kCls];
      if (n_all == 0) continue;                        // uniform
      // n_single: pdfs the tile's LAST frame can be asked for — the prefix the workgroup walks together (block copies and
      // barriers are collective).  n_mine: the shorter prefix this wavefront's own 64 frames can be asked for; beyond it
      // the wavefront only helps with the copies.
      int n_single = n_all, n_mine = n_all;
      if (p.first_frame) {
        const int t_last = min(T, (tl + 1) * kFramesPerTile) - 1 + p.ff_bias;
        const int t_mine = min(T, t_base + kFramesPerWave) - 1 + p.ff_bias;
        n_single = 0; n_mine = 0;
        for (int i0 = 0; i0 < n_all; i0 += 64) {
          const int i = i0 + lane;
          const int ff = i < n_all ? p.first_frame[l0 + base + i] : 0x7fffffff;
          n_single = max(n_single, prefix_end(__ballot(ff <= t_last), i0));   // (see gmm_bf16_kernel: superset for grouped plans)
          n_mine = max(n_mine, prefix_end(__ballot(ff <= t_mine), i0));
        }
      }
      float *out = p.out + p.ll_off[utt];
      if (n_single > 0) {
        // ---- x̃ = [x, x²] of this wavefront's 64 frames, split into bf16 triples: b[tile][step][piece], lane (frame, half)
        op8 b[kNT][kSteps][kPieces];
        // kHalf: `bad` = a scaled feature outside the f16 range (or NaN)
        const bool bad = split_features<kSteps, kPieces>(p, f0, T, t_base, col, h, b);
        if constexpr (kHalf) {
          if (__syncthreads_or(bad)) {                     // uniform: the whole tile goes to the bf16×3 pass
            if (threadIdx.x == 0) { p.redo[(size_t)utt * p.tiles + tl] = 1; atomicAdd(p.redo_count, 1); }
            continue;
          }
        }
        const uint4 *wsrc = kHalf ? p.wh : p.wb;
        const float *gsrc = kHalf ? p.gch : p.gc;
        const float inv_s = kHalf ? p.acc_scale_inv : 1.0f;
        const float l2e_s = 1.44269504088896341f * inv_s;  // inv_s is a power of two: (x·inv_s)·log2e == x·(log2e·inv_s)
        // Block copy global → LDS without a register stop (global_load_lds_dwordx4: every lane's 16 bytes land at a
        // wavefront-uniform LDS base + 16·lane, which is exactly the linear unit order of a block).
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // virtual block jb = pdfs [jb·kPdfs, (jb+1)·kPdfs) of the class; lane ↔ row ρ = lane mod 32 of every 32-unit group
        const int rho = lane & 31, my_k = rho / kSlot, my_r = rho % kSlot;
        auto fetch = [&](int jb, int buf, int c0, int c1) {
          const int idx = jb * kPdfs + my_k;               // pdf this lane's row belongs to (class-relative)
          const int row = idx < c1 ? blk_lds[idx - c0] + my_r : p.num_rows;   // past the needed prefix: the dummy row
          const uint4 *src = wsrc + (size_t)(row >> 5) * kUnits + (row & 31);
#pragma unroll
          for (int i = 0; i < kLoads; i++) {
            const int u0 = 64 * wave_u + 256 * i;        // first unit this wavefront moves in round i (uniform)
            if (u0 < kUnits)
              __builtin_amdgcn_global_load_lds((gptr_t)(src + ((u0 + lane) & ~31)), (lptr_t)&a_lds[buf][u0], 16, 0, 0);
          }
          if (wave_u == 0)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + row), (lptr_t)&gc_lds[buf][0], 4, 0, 0);
        };
        auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // ---- block loop: multiply, reduce per pdf, stage one column per pdf, flush every 32 columns
        const int nb_mine = (n_mine + kPdfs - 1) / kPdfs;   // virtual blocks this wavefront multiplies
        auto flush = [&](int col_last) {                     // columns [col_last − col_last%32, col_last] → HBM
          const int jj = col_last & 31;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int j0 = col_last - jj, cnt = min(jj + 1, n_mine - j0);
#pragma unroll 4
          for (int i = 0; i < 32; i++) {
            const int r = h + 2 * i, t = t_base + r;
            if (col < cnt && t < T) __builtin_nontemporal_store(stage[r * 33 + col], &out[(size_t)t * P + base + j0 + col]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        };
        // Σ exp(x − m) over `cnt` registers from r0, pairwise; m: their maximum
        auto group_max = [&](const f32x16 &v, int r0, int cnt) {
          float m = v[r0];
#pragma unroll
          for (int r = 1; r < cnt; r++) m = fmaxf(m, v[r0 + r]);
          return m;
        };
        auto group_expsum = [&](const f32x16 &v, int r0, int cnt, float m) {
          float e[8];
#pragma unroll
          for (int r = 0; r < cnt; r++) e[r] = __builtin_amdgcn_exp2f((v[r0 + r] - m) * l2e_s);
#pragma unroll
          for (int w = 1; w < cnt; w <<= 1)
#pragma unroll
            for (int r = 0; r + w < cnt; r += 2 * w) e[r] += e[r + w];
          return e[0];
        };
        int pending = -1;                                  // last column of a staged window waiting to be written out
        for (int c0 = 0; c0 < n_single; c0 += kBlkCache) {
        const int c1 = min(n_single, c0 + kBlkCache);
        __syncthreads();                                   // previous chunk's table is no longer read
        for (int i = c0 + threadIdx.x; i < c1; i += 256) blk_lds[i - c0] = p.row0[list[base + i]];
        __syncthreads();
        const int jb0 = c0 / kPdfs, jb1 = (c1 + kPdfs - 1) / kPdfs;
        fetch(jb0, 0, c0, c1);
        landed();
        __syncthreads();
        for (int jb = jb0; jb < jb1; jb++) {
          const int buf = (jb - jb0) & 1;
          if (pending >= 0) { flush(pending); pending = -1; }   // a block early: see "score stores" in gmm_split_single_kernel
          fetch(min(jb + 1, jb1 - 1), buf ^ 1, c0, c1);
          if (active && jb < nb_mine) {
            f32x16 acc[kNT];
            multiply_block<kSteps, kPieces>(a_lds[buf], gc_lds[buf], b, acc, col, h);
            // ---- per-pdf log-sum-exp.  Accumulator register r of half-wave h is row (r & 3) + 8 (r >> 2) + 4 h.
            const int colbase = (jb * kPdfs) & 31;           // first staging column of this block
#pragma unroll
            for (int n = 0; n < kNT; n++) {
              float *srow = stage + (32 * n + col) * 33 + colbase;
              if constexpr (kSlot == 16) {                   // pdf k: rows 16k..16k+15 = registers [8k, 8k+8) of both halves
                float ll[2];
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                  float m = group_max(acc[n], 8 * k2, 8);
                  m = fmaxf(m, swap32(m, h));
                  float sv = group_expsum(acc[n], 8 * k2, 8, m);
                  sv += swap32(sv, h);
                  ll[k2] = finish(m * inv_s, sv);
                }
                srow[h] = h ? ll[1] : ll[0];                 // each half-wave stores one of the two columns
              } else if constexpr (kSlot == 8) {             // pdf k: rows 8k..8k+7 = registers [4k, 4k+4) of both halves
                float ll[4];
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) {
                  float m = group_max(acc[n], 4 * k2, 4);
                  m = fmaxf(m, swap32(m, h));
                  float sv = group_expsum(acc[n], 4 * k2, 4, m);
                  sv += swap32(sv, h);
                  ll[k2] = finish(m * inv_s, sv);
                }
                srow[h] = h ? ll[1] : ll[0];
                srow[2 + h] = h ? ll[3] : ll[2];
              } else {                                       // kSlot 4: pdf 2i + h: rows 8i + 4h .. +3 = registers [4i, 4i+4)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                  const float m = group_max(acc[n], 4 * i, 4);
                  const float sv = group_expsum(acc[n], 4 * i, 4, m);
                  srow[2 * i + h] = finish(m * inv_s, sv);
                }
              }
            }
            const int col_last = min((jb + 1) * kPdfs, n_mine) - 1;   // last valid column this block produced
            if ((col_last & 31) == 31 || jb == nb_mine - 1) pending = col_last;
          }
          landed();
          __syncthreads();                                 // block jb+1 is in place; everybody is done with block jb
        }
        }
        if (pending >= 0) flush(pending);
      }
    }
  }
}

// Lazy scoring, once per batch: the first packed model row of every score column (saves the pdf id → row lookup, one
// dependent load per model block and in every wavefront's start-up chain).
__global__ void gmm_col_rows_kernel(GmmParams p, int32_t *out) {
  const int utt = blockIdx.x;
  const int64_t l0 = p.pdf_off[utt], l1 = p.pdf_off[utt + 1];
  for (int64_t j = l0 + threadIdx.x; j < l1; j += blockDim.x) {
    const int pdf = p.pdf_list[j];
    int r = p.row0[pdf];
    if (p.col_nb_packed) { const int nb = p.nblk[pdf]; if (nb > 1) r |= nb - 1; }
    out[j] = r;
  }
}

// Lazy scoring, once per window: the band's index range [lo, hi) in every run of class 0 (slots 0..groups-1; one run when the
// plan is not grouped), in classes 2, 3, 4 and in class 1 (the slots after the runs'), relative to the class's first column — what every scoring
// wavefront of the sub-tile would otherwise search for itself (two dependent memory trips each).  One wavefront per utterance.
constexpr int kRunSlots = kMfaRunSlots;   // slots 0..kRunSlots-1: runs of class 0; then classes 2, 3, 4; then class 1; then class 5
constexpr int kRangeSlots = kMfaRangeSlots;
__global__ __launch_bounds__(256) void gmm_band_ranges_kernel(GmmParams p) {
  const int lane = threadIdx.x & 63;
  const int utt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (utt >= p.n_utt) return;
  if (p.b_t_begin > 0 && p.b_done && p.b_done[(size_t)utt * p.b_done_stride + p.b_done_word] != 0) return;
  const int64_t l0 = p.pdf_off[utt];
  const int32_t *cc6 = p.class_counts + (size_t)utt * 6;
  const Band bd = band_of(p, utt);
  const int runs = p.groups > 1 ? p.groups : 1;
  int32_t *out = p.ranges + (size_t)utt * kRangeSlots * 2;
  int off = 0;
  for (int slot = 0; slot < kRangeSlots; slot++) {
    int cnt = 0, base = 0;     // the run searched, its first column relative to the class, the class's first column in `off`
    if (slot < kRunSlots) {
      if (slot >= runs) continue;
      if (p.groups > 1) { const int32_t *gc = p.group_counts + (size_t)utt * p.groups; for (int g = 0; g < slot; g++) base += gc[g]; cnt = gc[slot]; }
      else cnt = cc6[0];
      off = 0;
    } else if (slot == kRunSlots + 3) {
      off = cc6[0];
      cnt = cc6[1];
    } else if (slot == kRunSlots + 4) {
      off = cc6[0] + cc6[1] + cc6[2] + cc6[3] + cc6[4];
      cnt = cc6[5];
    } else {
      const int cls = slot - kRunSlots + 2;
      off = cc6[0] + cc6[1];
      for (int k = 2; k < cls; k++) off += cc6[k];
      cnt = cc6[cls];
    }
    int nh = 0, nl = 0;
    for (int i0 = 0; i0 < cnt; i0 += 256) {
      int ff[4], ld[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = min(i0 + 64 * u + lane, cnt - 1);
        ff[u] = p.first_frame[l0 + off + base + i]; ld[u] = p.last_depth[l0 + off + base + i];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool in = i0 + 64 * u + lane < cnt;
        nh += __popcll(__ballot(in && ff[u] <= bd.hi));
        nl += __popcll(__ballot(in && ld[u] < bd.lo));
      }
    }
    if (lane == 0) { out[2 * slot] = base + min(nl, nh); out[2 * slot + 1] = base + nh; }
  }
}

// Index of the 64-frame tile `tile` of utterance `utt` in the pre-split operand buffer: ⌊frame_off/64⌋ + utt + tile is
// monotone and leaves every utterance room for ⌈T/64⌉ tiles without a separate offset table.
__device__ __forceinline__ int64_t xsplit_tile_index(int64_t frame_off_u, int utt, int tile) { return (frame_off_u >> 6) + utt + tile; }

// Lazy scoring pre-pass: the f16 hi/lo operands [x, x²]·scale of every 64-frame tile, in the register layout the band
// kernel's MFMAs read (b[n][step][piece] of lane l at ((n·kSteps + step)·2 + piece)·64 + l), so that a wavefront starts
// a window with twenty coalesced 1 KiB loads instead of 160 strided 4-byte loads and the split arithmetic — the values are
// those of split_features bit for bit.  One wavefront per tile.
template <int kSteps>
__global__ __launch_bounds__(256) void gmm_presplit_kernel(GmmParams p, uint4 *out, int *bad_out, int tiles_per_utt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const int utt = (int)(item / tiles_per_utt), tile = (int)(item - (int64_t)utt * tiles_per_utt);
  if (utt >= p.n_utt) return;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  if (tile * 64 >= T) return;
  f16x8 b[2][kSteps][2];
  const bool bad = split_features<kSteps, 2>(p, f0, T, tile * 64, lane & 31, lane >> 5, b);
  const int64_t ti = xsplit_tile_index(f0, utt, tile);
  uint4 *dst = out + ti * (2 * kSteps * 2 * 64) + lane;
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int s_ = 0; s_ < kSteps; s_++)
#pragma unroll
      for (int q = 0; q < 2; q++) dst[((n * kSteps + s_) * 2 + q) * 64] = __builtin_bit_cast(uint4, b[n][s_][q]);
  const bool any_bad = __ballot(bad) != 0ull;
  if (lane == 0) bad_out[ti] = any_bad ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Lazy (windowed) scoring — mfa_gmm_score_window.  Kaldi evaluates its decodable lazily: a score exists only if a live
// token's arc asked for it.  The dense kernels above score every pdf of the utterance's graph for every frame from the
// pdf's first reachable frame on — measured, ≈9× more cells than the decoder reads.  Here the decoder runs in windows of
// K frames and publishes, at each window end, the band of graph depths its live tokens can reach within K arcs; the
// kernel below scores, for the window's frames, only the pdfs whose arcs leave states inside that band.
//
// With so few frames per (utterance, pdf) there is nothing to share a model block across: one wavefront owns one
// (utterance, 64-frame sub-tile), keeps its x̃ operands in registers (as above) and streams the band's model blocks
// straight from L2 / Infinity Cache into its A registers — 10 coalesced 1 KiB loads per 32-row block, each operand
// register re-loaded for the next block as soon as the MFMAs that read it have been issued.  Per block: 30 (60) MFMAs,
// the log-sum-exp, one staged score column.  Arithmetic per cell is that of gmm_split_single_kernel exactly (same operand
// split, same product order, same epilogue expressions): a cell scored here is bit-identical to the dense kernel's.
// Bound: the 10 KiB (15 KiB) of operands per block and 64 frames — fabric bandwidth, not the matrix pipe.
template <int kSteps, int kPieces>
__global__ __launch_bounds__(256, 2) void gmm_band_kernel(GmmParams p) {
  constexpr bool kHalf = kPieces == 2;
  using op8 = std::conditional_t<kHalf, f16x8, bf16x8>;
  constexpr int kUnits = kSteps * kPieces * 2 * 32;    // 16-byte units per block
  __shared__ float stage_all[4][64 * 33];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *stage = stage_all[wave];
  const int grp = p.b_split ? band_split_block(p).y : 0;                          // the run of class 0 this wavefront scores
#ifdef GMM_BAND_STAMPS
  // phase accounting (-DGMM_BAND_STAMPS, buffer from mfa_debug_gmm_trace): Σ 100 MHz ticks of {item + band search, feature
  // split, block loop}, wavefronts with work, blocks
  const unsigned long long st0 = wall_clock64();
#endif
  int utt, r, chunk = 0;
  if (!band_item(p, wave, utt, r, &chunk)) return;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int t_base = band_t_begin(p, utt) + 64 * r;
  if (t_base >= T) return;
  int *redo_flag = p.redo + ((size_t)utt * p.b_sub + r) * p.b_nchunk + chunk;
  if (!kHalf && p.redo_mode == 2 && *redo_flag == 0) return;          // only what the f16 pass declined
  const int col = lane & 31, h = lane >> 5;
  const int64_t l0 = p.pdf_off[utt];
  const int P = (int)(p.pdf_off[utt + 1] - l0);
  const int32_t *list = p.pdf_list + l0;
  const int32_t *cc6 = p.class_counts + (size_t)utt * 6;
  const Band bd = band_of(p, utt);
  // The pre-split operands depend on nothing but the sub-tile: requested first, they travel while the band is looked up
  // (volatile: the loads stay here instead of sinking below the early exit).
  op8 b[2][kSteps][kPieces];
  bool bad = false;
  const bool presplit = kHalf && p.xsplit != nullptr;
  if (presplit) {
    const int64_t ti = xsplit_tile_index(f0, utt, t_base >> 6);
    const uint4 *src = p.xsplit + ti * (2 * kSteps * 2 * 64) + lane;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++)
#pragma unroll
        for (int q = 0; q < kPieces; q++) {
#ifndef GMM_BAND_X_STREAMING
          uint4 v;
          const volatile uint4 *a4 = src + ((n * kSteps + s_) * 2 + (q & 1)) * 64;
          v.x = a4->x; v.y = a4->y; v.z = a4->z; v.w = a4->w;
#else     // measured (-DGMM_BAND_X_STREAMING): non-temporal tile loads keep more of the model in L2 (FETCH_SIZE 30 -> 24.5 GB per
          // step) but the kernel is 3 % slower, the throughput unchanged
          typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
          const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(src + ((n * kSteps + s_) * 2 + (q & 1)) * 64));
#endif
          b[n][s_][q] = __builtin_bit_cast(op8, v);
        }
    bad = p.xsplit_bad[ti] != 0;
  }
  // band range [lo, hi) of every class this kernel scores: 0 (one 32-row block per pdf) and 2, 3, 4 (16-, 8-, 4-row slots);
  // classes 1 (pdfs of more than 32 Gaussians) and 5 (single Gaussians) are the f32 band kernel's
  int lo_c[5], hi_c[5], base_c[5];
  if (p.ranges) {   // looked up once per utterance and window by gmm_band_ranges_kernel
    const int32_t *rg = p.ranges + (size_t)utt * kRangeSlots * 2;
    int off = 0;
#pragma unroll
    for (int cls = 0; cls < 5; cls++) {
      const int slot = cls == 0 ? grp : (cls == 1 ? kRunSlots + 3 : kRunSlots + cls - 2);
      base_c[cls] = off; off += cc6[cls];
      lo_c[cls] = rg[2 * slot]; hi_c[cls] = rg[2 * slot + 1];
    }
  } else {
    int off = 0;
#pragma unroll
    for (int cls = 0; cls < 5; cls++) {
      const int cnt_cls = cc6[cls];
      int cnt = cnt_cls, seg = 0;       // the run searched: the whole class, or (class 0 of a grouped plan) this wavefront's run
      if (cls == 0 && p.b_split) {
        const int32_t *gc = p.group_counts + (size_t)utt * p.groups;
        for (int g = 0; g < grp; g++) seg += gc[g];
        cnt = gc[grp];
      }
      int nh = 0, nl = 0;
      {
        // four chunks per trip, every load issued before the first ballot waits (one memory round trip per 256 pdfs, not four)
        for (int i0 = 0; i0 < cnt; i0 += 256) {
          int ff[4], ld[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = min(i0 + 64 * u + lane, cnt - 1);
            ff[u] = p.first_frame[l0 + off + seg + i]; ld[u] = p.last_depth[l0 + off + seg + i];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const bool in = i0 + 64 * u + lane < cnt;
            nh += __popcll(__ballot(in && ff[u] <= bd.hi));
            nl += __popcll(__ballot(in && ld[u] < bd.lo));
          }
        }
      }
      lo_c[cls] = seg + min(nl, nh); hi_c[cls] = seg + nh; base_c[cls] = off;
      off += cnt_cls;
    }
  }
  if (p.b_chunk > 0) {                                 // list passes: this wavefront's share of the band (class 0 in chunks,
    lo_c[0] += chunk * p.b_chunk;                      // the small-slot classes with chunk 0)
    hi_c[0] = min(hi_c[0], lo_c[0] + p.b_chunk);
    if (chunk != 0) { hi_c[2] = lo_c[2]; hi_c[3] = lo_c[3]; hi_c[4] = lo_c[4]; }
  }
  if (p.b_split) {
    // the small-slot classes have no runs: the virtual blocks (32 / slot pdfs each) of their three bands, laid end to end,
    // are cut into `groups` pieces, one per wavefront of the sub-tile — all of them on run 0's wavefront would be all of
    // them on one XCD, and a piece of every class on every wavefront (the first version) was three pipelines to fill and
    // drain per wavefront, three or four blocks each: a piece now lies inside one class, rarely two
    int nb_c[5], tot = 0;
#pragma unroll
    for (int cls = 2; cls < 5; cls++) {
      const int kp = cls == 2 ? 2 : (cls == 3 ? 4 : 8);
      nb_c[cls] = (hi_c[cls] + kp - 1) / kp - lo_c[cls] / kp;
      if (lo_c[cls] >= hi_c[cls]) nb_c[cls] = 0;
      tot += nb_c[cls];
    }
    const int per = (tot + p.groups - 1) / p.groups;
    const int w0 = grp * per, w1 = min(tot, w0 + per);
    int pos = 0;
#pragma unroll
    for (int cls = 2; cls < 5; cls++) {
      const int kp = cls == 2 ? 2 : (cls == 3 ? 4 : 8);
      const int jb0 = lo_c[cls] / kp;
      const int a = jb0 + max(w0 - pos, 0), b = jb0 + min(w1 - pos, nb_c[cls]);
      pos += nb_c[cls];
      if (a >= b) hi_c[cls] = lo_c[cls];
      else { lo_c[cls] = max(lo_c[cls], a * kp); hi_c[cls] = min(hi_c[cls], b * kp); }
    }
  }
  // class 1 (pdfs of more than 32 Gaussians, few): its band's columns go round the sub-tile's wavefronts one by one
  const int step1 = p.b_split ? p.groups : (p.b_chunk > 0 ? p.b_nchunk : 1);
  const int first1 = lo_c[1] + (p.b_split ? grp : (p.b_chunk > 0 ? chunk : 0));
  const int lo = lo_c[0], hi = hi_c[0];
  if (lo >= hi && first1 >= hi_c[1] && lo_c[2] >= hi_c[2] && lo_c[3] >= hi_c[3] && lo_c[4] >= hi_c[4]) {
    if (kHalf && lane == 0) *redo_flag = 0;
    return;
  }
#ifdef GMM_BAND_STAMPS
  const unsigned long long st1 = wall_clock64();
#endif
  if (!presplit) bad = split_features<kSteps, kPieces>(p, f0, T, t_base, col, h, b);
#ifdef GMM_BAND_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long st2 = wall_clock64();
#endif
  if constexpr (kHalf) {
    const bool any_bad = __ballot(bad) != 0ull;
    if (lane == 0) *redo_flag = any_bad ? 1 : 0;
    if (any_bad) return;                                               // a scaled feature left the f16 range: bf16×3 pass
  }
  const float inv_s = kHalf ? p.acc_scale_inv : 1.0f;
  const float l2e_s = 1.44269504088896341f * inv_s;
  float *out = p.out + p.ll_off[utt];
  constexpr int kProd = kHalf ? 3 : 6;
  constexpr int pa[6] = {kHalf ? 1 : 2, kHalf ? 0 : 1, 0, 1, 0, 0}, pb[6] = {0, 1, kHalf ? 0 : 2, 0, 1, 0};
  // `cnt` staged columns, the first of them score column c0 of the utterance's matrix → HBM as 128-byte row segments
  auto flush_cols = [&](int c0, int cnt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
      const int rr = h + 2 * i, t = t_base + rr;
      if (col < cnt && t < T) __builtin_nontemporal_store(stage[rr * 33 + col], &out[(size_t)t * P + c0 + col]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // ---------------------------------------------------------------- class 0: one pdf per 32-row block
  if (lo < hi) {
    const uint4 *wsrc = (kHalf ? p.wh : p.wb) + lane;
    const float *gsrc = (kHalf ? p.gch : p.gc) + 4 * h;
    const int last = hi - 1;
    // First model row per column, precomputed per batch: ONE load.  The pdf id → row chain it replaces (row0[list[j]]) made
    // the second load wait for the first — the youngest entry of the in-order vmcnt queue — i.e. drained every outstanding
    // operand load of the next block at the top of each block.
    const int32_t *crow = p.col_row0 + l0;
    auto row0_at = [&](int jj) { return crow[min(jj, last)]; };
    auto block_at = [&](int jj) {
      const int blk = __builtin_amdgcn_readfirstlane(row0_at(jj)) >> 5;
#ifdef BAND_DIAG_BLKMOD   // timing-only builds (-DBAND_DIAG_BLKMOD=8): every block from a cache-resident handful (wrong scores)
      return blk % BAND_DIAG_BLKMOD;
#else
      return blk;
#endif
    };
    op8 a[kSteps][kPieces];
    f32x4 g[4];
    {
      const int blk = block_at(lo);
      const uint4 *src = wsrc + (size_t)blk * kUnits;
#pragma unroll
      for (int q = 0; q < 4; q++) g[q] = *reinterpret_cast<const f32x4 *>(gsrc + (size_t)blk * 32 + 8 * q);
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++)
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
    }
    int blk_next = block_at(lo + 1);
#ifdef GMM_BAND_STAMPS
    unsigned long long ph[4] = {0, 0, 0, 0};   // shader-clock cycles: MFMA phase (issue + operand waits), lookup wait, log-sum-exp + stage, flush
#endif
    for (int j = lo; j < hi; j++) {
#ifdef GMM_BAND_STAMPS
      const unsigned long long c0_ = clock64();
#endif
      const int x_next2 = row0_at(j + 2);                      // lookup two blocks ahead (oldest entry of the vmcnt queue)
      f32x16 init, acc[2];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) init[rr] = g[rr >> 2][rr & 3];
      const uint4 *src = wsrc + (size_t)blk_next * kUnits;
      const float *gn = gsrc + (size_t)blk_next * 32;
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++) {
#pragma unroll
        for (int t6 = 0; t6 < kProd; t6++)
#pragma unroll
          for (int n = 0; n < 2; n++) {
            const f32x16 &cin = (s_ == 0 && t6 == 0) ? init : acc[n];
#ifdef BAND_DIAG_NO_MFMA   // timing-only builds: one product per step instead of six (results wrong by construction)
            if (t6 > 0) { if (s_ == 0 && t6 == 1) acc[n] = cin; continue; }
#endif
            if constexpr (kHalf) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
          }
        // this step's operand registers (and, after the first step, the gconst registers) are free: next block's rows.
        // (A second operand set — two blocks in flight per wavefront — was measured: 12.70 vs 12.76 ms per step; the
        //  kernel is bound by what the fabric delivers, ≈6.5 TB/s of 10 KiB blocks gathered from a 51 MB table.)
        if (s_ == 0) {
#pragma unroll
          for (int q = 0; q < 4; q++) g[q] = *reinterpret_cast<const f32x4 *>(gn + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef GMM_BAND_STAMPS
      const unsigned long long c1_ = clock64();
#endif
      blk_next = __builtin_amdgcn_readfirstlane(x_next2) >> 5;
#ifdef BAND_DIAG_BLKMOD
      blk_next %= BAND_DIAG_BLKMOD;
#endif
#ifdef GMM_BAND_STAMPS
      const unsigned long long c2_ = clock64();
#endif
      float mx[2], sum[2];
#pragma unroll
      for (int n = 0; n < 2; n++) {
#ifdef BAND_DIAG_NO_EPI   // timing-only builds: results are wrong by construction
        mx[n] = acc[n][0] + acc[n][15]; sum[n] = 1.0f;
#else
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, swap32(m, h));
        float sv = reg_expsum_fast(acc[n], m, l2e_s);
        sv += swap32(sv, h);
        mx[n] = m; sum[n] = sv;
#endif
      }
      const int jj = (j - lo) & 31;
      stage[(32 * h + col) * 33 + jj] = finish((h ? mx[1] : mx[0]) * inv_s, h ? sum[1] : sum[0]);
#ifdef GMM_BAND_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long c3_ = clock64();
#endif
      if (jj == 31 || j == last) flush_cols(j - jj, jj + 1);
#ifdef GMM_BAND_STAMPS
      const unsigned long long c4_ = clock64();
      ph[0] += c1_ - c0_; ph[1] += c2_ - c1_; ph[2] += c3_ - c2_; ph[3] += c4_ - c3_;
#endif
    }
#ifdef GMM_BAND_STAMPS
    if (p.trace && lane == 0 && kHalf) { atomicAdd(&p.trace[8], ph[0]); atomicAdd(&p.trace[9], ph[1]); atomicAdd(&p.trace[10], ph[2]); atomicAdd(&p.trace[11], ph[3]); }
#endif
  }

  // ---------------------------------------------------------------- class 1: several 32-row blocks per pdf
  // Online log-sum-exp over the pdf's blocks (running max M and sum S against it, per frame): (M, S) ← (max(M, m_b),
  // S·2^((M − M')·l2e) + s_b·2^((m_b − M')·l2e)).  Products and per-block reductions are class 0's; pad rows carry gconst
  // −1e30 and vanish in the sum.  No software pipeline: a trained model has a few such pdfs per band, if any.
  // Software pipeline as class 0's: the operands of the next block — the pdf's next one, or the first block of this
  // wavefront's next column — are requested as soon as a step's MFMAs have been issued; the column's (row, blocks) word is
  // looked up one column ahead.
  if (first1 < hi_c[1]) {
    const uint4 *wsrc = (kHalf ? p.wh : p.wb) + lane;
    const float *gsrc = (kHalf ? p.gch : p.gc) + 4 * h;
    const int32_t *crow1 = p.col_row0 + l0 + base_c[1];
    const int last1 = hi_c[1] - 1;
    auto col_word = [&](int jj) { return crow1[min(jj, last1)]; };     // row | (blocks − 1) when packed
    auto blocks_of = [&](int word, int jj) {
      return p.col_nb_packed ? (word & 31) + 1 : __builtin_amdgcn_readfirstlane(p.nblk[list[base_c[1] + min(jj, last1)]]);
    };
    int j1 = first1;
    int word = __builtin_amdgcn_readfirstlane(col_word(j1));
    int nb = blocks_of(word, j1), blk = word >> 5, bk = 0;
    int word_n = col_word(j1 + step1);                                 // stays a vector register until its column opens
    op8 a[kSteps][kPieces];
    f32x4 g[4];
    {
      const uint4 *src = wsrc + (size_t)blk * kUnits;
#pragma unroll
      for (int q = 0; q < 4; q++) g[q] = *reinterpret_cast<const f32x4 *>(gsrc + (size_t)blk * 32 + 8 * q);
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++)
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
    }
    float M[2] = {0.0f, 0.0f}, S[2] = {0.0f, 0.0f};
    for (;;) {
      const bool last_blk = bk + 1 == nb;
      const bool more_cols = j1 + step1 < hi_c[1];
      int blk_n = blk + 1;                                             // (past the pdf's last block only when nothing follows:
      if (last_blk) blk_n = more_cols ? __builtin_amdgcn_readfirstlane(word_n) >> 5 : blk;   //  then the same block again, unused)
      f32x16 init, acc[2];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) init[rr] = g[rr >> 2][rr & 3];
      const uint4 *src = wsrc + (size_t)blk_n * kUnits;
      const float *gn = gsrc + (size_t)blk_n * 32;
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++) {
#pragma unroll
        for (int t6 = 0; t6 < kProd; t6++)
#pragma unroll
          for (int n = 0; n < 2; n++) {
            const f32x16 &cin = (s_ == 0 && t6 == 0) ? init : acc[n];
            if constexpr (kHalf) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
          }
        if (s_ == 0) {
#pragma unroll
          for (int q = 0; q < 4; q++) g[q] = *reinterpret_cast<const f32x4 *>(gn + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n = 0; n < 2; n++) {
        float m = reg_max<0, 16>(acc[n]);
        m = fmaxf(m, swap32(m, h));
        float sv = reg_expsum_fast(acc[n], m, l2e_s);
        sv += swap32(sv, h);
        if (bk == 0) { M[n] = m; S[n] = sv; }
        else {
          const float mn = fmaxf(M[n], m);
          S[n] = S[n] * __builtin_amdgcn_exp2f((M[n] - mn) * l2e_s) + sv * __builtin_amdgcn_exp2f((m - mn) * l2e_s);
          M[n] = mn;
        }
      }
      blk = blk_n;
      if (!last_blk) { bk++; continue; }
      const int t = t_base + 32 * h + col;
      if (t < T) __builtin_nontemporal_store(finish((h ? M[1] : M[0]) * inv_s, h ? S[1] : S[0]), &out[(size_t)t * P + base_c[1] + j1]);
      if (!more_cols) break;
      j1 += step1;
      word = __builtin_amdgcn_readfirstlane(word_n);
      nb = blocks_of(word, j1); bk = 0;
      word_n = col_word(j1 + step1);
    }
  }

  // ---------------------------------------------------------------- classes 2, 3, 4: 32 / slot pdfs per virtual block
  // As gmm_split_small_kernel: the pdfs the list puts next to each other are gathered into one 32-row block (lane ↔ row
  // ρ = lane mod 32 → pdf ρ / slot, its row ρ mod slot; rows past the range come from the model's dummy row), the MFMAs
  // are those of class 0, the log-sum-exp runs over the slot's rows of each pdf — per pdf the very same expressions, so a
  // cell scored here carries the dense kernel's bits.  The gather costs nothing extra: every lane loads through its own
  // row pointer anyway.
  auto run_small = [&](auto slot_c, int base, int lo_s, int hi_s) {
    constexpr int kSlot = decltype(slot_c)::value, kPdfs = 32 / kSlot;
    if (lo_s >= hi_s) return;
    const uint4 *wsrc = kHalf ? p.wh : p.wb;
    const float *gsrc = kHalf ? p.gch : p.gc;
    const int rho = lane & 31, my_k = rho / kSlot, my_r = rho % kSlot;
    const int jb0 = lo_s / kPdfs, jb1 = (hi_s + kPdfs - 1) / kPdfs;
    auto row_of = [&](int jb) -> int {                 // this lane's packed row in virtual block jb (two dependent loads)
      const int idx = min(jb, jb1 - 1) * kPdfs + my_k;
      return idx < hi_s ? p.col_row0[l0 + base + idx] + my_r : p.num_rows;
    };
    auto src_of = [&](int row) { return wsrc + (size_t)(row >> 5) * kUnits + (row & 31) + 32 * h; };
    op8 a[kSteps][kPieces];
    int row_cur = row_of(jb0), row_next = row_of(jb0 + 1);
    float gcv = gsrc[row_cur];
    {
      const uint4 *src = src_of(row_cur);
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++)
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
    }
    const int col0 = base + jb0 * kPdfs;               // score column of the first staged column
    for (int jb = jb0; jb < jb1; jb++) {
      const int row_n2 = row_of(jb + 2);               // in flight during this block
      f32x16 init, acc[2];
#pragma unroll
      for (int rr = 0; rr < 16; rr++) init[rr] = __shfl(gcv, acc_row(rr, h));
      const uint4 *src = src_of(row_next);
#pragma unroll
      for (int s_ = 0; s_ < kSteps; s_++) {
#pragma unroll
        for (int t6 = 0; t6 < kProd; t6++)
#pragma unroll
          for (int n = 0; n < 2; n++) {
            const f32x16 &cin = (s_ == 0 && t6 == 0) ? init : acc[n];
            if constexpr (kHalf) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s_][pa[t6]], b[n][s_][pb[t6]], cin, 0, 0, 0);
          }
        if (s_ == 0) gcv = gsrc[row_next];
#pragma unroll
        for (int q = 0; q < kPieces; q++) a[s_][q] = __builtin_bit_cast(op8, src[(s_ * kPieces + q) * 64]);
        __builtin_amdgcn_sched_barrier(0);
      }
      row_next = row_n2;
      // per-pdf log-sum-exp.  Accumulator register r of half-wave h is row (r & 3) + 8 (r >> 2) + 4 h of the block.
      auto group_max = [&](const f32x16 &v, int r0, int cnt) {
        float m = v[r0];
#pragma unroll
        for (int rr = 1; rr < cnt; rr++) m = fmaxf(m, v[r0 + rr]);
        return m;
      };
      auto group_expsum = [&](const f32x16 &v, int r0, int cnt, float m) {
        float e[8];
#pragma unroll
        for (int rr = 0; rr < cnt; rr++) e[rr] = __builtin_amdgcn_exp2f((v[r0 + rr] - m) * l2e_s);
#pragma unroll
        for (int w = 1; w < cnt; w <<= 1)
#pragma unroll
          for (int rr = 0; rr + w < cnt; rr += 2 * w) e[rr] += e[rr + w];
        return e[0];
      };
      const int colbase = ((jb - jb0) * kPdfs) & 31;   // first staging column of this block
#pragma unroll
      for (int n = 0; n < 2; n++) {
        float *srow = stage + (32 * n + col) * 33 + colbase;
        if constexpr (kSlot == 16) {                   // pdf k: rows 16k..16k+15 = registers [8k, 8k+8) of both halves
          float ll[2];
#pragma unroll
          for (int k2 = 0; k2 < 2; k2++) {
            float m = group_max(acc[n], 8 * k2, 8);
            m = fmaxf(m, swap32(m, h));
            float sv = group_expsum(acc[n], 8 * k2, 8, m);
            sv += swap32(sv, h);
            ll[k2] = finish(m * inv_s, sv);
          }
          srow[h] = h ? ll[1] : ll[0];
        } else if constexpr (kSlot == 8) {             // pdf k: rows 8k..8k+7 = registers [4k, 4k+4) of both halves
          float ll[4];
#pragma unroll
          for (int k2 = 0; k2 < 4; k2++) {
            float m = group_max(acc[n], 4 * k2, 4);
            m = fmaxf(m, swap32(m, h));
            float sv = group_expsum(acc[n], 4 * k2, 4, m);
            sv += swap32(sv, h);
            ll[k2] = finish(m * inv_s, sv);
          }
          srow[h] = h ? ll[1] : ll[0];
          srow[2 + h] = h ? ll[3] : ll[2];
        } else {                                       // slot 4: pdf 2i + h: rows 8i + 4h .. +3 = registers [4i, 4i+4)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float m = group_max(acc[n], 4 * i, 4);
            const float sv = group_expsum(acc[n], 4 * i, 4, m);
            srow[2 * i + h] = finish(m * inv_s, sv);
          }
        }
      }
      const int done = (jb - jb0 + 1) * kPdfs;         // staged columns since col0 (whole blocks)
      if ((done & 31) == 0 || jb == jb1 - 1) {
        const int first = (done - 1) & ~31;            // first staged column of the open window
        const int valid = min(done, hi_s - jb0 * kPdfs) - first;   // columns of pdfs inside the class's range
        flush_cols(col0 + first, valid);
      }
    }
  };
  run_small(std::integral_constant<int, 16>{}, base_c[2], lo_c[2], hi_c[2]);
  run_small(std::integral_constant<int, 8>{}, base_c[3], lo_c[3], hi_c[3]);
  run_small(std::integral_constant<int, 4>{}, base_c[4], lo_c[4], hi_c[4]);
#ifdef GMM_BAND_STAMPS
  if (p.trace && lane == 0 && kHalf) {
    const unsigned long long st3 = wall_clock64();
    atomicAdd(&p.trace[0], st1 - st0); atomicAdd(&p.trace[1], st2 - st1); atomicAdd(&p.trace[2], st3 - st2);
    atomicAdd(&p.trace[3], 1ull); atomicAdd(&p.trace[4], (unsigned long long)(hi - lo));
  }
#endif
}

// Band-mode launch of the f32 kernel's tile walk: whatever slot classes gmm_band_kernel does not cover (single-Gaussian
// pdfs — bit-exact —, the 16/8/4-row classes, pdfs of more than 32 Gaussians; everything under MFA_GMM_BF16=0).
template <int M8>
__global__ __launch_bounds__(256, 2) void gmm_band_f32_kernel(GmmParams p) {
  __shared__ float stage_all[4][64 * 33];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int utt, r;
  if (!band_item(p, wave, utt, r)) return;
  score_tile<M8, 2>(p, utt, band_t_begin(p, utt) + 64 * r, lane, stage_all[wave], 0);
}

// max over the batch of the pdfs' first possible frames → *max_ff (the persistent kernel derives its phase split from it)
__global__ void gmm_max_first_frame_kernel(const int32_t *first_frame, const int64_t *pdf_off, int n_utt, int *out) {
  const int64_t n = pdf_off[n_utt];
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, first_frame[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
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
      float acc = p.gc[r0 + r];
      for (int k = 0; k < 2 * p.dim; k++) {
        float xv = k < p.dim ? x[k] : x[k - p.dim] * x[k - p.dim];
        acc = fmaf(p.w[mfa_packed_offset(r0 + r, k, p.kpad)], xv, acc);
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

// Block counts of multi-block pdfs ride in the five low bits of their columns' row words when every pdf has at most 32 blocks
// (MFA_GMM_PACK_NB=0: never — the lookup path, for tests)
static int col_nb_packed_for(const mfa_ctx *c) {
  const char *e = getenv("MFA_GMM_PACK_NB");
  if (e && e[0] == '0') return 0;
  return c->max_nblk <= 32 ? 1 : 0;
}

extern "C" {

MFA_API int mfa_load_gmm(mfa_ctx *c, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets, const float *h_gconsts,
                         const float *h_means_invvars, const float *h_inv_vars) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
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
  // whole 32-row blocks, plus room for the dummy row `rows` (zero weights, gconst −1e30) that idle lanes address
  const int blocks = (rows + 1 + 31) / 32;
  std::vector<float> w((size_t)blocks * 32 * kpad, 0.0f), gc((size_t)blocks * 32, kPadGconst);
  for (int p = 0; p < num_pdfs; p++) {
    int g0 = h_pdf_offsets[p], g = h_pdf_offsets[p + 1] - g0;
    for (int i = 0; i < g; i++) {
      const float *mi = h_means_invvars + (size_t)(g0 + i) * dim, *iv = h_inv_vars + (size_t)(g0 + i) * dim;
      for (int k = 0; k < 2 * dim; k++)
        w[mfa_packed_offset(row0[p] + i, k, kpad)] = k < dim ? mi[k] : -0.5f * iv[k - dim];
      gc[row0[p] + i] = h_gconsts[g0 + i];
    }
  }
  // bf16×3 split for the opt-in bf16 kernel: blocks of [step][piece][half][row] × 8 bf16 (natural k order, zero padded)
  std::vector<uint16_t> wb;
  const int steps = kpad / 16;
  const bool want_bf16 = (kpad == 80 || kpad == 96);
  if (want_bf16) {
    wb.assign((size_t)blocks * steps * 3 * 2 * 32 * 8, 0);
    auto to_bf16 = [](float f) -> uint16_t {   // round to nearest even, as the device's v_cvt_pk_bf16_f32
      uint32_t u; memcpy(&u, &f, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto from_bf16 = [](uint16_t b) -> float { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int row = 0; row < rows; row++) {
      for (int k = 0; k < 2 * dim; k++) {
        const float v = w[mfa_packed_offset(row, k, kpad)];
        const uint16_t v1 = to_bf16(v);
        const float r1 = v - from_bf16(v1);
        const uint16_t v2 = to_bf16(r1);
        const float r2 = r1 - from_bf16(v2);
        const uint16_t piece[3] = {v1, v2, to_bf16(r2)};
        const int s_ = k >> 4, hh = (k >> 3) & 1, e = k & 7;
        for (int qq = 0; qq < 3; qq++) {
          const size_t unit = (size_t)(row >> 5) * steps * 3 * 2 * 32 + (size_t)((s_ * 3 + qq) * 2 + hh) * 32 + (row & 31);
          wb[unit * 8 + e] = piece[qq];
        }
      }
    }
  }
  // f16×2 split (default scoring path of the 32-row classes without multi-block pdfs).  Column k of the weights is
  // multiplied by 2^e_k and column k of x̃ by S·2^-e_k, so every product — and the gconst, stored × S — carries the one
  // factor S and nothing is rounded differently.  The exponents balance the two operands inside the f16 range using the
  // model's own idea of how large a feature can get (|μ| + 10σ over all Gaussians); features beyond 65000 after scaling
  // are caught per tile on the device (see gmm_split_single_kernel).
  std::vector<uint16_t> wh;
  std::vector<float> fscale(kpad, 0.0f), gch;
  float acc_scale = 1.0f;
  if (want_bf16) {
    std::vector<double> wmax(2 * dim, 0.0), xmax(2 * dim, 0.0);
    for (int p = 0; p < num_pdfs; p++) {
      int g0 = h_pdf_offsets[p], g = h_pdf_offsets[p + 1] - g0;
      for (int i = 0; i < g; i++) {
        const float *mi = h_means_invvars + (size_t)(g0 + i) * dim, *iv = h_inv_vars + (size_t)(g0 + i) * dim;
        for (int k = 0; k < dim; k++) {
          const double v = iv[k], m = mi[k];
          if (std::isfinite(m)) wmax[k] = std::max(wmax[k], std::fabs(m));
          if (std::isfinite(v)) wmax[dim + k] = std::max(wmax[dim + k], 0.5 * std::fabs(v));
          if (std::isfinite(v) && std::isfinite(m) && v > 0) {
            const double reach = std::fabs(m / v) + 10.0 / std::sqrt(v);
            if (std::isfinite(reach)) { xmax[k] = std::max(xmax[k], reach); xmax[dim + k] = std::max(xmax[dim + k], reach * reach); }
          }
        }
      }
    }
    int log_s = 12;
    for (int k = 0; k < 2 * dim; k++)
      if (wmax[k] > 0 && xmax[k] > 0) log_s = std::min(log_s, (int)std::floor(26.0 - std::log2(wmax[k] * xmax[k])));
    log_s = std::max(log_s, -20);
    acc_scale = std::ldexp(1.0f, log_s);
    std::vector<int> e_w(2 * dim, 0);
    for (int k = 0; k < 2 * dim; k++) {
      if (!(wmax[k] > 0)) { fscale[k] = 0.0f; continue; }           // an all-zero weight column: x̃_k is irrelevant
      const double xm = xmax[k] > 0 ? xmax[k] : 1.0;
      int e = (int)std::lround(0.5 * (log_s + std::log2(xm) - std::log2(wmax[k])));
      while (std::ldexp(wmax[k], e) > 32768.0) e--;                 // never let the weights themselves leave the range
      e = std::max(-100, std::min(100, e));
      e_w[k] = e;
      fscale[k] = (float)std::ldexp(1.0, log_s - e);
    }
    wh.assign((size_t)blocks * steps * 2 * 2 * 32 * 8, 0);
    auto f16_bits = [](float f) -> uint16_t { _Float16 hv = (_Float16)f; uint16_t u; memcpy(&u, &hv, 2); return u; };
    for (int row = 0; row < rows; row++) {
      for (int k = 0; k < 2 * dim; k++) {
        const float v = std::ldexp(w[mfa_packed_offset(row, k, kpad)], e_w[k]);
        const _Float16 v1 = (_Float16)v;
        const float r1 = v - (float)v1;
        const uint16_t piece[2] = {f16_bits(v), f16_bits(r1)};
        const int s_ = k >> 4, hh = (k >> 3) & 1, e = k & 7;
        for (int qq = 0; qq < 2; qq++) {
          const size_t unit = (size_t)(row >> 5) * steps * 2 * 2 * 32 + (size_t)((s_ * 2 + qq) * 2 + hh) * 32 + (row & 31);
          wh[unit * 8 + e] = piece[qq];
        }
      }
    }
    gch.resize(gc.size());
    for (size_t i = 0; i < gc.size(); i++) gch[i] = gc[i] * acc_scale;
  }
  void *old[] = {c->d_w, c->d_gc, c->d_row0, c->d_nblk, c->d_slot, c->d_nrows, c->d_wb, c->d_wh, c->d_gch, c->d_fscale};
  c->d_wb = nullptr; c->d_wh = nullptr; c->d_gch = nullptr; c->d_fscale = nullptr;
  for (void *q : old) if (q) (void)hipFree(q);
  c->d_w = nullptr; c->d_gc = nullptr; c->d_row0 = nullptr; c->d_nblk = nullptr; c->d_slot = nullptr; c->d_nrows = nullptr;
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_w, w.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gc, gc.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_row0, row0.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_nblk, nblk.size() * 4));
  MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_slot, slot.size() * 4));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  if (want_bf16) {
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_wb, wb.size() * 2));
    MFA_HIP_CHECK(c, hipMemcpy(c->d_wb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_wh, wh.size() * 2));
    MFA_HIP_CHECK(c, hipMemcpy(c->d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gch, gch.size() * 4));
    MFA_HIP_CHECK(c, hipMemcpy(c->d_gch, gch.data(), gch.size() * 4, hipMemcpyHostToDevice));
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_fscale, fscale.size() * 4));
    MFA_HIP_CHECK(c, hipMemcpy(c->d_fscale, fscale.data(), fscale.size() * 4, hipMemcpyHostToDevice));
    c->gmm_acc_scale = acc_scale;
  }
  MFA_HIP_CHECK(c, hipMemcpy(c->d_gc, gc.data(), gc.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_row0, row0.data(), row0.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_nblk, nblk.data(), nblk.size() * 4, hipMemcpyHostToDevice));
  MFA_HIP_CHECK(c, hipMemcpy(c->d_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice));
  c->dim = dim; c->kpad = kpad; c->num_pdfs = num_pdfs; c->num_rows = rows;
  c->h_slot = slot;
  c->h_nblk = nblk;
  c->h_row0.assign(row0.begin(), row0.begin() + num_pdfs);
  c->h_ngauss.resize(num_pdfs);
  for (int p = 0; p < num_pdfs; p++) c->h_ngauss[p] = h_pdf_offsets[p + 1] - h_pdf_offsets[p];
  if (c->d_w_stats) { (void)hipFree(c->d_w_stats); c->d_w_stats = nullptr; }   // belonged to the previous model's layout
  if (c->d_nrows) { (void)hipFree(c->d_nrows); c->d_nrows = nullptr; }
  c->all_single_block = true;   // (name kept: "all pdfs are 32-row pdfs", single- or multi-block)
  c->has_multi_block = false;
  c->max_nblk = 1;
  for (int q = 0; q < 5; q++) c->has_slot_class[q] = false;
  c->has_single32 = false;
  for (int p = 0; p < num_pdfs; p++) {
    if (slot[p] != 32) c->all_single_block = false;
    if (nblk[p] > 1) c->has_multi_block = true;
    c->max_nblk = std::max(c->max_nblk, nblk[p]);
    if (slot[p] == 32 && nblk[p] == 1) c->has_single32 = true;
    c->has_slot_class[class_index(slot[p])] = true;
  }
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

MFA_API int mfa_gmm_sort_pdf_list_keyed(mfa_ctx *c, int32_t *h_pdfs, int32_t *h_first_frame, int32_t n,
                                        int32_t *h_class_counts) {
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  std::vector<std::pair<int32_t, int32_t>> bucket[6];  // (key, pdf)
  for (int i = 0; i < n; i++) {
    int p = h_pdfs[i];
    if (p < 0 || p >= c->num_pdfs) return c->fail("pdf id %d out of range [0,%d)", p, c->num_pdfs);
    int ci = class_index(c->h_slot[p]);
    bucket[ci == 0 ? (c->h_nblk[p] == 1 ? 0 : 1) : ci + 1].push_back({h_first_frame[i], p});
  }
  int k = 0;
  for (int b = 0; b < 6; b++) {
    std::stable_sort(bucket[b].begin(), bucket[b].end(),
                     [](const std::pair<int32_t, int32_t> &x, const std::pair<int32_t, int32_t> &y) { return x.first < y.first; });
    h_class_counts[b] = (int32_t)bucket[b].size();
    for (auto &e : bucket[b]) { h_first_frame[k] = e.first; h_pdfs[k++] = e.second; }
  }
  return 0;
}

MFA_API int mfa_fst_first_frames(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next, int32_t start,
                                 int32_t *h_depth) {
  if (n_states <= 0 || start < 0 || start >= n_states) return -1;
  for (int s = 0; s < n_states; s++) h_depth[s] = INT32_MAX;
  std::vector<int32_t> queue;
  queue.reserve(n_states);
  queue.push_back(start);
  h_depth[start] = 0;
  for (size_t q = 0; q < queue.size(); q++) {  // breadth-first: unit arc lengths
    const int s = queue[q];
    for (int a = h_arc_off[s]; a < h_arc_off[s + 1]; a++) {
      const int d = h_arc_next[a];
      if (d < 0 || d >= n_states) return -1;
      if (h_depth[d] == INT32_MAX) { h_depth[d] = h_depth[s] + 1; queue.push_back(d); }
    }
  }
  return 0;
}

// The same with epsilon input arcs (h_arc_pdf[a] < 0) counting for nothing: depth = fewest EMITTING arcs from the start state —
// the first frame a token can sit on the state, FasterDecoder's ProcessNonemitting moving tokens along epsilon arcs within a
// frame.  0-1 breadth-first search.
static int fst_first_frames_eps(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next, const int32_t *h_arc_pdf,
                                int32_t start, int32_t *h_depth) {
  if (n_states <= 0 || start < 0 || start >= n_states) return -1;
  for (int s = 0; s < n_states; s++) h_depth[s] = INT32_MAX;
  std::vector<int32_t> cur, nxt;
  cur.push_back(start);
  h_depth[start] = 0;
  int32_t level = 0;
  while (!cur.empty()) {
    for (size_t q = 0; q < cur.size(); q++) {            // (cur grows while epsilon arcs are followed)
      const int s = cur[q];
      if (h_depth[s] != level) continue;                 // reached more cheaply in the meantime
      for (int a = h_arc_off[s]; a < h_arc_off[s + 1]; a++) {
        const int d = h_arc_next[a];
        if (d < 0 || d >= n_states) return -1;
        const int32_t nd = level + (h_arc_pdf[a] < 0 ? 0 : 1);
        if (nd < h_depth[d]) { h_depth[d] = nd; (nd == level ? cur : nxt).push_back(d); }
      }
    }
    cur.swap(nxt); nxt.clear();
    level++;
  }
  return 0;
}

MFA_API int mfa_fst_last_depths(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next, int32_t start,
                                const int32_t *h_bfs_depth, int32_t *h_depth) {
  if (n_states <= 0 || start < 0 || start >= n_states) return -1;
  for (int a = 0; a < h_arc_off[n_states]; a++)
    if (h_arc_next[a] < 0 || h_arc_next[a] >= n_states) return -1;
  // h_depth[s] = the smallest BFS depth among the states reachable from s (s included).  Computed over the graph's
  // condensation: strongly connected components (self-loops, the small cycles of an ergodic silence topology) come out of
  // Tarjan's algorithm (iterative) in reverse topological order, i.e. sinks first — exactly the order this needs.
  std::vector<int32_t> index(n_states, -1), low(n_states, 0), comp(n_states, -1), stack, next_arc(n_states, 0);
  std::vector<char> on_stack(n_states, 0);
  std::vector<int32_t> call;   // DFS stack of states
  int32_t counter = 0, n_comp = 0;
  call.push_back(start);
  index[start] = low[start] = counter++;
  stack.push_back(start); on_stack[start] = 1;
  next_arc[start] = h_arc_off[start];
  std::vector<int32_t> comp_first;   // members of component k: comp_members[comp_first[k] .. comp_first[k+1])
  std::vector<int32_t> comp_members;
  while (!call.empty()) {
    const int s = call.back();
    if (next_arc[s] < h_arc_off[s + 1]) {
      const int d = h_arc_next[next_arc[s]++];
      if (index[d] < 0) {
        index[d] = low[d] = counter++;
        stack.push_back(d); on_stack[d] = 1;
        next_arc[d] = h_arc_off[d];
        call.push_back(d);
      } else if (on_stack[d]) {
        low[s] = std::min(low[s], index[d]);
      }
    } else {
      call.pop_back();
      if (!call.empty()) low[call.back()] = std::min(low[call.back()], low[s]);
      if (low[s] == index[s]) {
        comp_first.push_back((int32_t)comp_members.size());
        for (;;) {
          const int v = stack.back(); stack.pop_back(); on_stack[v] = 0;
          comp[v] = n_comp;
          comp_members.push_back(v);
          if (v == s) break;
        }
        n_comp++;
      }
    }
  }
  comp_first.push_back((int32_t)comp_members.size());
  // an arc s -> d between different components has comp[d] < comp[s]: ascending component order visits successors first
  std::vector<int32_t> cmin(n_comp, INT32_MAX);
  bool cyclic = false;
  for (int k = 0; k < n_comp; k++) {
    if (comp_first[k + 1] - comp_first[k] > 1) cyclic = true;
    int32_t m = INT32_MAX;
    for (int i = comp_first[k]; i < comp_first[k + 1]; i++) {
      const int s = comp_members[i];
      m = std::min(m, h_bfs_depth[s]);
      for (int a = h_arc_off[s]; a < h_arc_off[s + 1]; a++) {
        const int cd = comp[h_arc_next[a]];
        if (cd != k) m = std::min(m, cmin[cd]);
      }
    }
    cmin[k] = m;
  }
  for (int s = 0; s < n_states; s++) h_depth[s] = comp[s] >= 0 ? cmin[comp[s]] : 0;
  return cyclic ? 1 : 0;
}

// Score columns of one utterance for mfa_align_features_batch / mfa_gmm_score_batch — see include/mfa_hip.h.
MFA_API int mfa_build_score_plan(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next,
                                 const int32_t *h_arc_pdf, int32_t start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                 int32_t cluster_span, int32_t *h_state_depth, int32_t *h_arc_col, int32_t *h_col_pdf,
                                 int32_t *h_col_first, int32_t *h_col_last, int32_t *h_class_counts, int32_t *h_n_cols) {
  return mfa_build_score_plan_grouped(n_states, h_arc_off, h_arc_next, h_arc_pdf, start, num_pdfs, h_pdf_class, cluster_span,
                                      1, h_state_depth, h_arc_col, h_col_pdf, h_col_first, h_col_last, h_class_counts, nullptr,
                                      h_n_cols);
}

MFA_API int mfa_build_score_plans_batch(int32_t n_utt, const int64_t *h_state_off, const int64_t *h_arc_base,
                                        const int32_t *h_arc_off, const int32_t *h_arc_next, const int32_t *h_arc_pdf,
                                        const int32_t *h_start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                        int32_t cluster_span, int32_t groups, int32_t n_threads, int32_t *h_state_depth,
                                        int32_t *h_arc_col, int32_t *h_col_pdf, int32_t *h_col_first, int32_t *h_col_last,
                                        int32_t *h_class_counts, int32_t *h_group_counts, int32_t *h_n_cols,
                                        int32_t *h_bad_utt) {
  if (n_utt < 0) return -1;
  std::atomic<int> next(0), first_bad(INT32_MAX);
  std::vector<int> codes((size_t)std::max(n_utt, 1), 0);
  auto work = [&]() {
    for (;;) {
      const int u = next.fetch_add(1);
      if (u >= n_utt) break;
      const int64_t s0 = h_state_off[u], a0 = h_arc_base[u];
      const int32_t ns = (int32_t)(h_state_off[u + 1] - s0);
      const int rc = mfa_build_score_plan_grouped(ns, h_arc_off + s0 + u, h_arc_next + a0, h_arc_pdf + a0, h_start[u], num_pdfs,
                                                  h_pdf_class, cluster_span, groups, h_state_depth + 2 * s0, h_arc_col + a0,
                                                  h_col_pdf + a0, h_col_first + a0, h_col_last + a0, h_class_counts + 6 * (size_t)u,
                                                  groups > 1 ? h_group_counts + (size_t)groups * u : nullptr, h_n_cols + u);
      codes[u] = rc;
      if (rc != 0) { int cur = first_bad.load(); while (u < cur && !first_bad.compare_exchange_weak(cur, u)) {} }
    }
  };
  const int nt = std::max(1, std::min(n_threads, n_utt));
  if (nt == 1) work();
  else {
    std::vector<std::thread> ts;
    for (int t = 0; t < nt; t++) ts.emplace_back(work);
    for (auto &t : ts) t.join();
  }
  const int bad = first_bad.load();
  if (bad != INT32_MAX) { if (h_bad_utt) *h_bad_utt = bad; return codes[bad]; }
  return 0;
}

MFA_API int mfa_build_score_plan_grouped(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next,
                                         const int32_t *h_arc_pdf, int32_t start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                         int32_t cluster_span, int32_t groups, int32_t *h_state_depth, int32_t *h_arc_col,
                                         int32_t *h_col_pdf, int32_t *h_col_first, int32_t *h_col_last, int32_t *h_class_counts,
                                         int32_t *h_group_counts, int32_t *h_n_cols) {
  if (groups < 1 || groups > MFA_PLAN_MAX_GROUPS || (groups > 1 && !h_group_counts)) return -3;
  if (n_states <= 0 || start < 0 || start >= n_states) return -1;
  const int n_arcs = h_arc_off[n_states];
  std::vector<int32_t> bfs(n_states), low(n_states);
  bool has_eps = false;
  for (int a = 0; a < n_arcs; a++) if (h_arc_pdf[a] < 0) { has_eps = true; break; }
  // (an arc with pdf -1 is an epsilon input arc: no score column, no frame consumed)
  if ((has_eps ? fst_first_frames_eps(n_states, h_arc_off, h_arc_next, h_arc_pdf, start, bfs.data())
               : mfa_fst_first_frames(n_states, h_arc_off, h_arc_next, start, bfs.data())) != 0) return -1;
  if (mfa_fst_last_depths(n_states, h_arc_off, h_arc_next, start, bfs.data(), low.data()) < 0) return -1;
  for (int s = 0; s < n_states; s++) {
    h_state_depth[2 * s] = bfs[s] == INT32_MAX ? 0 : bfs[s];
    h_state_depth[2 * s + 1] = bfs[s] == INT32_MAX ? 0 : low[s];
  }
  // arcs by (pdf, BFS depth of the source state); a column = a run of one pdf's arcs whose depths stay within
  // cluster_span of the run's first (cluster_span <= 0: one column per pdf)
  std::vector<int32_t> src(n_arcs), order(n_arcs);
  for (int s = 0; s < n_states; s++)
    for (int a = h_arc_off[s]; a < h_arc_off[s + 1]; a++) src[a] = s;
  order.clear();
  for (int a = 0; a < n_arcs; a++) {
    if (h_arc_pdf[a] == -1) continue;                    // epsilon input arc
    if (h_arc_pdf[a] < 0 || h_arc_pdf[a] >= num_pdfs) return -2;
    if (h_pdf_class[h_arc_pdf[a]] < 0 || h_pdf_class[h_arc_pdf[a]] > 5) return -2;
    order.push_back(a);
  }
  const int n_emit = (int)order.size();
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
    if (h_arc_pdf[x] != h_arc_pdf[y]) return h_arc_pdf[x] < h_arc_pdf[y];
    return bfs[src[x]] < bfs[src[y]];
  });
  struct Col { int32_t pdf, first, last, cls; };
  std::vector<Col> cols;
  std::vector<int32_t> col_of_arc(n_arcs, -1);
  for (int i = 0; i < n_emit; i++) {
    const int a = order[i], pdf = h_arc_pdf[a], d = bfs[src[a]];
    const bool fresh = cols.empty() || cols.back().pdf != pdf ||
                       (cluster_span > 0 && ((int64_t)d - cols.back().first > cluster_span));
    if (fresh) cols.push_back({pdf, d, d, h_pdf_class[pdf]});
    cols.back().last = std::max(cols.back().last, d);
    col_of_arc[a] = (int32_t)cols.size() - 1;
  }
  // kernel order: slot class, (class 0 only: pdf id mod `groups` — the XCD whose L2 keeps that part of the model), then
  // ascending first depth (ties: pdf id, then depth — the creation order)
  const int n_cols = (int)cols.size();
  std::vector<int32_t> perm(n_cols), rank(n_cols);
  for (int i = 0; i < n_cols; i++) perm[i] = i;
  auto group_of = [&](const Col &c) { return c.cls == 0 ? c.pdf % groups : 0; };
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
    if (cols[x].cls != cols[y].cls) return cols[x].cls < cols[y].cls;
    const int gx = group_of(cols[x]), gy = group_of(cols[y]);
    if (gx != gy) return gx < gy;
    return cols[x].first < cols[y].first;
  });
  for (int k = 0; k < 6; k++) h_class_counts[k] = 0;
  if (h_group_counts) for (int k = 0; k < groups; k++) h_group_counts[k] = 0;
  int32_t run_cls = -1, run_grp = -1, run_max = 0;
  for (int i = 0; i < n_cols; i++) {
    const Col &cl = cols[perm[i]];
    const int grp = group_of(cl);
    rank[perm[i]] = i;
    h_col_pdf[i] = cl.pdf;
    h_col_first[i] = cl.first;
    if (cl.cls != run_cls || grp != run_grp) { run_cls = cl.cls; run_grp = grp; run_max = cl.last; }
    run_max = std::max(run_max, cl.last);
    h_col_last[i] = run_max;            // running max inside the class (class 0: inside the group): non-decreasing along it
    h_class_counts[cl.cls]++;
    if (h_group_counts && cl.cls == 0) h_group_counts[grp]++;
  }
  for (int a = 0; a < n_arcs; a++) h_arc_col[a] = col_of_arc[a] >= 0 ? rank[col_of_arc[a]] : 0;
  *h_n_cols = n_cols;
  return 0;
}

MFA_API int mfa_debug_gmm_trace(mfa_ctx *c, void *d_trace) {
  c->gmm_trace = d_trace;
  return 0;
}

MFA_API int mfa_gmm_score_batch(mfa_ctx *c, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                int32_t max_frames, const int32_t *d_pdf_list, const int64_t *d_pdf_off,
                                const int32_t *d_class_counts, const int32_t *d_pdf_first_frame, const int64_t *d_ll_off,
                                float *d_loglikes) {
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  if (n_utt <= 0 || max_frames <= 0) return 0;
  if (n_utt > 65535) return c->fail("at most 65535 utterances per scoring launch (got %d)", n_utt);
  GmmParams p;
  memset(&p, 0, sizeof(p));   // every field this function does not set (the band-mode ones) must read as "off"
  p.dim = c->dim; p.kpad = c->kpad; p.num_rows = c->num_rows;
  p.w = c->d_w; p.gc = c->d_gc; p.row0 = c->d_row0; p.nblk = c->d_nblk; p.slot = c->d_slot;
  p.feats = d_feats; p.frame_off = d_frame_off; p.pdf_list = d_pdf_list; p.pdf_off = d_pdf_off;
  p.class_counts = d_class_counts; p.ll_off = d_ll_off; p.out = d_loglikes;
  p.min_log_diff = logf(1.1920928955078125e-07f);
  p.first_frame = d_pdf_first_frame;
  p.trace = (unsigned long long *)c->gmm_trace;
  { const char *fb = getenv("MFA_GMM_FF_BIAS"); p.ff_bias = fb ? atoi(fb) : 0; }
  const char *naive = getenv("MFA_GMM_NAIVE");
  const int m8 = c->kpad / 8;
  KernelTimer kt(c, MFA_K_GMM);
  if ((naive && naive[0] == '1') || m8 > 12) {
    // worst-case P is not known on the host side of this call: cover max_frames * num_pdfs threads per utterance
    int64_t per_utt = (int64_t)max_frames * c->num_pdfs;
    dim3 grid((unsigned)((per_utt + 255) / 256), n_utt);
    hipLaunchKernelGGL(gmm_naive_kernel, grid, dim3(256), 0, c->stream, p);
  } else {
    // 4 wavefronts per workgroup, 2 frame tiles (64 frames) per wavefront, two workgroups resident per CU (206 VGPRs).
    // Variants measured and dropped in round 1: 1 tile per wavefront (3 per SIMD: 30 TFLOP/s, spills), 8-wavefront
    // workgroups with or without a static s_setprio split (103 TFLOP/s), one wavefront per SIMD with 4 tiles and an
    // in-wavefront MFMA/epilogue software pipeline (97 TFLOP/s).
    constexpr int kFramesPerItem = 256;
    p.n_utt = n_utt;
    p.tiles = (max_frames + kFramesPerItem - 1) / kFramesPerItem;
    constexpr int kQueueInts = 64 + 8 * 16;   // counters of the main launches + six small-slot launches + the multi-block one
    if (!c->d_gmm_queue) MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmm_queue, kQueueInts * sizeof(int)));
    MFA_HIP_CHECK(c, hipMemsetAsync(c->d_gmm_queue, 0, kQueueInts * sizeof(int), c->stream));
    p.queue = c->d_gmm_queue;
    p.max_ff = c->d_gmm_queue + 16;
    if (d_pdf_first_frame)
      hipLaunchKernelGGL(gmm_max_first_frame_kernel, dim3(64), dim3(256), 0, c->stream, d_pdf_first_frame, d_pdf_off, n_utt,
                         c->d_gmm_queue + 16);
    if (c->num_cus <= 0) {
      hipDeviceProp_t prop;
      MFA_HIP_CHECK(c, hipGetDeviceProperties(&prop, c->device));
      c->num_cus = prop.multiProcessorCount;
    }
    const int64_t items = (int64_t)n_utt * p.tiles;
    const int64_t wgs = std::min<int64_t>((int64_t)c->num_cus * 2, items);
    dim3 grid((unsigned)std::max<int64_t>(wgs, 1));
    const char *bf = getenv("MFA_GMM_BF16");
    p.wb = (const uint4 *)c->d_wb;
    p.wh = nullptr; p.gch = nullptr; p.fscale = nullptr; p.acc_scale_inv = 1.0f; p.redo = nullptr; p.redo_mode = 0; p.redo_count = nullptr; p.skip_cc0 = 0;
    p.skip_single = 0;
    if (!(bf && bf[0] == '0') && c->d_wb) {   // default on; MFA_GMM_BF16=0 keeps every class on the f32 kernel
      // class 0 on the bf16×3 kernel, then the f32 kernel for whatever other slot classes the lists hold (second set of
      // queue counters; an item with nothing left returns at once)
      const char *hf = getenv("MFA_GMM_F16");
      const bool f16_ok = !(hf && hf[0] == '0') && c->d_wh;
      const bool use_f16 = f16_ok && c->has_single32;   // single-block 32-row class on the lean f16 kernel
      p.skip_cc0 = 0;
      if (f16_ok) {
        // an f16×2 pass scores every tile it can and flags the others for the bf16×3 pass that follows it
        if (c->gmm_redo_cap < items) {
          if (c->d_gmm_redo) (void)hipFree(c->d_gmm_redo);
          c->d_gmm_redo = nullptr; c->gmm_redo_cap = 0;
          MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmm_redo, items * sizeof(int)));
          c->gmm_redo_cap = items;
        }
        MFA_HIP_CHECK(c, hipMemsetAsync(c->d_gmm_redo, 0, items * sizeof(int), c->stream));
        p.wh = (const uint4 *)c->d_wh; p.gch = c->d_gch; p.fscale = c->d_fscale;
        p.acc_scale_inv = 1.0f / c->gmm_acc_scale;
        p.redo = c->d_gmm_redo; p.redo_mode = 0; p.redo_count = c->d_gmm_queue + 51;
      }
      if (use_f16) {
        if (m8 == 10) hipLaunchKernelGGL((gmm_split_single_kernel<5, 2>), grid, dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((gmm_split_single_kernel<6, 2>), grid, dim3(256), 0, c->stream, p);
        p.redo_mode = 2;
        p.queue = c->d_gmm_queue + 34;
        if (m8 == 10) hipLaunchKernelGGL((gmm_split_single_kernel<5, 3>), grid, dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((gmm_split_single_kernel<6, 3>), grid, dim3(256), 0, c->stream, p);
        p.redo_mode = 0;
        p.skip_cc0 = 1;
        p.queue = c->d_gmm_queue + 64 + 6 * 16;
      }
      if (c->has_multi_block) {                           // pdfs of more than 32 Gaussians (and, without f16, the whole 32-row class)
        if (f16_ok) {                                     // f16×2 pass, then the bf16×3 pass over the tiles it declined
          if (m8 == 10) hipLaunchKernelGGL((gmm_bf16_kernel<5, 2>), grid, dim3(256), 0, c->stream, p);
          else hipLaunchKernelGGL((gmm_bf16_kernel<6, 2>), grid, dim3(256), 0, c->stream, p);
          p.redo_mode = 2;
          p.queue = c->d_gmm_queue + 64 + 7 * 16;
        }
        if (m8 == 10) hipLaunchKernelGGL((gmm_bf16_kernel<5, 3>), grid, dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((gmm_bf16_kernel<6, 3>), grid, dim3(256), 0, c->stream, p);
        p.redo_mode = 0;
      } else if (!use_f16 && c->has_single32) {
        if (m8 == 10) hipLaunchKernelGGL((gmm_split_single_kernel<5, 3>), grid, dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL((gmm_split_single_kernel<6, 3>), grid, dim3(256), 0, c->stream, p);
      }
      p.skip_single = 1;
      {
        // the 16- / 8- / 4-row classes on the same pipe (f16×2 pass, then the bf16×3 pass over declined tiles), each launch
        // with its own queue counters; classes the model does not have are not launched
        int qbase = 64;
        auto small = [&](int slot_rows, int cls_idx) {
          if (!c->has_slot_class[cls_idx]) return;
          for (int pass = f16_ok ? 0 : 1; pass < 2; pass++) {
            p.redo_mode = f16_ok ? (pass == 0 ? 0 : 2) : 0;
            p.queue = c->d_gmm_queue + qbase; qbase += 16;
#define MFA_LAUNCH_SMALL(STEPS, PIECES)                                                                                   \
            do {                                                                                                        \
              if (slot_rows == 16) hipLaunchKernelGGL((gmm_split_small_kernel<STEPS, PIECES, 16>), grid, dim3(256), 0, c->stream, p); \
              else if (slot_rows == 8) hipLaunchKernelGGL((gmm_split_small_kernel<STEPS, PIECES, 8>), grid, dim3(256), 0, c->stream, p); \
              else hipLaunchKernelGGL((gmm_split_small_kernel<STEPS, PIECES, 4>), grid, dim3(256), 0, c->stream, p); \
            } while (0)
            if (pass == 0) { if (m8 == 10) MFA_LAUNCH_SMALL(5, 2); else MFA_LAUNCH_SMALL(6, 2); }
            else { if (m8 == 10) MFA_LAUNCH_SMALL(5, 3); else MFA_LAUNCH_SMALL(6, 3); }
#undef MFA_LAUNCH_SMALL
          }
        };
        small(16, 1); small(8, 2); small(4, 3);
        p.skip_single = 2;
      }
      p.queue = c->d_gmm_queue + 17;
    }
    const bool only_split_classes = !c->has_slot_class[4] && p.skip_single == 2;   // no single-Gaussian pdfs left over
    if (p.skip_single && (c->all_single_block || only_split_classes)) {
      // every pdf of the model is a single 32-row block: nothing is left for the f32 kernel
    } else if (m8 <= 10) hipLaunchKernelGGL((gmm_kernel<10, 2, 2, 4>), grid, dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL((gmm_kernel<12, 2, 2, 4>), grid, dim3(256), 0, c->stream, p);
  }
  MFA_HIP_CHECK(c, hipGetLastError());
  MFA_DEBUG_POINT(c, "dense scoring of %d utterances", n_utt);
  return 0;
}

}  // extern "C"

int mfa_gmm_lazy_supported(mfa_ctx *c) { return c->gmm_ready && (c->kpad == 80 || c->kpad == 96); }

// Pre-split f16 operands of the whole batch (see gmm_presplit_kernel); mfa_gmm_score_window then hands them to the band kernel.
int mfa_gmm_presplit(mfa_ctx *c, const MfaLazyScoring *lazy, const int64_t *d_frame_off, int n_utt, int64_t total_frames) {
  c->xsplit_ready = false;
  {   // first model row of every score column of the batch (the band kernels read nothing else to find a block)
    const int64_t cols_cap = (int64_t)n_utt * std::max(1, lazy->plan.max_cols);
    if (c->col_row0_cap < cols_cap) {
      MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (c->d_col_row0) (void)hipFree(c->d_col_row0);
      c->d_col_row0 = nullptr; c->col_row0_cap = 0;
      MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_col_row0, (size_t)cols_cap * sizeof(int32_t)));
      c->col_row0_cap = cols_cap;
    }
    GmmParams q;
    memset(&q, 0, sizeof(q));
    q.row0 = c->d_row0; q.pdf_list = lazy->plan.d_pdf_list; q.pdf_off = lazy->plan.d_pdf_off; q.n_utt = n_utt;
    q.nblk = c->d_nblk; q.col_nb_packed = col_nb_packed_for(c);
    hipLaunchKernelGGL(gmm_col_rows_kernel, dim3(n_utt), dim3(256), 0, c->stream, q, c->d_col_row0);
    MFA_HIP_CHECK(c, hipGetLastError());
  }
  const char *bf = getenv("MFA_GMM_BF16");
  const char *hf = getenv("MFA_GMM_F16");
  const char *ps = getenv("MFA_GMM_PRESPLIT");
  if ((bf && bf[0] == '0') || (hf && hf[0] == '0') || (ps && ps[0] == '0') || !c->d_wh || !c->d_wb) return 0;   // no f16 pass: nothing to prepare
  const int ksteps = c->kpad / 16;
  if (ksteps != 5 && ksteps != 6) return 0;
  const int64_t tiles = (total_frames >> 6) + n_utt + 1;
  if (c->xsplit_tiles < tiles) {
    MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_xsplit) (void)hipFree(c->d_xsplit);
    if (c->d_xsplit_bad) (void)hipFree(c->d_xsplit_bad);
    c->d_xsplit = nullptr; c->d_xsplit_bad = nullptr; c->xsplit_tiles = 0;
    MFA_HIP_CHECK(c, hipMalloc(&c->d_xsplit, (size_t)tiles * 2 * 6 * 2 * 64 * 16));
    MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_xsplit_bad, (size_t)tiles * sizeof(int)));
    c->xsplit_tiles = tiles;
  }
  GmmParams p;
  memset(&p, 0, sizeof(p));
  p.dim = c->dim; p.kpad = c->kpad; p.feats = lazy->d_feats; p.frame_off = d_frame_off; p.n_utt = n_utt; p.fscale = c->d_fscale;
  const int tiles_per_utt = (lazy->max_frames + 63) / 64;
  const int64_t waves = (int64_t)n_utt * tiles_per_utt;
  const dim3 grid((unsigned)((waves + 3) / 4));
  KernelTimer kt(c, MFA_K_GMM);
  if (ksteps == 5) hipLaunchKernelGGL((gmm_presplit_kernel<5>), grid, dim3(256), 0, c->stream, p, (uint4 *)c->d_xsplit, c->d_xsplit_bad, tiles_per_utt);
  else hipLaunchKernelGGL((gmm_presplit_kernel<6>), grid, dim3(256), 0, c->stream, p, (uint4 *)c->d_xsplit, c->d_xsplit_bad, tiles_per_utt);
  MFA_HIP_CHECK(c, hipGetLastError());
  c->xsplit_ready = true;
  return 0;
}

const int32_t *mfa_band_ranges(mfa_ctx *c) { return c->d_band_ranges; }

int mfa_gmm_score_window(mfa_ctx *c, const MfaLazyScoring *lazy, const MfaWindowScore *ws, const int64_t *d_frame_off,
                         int n_utt, const int64_t *d_ll_off, float *d_loglikes) {
  if (!c->gmm_ready) return c->fail("mfa_load_gmm has not been called");
  if (!mfa_gmm_lazy_supported(c)) return c->fail("lazy scoring needs a model of at most 48 dimensions");
  if (ws->window <= 0 || ws->window % 64 != 0) return c->fail("scoring window must be a multiple of 64 frames");
  GmmParams p;
  memset(&p, 0, sizeof(p));
  p.dim = c->dim; p.kpad = c->kpad; p.num_rows = c->num_rows;
  p.w = c->d_w; p.gc = c->d_gc; p.row0 = c->d_row0; p.nblk = c->d_nblk; p.slot = c->d_slot;
  p.feats = lazy->d_feats; p.frame_off = d_frame_off; p.pdf_list = lazy->plan.d_pdf_list; p.pdf_off = lazy->plan.d_pdf_off;
  p.class_counts = lazy->plan.d_class_counts; p.ll_off = d_ll_off; p.out = d_loglikes;
  p.min_log_diff = logf(1.1920928955078125e-07f);
  p.first_frame = lazy->plan.d_pdf_first_frame; p.last_depth = lazy->plan.d_pdf_last_depth;
  p.n_utt = n_utt; p.tiles = 0;
  p.acc_scale_inv = 1.0f;
  p.trace = (unsigned long long *)c->gmm_trace;
  p.b_hi_slack = ws->hi_slack > 0 ? ws->hi_slack : 0;
  p.b_mode = 1; p.b_t_begin = ws->t_begin; p.b_sub = ws->window / 64; p.b_band = ws->band;
  p.b_utt_list = ws->utt_list; p.b_n_list = ws->n_list;
  p.b_done = ws->done; p.b_done_stride = ws->done_stride; p.b_done_word = ws->done_word;
  p.b_lag = ws->lag; p.b_lag_stride = ws->lag_stride; p.b_lag_word = ws->lag_word; p.b_lag_frames = ws->window;
  const int64_t waves = (int64_t)n_utt * p.b_sub;
  const dim3 grid((unsigned)((waves + 3) / 4));
  p.groups = lazy->plan.groups > 1 && lazy->plan.d_group_counts ? lazy->plan.groups : 0;
  p.group_counts = p.groups ? lazy->plan.d_group_counts : nullptr;
  p.b_chunk = (ws->cols_per_wave > 0 && !p.groups) ? ws->cols_per_wave : 0;   // (a grouped plan already spreads the band over `groups` wavefronts)
  p.b_nchunk = p.b_chunk > 0 ? (lazy->plan.max_cols + p.b_chunk - 1) / p.b_chunk : 1;
  const int64_t split_waves = waves * p.b_nchunk;
  p.b_split = p.groups > 1 ? 1 : 0;
  const dim3 split_grid((unsigned)((split_waves + 3) / 4) * (unsigned)(p.b_split ? p.groups : 1));
  const int m8 = c->kpad / 8;
  const char *bf = getenv("MFA_GMM_BF16");
  const char *hf = getenv("MFA_GMM_F16");
  KernelTimer kt(c, MFA_K_GMM);
  {   // the band's index ranges, once per utterance (instead of once per scoring wavefront)
    const int64_t need = (int64_t)n_utt * kRangeSlots * 2;
    if (c->band_ranges_cap < need) {
      MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (c->d_band_ranges) (void)hipFree(c->d_band_ranges);
      c->d_band_ranges = nullptr; c->band_ranges_cap = 0;
      MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_band_ranges, (size_t)need * sizeof(int32_t)));
      c->band_ranges_cap = need;
    }
    p.ranges = c->d_band_ranges;
    hipLaunchKernelGGL(gmm_band_ranges_kernel, dim3((unsigned)((n_utt + 3) / 4)), dim3(256), 0, c->stream, p);
  }
  const bool split_classes = c->has_single32 || c->has_multi_block || c->has_slot_class[1] || c->has_slot_class[2] || c->has_slot_class[3];
  if (!(bf && bf[0] == '0') && c->d_wb && split_classes) {
    const bool f16_ok = !(hf && hf[0] == '0') && c->d_wh;
    if (c->gmm_redo_cap < split_waves) {
      if (c->d_gmm_redo) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_gmm_redo); }
      c->d_gmm_redo = nullptr; c->gmm_redo_cap = 0;
      MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmm_redo, split_waves * sizeof(int)));
      c->gmm_redo_cap = split_waves;
    }
    p.redo = c->d_gmm_redo;
    p.wb = (const uint4 *)c->d_wb;
    p.col_row0 = c->d_col_row0;
    p.col_nb_packed = col_nb_packed_for(c);
    if (f16_ok) {
      p.wh = (const uint4 *)c->d_wh; p.gch = c->d_gch; p.fscale = c->d_fscale;
      p.acc_scale_inv = 1.0f / c->gmm_acc_scale;
      p.redo_mode = 0;
      if (c->xsplit_ready) { p.xsplit = (const uint4 *)c->d_xsplit; p.xsplit_bad = c->d_xsplit_bad; }
      if (m8 == 10) hipLaunchKernelGGL((gmm_band_kernel<5, 2>), split_grid, dim3(256), 0, c->stream, p);
      else hipLaunchKernelGGL((gmm_band_kernel<6, 2>), split_grid, dim3(256), 0, c->stream, p);
      p.redo_mode = 2;   // the sub-tiles the f16 pass flagged
    }
    if (m8 == 10) hipLaunchKernelGGL((gmm_band_kernel<5, 3>), split_grid, dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL((gmm_band_kernel<6, 3>), split_grid, dim3(256), 0, c->stream, p);
    p.redo_mode = 0;
    p.b_skip0 = 1;
  }
  p.b_split = 0;   // (the f32 band kernel keeps one wavefront per sub-tile and walks the runs of class 0 itself)
  const bool f32_classes = c->has_slot_class[4];   // single Gaussians stay on the f32 pipe (bit-exact); everything else was scored above
  if (!p.b_skip0 || f32_classes) {
    if (m8 <= 10) hipLaunchKernelGGL((gmm_band_f32_kernel<10>), grid, dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL((gmm_band_f32_kernel<12>), grid, dim3(256), 0, c->stream, p);
  }
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}
