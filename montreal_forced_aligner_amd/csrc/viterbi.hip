// Batched beam-pruned token-passing Viterbi for gfx950: one wavefront per utterance, frame loop inside the kernel.
// Replaces GmmAligner.align_utterance / export_alignments (MFA/alignment/multiprocessing.py:846-853;
// MFA/online/alignment.py:107) = Kaldi AlignUtteranceWrapper + FasterDecoder.
//
// The result is REQUIRED to equal the sequential decoder's, so the kernel reproduces its order-dependent rules in
// parallel form rather than approximating them:
//   * tokens live in an ORDERED list (Kaldi HashList order: buckets by first occupancy, insertion order inside);
//   * a candidate (token i, arc k) is created iff its cost is below the running cutoff
//       min(seed from the best token, min over all EARLIER candidates) + adaptive_beam
//     — an exclusive prefix-min over candidates in list×arc order (wave scan, no serial loop);
//   * per destination state the cheapest candidate wins, ties to the earliest candidate (two-phase LDS atomics);
//   * the next list is produced by a prefix-sum compaction keyed by each state's first creating candidate
//     (and its hash bucket's, when the graph has more states than hash buckets);
//   * GetCutoff's min_active=20 rule is evaluated exactly (counting selection of the 21st smallest cost).
// Costs are float64 exactly as Kaldi's tokens; arc weights / acoustic costs float32.
//
// Memory: per-wavefront LDS holds only the atomically updated tables (state→slot map, per-slot cost/first/winner, the
// candidate-ordinal counter array); token lists, the candidate stash and the back-pointer records stream through a
// per-utterance HBM workspace (288 GB lets every in-flight utterance keep its own).  No MFMA: this is min-plus DP.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <vector>

#include "ctx.hpp"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u32 kEmpty = 0xFFFFFFFFu;
constexpr u32 kClaim = 0xFFFFFFFEu;
constexpr u32 kOver = 0xFFFFFFFDu;   // hash bucket of a state that found no free slot (the frame is about to report overflow)
constexpr u64 kKeyInf = 0xFFFFFFFFFFFFFFFFull;
constexpr int kMinActive = 20;
constexpr float kBeamDelta = 0.5f;
constexpr float kHashRatio = 2.0f;
constexpr int kArcBits = 6;               // at most 64 arcs per state
constexpr int kMaxArcsPerState = 1 << kArcBits;

enum { ST_OK = 0, ST_RETRIED = 1, ST_FAILED = 2, ST_TOKEN_OVERFLOW = 3, ST_BP_OVERFLOW = 4, ST_UNSUPPORTED = 5, ST_INTERNAL = 6, ST_WORDS = 7, ST_PENDING = -1, ST_GROW = -2 };

// decoder state of one utterance between two windows (the token list itself is parked in w_state / w_cost)
struct VitState { int32_t n, cur, done, pad0; u32 H, pad1; u64 bp_used; };

struct VitParams {
  mfa_graph_batch g;
  const float *ll; const int64_t *ll_off; const int32_t *ll_cols; const int64_t *frame_off;
  float beam, scale;
  int nmax, cmax, bpf;        // live-token capacity, candidate capacity, back-pointer tokens per frame
  int hbits;                  // log2 of the state→slot hash table size (>= 4 x nmax entries); 0: direct map, one entry per state
  const int32_t *utt_list;    // utterances to decode (NULL: identity)
  const int32_t *n_list;      // number of entries in utt_list (device scalar) or NULL
  int pass;                   // 0 first beam, 1 retry
  int grow;                   // 1: a token/candidate overflow is not final — the utterance is re-run with larger tables
  // workspace
  u32 *w_state; double *w_cost;        // [n_utt][2][nmax]
  u32 *w_stash_a; u64 *w_stash_key;    // [n_utt][cmax]: (slot<<32|cidx) packed in stash_a pair → two arrays
  u32 *w_stash_b;
  u64 *w_bp;                           // [total_frames*bpf] (arc index <<32 | prev pos)
  u32 *w_tokoff;                       // [total_frames + n_utt]
  u32 *w_hash;                         // [n_utt] hash size carried from pass 0 to the retry pass
  const uint4 *w_arcnext;              // [total_arcs] {next, (arc_off[next] << 7) | out-degree(next), col, weight}, built once per call
  // Graphs with epsilon input arcs (g.d_state_nemit != NULL: every state's arcs are stored [emitting | epsilon] and the
  // out-degree above counts the emitting ones): per state {first epsilon arc << 7 | number of epsilon arcs}, built once per call
  const u32 *w_epsinfo;                // [n_utt * max_states] at (utt * max_states + state), or NULL
  int eps_stride;                      // max_states
  int lagmode;                         // windowed first-beam pass with the 64-token first tier: VitState.pad0 is the utterance's lag —
                                       // 1: its window in this launch is the previous one (a failed speculation being redone with the
                                       // proven band, by the first tier itself); the utterance stays one window behind from then on
  int eps_pops;                        // pops of one frame's epsilon closure before the utterance is handed back with a capacity status (64 per token slot; Kaldi has no budget: the caller's last resort is the general decoder)
  unsigned long long *stamps;          // -DVIT_STAMPS builds: per-utterance phase cycles (mfa_debug_viterbi_stamps) or NULL
  int llcap;                           // score-row cache capacity in LDS (floats); rows longer than this are read from HBM
  // windowed (resumable) decoding — mfa_align_features_batch: one launch decodes frames [t_begin, t_end) of every utterance,
  // parks the live token list in the HBM workspace and leaves the band of graph depths the NEXT window can touch
  int windowed, t_begin, t_end, next_window;
  // Two table sizes per window (first-beam pass of mfa_align_features_batch): redo_mode 1 = the small first tier — an
  // utterance that runs out of token slots (or enters the window with more tokens than the tier holds) is flagged in
  // w_redo and left exactly as it was parked at the window's start; redo_mode 2 = the large tier, launched right after for
  // the SAME window: only flagged utterances run.  0: a token overflow is final (or ST_GROW).
  int redo_mode;
  u32 *w_redo;                         // [n_utt]
  int npark;                           // stride (tokens) of the parked lists: the large tier's capacity, whatever tier runs
  VitState *w_vstate;                  // [n_utt]
  // Speculative look-ahead (first-beam windowed pass): the window was scored for a band narrower than the proven one; the
  // decoder checks every score it reads against the column ranges that were scored (spec_ranges, see mfa_band_ranges) and
  // gives the utterance up (ST_GROW: decoded again from frame 0 by the list pass, proven bands) the moment one lies outside.
  int spec;                            // 1: check
  const int32_t *spec_ranges;          // [n_utt][kMfaRangeSlots][2]
  const int32_t *spec_class_counts;    // [n_utt][6]
  int spec_groups;                     // runs of class 0 in the plan (0/1: one)
  const int32_t *state_depth;          // [total_states][2] {fewest arcs from start, most arcs from start} (mfa_score_plan)
  int32_t *band;                       // [n_utt][2] out: {min longest-path depth of a live token, max BFS depth + next_window - 1}
  // outputs
  int32_t *ali; int32_t *words; int32_t *n_words; float *like; float *frame_like; int32_t *status;
};

__device__ __forceinline__ u64 dkey(double d) {
  u64 b = (u64)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(u64 k) {
  u64 b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// ---- wavefront primitives on the DPP crossbar (row_shr / row_bcast / wave_shr are single VALU operand modifiers on
// gfx950; the ds_bpermute-based __shfl costs an LDS round trip per step).  Inclusive scan: Kogge-Stone inside each row
// of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals; lane 63 ends up holding the reduction.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ u32 dpp_u32(u32 old, u32 v) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ u32 incl_scan_sum(u32 v) {
  v += dpp_u32<0x111>(0, v); v += dpp_u32<0x112>(0, v); v += dpp_u32<0x114>(0, v); v += dpp_u32<0x118>(0, v);
  v += dpp_u32<0x142, 0xA>(0, v); v += dpp_u32<0x143, 0xC>(0, v);
  return v;
}
__device__ __forceinline__ u32 incl_scan_max(u32 v) {
  v = max(v, dpp_u32<0x111>(0, v)); v = max(v, dpp_u32<0x112>(0, v)); v = max(v, dpp_u32<0x114>(0, v));
  v = max(v, dpp_u32<0x118>(0, v)); v = max(v, dpp_u32<0x142, 0xA>(0, v)); v = max(v, dpp_u32<0x143, 0xC>(0, v));
  return v;
}
// min of two costs as ONE instruction.  fmin() is llvm.minnum: with IEEE mode on it first canonicalises both operands
// (v_max_f64 x, x, x) in case one is a signalling NaN — the decoder's costs are finite or +inf, never NaN, and every DPP scan
// step paid two extra double-rate instructions for it (62 canonicalisations against 50 minima in the first-tier kernel).
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double incl_scan_min(double v) {
  const double inf = INFINITY;
  v = min_f64(v, dpp_f64<0x111>(inf, v)); v = min_f64(v, dpp_f64<0x112>(inf, v)); v = min_f64(v, dpp_f64<0x114>(inf, v));
  v = min_f64(v, dpp_f64<0x118>(inf, v)); v = min_f64(v, dpp_f64<0x142, 0xA>(inf, v)); v = min_f64(v, dpp_f64<0x143, 0xC>(inf, v));
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) { return readlane_f64(incl_scan_min(v), 63); }
__device__ __forceinline__ u32 wave_max_u32(u32 v) { return (u32)__builtin_amdgcn_readlane((int)incl_scan_max(v), 63); }
// exclusive prefixes from an inclusive scan: shift the wavefront right by one lane (wave_shr:1), identity into lane 0
__device__ __forceinline__ double shift_in_min(double incl) { return dpp_f64<0x138>((double)INFINITY, incl); }

// Kaldi: ac_cost = -(scale * loglike) in float; new_weight = (double)arc.weight + tok.cost + ac_cost
__device__ __forceinline__ double cand_cost(float w, double cost, float ll, float scale) {
  float ac = -(scale * ll);
  return ((double)w + cost) + (double)ac;
}

// hand-over point between lanes of one wavefront (see the LDS carve comment in the kernel)
// Optional per-phase cycle accounting (-DVIT_STAMPS): s_memtime deltas accumulated per phase over all frames of an
// utterance, written to the buffer given to mfa_debug_viterbi_stamps ([n_utt][12] uint64; tools/viterbi_phases.py).
#ifdef VIT_STAMPS
#define STAMP(k)                                                                   \
  do {                                                                             \
    unsigned long long _t;                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");      \
    stamp_acc[k] += _t - stamp_last;                                               \
    stamp_last = _t;                                                               \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

#define WSYNC()                                            \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// Columns scored for the current window as a bitmap in LDS (one wavefront; kBmWords × 32 columns).
constexpr int kBmWords = 64;
__device__ __forceinline__ void build_scored_bitmap(const VitParams &p, int utt, int lane, u32 *bm) {
  for (int i = lane; i < kBmWords; i += 64) bm[i] = 0u;
  WSYNC();
  const int32_t *rg = p.spec_ranges + (size_t)utt * kMfaRangeSlots * 2;
  const int32_t *cc6 = p.spec_class_counts + (size_t)utt * 6;
  const int runs = p.spec_groups > 1 ? p.spec_groups : 1;
  for (int slot = 0; slot < kMfaRangeSlots; slot++) {
    if (slot < kMfaRunSlots && slot >= runs) continue;
    int base = 0;                                                            // first column of the slot's class
    if (slot == kMfaRunSlots + 3) base = cc6[0];                             // class 1
    else if (slot == kMfaRunSlots + 4) base = cc6[0] + cc6[1] + cc6[2] + cc6[3] + cc6[4];   // class 5
    else if (slot >= kMfaRunSlots) { base = cc6[0] + cc6[1]; for (int k = 2; k < slot - kMfaRunSlots + 2; k++) base += cc6[k]; }
    const int a = base + rg[2 * slot], b = base + rg[2 * slot + 1];
    for (int c = a + lane; c < b; c += 64)
      if (c >= 0 && c < 32 * kBmWords) atomicOr(&bm[c >> 5], 1u << (c & 31));
  }
  WSYNC();
}
__device__ __forceinline__ bool column_scored(const u32 *bm, int col) {
  return (u32)col < (u32)(32 * kBmWords) && ((bm[col >> 5] >> (col & 31)) & 1u) != 0u;
}

// ReachedFinal / best final token, traceback, outputs (transition-ids, words, likelihood) of one utterance whose frame
// loop has ended with `n` tokens in (c_state, c_cost) after `t` frames.  One wavefront; shared by the frame-loop kernels
// and by viterbi_finish_kernel.
template <class StateP, class CostP>
__device__ __forceinline__ void finalize_utterance(const VitParams &p, int utt, int lane, int status, int t, int T, int n,
                                                   StateP c_state, CostP c_cost, const float *final_w, const u64 *bp,
                                                   const u32 *tokoff, int64_t f0, int64_t ab_, const float *a_w,
                                                   const int32_t *a_col, const float *ll, int P, bool eps = false,
                                                   u64 bp_used = 0, u64 bp_cap = 0) {
  // ---------------- ReachedFinal / best final token (first in list order on ties)
  int32_t out_status = status;
  double bestc = INFINITY; u32 bpos = kEmpty;
  if (status == ST_OK) {
    if (t < T || n == 0) out_status = ST_FAILED;
    else {
      for (int c0 = 0; c0 < n; c0 += 64) {
        int i = c0 + lane;
        double tc = INFINITY;
        if (i < n) {
          float fw = final_w[c_state[i]];
          if (fw != INFINITY) tc = c_cost[i] + (double)fw;
        }
        double m = wave_min_f64(tc);
        if (m < bestc) {
          const u64 hit = __ballot(i < n && tc == m);
          bpos = (u32)c0 + (u32)__ffsll((long long)hit) - 1u;
          bestc = m;
        }
      }
      if (bpos == kEmpty) out_status = ST_FAILED;
    }
  }
  if (out_status != ST_OK) {
    if (lane == 0) {
      // a first-pass failure stays pending for the retry pass; other codes are final
      p.status[utt] = (p.pass == 0 && out_status == ST_FAILED) ? ST_PENDING
                      : (p.grow && out_status == ST_TOKEN_OVERFLOW) ? ST_GROW : out_status;
      p.n_words[utt] = 0; p.like[utt] = 0.0f;
    }
    return;
  }

  int32_t *ali = p.ali + f0;
  const u32 fstate = c_state[bpos];
  const int32_t *a_il = p.g.d_arc_ilabel + ab_, *a_ol = p.g.d_arc_olabel + ab_;
  int32_t *words = p.words + f0;
  float *flike = p.frame_like ? p.frame_like + f0 : nullptr;
  u32 nw_out = 0;
  double cost = 0.0; float w1 = 0.0f, w2 = 0.0f;
  const float inv_scale = -1.0f / p.scale;
  if (eps) {
    // ---------------- graphs with epsilon input arcs: a frame's record may be followed by records of the SAME frame's list
    // (tokens that came over epsilon arcs), so the path has T + E arcs.  Traceback writes their arc indices, back to front,
    // into the unused tail of the utterance's back-pointer area; the forward pass below reads them in path order: words from
    // any arc, a transition-id and a frame only from the emitting ones, the float accumulation of Kaldi's
    // GetLinearSymbolSequence over all of them (an epsilon arc adds its weight and the rounding residue of its cost step).
    u32 *path = (u32 *)(bp + bp_used);
    const u64 pcap64 = (bp_cap - bp_used) * 2ull;
    const u32 pcap = pcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (u32)pcap64;
    u32 wpos = pcap, pos = bpos;
    bool full = false;
    for (int tt = T - 1; tt >= 0 && !full; tt--) {
      const u32 to = tokoff[tt];
      for (int hop = 0; ; hop++) {
        const u64 rec = bp[(u64)to + pos];
        const u32 arc = (u32)(rec >> 32);
        pos = (u32)(rec & 0xFFFFFFFFu);
        if (wpos == 0u || hop > 4096) { full = true; break; }
        wpos--;
        if (lane == 0) path[wpos] = arc;
        if (a_il[arc] != 0) break;              // an emitting arc: `pos` now refers to the previous frame's list
      }
    }
    // the initial list (InitDecoding's closure): records bp[0 .. n_init), the start token's carries arc 0xFFFFFFFF
    for (int hop = 0; !full; hop++) {
      const u64 rec = bp[pos];
      const u32 arc = (u32)(rec >> 32);
      if (arc == 0xFFFFFFFFu) break;
      if (wpos == 0u || hop > 4096) { full = true; break; }
      wpos--;
      if (lane == 0) path[wpos] = arc;
      pos = (u32)(rec & 0xFFFFFFFFu);
    }
    if (full) {
      if (lane == 0) { p.status[utt] = ST_BP_OVERFLOW; p.n_words[utt] = 0; p.like[utt] = 0.0f; }
      return;
    }
    __threadfence_block();
    WSYNC();
    const u32 L = pcap - wpos;
    u32 frames_done = 0;
    for (u32 c0 = 0; c0 < L; c0 += 64) {
      const u32 i = c0 + (u32)lane;
      int arc = 0, il = 0, ol = 0; float w = 0.0f, ac = 0.0f;
      if (i < L) { arc = (int)path[wpos + i]; il = a_il[arc]; ol = a_ol[arc]; w = a_w[arc]; }
      const u64 em = __ballot(i < L && il != 0);
      const u32 tt = frames_done + (u32)__popcll(em & ((1ull << lane) - 1ull));
      if (i < L && il != 0 && tt < (u32)T) ac = -(p.scale * ll[(size_t)tt * P + a_col[arc]]);
      const u64 mask = __ballot(ol != 0);
      const u32 wat = nw_out + (u32)__popcll(mask & ((1ull << lane) - 1ull));
      if (ol != 0 && wat < (u32)T) words[wat] = ol;
      nw_out += (u32)__popcll(mask);
      float my_fl = 0.0f;
      const int lim = (int)min(64u, L - c0);
      for (int j = 0; j < lim; j++) {
        float wj = __shfl(w, j), acj = __shfl(ac, j);
        double nc = ((double)wj + cost) + (double)acj;
        float tot = (float)(nc - cost);
        float acost = tot - wj;
        w1 += wj; w2 += acost;
        cost = nc;
        if (lane == j) my_fl = acost * inv_scale;
      }
      if (i < L && il != 0 && tt < (u32)T) { ali[tt] = il; if (flike) flike[tt] = my_fl; }
      frames_done += (u32)__popcll(em);
    }
    if (nw_out > (u32)T) {   // more word labels than frames (output labels on epsilon arcs): the output layout cannot hold them
      if (lane == 0) { p.status[utt] = ST_WORDS; p.n_words[utt] = 0; p.like[utt] = 0.0f; }
      return;
    }
  } else {
  // ---------------- traceback, arc index per frame parked in ali[].  The chain is pos → record → pos; the per-frame offsets
  // do not depend on it, so 64 of them are fetched at once and handed out by v_readlane: one dependent load per frame
  // instead of two (every lane walks the same chain on broadcast addresses; lane 0 stores).
  {
    u32 pos = bpos;
    for (int c0 = T - 1; c0 >= 0; c0 -= 64) {
      const int tl = c0 - lane;
      const u32 tokv = tl >= 0 ? tokoff[tl] : 0u;
      const int cnt = min(64, c0 + 1);
      for (int k = 0; k < cnt; k++) {
        const u32 to = (u32)__builtin_amdgcn_readlane((int)tokv, k);
        const u64 rec = bp[(u64)to + pos];
        if (lane == 0) ali[c0 - k] = (int32_t)(rec >> 32);
        pos = (u32)(rec & 0xFFFFFFFFu);
      }
    }
  }
  __threadfence_block();
  WSYNC();
  // ---------------- outputs: transition-ids, words (ordered compaction), likelihood (Kaldi's float accumulation)
  for (int c0 = 0; c0 < T; c0 += 64) {
    const int tt = c0 + lane;
    int arc = tt < T ? ali[tt] : 0;
    int il = 0, ol = 0; float w = 0.0f, ac = 0.0f;
    if (tt < T) {
      il = a_il[arc]; ol = a_ol[arc]; w = a_w[arc];
      ac = -(p.scale * ll[(size_t)tt * P + a_col[arc]]);
    }
    // words in path order
    const u64 mask = __ballot(ol != 0);
    if (ol != 0) words[nw_out + __popcll(mask & ((1ull << lane) - 1ull))] = ol;
    nw_out += (u32)__popcll(mask);
    // cost chain, sequential in frame order (every lane runs the same chain on broadcast operands)
    float my_fl = 0.0f;
    const int lim = min(64, T - c0);
    for (int j = 0; j < lim; j++) {
      float wj = __shfl(w, j), acj = __shfl(ac, j);
      double nc = ((double)wj + cost) + (double)acj;
      float tot = (float)(nc - cost);
      float acost = tot - wj;
      w1 += wj; w2 += acost;
      cost = nc;
      if (lane == j) my_fl = acost * inv_scale;
    }
    if (tt < T) { ali[tt] = il; if (flike) flike[tt] = my_fl; }
  }
  }
  if (lane == 0) {
    w1 += final_w[fstate];
    p.like[utt] = -(w1 + w2) / p.scale;
    p.n_words[utt] = (int32_t)nw_out;
    p.status[utt] = p.pass == 0 ? ST_OK : ST_RETRIED;
  }
}

constexpr int kArcCache = 8;  // arcs per token kept in registers during expansion (deeper states take a slow tail loop)

// kListsInLds: the two token lists (state, cost) live in LDS (fast path) or, for graphs/beams whose tables would not
// fit in 160 KiB, in the per-utterance HBM workspace.
// (waves_per_eu 4: at most 128 VGPRs, so that the 9.5 KB first tier really gets its 16 wavefronts per CU)
// kEps: the instantiation for batches that hold graphs with epsilon input arcs (g.d_state_nemit): every frame's emitting phase
// is followed by FasterDecoder::ProcessNonemitting — see the closure block in the frame loop.  The epsilon-free instantiation
// is the code it always was.
template <bool kListsInLds, bool kEps = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void viterbi_kernel(VitParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  int utt = blockIdx.x;
  if (p.utt_list) {
    if (p.n_list && (int)blockIdx.x >= *p.n_list) return;
    utt = p.utt_list[blockIdx.x];
  }
  if (p.redo_mode == 2 && p.w_redo[utt] == 0u) return;   // large tier: only what the small tier handed over
  const int64_t so = p.g.d_state_off[utt];
  const int S = (int)(p.g.d_state_off[utt + 1] - so);
  const int64_t ab_ = p.g.d_arc_base[utt];
  const int32_t *arc_off = p.g.d_arc_off + so + utt;
  const float *final_w = p.g.d_final + so;
  const float *a_w = p.g.d_arc_weight + ab_;
  const int32_t *a_col = p.g.d_arc_col + ab_;
  const uint4 *a_rec = p.w_arcnext + ab_;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const float *ll = p.ll + p.ll_off[utt];
  const int P = p.ll_cols[utt];
  const int N = p.nmax, C = p.cmax;

  // ---- LDS carve (8-byte items first).  These tables carry values between lanes of ONE wavefront: LDS operations of
  // a wavefront execute in program order, so plain accesses are enough provided the compiler keeps them on the right
  // side of each hand-over point — that is what WSYNC() (a wavefront-scope fence pair + scheduling barrier) is for.
  // Within a phase the loads stay free to be issued back to back (the volatile version of round 1 waited on every one).
  u64 *s_cost = (u64 *)smem;                  // [N] best cost key per slot
  double *l_cost0 = kListsInLds ? (double *)(s_cost + N)   // [2][N] token costs (current / next list)
                                         : (double *)(p.w_cost + (size_t)utt * 2 * N);
  // state → slot: an open-addressing hash table (linear probing) over the states that received a candidate THIS frame —
  // at most N of them, whatever the size of the graph, so the table is 4N entries instead of one per graph state
  // (round 1: a direct map, 10.8 KB of the 23 KB a 2 700-state graph needed → 6 wavefronts per CU; now 16).
  u32 *hmap = (u32 *)(s_cost + (kListsInLds ? 3 : 1) * (size_t)N);  // [HM] slot index | kEmpty | kClaim | kOver
  // (large tiers, whose 4N-entry table would be bigger than one entry per graph state, address the table by state id:
  //  hbits = 0 — same code, no collisions)
  const bool hdirect = p.hbits == 0;
  const u32 HM = hdirect ? (u32)((S + 1) & ~1) : 1u << p.hbits, hmask = hdirect ? 0xFFFFFFFFu : HM - 1u;
  const int hshift = hdirect ? 0 : 32 - p.hbits;
  u32 *s_state = hmap + HM;         // [N]
  u32 *s_F = s_state + N;           // [N] first creating candidate (pos<<6|k)
  u32 *s_W = s_F + N;               // [N] winning candidate
  u32 *s_aux = s_W + N;             // [N] (rank<<24)|ordinal of the bucket leader's first candidate
  u32 *t_cbase = s_aux + N;         // [N] candidate ordinal base per source token
  u32 *s_an = t_cbase + N;          // [N] (first arc << 7 | out-degree) of the slot's state
  u32 *s_bucket = s_an + N;         // [N] hash bucket the slot's state was filed under (reset at the end of the frame)
  u32 *l_state0 = kListsInLds ? s_bucket + N : (u32 *)(p.w_state + (size_t)utt * 2 * N);  // [2][N] token states
  u32 *l_an0 = kListsInLds ? l_state0 + 2 * N : (u32 *)(p.w_state + (size_t)p.g.n_utt * 2 * N + (size_t)utt * 2 * N);
  u32 *cntord = s_bucket + N + (kListsInLds ? 4 * N : 0);  // [C] bucket sizes at leader ordinals → exclusive sums
  float *ll_row = (float *)(cntord + C);      // [llcap] this frame's score row
  u32 *ctr = (u32 *)(ll_row + p.llcap);       // [2]: nslots, nstash
  u32 *bm = ctr + 4;                          // [kBmWords] columns scored for this window (speculative look-ahead only)
  // epsilon closure (kEps): per slot its list position, the inverse, the winning epsilon arc, a scratch word for the winner
  // vote, the state's epsilon-arc info; and the stack of ProcessNonemitting
  u32 *e_pos = bm + kBmWords;                 // [N]
  u32 *e_inv = e_pos + N;                     // [N]
  u32 *e_arc = e_inv + N;                     // [N]
  u32 *e_tmp = e_arc + N;                     // [N]
  u32 *e_info = e_tmp + N;                    // [N]
  u32 *e_stk = e_info + N;                    // [2N]
  if constexpr (kEps) {
    for (int i = lane; i < N; i += 64) e_tmp[i] = 0xFFFFFFFFu;
  }

  u32 *st_a = p.w_stash_a + (size_t)utt * C;
  u32 *st_b = p.w_stash_b + (size_t)utt * C;
  u64 *st_key = p.w_stash_key + (size_t)utt * C;
  u64 *bp = p.w_bp + (size_t)f0 * p.bpf;
  const u64 bp_cap = (u64)T * (u64)p.bpf;
  u32 *tokoff = p.w_tokoff + f0 + utt;

  for (u32 i = lane; i < HM; i += 64) hmap[i] = kEmpty;
  for (int i = lane; i < C; i += 64) cntord[i] = 0;
  if (lane == 0) { ctr[0] = 0; ctr[1] = 0; }

  int status = ST_OK;
  const int start = p.g.d_start[utt];
  if (S <= 0 || start < 0 || start >= S || T <= 0) status = ST_FAILED;
  // (lagmode: an utterance one window behind — VitState.pad0 — works on [t_begin − K, t_begin))
  const int lag = (p.lagmode && p.windowed && p.t_begin > 0) ? (p.w_vstate[utt].pad0 != 0 ? 1 : 0) : 0;
  const int t_begin_u = p.t_begin - lag * (p.t_end - p.t_begin);
  const bool resume = p.windowed && t_begin_u > 0;
  int cur = 0, n = 1;
  u32 H = p.pass == 0 ? 1000u : p.w_hash[utt];
  u64 bp_used = 0;
  int t = 0;
  // token lists parked in HBM between windows (the kListsInLds = false variant keeps them there all the time)
  const int NP = kListsInLds ? p.npark : N;
  u32 *park_state = p.w_state + (size_t)utt * 2 * NP;
  u32 *park_an = p.w_state + (size_t)p.g.n_utt * 2 * NP + (size_t)utt * 2 * NP;
  double *park_cost = p.w_cost + (size_t)utt * 2 * NP;
  if (!resume) {
    // InitDecoding: one token at the start state with cost 0 (epsilon-free graphs: ProcessNonemitting is a no-op)
    if (lane == 0) {
      const int s0 = start < 0 || start >= S ? 0 : start;
      l_state0[0] = (u32)s0; l_cost0[0] = 0.0;
      u32 deg0 = S > 0 ? (u32)(arc_off[s0 + 1] - arc_off[s0]) : 0u;
      if constexpr (kEps) { if (S > 0 && p.g.d_state_nemit) deg0 = (u32)p.g.d_state_nemit[so + s0]; }
      l_an0[0] = S > 0 ? ((u32)arc_off[s0] << 7) | min(deg0, 127u) : 0u;
    }
    if constexpr (kEps) {
      // ---------------- InitDecoding's ProcessNonemitting(cutoff = FLT_MAX), as Kaldi runs it: a stack, the popped token's
      // epsilon arcs one after the other.  Once per utterance and a handful of tokens, so the wavefront simply walks the
      // sequential algorithm (every lane the same scalars, lane 0 stores, destinations looked up by a ballot over the tokens
      // created so far).  Tokens in creation order: s_state / s_cost (cost bits) / e_arc / e_inv (creator) / s_an; then the
      // hash-list order (buckets by first occupancy, creation order inside) gives the initial list and its back-pointer
      // records — bp[0 .. n): the start token's carries arc 0xFFFFFFFF, the others an epsilon arc and a position in this list.
      if (status == ST_OK) {
        u32 nc_ = 1u;
        if (lane == 0) {
          s_state[0] = (u32)start; s_cost[0] = (u64)__double_as_longlong(0.0); e_arc[0] = 0xFFFFFFFFu; e_inv[0] = 0u;
          s_an[0] = l_an0[0]; e_stk[0] = 0u;
        }
        WSYNC();
        u32 sp = 1u;
        int guard = 0;
        bool over = false;
        while (sp > 0u && !over) {
          if (++guard > p.eps_pops) { over = true; break; }
          const u32 e = e_stk[sp - 1u];
          sp--;
          const double ce = __longlong_as_double((long long)s_cost[e]);
          const u32 ei = p.w_epsinfo[(size_t)utt * p.eps_stride + s_state[e]];
          const u32 n_eps = ei & 127u, first = ei >> 7;
          for (u32 k = 0; k < n_eps && !over; k++) {
            const uint4 rec = a_rec[first + k];
            const u32 d = rec.x;
            const double ncst = ce + (double)__uint_as_float(rec.w);
            if (ncst > (double)3.4028234663852886e38f) continue;      // cutoff = numeric_limits<float>::max()
            u32 found = kEmpty;
            for (u32 c0 = 0; c0 < nc_; c0 += 64) {
              const u32 c_ = c0 + (u32)lane;
              const u64 hit = __ballot(c_ < nc_ && s_state[c_] == d);
              if (hit) { found = c0 + (u32)__ffsll((long long)hit) - 1u; break; }
            }
            bool pushed = false; u32 who = 0u;
            if (found == kEmpty) {
              if (nc_ >= (u32)N) { over = true; break; }
              if (lane == 0) {
                s_state[nc_] = d; s_cost[nc_] = (u64)__double_as_longlong(ncst); e_arc[nc_] = first + k; e_inv[nc_] = e; s_an[nc_] = rec.y;
              }
              who = nc_; nc_++; pushed = true;
            } else if (__longlong_as_double((long long)s_cost[found]) > ncst) {
              if (lane == 0) { s_cost[found] = (u64)__double_as_longlong(ncst); e_arc[found] = first + k; e_inv[found] = e; }
              who = found; pushed = true;
            }
            if (pushed) {
              if (sp >= 2u * (u32)N) { over = true; break; }
              if (lane == 0) e_stk[sp] = who;
              sp++;
            }
            WSYNC();
          }
        }
        if (over) status = ST_TOKEN_OVERFLOW;
        else {
          // hash-list order: position of token c = number of tokens whose (bucket's first creator, own index) is smaller
          for (u32 c0 = 0; c0 < nc_; c0 += 64) {
            const u32 c_ = c0 + (u32)lane;
            if (c_ < nc_) {
              const u32 bc = s_state[c_] % H;
              u32 lead_c = c_;
              for (u32 x = 0; x < c_; x++) if (s_state[x] % H == bc) { lead_c = x; break; }
              u32 pos_ = 0;
              for (u32 x = 0; x < nc_; x++) {
                if (x == c_) continue;
                const u32 bx = s_state[x] % H;
                u32 lead_x = x;
                for (u32 y = 0; y < x; y++) if (s_state[y] % H == bx) { lead_x = y; break; }
                if (lead_x < lead_c || (lead_x == lead_c && x < c_)) pos_++;
              }
              e_pos[c_] = pos_;
            }
          }
          WSYNC();
          for (u32 c0 = 0; c0 < nc_; c0 += 64) {
            const u32 c_ = c0 + (u32)lane;
            if (c_ < nc_) {
              const u32 pos_ = e_pos[c_];
              l_state0[pos_] = s_state[c_];
              l_cost0[pos_] = __longlong_as_double((long long)s_cost[c_]);
              l_an0[pos_] = s_an[c_];
              bp[pos_] = ((u64)e_arc[c_] << 32) | (u64)(c_ == 0u ? 0u : e_pos[e_inv[c_]]);
            }
          }
          n = (int)nc_;
          bp_used = nc_;
          __threadfence_block();
          WSYNC();
        }
      }
    }
  } else {
    const VitState vs = p.w_vstate[utt];
    if (vs.done) return;                           // finished (or failed) in an earlier window: outputs are final
    n = vs.n; H = vs.H; bp_used = vs.bp_used; t = t_begin_u;
    if (p.redo_mode == 1 && n > N) {               // more live tokens than this tier holds: the large tier takes the window
      if (lane == 0) p.w_redo[utt] = 1u;           // (cannot happen in lag mode: an utterance that outgrows the tier has left it)
      return;
    }
    if (n < 0 || n > N) { n = 0; status = ST_INTERNAL; }
    if (kListsInLds) {
      cur = 0;
      for (int i = lane; i < n; i += 64) { l_state0[i] = park_state[i]; l_an0[i] = park_an[i]; l_cost0[i] = park_cost[i]; }
    } else {
      cur = vs.cur & 1;
    }
  }
  const int t_stop = p.windowed ? min(T, t_begin_u + (p.t_end - p.t_begin)) : T;
  // score rows are staged through LDS one frame ahead (registers hold row t+1 while frame t is processed)
  constexpr int kPre = 8;
  const bool row_cached = P <= p.llcap && P <= 64 * kPre;
  float pre[kPre];
#pragma unroll
  for (int r = 0; r < kPre; r++) pre[r] = (row_cached && t < T && lane + 64 * r < P) ? ll[(size_t)t * P + lane + 64 * r] : 0.0f;
  WSYNC();
  const bool spec = p.spec != 0 && p.windowed && lag == 0;
  if (spec) build_scored_bitmap(p, utt, lane, bm);
  bool viol = false;   // a score outside the scored columns was read this window
  bool spec_failed = false;

#ifdef VIT_STAMPS
  unsigned long long stamp_acc[12] = {0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  for (; t < t_stop && status == ST_OK; t++) {
    const float *llt = ll + (size_t)t * P;
    // The frame's score row moves registers → LDS right before its first use, not here: on this target stores count in
    // vmcnt like loads and retire in order, so a wait for the row (requested during the previous frame) at the top of the
    // frame would also sit out the previous frame's back-pointer stores.  By the arc gather's wait they have long landed
    // (5.91 → 5.80 ms per 2 048 utterances; -DVIT_STAGE_ROW_AT_TOP restores the old place).
    auto stage_row = [&]() {
#ifndef VIT_STAGE_ROW_AT_TOP
      if (row_cached) {
#pragma unroll
        for (int r = 0; r < kPre; r++) if (lane + 64 * r < P) ll_row[lane + 64 * r] = pre[r];
        WSYNC();
      }
#endif
    };
#ifdef VIT_STAGE_ROW_AT_TOP
    if (row_cached) {
#pragma unroll
      for (int r = 0; r < kPre; r++) if (lane + 64 * r < P) ll_row[lane + 64 * r] = pre[r];
      WSYNC();
    }
#endif
#ifdef VIT_STAMPS
    stamp_acc[10] += (unsigned long long)n;   // tokens entering the frame
    stamp_acc[11] += 1;                       // frames
#endif
    STAMP(0);   // score row staged
    // Two address spaces, two loads, never a pointer select: a select turns into a FLAT load, whose wait
    // (vmcnt(0) lgkmcnt(0)) drains every outstanding vector-memory operation — including the next row's prefetch.
    auto score = [&](int col) -> float {
      float v = ll_row[row_cached ? col : 0];   // LDS read, always (column 0 when the row is not staged)
      if (!row_cached) v = *(const volatile float *)&llt[col];  // rows wider than the LDS cache: straight from HBM/L2
      // (volatile: otherwise the two loads are merged back into one FLAT load of a selected address)
      if (spec) viol |= !column_scored(bm, col);
      return v;
    };
    // Next frame's score row: requested after this frame's last dependent global load (vmcnt retires in order, so a
    // prefetch issued before the arc gather would have to land before the gather's wait returns); it then has the
    // claim / order / write phases and the next GetCutoff to arrive.
    auto prefetch_next_row = [&]() {
      if (row_cached && t + 1 < t_stop) {   // (rows past the window are not scored yet)
        const float *nx_row = llt + P;
#pragma unroll
        for (int r = 0; r < kPre; r++) if (lane + 64 * r < P) pre[r] = nx_row[lane + 64 * r];
      }
    };
    u32 *n_state = l_state0 + (cur ^ 1) * N;
    u32 *c_an = l_an0 + cur * N, *n_an = l_an0 + (cur ^ 1) * N;
    double *c_cost = l_cost0 + cur * N, *n_cost = l_cost0 + (cur ^ 1) * N;
    // ---------------- GetCutoff: best cost (first index on ties), count
    double best = INFINITY; u32 best_i = kEmpty;
    for (int c0 = 0; c0 < n; c0 += 64) {
      int i = c0 + lane;
      double cst = i < n ? c_cost[i] : INFINITY;
      double m = wave_min_f64(cst);
      if (m < best) {  // uniform
        const u64 hit = __ballot(i < n && cst == m);
        best_i = (u32)c0 + (u32)__ffsll((long long)hit) - 1u;  // first index holding the minimum
        best = m;
      }
    }
    double wcut; float abeam;
    if (n <= kMinActive) { wcut = INFINITY; abeam = INFINITY; }
    else {
      const double beam_cut = best + p.beam;
      u32 kle = 0;
      for (int c0 = 0; c0 < n; c0 += 64) {
        int i = c0 + lane;
        kle += (u32)__popcll(__ballot(i < n && c_cost[i] <= beam_cut));
      }
      if (kle > (u32)kMinActive) { wcut = beam_cut; abeam = p.beam; }
      else {
#ifdef VIT_STAMPS
        stamp_acc[9] += 1;   // frames that needed the exact min_active selection
#endif
        // sorted[min_active] (> beam_cut) = the smallest cost that has at least min_active+1 costs ≤ it
        double v = INFINITY;
        if (n <= 64) {  // costs are in registers: broadcast each with v_readlane, no LDS traffic
          const double cst = lane < n ? c_cost[lane] : INFINITY;
          u32 le = 0;
          for (int j = 0; j < n; j++) le += (readlane_f64(cst, j) <= cst) ? 1u : 0u;
          if (lane < n && le > (u32)kMinActive) v = cst;
        } else {
          for (int c0 = 0; c0 < n; c0 += 64) {
            int i = c0 + lane;
            double cst = i < n ? c_cost[i] : INFINITY;
            u32 le = 0;
            for (int j = 0; j < n; j++) le += (c_cost[j] <= cst) ? 1u : 0u;  // LDS broadcast reads
            if (i < n && le > (u32)kMinActive) v = min_f64(v, cst);
          }
        }
        v = wave_min_f64(v);
        wcut = v;
        abeam = (float)(v - best + (double)kBeamDelta);
      }
    }
    STAMP(1);   // GetCutoff
    // PossiblyResizeHash
    { u32 want = (u32)((float)n * kHashRatio); if (want > H) H = want; }

    // Candidate creation in three wavefront phases (each phase's LDS operations are issued back to back):
    //   look up the destination's slot → claim missing slots (CAS; the winner allocates, initialises, publishes)
    //   → re-read the published slot and lower its cost / first-creator with LDS atomics.
    auto hash_of = [&](u32 d) -> u32 { return hdirect ? d : (d * 2654435761u) >> hshift; };
    // read-only lookup (after every candidate of the frame has been filed): slot of state d, or kEmpty
    auto find = [&](u32 d) -> u32 {
      u32 h = hash_of(d);
      for (;;) {
        const u32 v = hmap[h];
        if (v == kEmpty) return kEmpty;
        if (v < (u32)N && s_state[v] == d) return v;
        h = (h + 1u) & hmask;
      }
    };
    // One probe step of find-or-insert for a pending candidate (wavefront-collective loops below call it in rounds with
    // a hand-over point between rounds): an empty bucket is claimed by compare-and-swap, the winner allocates a slot,
    // initialises it and publishes the index; a lane that lost the race (kClaim seen, or its own CAS failed) looks again
    // next round and finds either its own state (done) or a different one (moves on to the next bucket).
    auto probe = [&](bool &pend, u32 &h, u32 &res, u32 d, u32 dan) {
      if (!pend) return;
      const u32 v = hmap[h];
      if (v == kEmpty) {
        if (atomicCAS(&hmap[h], kEmpty, kClaim) == kEmpty) {
          const u32 my = atomicAdd(&ctr[0], 1u);
          if (my < (u32)N) {
            s_state[my] = d; s_an[my] = dan; s_cost[my] = kKeyInf; s_F[my] = kEmpty; s_W[my] = kEmpty; s_bucket[my] = h;
            hmap[h] = my;
            res = my;
          } else {
            hmap[h] = kOver;            // no slot left: nslots > N is reported right after the expansion
            res = kEmpty;
          }
          pend = false;
        }
      } else if (v == kOver) {
        res = kEmpty; pend = false;
      } else if (v != kClaim) {
        if (s_state[v] == d) { res = v; pend = false; }
        else h = (h + 1u) & hmask;
      }
    };
    auto lower = [&](u32 s, double cnw, u32 cidx) {
      atomicMin(&s_cost[s], dkey(cnw));
      atomicMin(&s_F[s], cidx);
    };
    u32 cand_base = 0;
    bool bad_degree = false;
    bool used_stash = false;
    bool fast = false;
    double frame_min = INFINITY;   // (kEps) cheapest candidate of the frame: next_weight_cutoff = frame_min + adaptive_beam
    // ---------------- fast path (the common case): at most 64 tokens and at most 64 candidates this frame → ONE
    // candidate per lane.  The running cutoff is then a plain exclusive prefix-min across lanes, every candidate does one
    // arc fetch, one slot lookup, one claim/lower, and the winner check comes straight from its registers.
    if (n <= 64) {
      const double cst = lane < n ? c_cost[lane] : INFINITY;
      const u32 an = lane < n ? c_an[lane] : 0u;
      const bool act = lane < n && cst < wcut;
      const u32 narc = act ? (an & 127u) : 0u;
      const u32 narc_incl = incl_scan_sum(narc);
      const u32 cb = narc_incl - narc;
      const u32 ctot = (u32)__builtin_amdgcn_readlane((int)narc_incl, 63);
      if (ctot <= 64u) {
        fast = true;
        cand_base = ctot;
        if (lane < n) t_cbase[lane] = cb;
        s_aux[lane] = 0u;                       // s_aux is free until the ordering pass: owner map of the 64 ordinals
        WSYNC();
        if (narc > 0u) s_aux[cb] = (u32)lane + 1u;  // head of each token's candidate run
        WSYNC();
        STAMP(2);   // candidate layout (scan, owner map)
        const u32 tok1 = incl_scan_max(s_aux[lane]);
        const bool valid = (u32)lane < ctot;
        const u32 tok = valid ? tok1 - 1u : 0u;
        const double tcost = c_cost[tok];
        const u32 tan = c_an[tok];
        const u32 k = valid ? (u32)lane - t_cbase[tok] : 0u;
        const u32 a = (tan >> 7) + k;
        float w = 0.0f; int col = 0; u32 nx = 0u, nan_ = 0u;
        if (valid) { const uint4 rec = a_rec[a]; nx = rec.x; nan_ = rec.y; col = (int)rec.z; w = __uint_as_float(rec.w); }
        stage_row();
        const double nw = valid ? cand_cost(w, tcost, score(col), p.scale) : INFINITY;
        STAMP(3);   // arc gather + score + cost
        const double seed = wave_min_f64((valid && tok == best_i) ? nw : INFINITY);  // the best token's candidates
        const double m_incl = incl_scan_min(nw);
        const double local = min_f64(seed, shift_in_min(m_incl));
        if constexpr (kEps) frame_min = min_f64(seed, readlane_f64(m_incl, 63));
        const bool created = valid && nw < local + (double)abeam;
        const u32 cidx = (tok << kArcBits) | k;
        STAMP(4);   // running cutoff (seed, prefix-min)
        u32 sl = kEmpty;
        {
          bool pend = created; u32 h = hash_of(nx);
          while (__any(pend)) {                           // slots are published before anybody re-reads
            probe(pend, h, sl, nx, nan_);
            WSYNC();
            if (ctr[0] > (u32)N) break;                   // out of slots: the frame reports the overflow below
          }
        }
        if (sl != kEmpty) lower(sl, nw, cidx);
        WSYNC();  // every candidate of the frame has lowered its slot's cost
        if (sl != kEmpty && dkey(nw) == s_cost[sl]) atomicMin(&s_W[sl], cidx);
        STAMP(5);   // claim / lower / winner
      }
    }
    if (!fast) {
      stage_row();
      // ---------------- seed of the running cutoff: the best token's cheapest candidate.  With a single chunk it is
      // taken from the expansion's registers below; otherwise computed here.
      const bool single = n <= 64;
      double run = INFINITY;  // min over candidate costs seen so far (seed + earlier candidates)
      if (!single && best_i != kEmpty) {
        const u32 ban = c_an[best_i];
        const int a0 = (int)(ban >> 7), a1 = a0 + (int)(ban & 127u);
        double m = INFINITY;
        for (int a = a0 + lane; a < a1; a += 64) m = min_f64(m, cand_cost(a_w[a], best, score(a_col[a]), p.scale));
        run = wave_min_f64(m);
      }

      // ---------------- expand tokens in list order (general path: token per lane, arcs in a per-lane loop)
      used_stash = !single;
      for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        const double cst = i < n ? c_cost[i] : INFINITY;
        const bool act = i < n && cst < wcut;
        int a0 = 0, narc = 0;
        if (act) { const u32 an = c_an[i]; a0 = (int)(an >> 7); narc = (int)(an & 127u); }
        if (narc > kMaxArcsPerState) bad_degree = true;
        const int maxarc = (int)wave_max_u32((u32)narc);
        if (maxarc > kArcCache) used_stash = true;
        const u32 narc_incl = incl_scan_sum((u32)narc);
        const u32 cb = cand_base + narc_incl - (u32)narc;
        if (i < n) t_cbase[i] = cb;
        cand_base += (u32)__builtin_amdgcn_readlane((int)narc_incl, 63);
        // arcs → registers (independent loads, one round trip), then their scores (second round trip)
        float w[kArcCache]; int col[kArcCache]; u32 nx[kArcCache]; u32 nan_[kArcCache]; double nw[kArcCache]; u32 sl[kArcCache];
  #pragma unroll
        for (int k = 0; k < kArcCache; k++) {
          w[k] = 0.0f; col[k] = 0; nx[k] = 0; nan_[k] = 0;
          if (k < narc) { const uint4 rec = a_rec[a0 + k]; nx[k] = rec.x; nan_[k] = rec.y; col[k] = (int)rec.z; w[k] = __uint_as_float(rec.w); }
        }
        double m = INFINITY;
  #pragma unroll
        for (int k = 0; k < kArcCache; k++) {
          nw[k] = (k < narc) ? cand_cost(w[k], cst, score(col[k]), p.scale) : INFINITY;
          m = min_f64(m, nw[k]);
          sl[k] = kEmpty;
        }
        for (int k = kArcCache; k < maxarc; k++)
          if (k < narc) m = min_f64(m, cand_cost(a_w[a0 + k], cst, score(a_col[a0 + k]), p.scale));
        if (single) run = best_i != kEmpty ? readlane_f64(m, __builtin_amdgcn_readfirstlane((int)best_i)) : INFINITY;  // best token is always expanded
        const double m_incl = incl_scan_min(m);
        double local = min_f64(run, shift_in_min(m_incl));
        run = min_f64(run, readlane_f64(m_incl, 63));

        // Candidate creation in three wavefront phases (each phase's LDS operations are issued back to back):
        //   look up the destination's slot → claim missing slots (CAS; the winner allocates, initialises, publishes)
        //   → re-read the published slot and lower its cost / first-creator with LDS atomics.
        bool cr[kArcCache]; u32 hk[kArcCache];
        bool any_pend = false;
  #pragma unroll
        for (int k = 0; k < kArcCache; k++) {
          cr[k] = (k < narc) && (nw[k] < local + (double)abeam);
          if (k < narc) local = min_f64(local, nw[k]);
          hk[k] = hash_of(nx[k]);
          any_pend |= cr[k];
        }
        {
          bool pend[kArcCache];
  #pragma unroll
          for (int k = 0; k < kArcCache; k++) pend[k] = cr[k];
          while (__any(any_pend)) {     // all arcs of all tokens of the chunk are filed in the same rounds
            any_pend = false;
  #pragma unroll
            for (int k = 0; k < kArcCache; k++) {
              probe(pend[k], hk[k], sl[k], nx[k], nan_[k]);
              any_pend |= pend[k];
            }
            WSYNC();
            if (ctr[0] > (u32)N) break;                   // out of slots: the frame reports the overflow below
          }
        }
  #pragma unroll
        for (int k = 0; k < kArcCache; k++) {
          if (k < maxarc) {  // uniform
            if (sl[k] != kEmpty) lower(sl[k], nw[k], ((u32)i << kArcBits) | (u32)k);
            if (!single && sl[k] != kEmpty) {
              u32 q = atomicAdd(&ctr[1], 1u);
              if (q < (u32)C) { st_a[q] = sl[k]; st_b[q] = ((u32)i << kArcBits) | (u32)k; st_key[q] = dkey(nw[k]); }
            }
          }
        }
        for (int k = kArcCache; k < maxarc; k++) {  // slow tail: states with more than kArcCache arcs
          bool created = false; double cnw = 0.0; u32 d = 0, dan = 0;
          if (k < narc) {
            cnw = cand_cost(a_w[a0 + k], cst, score(a_col[a0 + k]), p.scale);
            created = cnw < local + (double)abeam;
            local = min_f64(local, cnw);
            d = a_rec[a0 + k].x;
            dan = a_rec[a0 + k].y;
          }
          u32 s = kEmpty;
          {
            bool pend = created; u32 h = hash_of(d);
            while (__any(pend)) { probe(pend, h, s, d, dan); WSYNC(); if (ctr[0] > (u32)N) break; }
          }
          if (s != kEmpty) lower(s, cnw, ((u32)i << kArcBits) | (u32)k);
          if (s != kEmpty) {  // tail candidates always go through the stash
            u32 q = atomicAdd(&ctr[1], 1u);
            if (q < (u32)C) { st_a[q] = s; st_b[q] = ((u32)i << kArcBits) | (u32)k; st_key[q] = dkey(cnw); }
          }
        }
        if (single) {
          WSYNC();  // every candidate of the frame has lowered its slot's cost
          // winners straight from registers: earliest candidate among those that reached the slot's final best cost
  #pragma unroll
          for (int k = 0; k < kArcCache; k++)
            if (sl[k] != kEmpty && dkey(nw[k]) == s_cost[sl[k]]) atomicMin(&s_W[sl[k]], ((u32)i << kArcBits) | (u32)k);
        }
      }
      if constexpr (kEps) frame_min = run;
    }
    prefetch_next_row();
    // the stash lives in HBM: make its stores visible before other lanes read them back (workgroup-scope fence waits for
    // them); frames that kept everything in registers/LDS only need the wavefront hand-over
    if (used_stash) __threadfence_block();
    WSYNC();
    const u32 nslots = ctr[0], nstash = ctr[1];
    if (__any(bad_degree)) { status = ST_UNSUPPORTED; break; }
    if (spec && __any(viol)) { status = ST_TOKEN_OVERFLOW; spec_failed = true; break; }   // first tier: the window is redone (large tier, or lag mode)
    if (nslots > (u32)N || nstash > (u32)C || cand_base > (u32)C) { status = ST_TOKEN_OVERFLOW; break; }
    if (nslots == 0) { n = 0; t++; break; }  // everything pruned: no surviving token

    // ---------------- winners for stashed candidates (multi-chunk frames and deep states)
    for (u32 q0 = 0; q0 < nstash; q0 += 64) {
      u32 q = q0 + lane;
      if (q < nstash) {
        u32 s = st_a[q];
        if (st_key[q] == s_cost[s]) atomicMin(&s_W[s], st_b[q]);
      }
    }
    WSYNC();  // winners settled
    STAMP(6);   // general path + stash winners (zero when the fast path ran)
    // ---------------- Kaldi list order of the new tokens: for every slot the ordinal of its hash bucket's first creator and
    // its rank inside the bucket; bucket sizes at the leaders' ordinals, then exclusive sums = where every bucket starts.
    // (kEps: slots created by the epsilon closure carry ordinals past the candidates': s_F = 0x80000000 | k.)
    auto ord_of = [&](u32 F) -> u32 {
      if constexpr (kEps) { if (F >> 31) return cand_base + (F & 0x7FFFFFFFu); }
      return t_cbase[F >> kArcBits] + (F & (kMaxArcsPerState - 1));
    };
    auto order_pass = [&](u32 ns, u32 n_ord) {
      for (u32 j0 = 0; j0 < ns; j0 += 64) {
        u32 j = j0 + lane;
        if (j < ns) {
          const u32 d = s_state[j], Fj = s_F[j];
          u32 Fb = Fj, nb = 1, rank = 0;
          if ((u32)S > H) {
            nb = 0;
            for (u32 m = d % H; m < (u32)S; m += H) {
              u32 sm = find(m);
              if (sm < (u32)N) {
                u32 Fm = s_F[sm];
                nb++;
                if (Fm < Fj) rank++;
                if (Fm < Fb) Fb = Fm;
              }
            }
          }
          const u32 ord_b = ord_of(Fb);
          s_aux[j] = (rank << 24) | ord_b;
          if (Fb == Fj) cntord[ord_b] = nb;
        }
      }
      WSYNC();
      {
        u32 carry = 0;
        for (u32 o0 = 0; o0 < n_ord; o0 += 64) {
          u32 o = o0 + lane;
          u32 v = o < n_ord ? cntord[o] : 0u;
          const u32 inc = incl_scan_sum(v);
          if (o < n_ord && v != 0) cntord[o] = carry + inc - v;  // only leader ordinals are ever non-zero (and reset below)
          carry += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        }
      }
      WSYNC();
    };
    order_pass(nslots, cand_base);
    STAMP(7);   // list order (bucket ranks, ordinal scan)
    u32 nslots_f = nslots;      // slots after the epsilon closure (kEps)
    if constexpr (kEps) {
      // ---------------- FasterDecoder::ProcessNonemitting(next_weight_cutoff).  The new tokens live in the slot table
      // (state, cost key, first creator, winner) and the pass above has just given each its list position.  Kaldi pushes the
      // list on a stack (last token on top) and pops: a popped token's epsilon arcs, in arc order, insert their destination
      // (end of its hash bucket's chain) or replace its token when strictly cheaper, and push it.  Costs come out the same
      // whatever the order (label-correcting search, epsilon weights >= 0); the order of insertion — hence the list order the
      // next frame walks — and the back-pointer on ties do not, so the pops run one after the other as Kaldi's do; the popped
      // state's epsilon arcs are relaxed by the lanes in parallel with arc order restored where it matters (first creator
      // by atomicMin on the ordinal, earlier duplicates of a destination by a lane loop, pushes by ballot prefix).  States
      // without epsilon arcs are never pushed: popping them does nothing.
      const double eps_cut = frame_min + (double)abeam;
      bool any_eps = false;
      for (u32 j0 = 0; j0 < nslots; j0 += 64) {
        const u32 j = j0 + lane;
        u32 ei = 0;
        if (j < nslots) { ei = p.w_epsinfo[(size_t)utt * p.eps_stride + s_state[j]]; e_info[j] = ei; }
        if (__any((ei & 127u) != 0u)) any_eps = true;
      }
      if (any_eps) {
        bool eps_broken = false;
        for (u32 j0 = 0; j0 < nslots; j0 += 64) {
          const u32 j = j0 + lane;
          if (j < nslots) {
            const u32 aux = s_aux[j];
            const u32 pos = cntord[aux & 0xFFFFFFu] + (aux >> 24);
            if (pos < nslots) e_inv[pos] = j; else eps_broken = true;
          }
        }
        if (__any(eps_broken)) { status = ST_INTERNAL; break; }
        WSYNC();
        u32 sp = 0;
        for (u32 q0 = 0; q0 < nslots; q0 += 64) {
          const u32 q = q0 + lane;
          const u32 j = q < nslots ? e_inv[q] : 0u;
          const bool has = q < nslots && (e_info[j] & 127u) != 0u;
          const u64 m = __ballot(has);
          if (has) e_stk[sp + (u32)__popcll(m & ((1ull << lane) - 1ull))] = j;
          sp += (u32)__popcll(m);
        }
        WSYNC();
        u32 eord = 0;
        int guard = 0;
        bool eps_over = false;
        while (sp > 0u) {
          if (++guard > p.eps_pops) { eps_over = true; break; }
          const u32 e = e_stk[sp - 1u];
          sp--;
          const double ce = dunkey(s_cost[e]);
          if (ce > eps_cut) continue;
          const u32 ei = e_info[e];
          const u32 n_eps = ei & 127u, first = ei >> 7;
          if (n_eps == 0u) continue;
          if (n_eps > 64u) { bad_degree = true; break; }
          const bool valid = (u32)lane < n_eps;
          u32 nx = 0u, nan_ = 0u; float w = 0.0f;
          if (valid) { const uint4 rec = a_rec[first + (u32)lane]; nx = rec.x; nan_ = rec.y; w = __uint_as_float(rec.w); }
          const double nc = ce + (double)w;           // Kaldi: new_tok->cost_ = tok->cost_ + arc.weight (no acoustic term)
          const bool ok0 = valid && !(nc > eps_cut);
          u32 sl = kEmpty;
          {
            bool pend = ok0; u32 h = hash_of(nx);
            while (__any(pend)) { probe(pend, h, sl, nx, nan_); WSYNC(); if (ctr[0] > (u32)N) break; }
          }
          if (ctr[0] > (u32)N) { eps_over = true; break; }
          const bool ok = ok0 && sl != kEmpty;
          const u64 okm = __ballot(ok);
          const u64 pre = ok ? s_cost[sl] : kKeyInf;  // before this pop: infinite = the state was not in the list
          const bool is_new = ok && pre == kKeyInf;
          if (is_new) e_info[sl] = p.w_epsinfo[(size_t)utt * p.eps_stride + nx];
          // earlier arcs of this pop into the same state (rare): what Kaldi's sequential loop would have left there
          double pm = INFINITY; bool first_dup = true;
          for (u32 j = 0; j < n_eps; j++) {
            const u32 nxj = (u32)__builtin_amdgcn_readlane((int)nx, (int)j);
            const double ncj = readlane_f64(nc, (int)j);
            if (((okm >> j) & 1ull) && (u32)lane > j && nxj == nx) { pm = min_f64(pm, ncj); first_dup = false; }
          }
          const bool push = ok && (is_new ? (first_dup || nc < pm) : (nc < min_f64(dunkey(pre), pm)));
          WSYNC();                                    // every lane has read `pre`
          if (push) atomicMin(&s_cost[sl], dkey(nc));
          if (is_new) atomicMin(&s_F[sl], 0x80000000u | (eord + (u32)__popcll(okm & ((1ull << lane) - 1ull))));
          WSYNC();
          const bool win = push && dkey(nc) == s_cost[sl];
          if (win) atomicMin(&e_tmp[sl], (u32)lane);
          WSYNC();
          if (win && e_tmp[sl] == (u32)lane) { s_W[sl] = 0x80000000u | e; e_arc[sl] = first + (u32)lane; }
          WSYNC();
          if (win) e_tmp[sl] = 0xFFFFFFFFu;
          const bool pp = push && (e_info[sl] & 127u) != 0u;
          const u64 pmk = __ballot(pp);
          const u32 at = sp + (u32)__popcll(pmk & ((1ull << lane) - 1ull));
          if (pp && at < 2u * (u32)N) e_stk[at] = sl;
          sp += (u32)__popcll(pmk);
          if (sp > 2u * (u32)N) { eps_over = true; break; }
          eord += (u32)__popcll(okm);
          if (cand_base + eord > (u32)C) { eps_over = true; break; }
          WSYNC();
        }
        if (__any(bad_degree)) { status = ST_UNSUPPORTED; break; }
        if (eps_over) { status = ST_TOKEN_OVERFLOW; break; }
        nslots_f = ctr[0];
        if (nslots_f > nslots) {
          // new states: the list order is worked out again over all slots (a new state goes to the end of its bucket's
          // chain, which may lie in the middle of the list)
          for (u32 j0 = 0; j0 < nslots; j0 += 64) {
            const u32 j = j0 + lane;
            if (j < nslots) cntord[s_aux[j] & 0xFFFFFFu] = 0u;
          }
          WSYNC();
          order_pass(nslots_f, cand_base + eord);
        }
      }
    }
    // ---------------- write the new list + back-pointers, reset the tables
    if (bp_used + nslots_f > bp_cap) { status = ST_BP_OVERFLOW; break; }
    bool broken = false;  // defensive: an inconsistent table must never turn into an out-of-range store
    if constexpr (kEps) {
      for (u32 j0 = 0; j0 < nslots_f; j0 += 64) {
        const u32 j = j0 + lane;
        if (j < nslots_f) { const u32 aux = s_aux[j]; e_pos[j] = cntord[aux & 0xFFFFFFu] + (aux >> 24); }
      }
      WSYNC();
    }
    for (u32 j0 = 0; j0 < nslots_f; j0 += 64) {
      u32 j = j0 + lane;
      if (j < nslots_f) {
        const u32 aux = s_aux[j];
        const u32 pos = cntord[aux & 0xFFFFFFu] + (aux >> 24);
        const u32 d = s_state[j], W = s_W[j];
        bool eps_w = false;
        if constexpr (kEps) eps_w = (W >> 31) != 0u;
        if (eps_w) {
          // the token came over an epsilon arc: its predecessor is a token of THIS frame's list (no frame consumed)
          const u32 src = W & 0x7FFFFFFFu;
          if (pos >= nslots_f || src >= nslots_f || d >= (u32)S) broken = true;
          else {
            n_state[pos] = d;
            n_an[pos] = s_an[j];
            n_cost[pos] = dunkey(s_cost[j]);
            bp[bp_used + pos] = ((u64)e_arc[j] << 32) | (u64)e_pos[src];
          }
        } else {
          const u32 ppos = W >> kArcBits, k = W & (kMaxArcsPerState - 1);
          if (pos >= nslots_f || ppos >= (u32)n || d >= (u32)S) broken = true;
          else {
            const u32 arc = (c_an[ppos] >> 7) + k;
            n_state[pos] = d;
            n_an[pos] = s_an[j];
            n_cost[pos] = dunkey(s_cost[j]);
            bp[bp_used + pos] = ((u64)arc << 32) | (u64)ppos;
          }
        }
      }
    }
    if (__any(broken)) { status = ST_INTERNAL; break; }
    WSYNC();
    for (u32 j0 = 0; j0 < nslots_f; j0 += 64) {
      u32 j = j0 + lane;
      if (j < nslots_f) { hmap[s_bucket[j]] = kEmpty; cntord[s_aux[j] & 0xFFFFFFu] = 0; }
    }
    if (lane == 0) { tokoff[t] = (u32)bp_used; ctr[0] = 0; ctr[1] = 0; }
    bp_used += nslots_f;
    n = (int)nslots_f;
    cur ^= 1;
    if (!kListsInLds) __threadfence_block();  // token lists in HBM: stores must land before the next frame reads them
    WSYNC();
    STAMP(8);   // new list, back-pointers, table reset
  }
#ifdef VIT_STAMPS
  // accumulated over the windows of the first tier (the caller zeroes the buffer); [11] = frames decoded
  if (lane == 0 && p.pass == 0 && p.stamps && p.redo_mode != 2 && p.utt_list == nullptr)
    for (int k = 0; k < 12; k++) p.stamps[(size_t)utt * 12 + k] += stamp_acc[k];
#endif
  __threadfence_block();  // back-pointer records (HBM) are read back by the traceback below
  if (p.redo_mode == 1 && status == ST_TOKEN_OVERFLOW) {
    // small tier out of slots: nothing parked has been touched (lists and decoder state are written at a window's END
    // only; the back-pointer records of this window are simply written again) — the large tier redoes the window
    if (p.lagmode) {
      // lag mode (see viterbi_small_kernel): a failed speculation puts the utterance one window behind — this kernel redoes the
      // window in its next launch, on the proven band; a capacity overflow sends it to the from-scratch list pass
      if (lane == 0) {
        VitState vs;
        if (spec_failed) {
          if (resume) vs = p.w_vstate[utt];
          else { vs.n = 1; vs.cur = 0; vs.H = 1000u; vs.pad1 = 0; vs.bp_used = 0; }     // (window 0: nothing was parked yet)
          vs.done = 0; vs.pad0 = 1;
        } else {
          vs.n = 0; vs.cur = 0; vs.done = 1; vs.pad0 = 0; vs.H = 1000u; vs.pad1 = 0; vs.bp_used = 0;
          p.status[utt] = ST_GROW; p.n_words[utt] = 0; p.like[utt] = 0.0f;
        }
        p.w_vstate[utt] = vs;
      }
      return;
    }
    if (lane == 0) p.w_redo[utt] = 1u;
    return;
  }
  if (p.redo_mode == 2 && lane == 0) p.w_redo[utt] = 0u;
  u32 *c_state = l_state0 + cur * N;
  double *c_cost = l_cost0 + cur * N;
  if (p.windowed && status == ST_OK && t < T && n > 0) {
    // ---------------- end of a window, utterance not finished: park the token list and publish the band of depths the
    // next window's frames can reach.  A token on state s at frame t' >= t descends from a live token l of frame t, so
    //   bfs_depth(s) <= bfs_depth(l) + (t' - t)   and   longest_depth(s) >= longest_depth(l):
    // a pdf can be asked for in [t, t + K) only if some arc emitting it leaves a state inside those two bounds.
    const u32 *c_an = l_an0 + cur * N;
    u32 dmax = 0, dmin_inv = 0;   // max of bfs depth; max of ~longest (= min of longest)
    for (int i = lane; i < n; i += 64) {
      const u32 s_ = c_state[i];
      if (kListsInLds) { park_state[i] = s_; park_an[i] = c_an[i]; park_cost[i] = c_cost[i]; }
      if (p.state_depth) {
        const int32_t *sd = p.state_depth + 2 * (so + (int64_t)s_);
        dmax = max(dmax, (u32)sd[0]);
        dmin_inv = max(dmin_inv, ~(u32)sd[1]);
      }
    }
    dmax = wave_max_u32(dmax);
    dmin_inv = wave_max_u32(dmin_inv);
    if (lane == 0) {
      VitState vs;
      vs.n = n; vs.cur = kListsInLds ? 0 : cur; vs.done = 0; vs.pad0 = lag; vs.H = H; vs.pad1 = 0; vs.bp_used = bp_used;
      p.w_vstate[utt] = vs;
      if (p.band) {
        const long long hi = (long long)dmax + (long long)p.next_window - 1;
        p.band[2 * utt] = p.state_depth ? (int32_t)~dmin_inv : 0;
        p.band[2 * utt + 1] = p.state_depth ? (int32_t)min(hi, (long long)INT32_MAX) : INT32_MAX;
      }
    }
    return;
  }
  if (p.windowed && lane == 0) {   // finished one way or the other: later windows of this pass skip the utterance
    VitState vs;
    vs.n = 0; vs.cur = 0; vs.done = 1; vs.pad0 = lag; vs.H = H; vs.pad1 = 0; vs.bp_used = bp_used;
    p.w_vstate[utt] = vs;
  }
  if (p.pass == 0 && lane == 0) p.w_hash[utt] = H;

  finalize_utterance(p, utt, lane, status, t, T, n, c_state, c_cost, final_w, bp, tokoff, f0, ab_, a_w, a_col, ll, P, kEps, bp_used,
                     bp_cap);
}

// ---------------------------------------------------------------------------------------------------------------------
// First tier of the windowed first-beam pass (mfa_align_features_batch), written for what that tier actually sees: at
// most 64 live tokens (one per lane) and at most 64·kRounds candidates per frame (one per lane and round).  Same decoder,
// same decisions, bit for bit — but straight-line wavefront code: the general kernel above carries a token-chunk loop, an
// arc cache of eight per lane, an HBM candidate stash and the retry/grow bookkeeping through every frame (6 000
// instructions, 139 spilled scalars), this one a third of that.  Anything outside its envelope (more tokens, more
// candidates, a state of more than 64 arcs) flags the utterance for the large tier, which redoes the window from the
// state parked at its start — exactly the hand-over the general kernel's first tier uses.  An utterance that reaches its
// last frame is parked with done = 2; viterbi_finish_kernel then does ReachedFinal, traceback and outputs.
//   GetCutoff's min_active rule: the (min_active + 1 − k)-th smallest cost outside the beam by ballot quickselect (a
//   handful of compare+ballot steps) instead of ranking every token against every other.
constexpr int kSmallN = 64;
template <int kRounds, bool kEps = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void viterbi_small_kernel(VitParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int N = kSmallN, C = 64 * kRounds;
  constexpr u32 HM = 256u, hmask = HM - 1u;
  constexpr int hshift = 24;
  const int lane = threadIdx.x;
  const int utt = blockIdx.x;
  const VitState vs0 = p.w_vstate[utt];
  // lagmode: an utterance whose speculative window failed is one window behind from then on (VitState.pad0): this launch redoes
  // that window for it — scored again with the proven band by the scoring launch before this one — without the check
  const int K_ = p.t_end - p.t_begin;
  const int lag = (p.lagmode && p.t_begin > 0 && vs0.pad0 != 0) ? 1 : 0;
  const int t_begin_u = p.t_begin - lag * K_;
  const bool resume = t_begin_u > 0;
  if (p.t_begin > 0 && vs0.done) return;           // finished (or failed, or waiting for the finish kernel)
  const int64_t so = p.g.d_state_off[utt];
  const int S = (int)(p.g.d_state_off[utt + 1] - so);
  const int64_t ab_ = p.g.d_arc_base[utt];
  const int32_t *arc_off = p.g.d_arc_off + so + utt;
  const uint4 *a_rec = p.w_arcnext + ab_;
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const float *ll = p.ll + p.ll_off[utt];
  const int P = p.ll_cols[utt];
  const int NP = p.npark;
  u32 *park_state = p.w_state + (size_t)utt * 2 * NP;
  u32 *park_an = p.w_state + (size_t)p.g.n_utt * 2 * NP + (size_t)utt * 2 * NP;
  double *park_cost = p.w_cost + (size_t)utt * 2 * NP;
  // Outside this kernel's envelope (more than 64 tokens, more than 64·kRounds candidates, a malformed graph).  Without lagmode:
  // nothing parked has been touched, the large tier — launched right after — redoes the window.  With lagmode there is no such
  // launch: the utterance leaves the fast track for good and is decoded from its first frame by the table-growth list pass that
  // follows the windowed pass (general kernel, the caller's full capacity) — rare: none of the 4 096 utterances of the bench
  // workload ever holds more than 64 tokens at beam 10.
  auto hand_over = [&]() {
    if (lane != 0) return;
    if (p.lagmode) {
      VitState vs; vs.n = 0; vs.cur = 0; vs.done = 1; vs.pad0 = 0; vs.H = 1000u; vs.pad1 = 0; vs.bp_used = 0;
      p.w_vstate[utt] = vs;
      p.status[utt] = ST_GROW; p.n_words[utt] = 0; p.like[utt] = 0.0f;
    } else {
      p.w_redo[utt] = 1u;
    }
  };

  int n = 1, t = 0;
  u32 H = 1000u;
  u64 bp_used = 0;
  const int start = p.g.d_start[utt];
  if (S <= 0 || start < 0 || start >= S || T <= 0) { hand_over(); return; }   // (the general kernel reports the failure)
  if (resume) {
    n = vs0.n; H = vs0.H; bp_used = vs0.bp_used; t = t_begin_u;
    if (n > N || n <= 0) { hand_over(); return; }
  }

  // ---- LDS carve (as the general kernel's, with its token capacity fixed at 64)
  u64 *s_cost = (u64 *)smem;                 // [N]
  double *l_cost0 = (double *)(s_cost + N);  // [2][N]
  u32 *hmap = (u32 *)(l_cost0 + 2 * N);      // [HM]
  u32 *s_state = hmap + HM;                  // [N]
  u32 *s_F = s_state + N;
  u32 *s_W = s_F + N;
  u32 *t_cbase = s_W + N;
  u32 *s_an = t_cbase + N;
  u32 *s_bucket = s_an + N;
  u32 *l_state0 = s_bucket + N;              // [2][N]
  u32 *l_an0 = l_state0 + 2 * N;             // [2][N]
  u32 *cntord = l_an0 + 2 * N;               // [C]: owner map of the candidate ordinals, then bucket sizes → exclusive sums
  u32 *ctr = cntord + C;                     // [2]
  u32 *bm = ctr + 4;                         // [kBmWords] columns scored for this window (speculative look-ahead only)
  // epsilon closure (kEps; as the general kernel's): per slot its list position, the inverse, the winning epsilon arc, a scratch
  // word for the winner vote, the state's epsilon-arc info; and the stack of ProcessNonemitting
  u32 *e_pos = bm + kBmWords;                // [N]
  u32 *e_inv = e_pos + N;                    // [N]
  u32 *e_arc = e_inv + N;                    // [N]
  u32 *e_tmp = e_arc + N;                    // [N]
  u32 *e_info = e_tmp + N;                   // [N]
  u32 *e_stk = e_info + N;                   // [2N]
  // (no staged score row: a candidate reads its score straight from L2 — measured faster than the general kernel's LDS
  //  row cache here, and 2 KB less LDS per wavefront leaves room for a scoring workgroup next to sixteen of these)

  u64 *bp = p.w_bp + (size_t)f0 * p.bpf;
  const u64 bp_cap = (u64)T * (u64)p.bpf;
  u32 *tokoff = p.w_tokoff + f0 + utt;

  for (u32 i = lane; i < HM; i += 64) hmap[i] = kEmpty;
  for (int i = lane; i < C; i += 64) cntord[i] = 0;
  if (lane == 0) ctr[0] = 0;
  if constexpr (kEps) e_tmp[lane] = 0xFFFFFFFFu;
  int cur = 0;
  if (!resume) {
    if constexpr (kEps) {
      // InitDecoding's ProcessNonemitting(cutoff = FLT_MAX), as in viterbi_kernel<·, true>: the wavefront walks Kaldi's
      // sequential algorithm (a stack; the popped token's epsilon arcs one after the other; destinations looked up by a ballot
      // over the tokens created so far), then the hash-list order gives the initial list and its back-pointer records —
      // bp[0 .. n): the start token's carries arc 0xFFFFFFFF, the others an epsilon arc and a position in this list.
      u32 nc_ = 1u;
      if (lane == 0) {
        s_state[0] = (u32)start; s_cost[0] = (u64)__double_as_longlong(0.0); e_arc[0] = 0xFFFFFFFFu; e_inv[0] = 0u;
        s_an[0] = ((u32)arc_off[start] << 7) | min((u32)p.g.d_state_nemit[so + start], 127u);
        e_stk[0] = 0u;
      }
      WSYNC();
      {
        u32 sp = 1u;
        int guard = 0;
        bool over = false;
        while (sp > 0u && !over) {
          if (++guard > p.eps_pops) { over = true; break; }
          const u32 e = e_stk[sp - 1u];
          sp--;
          const double ce = __longlong_as_double((long long)s_cost[e]);
          const u32 ei = p.w_epsinfo[(size_t)utt * p.eps_stride + s_state[e]];
          const u32 n_eps = ei & 127u, first = ei >> 7;
          for (u32 k = 0; k < n_eps && !over; k++) {
            const uint4 rec = a_rec[first + k];
            const u32 d = rec.x;
            const double ncst = ce + (double)__uint_as_float(rec.w);
            if (ncst > (double)3.4028234663852886e38f) continue;      // cutoff = numeric_limits<float>::max()
            const u64 hit = __ballot((u32)lane < nc_ && s_state[lane] == d);
            const u32 found = hit ? (u32)__ffsll((long long)hit) - 1u : kEmpty;
            bool pushed = false; u32 who = 0u;
            if (found == kEmpty) {
              if (nc_ >= (u32)N) { over = true; break; }
              if (lane == 0) {
                s_state[nc_] = d; s_cost[nc_] = (u64)__double_as_longlong(ncst); e_arc[nc_] = first + k; e_inv[nc_] = e; s_an[nc_] = rec.y;
              }
              who = nc_; nc_++; pushed = true;
            } else if (__longlong_as_double((long long)s_cost[found]) > ncst) {
              if (lane == 0) { s_cost[found] = (u64)__double_as_longlong(ncst); e_arc[found] = first + k; e_inv[found] = e; }
              who = found; pushed = true;
            }
            if (pushed) {
              if (sp >= 2u * (u32)N) { over = true; break; }
              if (lane == 0) e_stk[sp] = who;
              sp++;
            }
            WSYNC();
          }
        }
        if (over) { hand_over(); return; }
      }
      // hash-list order: position of token c = number of tokens whose (bucket's first creator, own index) is smaller
      if ((u32)lane < nc_) {
        const u32 c_ = (u32)lane;
        const u32 bc = s_state[c_] % H;
        u32 lead_c = c_;
        for (u32 x = 0; x < c_; x++) if (s_state[x] % H == bc) { lead_c = x; break; }
        u32 pos_ = 0;
        for (u32 x = 0; x < nc_; x++) {
          if (x == c_) continue;
          const u32 bx = s_state[x] % H;
          u32 lead_x = x;
          for (u32 y = 0; y < x; y++) if (s_state[y] % H == bx) { lead_x = y; break; }
          if (lead_x < lead_c || (lead_x == lead_c && x < c_)) pos_++;
        }
        e_pos[c_] = pos_;
      }
      WSYNC();
      if ((u32)lane < nc_) {
        const u32 c_ = (u32)lane, pos_ = e_pos[c_];
        l_state0[pos_] = s_state[c_];
        l_cost0[pos_] = __longlong_as_double((long long)s_cost[c_]);
        l_an0[pos_] = s_an[c_];
        bp[pos_] = ((u64)e_arc[c_] << 32) | (u64)(c_ == 0u ? 0u : e_pos[e_inv[c_]]);
      }
      n = (int)nc_;
      bp_used = nc_;
      __threadfence_block();
      WSYNC();
    } else {
      if (lane == 0) { l_state0[0] = (u32)start; l_cost0[0] = 0.0; l_an0[0] = ((u32)arc_off[start] << 7) | (u32)(arc_off[start + 1] - arc_off[start]); }
    }
  } else if (lane < n) {
    l_state0[lane] = park_state[lane]; l_an0[lane] = park_an[lane]; l_cost0[lane] = park_cost[lane];
  }
  const int t_stop = min(T, t_begin_u + K_);
  WSYNC();
  const bool spec = p.spec != 0 && lag == 0;
  if (spec) build_scored_bitmap(p, utt, lane, bm);

  bool overflow = false, spec_fail = false;
  bool have_best = false;
  double best_carry = 0.0;
  for (; t < t_stop; t++) {
    const float *llt = ll + (size_t)t * P;
    u32 *n_state = l_state0 + (cur ^ 1) * N;
    u32 *c_an = l_an0 + cur * N, *n_an = l_an0 + (cur ^ 1) * N;
    double *c_cost = l_cost0 + cur * N, *n_cost = l_cost0 + (cur ^ 1) * N;
    // ---------------- GetCutoff
    const double cst = lane < n ? c_cost[lane] : INFINITY;
    const u32 an = lane < n ? c_an[lane] : 0u;
    // The cheapest token's cost is the cheapest candidate of the previous frame (the global minimum is always created and
    // wins its slot, and a slot's cost is its candidate's, bit for bit): carried over instead of a wavefront reduction.
    const double best = (have_best && !kEps) ? best_carry : wave_min_f64(cst);   // (a closure token can undercut every candidate)
    const u32 best_i = (u32)__ffsll((long long)__ballot(lane < n && cst == best)) - 1u;
    double wcut = INFINITY; float abeam = INFINITY;
    if (n > kMinActive) {
      const double beam_cut = best + p.beam;
      const u64 inside = __ballot(lane < n && cst <= beam_cut);
      const int kle = __popcll(inside);
      if (kle > kMinActive) { wcut = beam_cut; abeam = p.beam; }
      else {
        // the (min_active + 1 − kle)-th smallest cost outside the beam: ballot quickselect (ties resolved by counting < and <=)
        u64 A = __ballot(lane < n) & ~inside;
        int r = kMinActive + 1 - kle;
        double v = INFINITY;
        while (A != 0ull) {
          const int pl = __ffsll((long long)A) - 1;
          const double pv = readlane_f64(cst, pl);
          const u64 lt = __ballot(cst < pv) & A, le = __ballot(cst <= pv) & A;
          const int clt = __popcll(lt), cle = __popcll(le);
          if (r <= clt) A = lt;
          else if (r <= cle) { v = pv; break; }
          else { A &= ~le; r -= cle; }
        }
        wcut = v;
        abeam = (float)(v - best + (double)kBeamDelta);
      }
    }
    { u32 want = (u32)((float)n * kHashRatio); if (want > H) H = want; }
    // ---------------- candidate layout: ordinal base per token, owner of every ordinal
    const bool act = lane < n && cst < wcut;
    const u32 narc = act ? (an & 127u) : 0u;
    const u32 narc_incl = incl_scan_sum(narc);
    const u32 cb = narc_incl - narc;
    const u32 ctot = (u32)__builtin_amdgcn_readlane((int)narc_incl, 63);
    if (__any(narc > (u32)kMaxArcsPerState) || ctot > (u32)C) { overflow = true; break; }
    if (lane < n) t_cbase[lane] = cb;
    if (narc > 0u) cntord[cb] = (u32)lane + 1u;       // head of each token's candidate run (cntord is all zero between frames)
    WSYNC();
    const int rounds = (int)((ctot + 63u) >> 6);
    // ---------------- arc gather (all rounds' loads in flight together)
    u32 cidx[kRounds], nx[kRounds], nan_[kRounds]; int colv[kRounds]; float wv[kRounds]; double tcost[kRounds];
    {
      u32 carry = 0;
#pragma unroll
      for (int r = 0; r < kRounds; r++) {
        cidx[r] = 0; nx[r] = 0; nan_[r] = 0; colv[r] = 0; wv[r] = 0.0f; tcost[r] = INFINITY;
        if (r < rounds) {   // uniform
          const u32 c = (u32)lane + 64u * r;
          const u32 own = max(incl_scan_max(cntord[c]), carry);
          carry = (u32)__builtin_amdgcn_readlane((int)own, 63);
          const bool valid = c < ctot;
          const u32 tok = valid ? own - 1u : 0u;
          tcost[r] = valid ? c_cost[tok] : INFINITY;
          const u32 tan = c_an[tok];
          const u32 k_ = valid ? c - t_cbase[tok] : 0u;
          cidx[r] = (tok << kArcBits) | k_;            // (token, arc) of the candidate: what the winner records
          if (valid) { const uint4 rec = a_rec[(tan >> 7) + k_]; nx[r] = rec.x; nan_[r] = rec.y; colv[r] = (int)rec.z; wv[r] = __uint_as_float(rec.w); }
        }
      }
    }
    WSYNC();
    if (spec) {   // every score read below must be one that was computed for this window
      bool viol = false;
#pragma unroll
      for (int r = 0; r < kRounds; r++) viol |= (u32)lane + 64u * r < ctot && !column_scored(bm, colv[r]);
      if (__any(viol)) { spec_fail = true; break; }
    }
    const u32 cb2 = t_cbase[min(lane, N - 1)];        // (= cb for the lanes that hold a token; re-read: one register less across the gather)
    if (narc > 0u) cntord[cb2] = 0u;                  // owner map read by every round: back to zero for the ordering pass
    double nw[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++)
      nw[r] = ((u32)lane + 64u * r < ctot) ? cand_cost(wv[r], tcost[r], llt[colv[r]], p.scale) : INFINITY;
    // ---------------- running cutoff: seed from the best token's candidates, then an exclusive prefix-min in ordinal order
    // (the best token's candidates sit at ordinals cb[best] .. + narc[best]: a few broadcast reads instead of a masked
    //  wavefront reduction per round)
    double run = INFINITY;
    {
      const u32 ob = (u32)__builtin_amdgcn_readlane((int)cb2, (int)best_i), nb_ = (u32)__builtin_amdgcn_readlane((int)narc, (int)best_i);
      for (u32 k = 0; k < nb_; k++) {
        const u32 ord = ob + k;
        const int ln = (int)(ord & 63u);
        double v = readlane_f64(nw[0], ln);
#pragma unroll
        for (int r = 1; r < kRounds; r++) if ((ord >> 6) == (u32)r) v = readlane_f64(nw[r], ln);
        run = min_f64(run, v);
      }
    }
    bool created[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      created[r] = false;
      if (r < rounds) {
        const double m_incl = incl_scan_min(nw[r]);
        const double local = min_f64(run, shift_in_min(m_incl));
        run = min_f64(run, readlane_f64(m_incl, 63));
        created[r] = ((u32)lane + 64u * r < ctot) && nw[r] < local + (double)abeam;
      }
    }
    // ---------------- find-or-insert the destination's slot, lower its cost / first creator, settle the winner
    u32 sl[kRounds], hk[kRounds];
    bool pend[kRounds];
    bool any_pend = false;
#pragma unroll
    for (int r = 0; r < kRounds; r++) { sl[r] = kEmpty; pend[r] = created[r]; hk[r] = (nx[r] * 2654435761u) >> hshift; any_pend |= pend[r]; }
    while (__any(any_pend)) {
      any_pend = false;
#pragma unroll
      for (int r = 0; r < kRounds; r++) {
        if (r >= rounds) continue;      // (uniform: a frame of one round does not walk the other rounds' masks)
        if (pend[r]) {
          const u32 v = hmap[hk[r]];
          if (v == kEmpty) {
            if (atomicCAS(&hmap[hk[r]], kEmpty, kClaim) == kEmpty) {
              const u32 my = atomicAdd(&ctr[0], 1u);
              if (my < (u32)N) {
                s_state[my] = nx[r]; s_an[my] = nan_[r]; s_cost[my] = kKeyInf; s_F[my] = kEmpty; s_W[my] = kEmpty; s_bucket[my] = hk[r];
                hmap[hk[r]] = my;
                sl[r] = my;
              } else {
                hmap[hk[r]] = kOver;
              }
              pend[r] = false;
            }
          } else if (v == kOver) {
            pend[r] = false;
          } else if (v != kClaim) {
            if (s_state[v] == nx[r]) { sl[r] = v; pend[r] = false; }
            else hk[r] = (hk[r] + 1u) & hmask;
          }
        }
        any_pend |= pend[r];
      }
      WSYNC();
      if (ctr[0] > (u32)N) break;
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      if (r < rounds && sl[r] != kEmpty) { atomicMin(&s_cost[sl[r]], dkey(nw[r])); atomicMin(&s_F[sl[r]], cidx[r]); }
    }
    WSYNC();
#pragma unroll
    for (int r = 0; r < kRounds; r++)
      if (r < rounds && sl[r] != kEmpty && dkey(nw[r]) == s_cost[sl[r]]) atomicMin(&s_W[sl[r]], cidx[r]);
    WSYNC();
    const u32 nslots = ctr[0];
    if (nslots > (u32)N || bp_used + nslots > bp_cap) { overflow = true; break; }
    if (nslots == 0) { n = 0; t++; break; }            // everything pruned: no surviving token
    // ---------------- Kaldi list order of the new tokens (one slot per lane): ordinal of the hash bucket's first creator and
    // rank inside the bucket; bucket sizes at the leaders' ordinals, then exclusive sums = where every bucket starts.
    // (kEps: slots created by the epsilon closure carry ordinals past the candidates': s_F = 0x80000000 | k.)
    u32 aux = 0;
    auto order_lanes = [&](u32 ns, u32 n_ord) {
      aux = 0;
      if ((u32)lane < ns) {
        const u32 d = s_state[lane], Fj = s_F[lane];
        u32 Fb = Fj, nb = 1, rank = 0;
        if ((u32)S > H) {
          nb = 0;
          for (u32 m = d % H; m < (u32)S; m += H) {
            u32 h = (m * 2654435761u) >> hshift, sm = kEmpty;
            for (;;) {
              const u32 v = hmap[h];
              if (v == kEmpty) break;
              if (v < (u32)N && s_state[v] == m) { sm = v; break; }
              h = (h + 1u) & hmask;
            }
            if (sm < (u32)N) {
              const u32 Fm = s_F[sm];
              nb++;
              if (Fm < Fj) rank++;
              if (Fm < Fb) Fb = Fm;
            }
          }
        }
        u32 ord_b = t_cbase[(Fb >> kArcBits) & (u32)(N - 1)] + (Fb & (kMaxArcsPerState - 1));
        if constexpr (kEps) { if (Fb >> 31) ord_b = ctot + (Fb & 0x7FFFFFFFu); }
        aux = (rank << 24) | ord_b;
        if (Fb == Fj) cntord[ord_b] = nb;
      }
      WSYNC();
      {
        u32 carry = 0;
        const int rounds_ord = (int)((n_ord + 63u) >> 6);
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
          if (r < rounds_ord) {
            const u32 o = (u32)lane + 64u * r;
            const u32 v = cntord[o];
            const u32 inc = incl_scan_sum(v);
            if (v != 0) cntord[o] = carry + inc - v;
            carry += (u32)__builtin_amdgcn_readlane((int)inc, 63);
          }
        }
      }
      WSYNC();
    };
    order_lanes(nslots, ctot);
    u32 nslots_f = nslots;
    if constexpr (kEps) {
      // ---------------- FasterDecoder::ProcessNonemitting(next_weight_cutoff), as in viterbi_kernel<·, true>: the pops one after
      // the other in Kaldi's order (stack of the list, last token on top), the popped state's epsilon arcs relaxed by the lanes.
      const double eps_cut = run + (double)abeam;
      u32 my_info = 0;
      if ((u32)lane < nslots) { my_info = p.w_epsinfo[(size_t)utt * p.eps_stride + s_state[lane]]; e_info[lane] = my_info; }
      if (__any((my_info & 127u) != 0u)) {
        bool eps_broken = false;
        if ((u32)lane < nslots) {
          const u32 pos = cntord[aux & 0xFFFFFFu] + (aux >> 24);
          if (pos < nslots) e_inv[pos] = (u32)lane; else eps_broken = true;
        }
        if (__any(eps_broken)) { overflow = true; break; }
        WSYNC();
        u32 sp = 0;
        {
          const u32 j = (u32)lane < nslots ? e_inv[lane] : 0u;
          const bool has = (u32)lane < nslots && (e_info[j] & 127u) != 0u;
          const u64 m = __ballot(has);
          if (has) e_stk[(u32)__popcll(m & ((1ull << lane) - 1ull))] = j;
          sp = (u32)__popcll(m);
        }
        WSYNC();
        u32 eord = 0;
        int guard = 0;
        bool eps_over = false;
        while (sp > 0u) {
          if (++guard > p.eps_pops) { eps_over = true; break; }
          const u32 e = e_stk[sp - 1u];
          sp--;
          const double ce = dunkey(s_cost[e]);
          if (ce > eps_cut) continue;
          const u32 ei = e_info[e];
          const u32 n_eps = ei & 127u, first = ei >> 7;
          if (n_eps == 0u) continue;
          if (n_eps > 64u) { eps_over = true; break; }
          const bool valid = (u32)lane < n_eps;
          u32 nxe = 0u, nane = 0u; float w = 0.0f;
          if (valid) { const uint4 rec = a_rec[first + (u32)lane]; nxe = rec.x; nane = rec.y; w = __uint_as_float(rec.w); }
          const double nc = ce + (double)w;           // Kaldi: new_tok->cost_ = tok->cost_ + arc.weight (no acoustic term)
          const bool ok0 = valid && !(nc > eps_cut);
          u32 sle = kEmpty;
          {
            bool pend = ok0; u32 h = (nxe * 2654435761u) >> hshift;
            while (__any(pend)) {
              if (pend) {
                const u32 v = hmap[h];
                if (v == kEmpty) {
                  if (atomicCAS(&hmap[h], kEmpty, kClaim) == kEmpty) {
                    const u32 my = atomicAdd(&ctr[0], 1u);
                    if (my < (u32)N) {
                      s_state[my] = nxe; s_an[my] = nane; s_cost[my] = kKeyInf; s_F[my] = kEmpty; s_W[my] = kEmpty; s_bucket[my] = h;
                      hmap[h] = my;
                      sle = my;
                    } else {
                      hmap[h] = kOver;
                    }
                    pend = false;
                  }
                } else if (v == kOver) {
                  pend = false;
                } else if (v != kClaim) {
                  if (s_state[v] == nxe) { sle = v; pend = false; }
                  else h = (h + 1u) & hmask;
                }
              }
              WSYNC();
              if (ctr[0] > (u32)N) break;
            }
          }
          if (ctr[0] > (u32)N) { eps_over = true; break; }
          const bool ok = ok0 && sle != kEmpty;
          const u64 okm = __ballot(ok);
          const u64 pre = ok ? s_cost[sle] : kKeyInf;  // before this pop: infinite = the state was not in the list
          const bool is_new = ok && pre == kKeyInf;
          if (is_new) e_info[sle] = p.w_epsinfo[(size_t)utt * p.eps_stride + nxe];
          // earlier arcs of this pop into the same state (rare): what Kaldi's sequential loop would have left there
          double pm = INFINITY; bool first_dup = true;
          for (u32 j = 0; j < n_eps; j++) {
            const u32 nxj = (u32)__builtin_amdgcn_readlane((int)nxe, (int)j);
            const double ncj = readlane_f64(nc, (int)j);
            if (((okm >> j) & 1ull) && (u32)lane > j && nxj == nxe) { pm = min_f64(pm, ncj); first_dup = false; }
          }
          const bool push = ok && (is_new ? (first_dup || nc < pm) : (nc < min_f64(dunkey(pre), pm)));
          WSYNC();                                    // every lane has read `pre`
          if (push) atomicMin(&s_cost[sle], dkey(nc));
          if (is_new) atomicMin(&s_F[sle], 0x80000000u | (eord + (u32)__popcll(okm & ((1ull << lane) - 1ull))));
          WSYNC();
          const bool win = push && dkey(nc) == s_cost[sle];
          if (win) atomicMin(&e_tmp[sle], (u32)lane);
          WSYNC();
          if (win && e_tmp[sle] == (u32)lane) { s_W[sle] = 0x80000000u | e; e_arc[sle] = first + (u32)lane; }
          WSYNC();
          if (win) e_tmp[sle] = 0xFFFFFFFFu;
          const bool pp = push && (e_info[sle] & 127u) != 0u;
          const u64 pmk = __ballot(pp);
          const u32 at = sp + (u32)__popcll(pmk & ((1ull << lane) - 1ull));
          if (pp && at < 2u * (u32)N) e_stk[at] = sle;
          sp += (u32)__popcll(pmk);
          if (sp > 2u * (u32)N) { eps_over = true; break; }
          eord += (u32)__popcll(okm);
          if (ctot + eord > (u32)C) { eps_over = true; break; }
          WSYNC();
        }
        if (eps_over) { overflow = true; break; }
        nslots_f = ctr[0];
        if (nslots_f > (u32)N || bp_used + nslots_f > bp_cap) { overflow = true; break; }
        if (nslots_f > nslots) {
          // new states: the list order is worked out again over all slots (a new state goes to the end of its bucket's chain)
          if ((u32)lane < nslots) cntord[aux & 0xFFFFFFu] = 0u;
          WSYNC();
          order_lanes(nslots_f, ctot + eord);
        }
      }
    }
    // ---------------- write the new list + back-pointers, reset the tables
    bool broken = false;
    if constexpr (kEps) {
      if ((u32)lane < nslots_f) e_pos[lane] = cntord[aux & 0xFFFFFFu] + (aux >> 24);
      WSYNC();
    }
    if ((u32)lane < nslots_f) {
      const u32 pos = cntord[aux & 0xFFFFFFu] + (aux >> 24);
      const u32 d = s_state[lane], W = s_W[lane];
      bool eps_w = false;
      if constexpr (kEps) eps_w = (W >> 31) != 0u;
      if (eps_w) {
        // the token came over an epsilon arc: its predecessor is a token of THIS frame's list (no frame consumed)
        const u32 src = W & 0x7FFFFFFFu;
        if (pos >= nslots_f || src >= nslots_f || d >= (u32)S) broken = true;
        else {
          n_state[pos] = d;
          n_an[pos] = s_an[lane];
          n_cost[pos] = dunkey(s_cost[lane]);
          bp[bp_used + pos] = ((u64)e_arc[lane] << 32) | (u64)e_pos[src];
        }
      } else {
        const u32 ppos = W >> kArcBits, k = W & (kMaxArcsPerState - 1);
        if (pos >= nslots_f || ppos >= (u32)n || d >= (u32)S) broken = true;
        else {
          const u32 arc = (c_an[ppos] >> 7) + k;
          n_state[pos] = d;
          n_an[pos] = s_an[lane];
          n_cost[pos] = dunkey(s_cost[lane]);
          bp[bp_used + pos] = ((u64)arc << 32) | (u64)ppos;
        }
      }
    }
    if (__any(broken)) { overflow = true; break; }     // (cannot happen; the large tier would report ST_INTERNAL)
    WSYNC();
    if ((u32)lane < nslots_f) { hmap[s_bucket[lane]] = kEmpty; cntord[aux & 0xFFFFFFu] = 0; }
    if (lane == 0) { tokoff[t] = (u32)bp_used; ctr[0] = 0; }
    bp_used += nslots_f;
    n = (int)nslots_f;
    cur ^= 1;
    best_carry = run; have_best = true;
    WSYNC();
  }
  if (overflow) { hand_over(); return; }
  // the narrow band did not hold: nothing parked has been touched, so the window is scored again with the proven band and
  // redone from its parked state by the large tier — the same hand-over as a capacity overflow
  if (spec_fail) {
    if (p.lagmode) {
      // the parked state is as it was at the window's start: mark the utterance as one window behind — the next scoring launch
      // scores this window again with the proven band, the next launch of this kernel redoes it (no separate launch for a
      // handful of wavefronts, which with several batches in flight left the chip empty a tenth of the time)
      if (lane == 0) {
        VitState vs = vs0;
        if (!resume) { vs.n = 1; vs.cur = 0; vs.H = 1000u; vs.pad1 = 0; vs.bp_used = 0; }   // (window 0: nothing was parked yet)
        vs.done = 0; vs.pad0 = 1;
        p.w_vstate[utt] = vs;
      }
      return;
    }
    hand_over(); return;
  }
  __threadfence_block();
  u32 *c_state = l_state0 + cur * N;
  double *c_costp = l_cost0 + cur * N;
  const u32 *c_anp = l_an0 + cur * N;
  if (n == 0) {   // no surviving token: pending for the retry pass, as the general kernel's finalisation reports it
    if (lane == 0) {
      VitState vs; vs.n = 0; vs.cur = 0; vs.done = 1; vs.pad0 = lag; vs.H = H; vs.pad1 = 0; vs.bp_used = bp_used;
      p.w_vstate[utt] = vs;
      p.w_hash[utt] = H;
      p.status[utt] = ST_PENDING; p.n_words[utt] = 0; p.like[utt] = 0.0f;
    }
    return;
  }
  // ---------------- park the list (window end, or last frame: done = 2 hands the utterance to viterbi_finish_kernel)
  u32 dmax = 0, dmin_inv = 0;
  if (lane < n) {
    const u32 s_ = c_state[lane];
    park_state[lane] = s_; park_an[lane] = c_anp[lane]; park_cost[lane] = c_costp[lane];
    if (p.state_depth && t < T) {
      const int32_t *sd = p.state_depth + 2 * (so + (int64_t)s_);
      dmax = (u32)sd[0]; dmin_inv = ~(u32)sd[1];
    }
  }
  dmax = wave_max_u32(dmax);
  dmin_inv = wave_max_u32(dmin_inv);
  if (lane == 0) {
    VitState vs;
    vs.n = n; vs.cur = 0; vs.done = t < T ? 0 : 2; vs.pad0 = lag; vs.H = H; vs.pad1 = 0; vs.bp_used = bp_used;
    p.w_vstate[utt] = vs;
    if (t >= T) p.w_hash[utt] = H;
    if (p.band && t < T) {
      const long long hi = (long long)dmax + (long long)p.next_window - 1;
      p.band[2 * utt] = p.state_depth ? (int32_t)~dmin_inv : 0;
      p.band[2 * utt + 1] = p.state_depth ? (int32_t)min(hi, (long long)INT32_MAX) : INT32_MAX;
    }
  }
}

// ReachedFinal, traceback and outputs for the utterances viterbi_small_kernel decoded to their last frame (done = 2).
__global__ __launch_bounds__(64) void viterbi_finish_kernel(VitParams p) {
  const int lane = threadIdx.x;
  const int utt = blockIdx.x;
  const VitState vs = p.w_vstate[utt];
  if (vs.done != 2) return;
  const int64_t so = p.g.d_state_off[utt];
  const int64_t ab_ = p.g.d_arc_base[utt];
  const int64_t f0 = p.frame_off[utt];
  const int T = (int)(p.frame_off[utt + 1] - f0);
  const int NP = p.npark;
  const u32 *c_state = p.w_state + (size_t)utt * 2 * NP;
  const double *c_cost = p.w_cost + (size_t)utt * 2 * NP;
  if (lane == 0) { VitState d = vs; d.done = 1; p.w_vstate[utt] = d; }
  finalize_utterance(p, utt, lane, ST_OK, T, T, vs.n, c_state, c_cost, p.g.d_final + so, p.w_bp + (size_t)f0 * p.bpf,
                     p.w_tokoff + f0 + utt, f0, ab_, p.g.d_arc_weight + ab_, p.g.d_arc_col + ab_, p.ll + p.ll_off[utt],
                     p.ll_cols[utt], p.w_epsinfo != nullptr, vs.bp_used, (u64)T * (u64)p.bpf);
}

// Build the retry list: utterances left pending by the first pass.
__global__ void collect_pending_kernel(const int32_t *status, int n_utt, int code, int32_t *list, int32_t *count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_utt && status[i] == code) list[atomicAdd(count, 1)] = i;
}
// utterances a first-tier launch flagged for the large tier (w_redo != 0), as a list for the scoring kernels
__global__ void collect_flagged_kernel(const u32 *flags, int n_utt, int32_t *list, int32_t *count) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_utt && flags[u] != 0u) list[atomicAdd(count, 1)] = u;
}
__global__ void finalize_pending_kernel(int32_t *status, int n_utt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_utt && status[i] == ST_PENDING) status[i] = ST_FAILED;
}

// One 16-byte record per arc for the frame loop: {destination state, (first arc << 7 | out-degree) of the destination,
// score column, weight}.  The destination's arc range folds the arc_off lookup of the NEXT frame into this frame's arc
// fetch (one dependent HBM/L2 round trip per frame instead of three), and the record makes that fetch one 16-byte load
// touching one line instead of four 4-byte gathers from four arrays.
__global__ void arcnext_kernel(mfa_graph_batch g, uint4 *out, u32 *epsinfo, int max_states) {
  const int utt = blockIdx.x;
  const int64_t so = g.d_state_off[utt], ab = g.d_arc_base[utt];
  const int64_t na = g.d_arc_base[utt + 1] - ab;
  const int S = (int)(g.d_state_off[utt + 1] - so);
  const int32_t *arc_off = g.d_arc_off + so + utt;
  const int32_t *nemit = g.d_state_nemit ? g.d_state_nemit + so : nullptr;   // emitting arcs per state (the rest are epsilon arcs)
  for (int64_t a = threadIdx.x; a < na; a += blockDim.x) {
    const int d = g.d_arc_next[ab + a];
    const u32 deg = nemit ? (u32)nemit[d] : (u32)(arc_off[d + 1] - arc_off[d]);
    const u32 an = ((u32)arc_off[d] << 7) | min(deg, 127u);
    out[ab + a] = make_uint4((u32)d, an, (u32)g.d_arc_col[ab + a], __float_as_uint(g.d_arc_weight[ab + a]));
  }
  if (epsinfo && nemit)
    for (int s_ = threadIdx.x; s_ < S && s_ < max_states; s_ += blockDim.x) {
      const int first = arc_off[s_] + nemit[s_], n_eps = arc_off[s_ + 1] - first;
      epsinfo[(size_t)utt * max_states + s_] = ((u32)first << 7) | (u32)min(max(n_eps, 0), 127);
    }
}

constexpr int kLlCap = 512;
// state→slot hash table: a power of two, at least four entries per live-token slot
// — unless one entry per graph state is smaller (the large tiers of the rare passes): then 0 = direct map
int hash_bits(int S, int N) {
  int b = 8;
  while ((1 << b) < 4 * N) b++;
  return ((size_t)4 << b) <= (size_t)((S + 1) & ~1) * 4 ? b : 0;
}
size_t lds_bytes(int S, int N, int C, bool lists_in_lds, bool eps = false) {
  const int hb = hash_bits(S, N);
  const size_t table = hb ? ((size_t)4 << hb) : (size_t)((S + 1) & ~1) * 4;
  return (size_t)N * 8 + table + (size_t)N * 7 * 4 + (size_t)C * 4 + (size_t)kLlCap * 4 + 16 + (size_t)kBmWords * 4 +
         (lists_in_lds ? (size_t)N * 32 : 0) + (eps ? (size_t)N * 7 * 4 : 0);   // eps: position / inverse / arc / vote / info + stack
}
constexpr size_t kLdsLimit = 160 * 1024;

struct WsLayout {
  size_t arcnext, state, cost, sta, stb, stkey, bp, tokoff, hash, list, count, vstate, band, redo, eps, total;
};
WsLayout ws_layout(int n_utt, int64_t total_frames, int N, int C, int bpf, int64_t total_arcs, int64_t eps_states = 0) {
  WsLayout w; size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  w.arcnext = take((size_t)total_arcs * 16);
  w.state = take((size_t)n_utt * 4 * N * 4);  // token states + their packed arc ranges
  w.cost = take((size_t)n_utt * 2 * N * 8);
  w.sta = take((size_t)n_utt * C * 4);
  w.stb = take((size_t)n_utt * C * 4);
  w.stkey = take((size_t)n_utt * C * 8);
  w.bp = take((size_t)total_frames * bpf * 8);
  w.tokoff = take((size_t)(total_frames + n_utt) * 4);
  w.hash = take((size_t)n_utt * 4);
  w.list = take((size_t)n_utt * 4);
  w.count = take(256);
  w.vstate = take((size_t)n_utt * sizeof(VitState));
  w.band = take((size_t)n_utt * 2 * 4);
  w.redo = take((size_t)n_utt * 4);
  w.eps = take((size_t)eps_states * 4);
  w.total = o;
  return w;
}

// N = live-token capacity, C = per-frame candidate capacity.  One token per state and one candidate per arc are hard
// upper bounds, so the retry pass (whose beam keeps most of the graph alive) can always be given room that cannot overflow.
int pick_caps(const mfa_align_opts *o, int max_states, int max_arcs, int pass, int *N, int *C) {
  int n = o->max_tokens > 0 ? o->max_tokens : 1024;
  if (pass == 1) n = n * 4;
  if (n > max_states) n = max_states;
  n = (n + 63) & ~63;
  if (n < 64) n = 64;
  int c = pass == 1 ? 8 * n : 4 * n;
  if (c > max_arcs) c = max_arcs;
  c = (c + 63) & ~63;
  if (c < 256) c = 256;
  *N = n; *C = c;
  return 0;
}

}  // namespace

extern "C" {

MFA_API int mfa_debug_viterbi_stamps(mfa_ctx *c, void *d_stamps) {
  c->vit_stamps = d_stamps;
  return 0;
}

MFA_API size_t mfa_align_workspace_bytes(mfa_ctx *c, int32_t n_utt, int64_t total_frames, const mfa_align_opts *o) {
  (void)c;
  int N = (o->max_tokens > 0 ? o->max_tokens : 1024) * 4, C = 8 * N;
  return ws_layout(n_utt, total_frames, N, C, o->bp_tokens_per_frame > 0 ? o->bp_tokens_per_frame : 512,
                   (int64_t)n_utt * 8 * N).total;
}

}  // extern "C"

namespace {

// Shared driver of mfa_align_batch (scores given) and mfa_align_features_batch (lazy: `lazy` non-NULL — every pass is a
// loop over windows of `window` frames: score the cells the live tokens can reach, then decode the window).
int align_impl(mfa_ctx *c, const mfa_graph_batch *g, const float *d_loglikes, const int64_t *d_ll_off,
               const int32_t *d_ll_cols, const int64_t *d_frame_off, int64_t total_frames,
               int64_t total_arcs, int32_t max_states, int32_t max_arcs, const mfa_align_opts *o, int32_t *d_ali, int32_t *d_words, int32_t *d_n_words,
               float *d_like, float *d_frame_like, int32_t *d_status, const MfaLazyScoring *lazy) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  const int n_utt = g->n_utt;
  if (n_utt <= 0) return 0;
  if (o->beam <= 0.0f || (o->retry_beam != 0.0f && o->retry_beam <= o->beam))
    return c->fail("Beams do not make sense: beam %f, retry-beam %f", o->beam, o->retry_beam);
  if (max_states <= 0 || max_arcs <= 0) return c->fail("max_states and max_arcs must be positive");
  if (total_frames <= 0 || total_arcs <= 0) return c->fail("total_frames and total_arcs must be positive");
  if (max_arcs >= (1 << 25)) return c->fail("graphs with more than 2^25 arcs are not supported");
  const int bpf = o->bp_tokens_per_frame > 0 ? o->bp_tokens_per_frame : 512;
  const int passes = o->retry_beam != 0.0f ? 2 : 1;
  int N[2], C[2];
  for (int ps = 0; ps < 2; ps++) pick_caps(o, max_states, max_arcs, ps, &N[ps], &C[ps]);
  const int Nw = passes == 2 ? N[1] : N[0], Cw = passes == 2 ? C[1] : C[0];
  // graphs with epsilon input arcs: the general kernel's kEps instantiation on every tier (the 64-token kernel knows nothing
  // of them), one {first epsilon arc, count} word per state in the workspace
  const bool eps = g->d_state_nemit != nullptr;
  if (eps && max_arcs >= (1 << 24)) return c->fail("graphs with epsilon arcs and more than 2^24 arcs are not supported");
  const int64_t eps_states = eps ? (int64_t)n_utt * max_states : 0;
  WsLayout w = ws_layout(n_utt, total_frames, Nw, Cw, bpf, total_arcs, eps_states);
  if (c->ws_bytes < w.total) {
    if (c->d_ws) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_ws); c->d_ws = nullptr; c->ws_bytes = 0; }
    MFA_HIP_CHECK(c, hipMalloc(&c->d_ws, w.total));
    c->ws_bytes = w.total;
  }
  unsigned char *base = (unsigned char *)c->d_ws;
  hipLaunchKernelGGL(arcnext_kernel, dim3(n_utt), dim3(256), 0, c->stream, *g, (uint4 *)(base + w.arcnext),
                     eps ? (u32 *)(base + w.eps) : (u32 *)nullptr, max_states);
  // Launch plan.  The decoder is latency-bound (one wavefront walks one utterance frame by frame), so throughput is the
  // number of wavefronts a CU can keep resident, and that is set by the LDS tables, which scale with the token capacity.
  // With the normal beam a frame rarely holds more than a few dozen tokens, so every utterance is first decoded with
  // small tables (kSmallTokens); the few that overflow them are marked ST_GROW and decoded again, from scratch and with
  // the same beam, at the caller's full capacity.  Then the retry-beam pass for utterances that did not reach a final state.
  if (lazy && mfa_gmm_presplit(c, lazy, d_frame_off, n_utt, total_frames) != 0) return -1;
  struct Launch { int pass, N, C, code, grow; int N2 = 0, C2 = 0; };   // N2 > 0: a large tier redoes single windows
  std::vector<Launch> plan;
  // First-tier capacity.  With the hashed state→slot table nothing in the decoder's LDS scales with the graph: 64 tokens
  // need 9.5 KB → 16 wavefronts per CU (a whole batch of 4 096 resident at once on 256 CUs), 128 tokens 14.6 KB → 10.
  // MFA_VIT_TIER overrides (diagnostics).
  int kSmallTokens = lazy ? 64 : 128;
  { const char *e = getenv("MFA_VIT_TIER"); if (e && atoi(e) >= 64) kSmallTokens = (atoi(e) + 63) & ~63; }
  // Speculative look-ahead of the windowed first-beam pass.  The proven band lets a token advance one arc per frame, K − 1
  // arcs by the window's last frame; speech advances a third of that (synthetic 10 s utterances: 21 states per 64 frames on
  // average, 34 at the 99th percentile, 38 at most).  The window is scored for a look-ahead of K / 2 arcs instead (32 for
  // K = 64: a fifth fewer model blocks gathered than with 48, tools/band_study.py), the decoder checks every score it reads
  // against what was scored, and a window in which it asks for more (about 1 % of them) is scored again with the proven band
  // and redone from the state parked at its start by the large tier — the hand-over capacity overflows already use; results
  // cannot differ.  (Round 2 re-decoded such an utterance from frame 0, which made anything below 48 arcs a loss.)
  // MFA_LAZY_LOOKAHEAD=n overrides (n >= K − 1: off).
  int spec_slack = 0;
  if (lazy) {
    int look = lazy->window / 2;
    { const char *e = getenv("MFA_LAZY_LOOKAHEAD"); if (e && atoi(e) > 0) look = atoi(e); }
    spec_slack = std::max(0, lazy->window - 1 - look);
    if (lazy->plan.max_cols > 32 * kBmWords) spec_slack = 0;   // (the decoder's bitmap of scored columns holds 2 048)
  }
  bool will_lag = false;
  if (lazy && N[0] > kSmallTokens) {
    // windowed first-beam pass: the small tier decodes every window; the few utterances it cannot hold in a window are
    // decoded again — that window only, from the state parked at its start — by the large tier (LDS-resident lists,
    // so at most 1 024 tokens; beyond that a from-scratch pass with HBM-resident lists follows, as in the dense path)
    const int nb = std::min(N[0], 1024);
    const int cb = std::min(C[0], 4 * nb);
    // lag mode (see the window loop): the 64-token first tier redoes its own failed speculations one window later and sends
    // capacity overflows to the from-scratch list pass instead of a large-tier launch per window
    will_lag = true;                             // (either first tier: the 64-token kernel or the general one — epsilon batches)
    { const char *e = getenv("MFA_VIT_LAG"); if (e && e[0] == '0') will_lag = false; }
    const bool second = nb < N[0] || will_lag;   // a from-scratch list pass: table growth beyond the large tier / the first tier
    Launch a{0, kSmallTokens, std::min(C[0], 4 * kSmallTokens), 0, second ? 1 : 0};
    a.N2 = nb; a.C2 = cb;
    plan.push_back(a);
    if (second) plan.push_back({0, N[0], C[0], ST_GROW, 0});
  } else if (N[0] > kSmallTokens) {
    int cs = std::min(C[0], 4 * kSmallTokens);
    plan.push_back({0, kSmallTokens, cs, 0, 1});
    plan.push_back({0, N[0], C[0], ST_GROW, 0});
  } else {
    plan.push_back({0, N[0], C[0], 0, 0});
  }
  if (passes == 2) plan.push_back({1, N[1], C[1], ST_PENDING, 0});
  for (Launch &L : plan) {
    const int ps = L.pass;
    // token lists in LDS when everything fits comfortably; in HBM for big graphs / the wide retry beam; and if even the
    // atomically updated tables do not fit, shrink the token capacity (an overflow is then reported per utterance)
    // (the list passes — table growth, retry beam — hold a handful of utterances: occupancy does not matter there, the
    //  per-frame latency of HBM-resident lists does)
    bool lists_in_lds = lds_bytes(max_states, L.N, L.C, true, eps) <= kLdsLimit / 2 ||
                        ((ps == 0 || L.code != 0) && lds_bytes(max_states, L.N, L.C, true, eps) <= kLdsLimit);
    while (lds_bytes(max_states, L.N, L.C, lists_in_lds, eps) > kLdsLimit && L.N > 64) {
      L.N = (L.N / 2 + 63) & ~63;
      if (L.C > 8 * L.N) L.C = 8 * L.N;
    }
    size_t lds = lds_bytes(max_states, L.N, L.C, lists_in_lds, eps);
    if (lds > kLdsLimit) return c->fail("Viterbi tables need %zu bytes of LDS (> 160 KiB): %d states, %d tokens", lds, max_states, L.N);
    VitParams p;
    memset(&p, 0, sizeof(p));
    p.g = *g; p.ll = d_loglikes; p.ll_off = d_ll_off; p.ll_cols = d_ll_cols; p.frame_off = d_frame_off;
    p.beam = ps == 0 ? o->beam : o->retry_beam; p.scale = o->acoustic_scale;
    p.nmax = L.N; p.cmax = L.C; p.bpf = bpf; p.pass = ps; p.grow = L.grow; p.hbits = hash_bits(max_states, L.N);
    // workspace strides follow this launch's capacities (lists and stash are per-launch scratch)
    WsLayout wp = ws_layout(n_utt, total_frames, L.N, L.C, bpf, total_arcs, eps_states);
    p.w_state = (u32 *)(base + wp.state); p.w_cost = (double *)(base + wp.cost);
    p.w_stash_a = (u32 *)(base + wp.sta); p.w_stash_b = (u32 *)(base + wp.stb); p.w_stash_key = (u64 *)(base + wp.stkey);
    p.w_bp = (u64 *)(base + wp.bp); p.w_tokoff = (u32 *)(base + wp.tokoff);
    p.w_hash = (u32 *)(base + w.hash);     // fixed location across launches
    p.w_arcnext = (const uint4 *)(base + w.arcnext);
    p.w_epsinfo = eps ? (const u32 *)(base + w.eps) : nullptr; p.eps_stride = max_states;
    int eps_pops_env = 0;                  // MFA_VIT_EPS_POPS (tests: forces the hand-over to the general decoder)
    { const char *e = getenv("MFA_VIT_EPS_POPS"); if (e && atoi(e) > 0) eps_pops_env = atoi(e); }
    p.eps_pops = eps_pops_env ? eps_pops_env : 64 * L.N;
    p.llcap = kLlCap;
    p.stamps = (unsigned long long *)c->vit_stamps;
    int32_t *d_list = (int32_t *)(base + w.list), *d_count = (int32_t *)(base + w.count);
    p.utt_list = nullptr; p.n_list = nullptr;
    p.ali = d_ali; p.words = d_words; p.n_words = d_n_words; p.like = d_like; p.frame_like = d_frame_like; p.status = d_status;
    if (L.code != 0) {
      MFA_HIP_CHECK(c, hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
      hipLaunchKernelGGL(collect_pending_kernel, dim3((n_utt + 255) / 256), dim3(256), 0, c->stream, d_status, n_utt, L.code, d_list, d_count);
      p.utt_list = d_list; p.n_list = d_count;
    }
    p.windowed = 0; p.t_begin = 0; p.t_end = 0x7fffffff; p.next_window = 0;
    p.w_vstate = (VitState *)(base + w.vstate); p.state_depth = nullptr; p.band = nullptr;
    p.redo_mode = 0; p.w_redo = (u32 *)(base + w.redo); p.npark = L.N2 > 0 ? L.N2 : L.N;
    // large tier of this pass (same workspace strides as a launch of its own would have)
    size_t lds2 = 0;
    VitParams p2 = p;
    if (L.N2 > 0) {
      lds2 = lds_bytes(max_states, L.N2, L.C2, true, eps);
      while (lds2 > kLdsLimit && L.N2 > 128) {         // (the epsilon tables can push a 1 024-token large tier over the limit)
        L.N2 = (L.N2 / 2 + 63) & ~63;
        if (L.C2 > 4 * L.N2) L.C2 = 4 * L.N2;
        lds2 = lds_bytes(max_states, L.N2, L.C2, true, eps);
      }
      if (lds2 > kLdsLimit) return c->fail("Viterbi large tier needs %zu bytes of LDS", lds2);
      WsLayout w2 = ws_layout(n_utt, total_frames, L.N2, L.C2, bpf, total_arcs, eps_states);
      p2.nmax = L.N2; p2.cmax = L.C2; p2.hbits = hash_bits(max_states, L.N2);
      p2.eps_pops = eps_pops_env ? eps_pops_env : 64 * L.N2;
      // park arrays (state / cost) and the back-pointer trail are SHARED between the tiers: the layout of the large one
      p2.w_state = (u32 *)(base + w2.state); p2.w_cost = (double *)(base + w2.cost);
      p2.w_stash_a = (u32 *)(base + w2.sta); p2.w_stash_b = (u32 *)(base + w2.stb); p2.w_stash_key = (u64 *)(base + w2.stkey);
      p2.w_bp = (u64 *)(base + w2.bp); p2.w_tokoff = (u32 *)(base + w2.tokoff);
      p.w_state = p2.w_state; p.w_cost = p2.w_cost; p.w_bp = p2.w_bp; p.w_tokoff = p2.w_tokoff;
      // the small tier's per-frame scratch lives inside the large tier's arrays too (its strides are smaller): a layout
      // of its own would put its candidate stash on top of the other tier's parked lists
      p.w_stash_a = p2.w_stash_a; p.w_stash_b = p2.w_stash_b; p.w_stash_key = p2.w_stash_key;
      MFA_HIP_CHECK(c, hipMemsetAsync(base + w.redo, 0, (size_t)n_utt * 4, c->stream));
    }
    // four instantiations: token lists in LDS or HBM × epsilon-free or not
    void (*k_lds)(VitParams) = eps ? viterbi_kernel<true, true> : viterbi_kernel<true, false>;
    void (*k_hbm)(VitParams) = eps ? viterbi_kernel<false, true> : viterbi_kernel<false, false>;
    if (lists_in_lds) MFA_HIP_CHECK(c, hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, lds2)));
    else MFA_HIP_CHECK(c, hipFuncSetAttribute((const void *)k_hbm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (lds2 > 0) MFA_HIP_CHECK(c, hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, lds2)));
    auto launch_decoder = [&]() {
      KernelTimer kt(c, MFA_K_VITERBI);
      if (lists_in_lds) hipLaunchKernelGGL(k_lds, dim3(n_utt), dim3(64), lds, c->stream, p);
      else hipLaunchKernelGGL(k_hbm, dim3(n_utt), dim3(64), lds, c->stream, p);
    };
    // first tier of the windowed pass: the dedicated 64-token kernel (MFA_VIT_LEAN=0: the general kernel as first tier)
    constexpr int kSmallRounds = 3;
    const size_t lds_small = (size_t)kSmallN * (8 + 16 + 6 * 4 + 8 + 8) + 256 * 4 + (size_t)64 * kSmallRounds * 4 + 16 + (size_t)kBmWords * 4 +
                             (eps ? (size_t)kSmallN * 7 * 4 : 0);      // (epsilon closure: position / inverse / arc / vote / info + stack)
    bool lean = lazy && L.N2 > 0 && lists_in_lds && L.code == 0 && L.N == kSmallN;
    { const char *e = getenv("MFA_VIT_LEAN"); if (e && e[0] == '0') lean = false; }
    if (!lazy) {
      launch_decoder();
      MFA_DEBUG_POINT(c, "decoded pass=%d code=%d N=%d C=%d lds=%zu in_lds=%d", ps, L.code, L.N, L.C, lds, (int)lists_in_lds);
    } else {
      // The rare passes (table growth, retry beam) keep far more of the graph alive, so a narrow band buys little there:
      // they take wide windows and few launches.
      const int K = L.code == 0 ? lazy->window : std::max(lazy->window, 256);
      p.windowed = 1; p.next_window = K;
      p.state_depth = lazy->plan.d_state_depth; p.band = (int32_t *)(base + w.band);
      // lagmode (the tiered first-beam pass, whichever kernel is its first tier; MFA_VIT_LAG=0 turns it off): an utterance whose
      // speculative window failed is not handed to the large tier — a launch that a handful of wavefronts can use, one workgroup
      // with up to 100 KB of LDS per utterance of the batch to find them — but falls one window behind: the next scoring launch
      // scores the failed window again for it, with the proven band, the next first-tier launch redoes it, and so on to the end,
      // where one extra round of launches finishes the stragglers.  Capacity overflows wait for the from-scratch list pass.
      const bool tiered_pass = L.N2 > 0 && lists_in_lds && L.code == 0;
      const bool lagmode = tiered_pass && will_lag;
      p.lagmode = lagmode ? 1 : 0; p2.lagmode = p.lagmode;
      const int t_loop_end = lazy->max_frames + (lagmode ? K : 0);
      for (int t0 = 0; t0 < t_loop_end; t0 += K) {
        MfaWindowScore ws;
        memset(&ws, 0, sizeof(ws));
        if (lagmode) { ws.lag = (const int32_t *)(base + w.vstate); ws.lag_stride = (int)(sizeof(VitState) / 4); ws.lag_word = 3; }
        ws.t_begin = t0; ws.window = K; ws.band = p.band; ws.utt_list = p.utt_list; ws.n_list = p.n_list;
        ws.cols_per_wave = L.code == 0 ? 0 : 32;   // list passes: few utterances, wide bands — spread the columns over wavefronts
        const bool spec = L.code == 0 && L.N2 > 0 && lists_in_lds && spec_slack > 0;   // (only the tiered first-beam pass speculates)
        ws.hi_slack = spec ? spec_slack : 0;
        ws.done = (const int32_t *)(base + w.vstate); ws.done_stride = (int)(sizeof(VitState) / 4); ws.done_word = 2;
        if (mfa_gmm_score_window(c, lazy, &ws, d_frame_off, n_utt, d_ll_off, (float *)d_loglikes) != 0) return -1;
        MFA_DEBUG_POINT(c, "scored window t0=%d K=%d pass=%d code=%d N=%d C=%d", t0, K, ps, L.code, L.N, L.C);
        p.t_begin = t0; p.t_end = t0 + K;
        p.spec = spec ? 1 : 0; p.spec_ranges = mfa_band_ranges(c); p.spec_class_counts = lazy->plan.d_class_counts;
        p.spec_groups = lazy->plan.groups;
        p2.spec = p.spec; p2.spec_ranges = p.spec_ranges; p2.spec_class_counts = p.spec_class_counts; p2.spec_groups = p.spec_groups;
        if (L.N2 > 0 && lists_in_lds) {
          p.redo_mode = 1;
          if (lean) {
            KernelTimer kt1(c, MFA_K_VITERBI);
            if (eps) hipLaunchKernelGGL((viterbi_small_kernel<kSmallRounds, true>), dim3(n_utt), dim3(64), lds_small, c->stream, p);
            else hipLaunchKernelGGL((viterbi_small_kernel<kSmallRounds, false>), dim3(n_utt), dim3(64), lds_small, c->stream, p);
          } else {
            launch_decoder();
          }
          if (lagmode) {      // nothing else per window: failed speculations lag, capacity overflows wait for the list pass below
            MFA_DEBUG_POINT(c, "decoded window t0=%d K=%d (first tier, lag mode)", t0, K);
            continue;
          }
          p2.windowed = 1; p2.next_window = K; p2.state_depth = p.state_depth; p2.band = p.band;
          p2.t_begin = t0; p2.t_end = t0 + K; p2.redo_mode = 2; p2.npark = p.npark; p2.grow = L.grow;
          p2.utt_list = p.utt_list; p2.n_list = p.n_list;
          if (spec) {
            // Window-level redo of a failed speculation: the utterances the first tier handed over (token overflow or a
            // score asked for outside the narrow band) get THIS window scored again with the proven band, then the large
            // tier redoes it from the parked state without the check.  Mostly empty launches: a fraction of a percent of
            // the (utterance, window) pairs are flagged.
            MFA_HIP_CHECK(c, hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(collect_flagged_kernel, dim3((n_utt + 255) / 256), dim3(256), 0, c->stream, p.w_redo, n_utt, d_list, d_count);
            MfaWindowScore ws2 = ws;
            ws2.hi_slack = 0; ws2.utt_list = d_list; ws2.n_list = d_count; ws2.cols_per_wave = 0;
            if (mfa_gmm_score_window(c, lazy, &ws2, d_frame_off, n_utt, d_ll_off, (float *)d_loglikes) != 0) return -1;
            p2.spec = 0;
          }
          {
            KernelTimer kt2(c, MFA_K_VITERBI);
            hipLaunchKernelGGL(k_lds, dim3(n_utt), dim3(64), lds2, c->stream, p2);
          }
        } else {
          launch_decoder();
        }
        MFA_DEBUG_POINT(c, "decoded window t0=%d K=%d pass=%d code=%d N=%d C=%d lds=%zu in_lds=%d", t0, K, ps, L.code, L.N, L.C, lds, (int)lists_in_lds);
      }
      if (lean) {   // utterances the first tier decoded to their last frame: ReachedFinal, traceback, outputs
        KernelTimer kt3(c, MFA_K_VITERBI);
        hipLaunchKernelGGL(viterbi_finish_kernel, dim3(n_utt), dim3(64), 0, c->stream, p);
      }
    }
    MFA_HIP_CHECK(c, hipGetLastError());
  }
  c->xsplit_ready = false;
  hipLaunchKernelGGL(finalize_pending_kernel, dim3((n_utt + 255) / 256), dim3(256), 0, c->stream, d_status, n_utt);
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

MFA_API int mfa_align_batch(mfa_ctx *c, const mfa_graph_batch *g, const float *d_loglikes, const int64_t *d_ll_off,
                            const int32_t *d_ll_cols, const int64_t *d_frame_off, int64_t total_frames,
                            int64_t total_arcs, int32_t max_states, int32_t max_arcs, const mfa_align_opts *o, int32_t *d_ali, int32_t *d_words, int32_t *d_n_words,
                            float *d_like, float *d_frame_like, int32_t *d_status) {
  return align_impl(c, g, d_loglikes, d_ll_off, d_ll_cols, d_frame_off, total_frames, total_arcs, max_states, max_arcs, o,
                    d_ali, d_words, d_n_words, d_like, d_frame_like, d_status, nullptr);
}

MFA_API int mfa_align_features_batch(mfa_ctx *c, const mfa_graph_batch *g, const mfa_score_plan *plan, const float *d_feats,
                                     const int64_t *d_frame_off, int32_t max_frames, int64_t total_frames, int64_t total_arcs,
                                     int32_t max_states, int32_t max_arcs, const mfa_align_opts *o, int32_t window,
                                     float *d_loglikes, const int64_t *d_ll_off, const int32_t *d_ll_cols, int32_t *d_ali,
                                     int32_t *d_words, int32_t *d_n_words, float *d_like, float *d_frame_like,
                                     int32_t *d_status) {
  if (!plan || !plan->d_pdf_list || !plan->d_pdf_off || !plan->d_class_counts || !plan->d_pdf_first_frame ||
      !plan->d_pdf_last_depth || !plan->d_state_depth)
    return c->fail("mfa_align_features_batch: incomplete score plan");
  if (window <= 0 || window % 64 != 0) return c->fail("mfa_align_features_batch: window must be a positive multiple of 64 frames (got %d)", window);
  if (max_frames <= 0) return c->fail("mfa_align_features_batch: max_frames must be positive");
  if (!mfa_gmm_lazy_supported(c)) {
    // feature dimensions beyond the MFMA kernels' instantiations: dense scoring (naive kernel), then the decoder
    if (mfa_gmm_score_batch(c, d_feats, d_frame_off, g->n_utt, max_frames, plan->d_pdf_list, plan->d_pdf_off,
                            plan->d_class_counts, plan->d_pdf_first_frame, d_ll_off, d_loglikes) != 0) return -1;
    return align_impl(c, g, d_loglikes, d_ll_off, d_ll_cols, d_frame_off, total_frames, total_arcs, max_states, max_arcs, o,
                      d_ali, d_words, d_n_words, d_like, d_frame_like, d_status, nullptr);
  }
  if (plan->max_cols <= 0) return c->fail("mfa_align_features_batch: plan.max_cols must be the largest column count of the batch");
  MfaLazyScoring lazy;
  lazy.plan = *plan; lazy.d_feats = d_feats; lazy.max_frames = max_frames; lazy.window = window;
  return align_impl(c, g, d_loglikes, d_ll_off, d_ll_cols, d_frame_off, total_frames, total_arcs, max_states, max_arcs, o,
                    d_ali, d_words, d_n_words, d_like, d_frame_like, d_status, &lazy);
}

}  // extern "C"
