// General-graph decoder for gfx950: training graphs WITH epsilon input arcs (kalpy-compiled fsts.*.ark can hold them).
// Replaces the same call as viterbi.hip — GmmAligner.align_utterance / export_alignments
// (MFA/alignment/multiprocessing.py:846-853; MFA/online/alignment.py:107) = Kaldi AlignUtteranceWrapper + FasterDecoder —
// for the utterances whose graph the wavefront-parallel decoder does not take.
//
// FasterDecoder's ProcessNonemitting walks a LIFO queue and inserts into the token hash list as it goes, so which token
// wins a tie and where a new state lands in the list depend on a strictly sequential order.  Rather than re-derive that
// order in parallel form (as viterbi.hip does for ProcessEmitting), this kernel runs the sequential algorithm as it is
// written — hash list with per-bucket insertion (util/hash-list-inl.h), token pool, GetCutoff with min_active, PossiblyResizeHash,
// ProcessEmitting, ProcessNonemitting, ReachedFinal, best path, retry beam — with ONE THREAD PER UTTERANCE and every data
// structure in a per-utterance HBM workspace.  Parallelism is across utterances only (a wavefront = 64 utterances, divergent);
// a 10 s utterance costs tens of milliseconds, a batch of thousands about as much.  This is the correctness path for
// graphs with epsilon arcs, not the throughput path: results are bit-identical to the oracle's (tests/test_gpu_general.py).
#include <cmath>
#include <cstdint>
#include <cstring>

#include <algorithm>
#include <vector>

#include "ctx.hpp"

namespace {

enum { G_OK = 0, G_RETRIED = 1, G_FAILED = 2, G_TOKEN_OVERFLOW = 3, G_BP_OVERFLOW = 4, G_INTERNAL = 6, G_WORDS = 7 };
constexpr int kMinActive = 20;
constexpr float kBeamDelta = 0.5f;
constexpr float kHashRatio = 2.0f;

struct GenParams {
  mfa_graph_batch g;
  const float *ll; const int64_t *ll_off; const int32_t *ll_cols; const int64_t *frame_off;
  float beam, retry_beam, scale;
  int ncap;            // live-token capacity (elements per hash-list generation)
  int hcap;            // hash buckets capacity
  int qcap;            // nonemitting queue capacity
  int ppf;             // token-pool entries per frame
  // per-utterance workspace (strides follow the capacities above; pool at pool_off[utt])
  double *tok_cost; int32_t *tok_arc; int32_t *tok_prev; const int64_t *pool_off;
  int32_t *el_key; int32_t *el_val; int32_t *el_tail;     // [n_utt][2][ncap]
  int32_t *bk_prev; int32_t *bk_last;                     // [n_utt][hcap]
  int32_t *queue;                                         // [n_utt][qcap]
  double *tmp;                                            // [n_utt][ncap]
  int32_t *rev;                                           // [n_utt][rev_cap] reversed best path (arc indices)
  int rev_cap;
  int32_t *ali; int32_t *words; int32_t *n_words; float *like; float *frame_like; int32_t *status;
  // work items [item_base, item_end) of this launch; item → utterance utt_list[item] (second tier: the utterances whose token
  // pool overflowed in the first) or the item itself; workspace slot = item − item_base
  int item_base, item_end;
  const int32_t *utt_list;
};

// One utterance, one thread: the oracle's FasterDecoder (oracle/mfa_oracle.cpp) with arrays in place of std::vector.
struct Decoder {
  const GenParams &p; int utt;
  // graph
  int S; int64_t ab; const int32_t *arc_off; const float *final_w;
  const int32_t *a_next; const float *a_w; const int32_t *a_col; const int32_t *a_il; const int32_t *a_ol;
  int start;
  // scores
  const float *ll; int P, T;
  // workspace
  double *tok_cost; int32_t *tok_arc; int32_t *tok_prev; int64_t pool_cap, pool_n;
  int32_t *el_key, *el_val, *el_tail; int el_base, el_n;     // current generation [el_base, el_base + ncap)
  int32_t *bk_prev, *bk_last; int32_t *queue; double *tmp;
  int list_head, bucket_list_tail; int hash_size;
  int status;

  __device__ Decoder(const GenParams &pp, int u, int w) : p(pp), utt(u) {
    const int64_t so = p.g.d_state_off[u];
    S = (int)(p.g.d_state_off[u + 1] - so);
    ab = p.g.d_arc_base[u];
    arc_off = p.g.d_arc_off + so + u; final_w = p.g.d_final + so;
    a_next = p.g.d_arc_next + ab; a_w = p.g.d_arc_weight + ab; a_col = p.g.d_arc_col + ab;
    a_il = p.g.d_arc_ilabel + ab; a_ol = p.g.d_arc_olabel + ab;
    start = p.g.d_start[u];
    const int64_t f0 = p.frame_off[u];
    T = (int)(p.frame_off[u + 1] - f0);
    ll = p.ll + p.ll_off[u]; P = p.ll_cols[u];
    const int64_t po = p.pool_off[w];
    pool_cap = p.pool_off[w + 1] - po;
    tok_cost = p.tok_cost + po; tok_arc = p.tok_arc + po; tok_prev = p.tok_prev + po;
    el_key = p.el_key + (size_t)w * 2 * p.ncap; el_val = p.el_val + (size_t)w * 2 * p.ncap; el_tail = p.el_tail + (size_t)w * 2 * p.ncap;
    bk_prev = p.bk_prev + (size_t)w * p.hcap; bk_last = p.bk_last + (size_t)w * p.hcap;
    queue = p.queue + (size_t)w * p.qcap; tmp = p.tmp + (size_t)w * p.ncap;
    hash_size = 1000 < p.hcap ? 1000 : p.hcap;
    status = G_OK;
  }

  // ---- HashList (util/hash-list-inl.h): list order = buckets by first occupancy, within a bucket by insertion
  __device__ int Clear() {
    for (int b = bucket_list_tail; b != -1; b = bk_prev[b]) bk_last[b] = -1;
    bucket_list_tail = -1;
    const int ans = list_head; list_head = -1; return ans;
  }
  __device__ int Insert(int key, int val, bool *inserted) {
    const int index = (int)((unsigned)key % (unsigned)hash_size);
    if (bk_last[index] != -1) {
      const int head = bk_prev[index] == -1 ? list_head : el_tail[bk_last[bk_prev[index]]];
      const int tail = el_tail[bk_last[index]];
      for (int e = head; e != tail; e = el_tail[e]) if (el_key[e] == key) { *inserted = false; return e; }
    }
    if (el_n >= p.ncap) { status = G_TOKEN_OVERFLOW; *inserted = false; return -1; }
    const int ei = el_base + el_n++;
    el_key[ei] = key; el_val[ei] = val; el_tail[ei] = -1;
    *inserted = true;
    if (bk_last[index] == -1) {
      if (bucket_list_tail == -1) list_head = ei; else el_tail[bk_last[bucket_list_tail]] = ei;
      el_tail[ei] = -1;
      bk_last[index] = ei;
      bk_prev[index] = bucket_list_tail;
      bucket_list_tail = index;
    } else {
      el_tail[ei] = el_tail[bk_last[index]];
      el_tail[bk_last[index]] = ei;
      bk_last[index] = ei;
    }
    return ei;
  }

  // NewToken: cost = prev.cost + arc.weight (+ ac_cost): Kaldi's double + float arithmetic (left to right)
  __device__ int NewToken(int arc, double cost, int prev) {
    if (pool_n >= pool_cap) { status = G_BP_OVERFLOW; return -1; }
    const int t = (int)pool_n++;
    tok_arc[t] = arc; tok_prev[t] = prev; tok_cost[t] = cost;
    return t;
  }

  __device__ void ProcessNonemitting(double cutoff) {
    int qn = 0;
    for (int e = list_head; e != -1; e = el_tail[e]) {
      if (qn >= p.qcap) { status = G_INTERNAL; return; }
      queue[qn++] = e;
    }
    while (qn > 0 && status == G_OK) {
      const int e = queue[--qn];
      const int state = el_key[e], tok = el_val[e];
      if (tok_cost[tok] > cutoff) continue;
      for (int a = arc_off[state]; a < arc_off[state + 1]; a++) {
        if (a_il[a] != 0) continue;
        const double nc = tok_cost[tok] + a_w[a];
        if (nc > cutoff) continue;
        // (the oracle allocates the token first and discards it when it loses; allocating only winners leaves the same
        //  surviving tokens, costs and back-pointers)
        bool ins = false;
        const int key = a_next[a];
        const int index = (int)((unsigned)key % (unsigned)hash_size);
        int found = -1;
        if (bk_last[index] != -1) {
          const int head = bk_prev[index] == -1 ? list_head : el_tail[bk_last[bk_prev[index]]];
          const int tail = el_tail[bk_last[index]];
          for (int x = head; x != tail; x = el_tail[x]) if (el_key[x] == key) { found = x; break; }
        }
        if (found == -1) {
          const int nt = NewToken(a, nc, tok);
          if (nt < 0) return;
          const int ef = Insert(key, nt, &ins);
          if (ef < 0) return;
          if (qn >= p.qcap) { status = G_INTERNAL; return; }
          queue[qn++] = ef;
        } else if (tok_cost[el_val[found]] > nc) {
          const int nt = NewToken(a, nc, tok);
          if (nt < 0) return;
          el_val[found] = nt;
          if (qn >= p.qcap) { status = G_INTERNAL; return; }
          queue[qn++] = found;
        }
      }
    }
  }

  __device__ void InitDecoding() {
    // hash list reset (Clear), new generation
    for (int b = 0; b < p.hcap; b++) { bk_last[b] = -1; bk_prev[b] = -1; }
    bucket_list_tail = -1; list_head = -1;
    el_base = 0; el_n = 0; pool_n = 0;
    const int t0 = NewToken(-1, 0.0, -1);
    bool ins;
    Insert(start, t0, &ins);
    ProcessNonemitting(3.4028234663852886e38);   // std::numeric_limits<float>::max()
  }

  // value at sorted position k of tmp[0..n) (std::nth_element's tmp[k]); tmp is permuted
  __device__ double NthElement(int n, int k) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const double pivot = tmp[(lo + hi) >> 1];
      int i = lo, j = hi;
      while (i <= j) {
        while (tmp[i] < pivot) i++;
        while (tmp[j] > pivot) j--;
        if (i <= j) { const double x = tmp[i]; tmp[i] = tmp[j]; tmp[j] = x; i++; j--; }
      }
      if (k <= j) hi = j; else if (k >= i) lo = i; else break;
    }
    return tmp[k];
  }

  __device__ double GetCutoff(int head, int *tok_count, float *adaptive_beam, int *best_elem, float beam) {
    double best_cost = INFINITY;
    int count = 0;
    for (int e = head; e != -1; e = el_tail[e], count++) {
      const double w = tok_cost[el_val[e]];
      if (count < p.ncap) tmp[count] = w;
      if (w < best_cost) { best_cost = w; *best_elem = e; }
    }
    *tok_count = count;
    const double beam_cutoff = best_cost + beam;
    double min_active_cutoff = INFINITY;
    if (count > kMinActive) min_active_cutoff = NthElement(count, kMinActive);
    if (min_active_cutoff > beam_cutoff) { *adaptive_beam = (float)(min_active_cutoff - best_cost + kBeamDelta); return min_active_cutoff; }
    *adaptive_beam = beam;
    return beam_cutoff;
  }

  __device__ double ProcessEmitting(int frame, float beam) {
    const int last_toks = Clear();
    // the new generation of elements lives in the other half of the element pool
    el_base = el_base == 0 ? p.ncap : 0; el_n = 0;
    int tok_cnt = 0, best_elem = -1; float adaptive_beam = 0.0f;
    const double weight_cutoff = GetCutoff(last_toks, &tok_cnt, &adaptive_beam, &best_elem, beam);
    int new_sz = (int)((float)tok_cnt * kHashRatio);    // PossiblyResizeHash
    if (new_sz > p.hcap) new_sz = p.hcap;
    if (new_sz > hash_size) hash_size = new_sz;
    double next_weight_cutoff = INFINITY;
    const float *row = ll + (size_t)frame * P;
    if (best_elem != -1) {
      const int state = el_key[best_elem], tok = el_val[best_elem];
      for (int a = arc_off[state]; a < arc_off[state + 1]; a++) {
        if (a_il[a] != 0) {
          const float ac_cost = -(p.scale * row[a_col[a]]);
          const double nw = ((double)a_w[a] + tok_cost[tok]) + (double)ac_cost;
          if (nw + adaptive_beam < next_weight_cutoff) next_weight_cutoff = nw + adaptive_beam;
        }
      }
    }
    for (int e = last_toks; e != -1 && status == G_OK; e = el_tail[e]) {
      const int state = el_key[e], tok = el_val[e];
      if (tok_cost[tok] < weight_cutoff) {
        for (int a = arc_off[state]; a < arc_off[state + 1]; a++) {
          if (a_il[a] == 0) continue;
          const float ac_cost = -(p.scale * row[a_col[a]]);
          const double nw = ((double)a_w[a] + tok_cost[tok]) + (double)ac_cost;
          if (nw < next_weight_cutoff) {
            if (nw + adaptive_beam < next_weight_cutoff) next_weight_cutoff = nw + adaptive_beam;
            // find-or-insert the destination (allocating a token only when it is new or strictly better)
            const int key = a_next[a];
            const int index = (int)((unsigned)key % (unsigned)hash_size);
            int found = -1;
            if (bk_last[index] != -1) {
              const int head = bk_prev[index] == -1 ? list_head : el_tail[bk_last[bk_prev[index]]];
              const int tail = el_tail[bk_last[index]];
              for (int x = head; x != tail; x = el_tail[x]) if (el_key[x] == key) { found = x; break; }
            }
            if (found == -1) {
              const int nt = NewToken(a, nw, tok);
              if (nt < 0) break;
              bool ins;
              if (Insert(key, nt, &ins) < 0) break;
            } else if (tok_cost[el_val[found]] > nw) {
              const int nt = NewToken(a, nw, tok);
              if (nt < 0) break;
              el_val[found] = nt;
            }
          }
        }
      }
    }
    return next_weight_cutoff;
  }

  __device__ void Decode(float beam) {
    InitDecoding();
    for (int t = 0; t < T && status == G_OK; t++) {
      const double c = ProcessEmitting(t, beam);
      if (status != G_OK) break;
      ProcessNonemitting(c);
    }
  }

  __device__ bool ReachedFinal() const {
    for (int e = list_head; e != -1; e = el_tail[e])
      if (tok_cost[el_val[e]] != INFINITY && final_w[el_key[e]] != INFINITY) return true;
    return false;
  }
};

__global__ __launch_bounds__(64) void viterbi_general_kernel(GenParams p) {
  const int item = p.item_base + blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= p.item_end) return;
  const int utt = p.utt_list ? p.utt_list[item] : item;
  Decoder d(p, utt, item - p.item_base);
  const int64_t f0 = p.frame_off[utt];
  int32_t *ali = p.ali + f0, *words = p.words + f0;
  float *flike = p.frame_like ? p.frame_like + f0 : nullptr;
  auto fail = [&](int st) { p.status[utt] = st; p.n_words[utt] = 0; p.like[utt] = 0.0f; };
  if (d.S <= 0 || d.start < 0 || d.start >= d.S || d.T <= 0) { fail(G_FAILED); return; }
  d.Decode(p.beam);
  int status = G_OK;
  bool ans = d.status == G_OK && d.ReachedFinal();
  if (d.status == G_OK && !ans && p.retry_beam != 0.0f) {
    status = G_RETRIED;
    d.Decode(p.retry_beam);
    ans = d.status == G_OK && d.ReachedFinal();
  }
  if (d.status != G_OK) { fail(d.status); return; }
  if (!ans) { fail(G_FAILED); return; }
  // ---- best final token (first in list order on ties), path back through the token pool
  int best_tok = -1; double best_cost = INFINITY;
  for (int e = d.list_head; e != -1; e = d.el_tail[e]) {
    const double c = d.tok_cost[d.el_val[e]] + (double)d.final_w[d.el_key[e]];
    if (c < best_cost && c != INFINITY) { best_cost = c; best_tok = d.el_val[e]; }
  }
  if (best_tok < 0) { fail(G_FAILED); return; }
  int32_t *rev = p.rev + (size_t)(item - p.item_base) * p.rev_cap;
  int n_rev = 0;
  for (int tok = best_tok; tok != -1; tok = d.tok_prev[tok]) {
    if (n_rev >= p.rev_cap) { fail(G_INTERNAL); return; }
    rev[n_rev++] = tok;
  }
  n_rev--;   // the start token carries no arc
  // GetLinearSymbolSequence: float accumulation of graph and acoustic costs in path order
  float w1 = 0.0f, w2 = 0.0f;
  int n_ali = 0, n_w = 0;
  const float inv_scale = -1.0f / p.scale;
  for (int i = n_rev - 1; i >= 0; i--) {
    const int tok = rev[i], prev = d.tok_prev[tok], a = d.tok_arc[tok];
    const float tot = (float)(d.tok_cost[tok] - (prev >= 0 ? d.tok_cost[prev] : 0.0));
    const float gc = d.a_w[a], ac = tot - gc;
    w1 += gc; w2 += ac;
    if (d.a_il[a] != 0) {
      if (n_ali < d.T) { ali[n_ali] = d.a_il[a]; if (flike) flike[n_ali] = ac * inv_scale; }
      n_ali++;
    }
    if (d.a_ol[a] != 0) { if (n_w < d.T) words[n_w] = d.a_ol[a]; n_w++; }
  }
  if (n_ali != d.T) { fail(G_INTERNAL); return; }
  if (n_w > d.T) { fail(G_WORDS); return; }   // more word labels than frames: d_words holds one per frame (mfa_hip.h, status 7)
  w1 += d.final_w[d.a_next[d.tok_arc[best_tok]]];
  p.like[utt] = -(w1 + w2) / p.scale;
  p.n_words[utt] = n_w;
  p.status[utt] = status;
}

}  // namespace

extern "C" {

MFA_API int mfa_align_general_batch(mfa_ctx *c, const mfa_graph_batch *g, const float *d_loglikes, const int64_t *d_ll_off,
                                    const int32_t *d_ll_cols, const int64_t *d_frame_off, const int64_t *h_frame_off,
                                    int32_t max_states, int32_t max_arcs, const mfa_align_opts *o, int32_t *d_ali,
                                    int32_t *d_words, int32_t *d_n_words, float *d_like, float *d_frame_like,
                                    int32_t *d_status) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  const int n_utt = g->n_utt;
  if (n_utt <= 0) return 0;
  if (o->beam <= 0.0f || (o->retry_beam != 0.0f && o->retry_beam <= o->beam))
    return c->fail("Beams do not make sense: beam %f, retry-beam %f", o->beam, o->retry_beam);
  if (max_states <= 0 || max_arcs <= 0) return c->fail("max_states and max_arcs must be positive");
  GenParams p;
  memset(&p, 0, sizeof(p));
  p.g = *g; p.ll = d_loglikes; p.ll_off = d_ll_off; p.ll_cols = d_ll_cols; p.frame_off = d_frame_off;
  p.beam = o->beam; p.retry_beam = o->retry_beam; p.scale = o->acoustic_scale;
  // one token per graph state is a hard upper bound on the live tokens, one bucket per two tokens on the hash size
  p.ncap = (max_states + 63) & ~63;
  p.hcap = 2 * p.ncap > 1000 ? 2 * p.ncap : 1000;
  p.qcap = 8 * p.ncap;
  int64_t max_frames = 0;
  for (int u = 0; u < n_utt; u++) max_frames = std::max<int64_t>(max_frames, h_frame_off[u + 1] - h_frame_off[u]);
  const int ppf = std::min<int64_t>(max_arcs, std::max(4 * (o->bp_tokens_per_frame > 0 ? o->bp_tokens_per_frame : 512), 256));
  p.ppf = ppf;
  p.rev_cap = (int)(4 * max_frames + 64);
  // Workspace per utterance: the token pool (frames × tokens-per-frame × 16 bytes) dominates — 33 MB for 10 s at the full
  // ppf.  A launch takes ≈0.5 s whatever its size (one thread walks one utterance), so throughput is utterances per launch:
  // the first tier gives every utterance a pool of 256 tokens per frame (4 MB per 10 s: what the first beam creates fits
  // several times over), the second tier decodes again, with the full pool, the few whose pool overflowed (wide retry
  // beams).  Each tier runs in chunks whose workspace stays under a cap (MFA_GENERAL_WS_GIB, default 64), one launch each.
  const size_t per_utt_fixed = (size_t)3 * 2 * p.ncap * 4 + (size_t)2 * p.hcap * 4 + (size_t)p.qcap * 4 + (size_t)p.ncap * 8 +
                               (size_t)p.rev_cap * 4 + 8 + 12 * 256;
  size_t cap_bytes = (size_t)64 << 30;
  { const char *e = getenv("MFA_GENERAL_WS_GIB"); if (e && atof(e) > 0.0) cap_bytes = (size_t)(atof(e) * (double)((size_t)1 << 30)); }
  p.ali = d_ali; p.words = d_words; p.n_words = d_n_words; p.like = d_like; p.frame_like = d_frame_like; p.status = d_status;
  int n_chunks = 0;
  // one tier: items [0, n_items) (utterance = list[item] or the item), `tpf` pool entries per frame
  auto run_tier = [&](int n_items, const std::vector<int32_t> *list, const int32_t *d_list, int tpf) -> int {
    p.ppf = tpf; p.utt_list = d_list;
    auto utt_of = [&](int item) { return list ? (*list)[item] : item; };
    for (int i0 = 0; i0 < n_items;) {
      int i1 = i0;
      size_t need = 0;
      while (i1 < n_items) {     // the chunk [i0, i1): at least one utterance, then as many as fit
        const int u = utt_of(i1);
        const size_t add = per_utt_fixed + (size_t)(2 + (h_frame_off[u + 1] - h_frame_off[u]) * (int64_t)tpf) * 16;
        if (i1 > i0 && need + add > cap_bytes) break;
        need += add; i1++;
      }
      const int nc = i1 - i0;
      std::vector<int64_t> pool_off(nc + 1, 0);
      for (int k = 0; k < nc; k++) {
        const int u = utt_of(i0 + k);
        pool_off[k + 1] = pool_off[k] + 2 + (h_frame_off[u + 1] - h_frame_off[u]) * (int64_t)tpf;
      }
      size_t off = 0;
      auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
      const size_t o_cost = take((size_t)pool_off[nc] * 8), o_arc = take((size_t)pool_off[nc] * 4), o_prev = take((size_t)pool_off[nc] * 4);
      const size_t o_poff = take((size_t)(nc + 1) * 8);
      const size_t o_ek = take((size_t)nc * 2 * p.ncap * 4), o_ev = take((size_t)nc * 2 * p.ncap * 4), o_et = take((size_t)nc * 2 * p.ncap * 4);
      const size_t o_bp = take((size_t)nc * p.hcap * 4), o_bl = take((size_t)nc * p.hcap * 4);
      const size_t o_q = take((size_t)nc * p.qcap * 4), o_tmp = take((size_t)nc * p.ncap * 8);
      const size_t o_rev = take((size_t)nc * p.rev_cap * 4);
      if (c->gen_ws_bytes < off) {
        if (c->d_gen_ws) { MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_gen_ws); c->d_gen_ws = nullptr; c->gen_ws_bytes = 0; }
        MFA_HIP_CHECK(c, hipMalloc(&c->d_gen_ws, off));
        c->gen_ws_bytes = off;
      }
      unsigned char *base = (unsigned char *)c->d_gen_ws;
      p.tok_cost = (double *)(base + o_cost); p.tok_arc = (int32_t *)(base + o_arc); p.tok_prev = (int32_t *)(base + o_prev);
      p.pool_off = (const int64_t *)(base + o_poff);
      p.el_key = (int32_t *)(base + o_ek); p.el_val = (int32_t *)(base + o_ev); p.el_tail = (int32_t *)(base + o_et);
      p.bk_prev = (int32_t *)(base + o_bp); p.bk_last = (int32_t *)(base + o_bl);
      p.queue = (int32_t *)(base + o_q); p.tmp = (double *)(base + o_tmp); p.rev = (int32_t *)(base + o_rev);
      MFA_HIP_CHECK(c, hipMemcpyAsync(base + o_poff, pool_off.data(), (size_t)(nc + 1) * 8, hipMemcpyHostToDevice, c->stream));
      MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));   // pool_off is a host vector about to go out of scope
      p.item_base = i0; p.item_end = i1;
      {
        KernelTimer kt(c, MFA_K_VITERBI);
        hipLaunchKernelGGL(viterbi_general_kernel, dim3((nc + 63) / 64), dim3(64), 0, c->stream, p);
      }
      MFA_HIP_CHECK(c, hipGetLastError());
      i0 = i1; n_chunks++;
    }
    return 0;
  };
  const int tpf_small = std::min(ppf, 256);
  if (run_tier(n_utt, nullptr, nullptr, tpf_small) != 0) return -1;
  int n_second = 0;
  if (tpf_small < ppf) {
    std::vector<int32_t> st((size_t)n_utt);
    MFA_HIP_CHECK(c, hipMemcpyAsync(st.data(), d_status, (size_t)n_utt * 4, hipMemcpyDeviceToHost, c->stream));
    MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> again;
    for (int u = 0; u < n_utt; u++) if (st[u] == G_BP_OVERFLOW) again.push_back(u);
    n_second = (int)again.size();
    if (n_second > 0) {
      if (c->gen_list_cap < (size_t)n_second) {
        if (c->d_gen_list) (void)hipFree(c->d_gen_list);
        c->d_gen_list = nullptr; c->gen_list_cap = 0;
        MFA_HIP_CHECK(c, hipMalloc((void **)&c->d_gen_list, (size_t)n_second * 4));
        c->gen_list_cap = (size_t)n_second;
      }
      MFA_HIP_CHECK(c, hipMemcpyAsync(c->d_gen_list, again.data(), (size_t)n_second * 4, hipMemcpyHostToDevice, c->stream));
      MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (run_tier(n_second, &again, c->d_gen_list, ppf) != 0) return -1;
    }
  }
  MFA_DEBUG_POINT(c, "general decoder: %d utterances in %d launches (%d again with the full pool), ncap %d hcap %d ppf %d", n_utt,
                  n_chunks, n_second, p.ncap, p.hcap, ppf);
  return 0;
}

}  // extern "C"
