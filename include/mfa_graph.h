/* Host-side training-graph compiler (C ABI, no GPU, no torch): transcript → HCLG-style training graph, batched and
 * multi-threaded.  Replaces, for whole batches, the call the reference makes per utterance:
 *   kalpy TrainingGraphCompiler.compile_fst(text) / .export_graphs(...)
 *   (MFA/online/alignment.py:96; MFA/alignment/multiprocessing.py:537-571 — kalpy's C++ over OpenFst).
 * The construction is the one montreal_forced_aligner_amd/graph.py documents (lexicon with optional silence → trim →
 * suffix sharing → context expansion → HMM expansion with AddSelfLoopsReorder semantics); this library produces the same
 * graphs bit for bit — state numbering, arc order, float32 weights (tests/test_graph_native_cpu.py) — at native speed.
 *
 * Division of labour with the Python host side: the lexicon arrives as flat tables (one "entry" per distinct word form:
 * its word id and pronunciations with their costs already in −log form), the tree / topology stay in Python and are
 * consulted only for context windows this compiler has not seen yet (mfa_gc_missing_windows → mfa_gc_add_windows).
 *
 * All functions return 0 on success, a negative value on error (mfa_gc_last_error).  Buffers belong to the caller.
 */
#ifndef MFA_GRAPH_H
#define MFA_GRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_GC_API __attribute__((visibility("default")))

typedef struct mfa_gc mfa_gc;

/* One pronunciation of a lexicon entry.  Costs are doubles computed by the host exactly as graph.py computes them:
 *   c0_ns / c0_s / c0_start: cost of the first arc when the pronunciation leaves NS_i / S_i / the start state,
 *   w_ns: −ln(1 − p_after) added on the arc into NS_{i+1}; w_sil: −ln p_after on the arc into the silence node;
 *   has_sil: p_after > 0. */
typedef struct {
  int32_t phone_off;      /* into mfa_gc_config.phones */
  int32_t n_phones;
  double c0_ns, c0_s, c0_start;
  double w_ns, w_sil;
  int32_t has_sil;
  int32_t pad;
} mfa_gc_pron;

typedef struct {
  int32_t context_width;            /* 1 or 3 (central position = width / 2) */
  int32_t share_suffixes;           /* LexiconCompiler(share_suffixes=) */
  int32_t sil_phone;                /* phone id of the optional-silence phone */
  int32_t n_entries;                /* lexicon entries (distinct word forms incl. the OOV entry) */
  const int32_t *entry_word;        /* [n_entries] word id (olabel) */
  const int32_t *entry_pron_off;    /* [n_entries + 1] into prons */
  const mfa_gc_pron *prons;
  const int32_t *phones;            /* phone ids of all pronunciations, position suffixes applied */
  double cost_init_sil;             /* −ln p_init (start —sil→ S_0) */
  double cost_init_eps;             /* −ln(1 − p_init) (folded into the arcs leaving the start state) */
  double final_ns, final_s;         /* final costs of NS_n / S_n */
  /* transition model */
  int32_t n_tids;                   /* transition-ids are 1..n_tids */
  const int32_t *id2state;          /* [n_tids + 1] transition-id → transition-state */
  int32_t n_tstates;
  const int32_t *self_loop_of;      /* [n_tstates + 1] transition-state → its self-loop transition-id, 0 = none */
  int32_t determinize;              /* != 0: DeterminizeStarInLog + MinimizeEncoded between the HMM expansion and the self-loops,
                                       as Kaldi's TrainingGraphCompiler::CompileGraph runs them (graph.py determinize_star_log,
                                       minimize_encoded) */
} mfa_gc_config;

MFA_GC_API mfa_gc *mfa_gc_create(const mfa_gc_config *cfg);
MFA_GC_API void mfa_gc_destroy(mfa_gc *gc);
MFA_GC_API const char *mfa_gc_last_error(const mfa_gc *gc);

/* HMM of a context window (graph.py TrainingGraphCompiler._hmm): its non-self-loop transitions (hs, dst, tid) and the
 * index of the final HMM state.  windows: [n][context_width]; trans: [trans_off[n]][3]. */
MFA_GC_API int mfa_gc_add_windows(mfa_gc *gc, int32_t n, const int32_t *windows, const int32_t *trans_off, const int32_t *trans,
                       const int32_t *n_final);

/* Phase 1: phone graphs and context expansion of a batch (entries: lexicon entry per transcript word, ragged by
 * word_off [n_utt + 1]); returns the number of context windows not yet registered (>= 0) or < 0 on error. */
MFA_GC_API int64_t mfa_gc_prepare(mfa_gc *gc, int32_t n_utt, const int64_t *word_off, const int32_t *entries, int32_t n_threads);
MFA_GC_API int mfa_gc_missing_windows(mfa_gc *gc, int32_t *windows /* [missing][context_width] */);

/* The context-dependency tree (Kaldi ContextDependency / EventMap), the HMM topology and the
 * transition-state table, flattened, so that the HMM of a context window is worked out HERE instead of by a callback into
 * the host for every window not seen before (a 5k-leaf triphone model meets ~10^5 distinct windows in a batch of 4 096
 * transcripts).  Optional: without it every window goes through mfa_gc_missing_windows / mfa_gc_add_windows.
 * Tree nodes: kind 0 = CE (answer), 1 = TE (key, children table[a .. a + b), -1 = NULL), 2 = SE (key, yes-set
 * yes_vals[yes_off[i] .. yes_off[i + 1]) sorted ascending, child a when the value is in it, else child b); event keys 0 ..
 * width - 1 are the window's phones, key -1 the pdf-class.
 * Topology: phone2entry[phone] (-1 = none) for phone <= max_phone; entry e has HMM states entry_state_off[e] ..
 * entry_state_off[e + 1]; state s: forward / self-loop pdf-class, transitions trans_off[s] .. trans_off[s + 1] with
 * destinations trans_dst (relative to the entry's first state).  tuples [n_tuples][4] = (phone, hmm state, forward pdf,
 * self-loop pdf) of transition-state 1 .. n_tuples; state2id [n_tuples + 2] first transition-id of a transition-state. */
typedef struct {
  int32_t n_nodes, root;
  const int32_t *kind, *key, *answer, *a, *b, *yes_off;
  const int32_t *table, *yes_vals;
  int32_t max_phone;
  const int32_t *phone2entry;
  int32_t n_entries;
  const int32_t *entry_state_off;
  const int32_t *fwd_class, *slf_class, *trans_off, *trans_dst;
  int32_t n_tuples;
  const int32_t *tuples;
  const int32_t *state2id;
} mfa_gc_model;
MFA_GC_API int mfa_gc_set_model(mfa_gc *gc, const mfa_gc_model *model);
/* After mfa_gc_prepare: works out the HMMs of the missing windows from the model given to mfa_gc_set_model; returns how many
 * are still missing (those the tree or the tuple table has no answer for — the host's code raises its error for them). */
MFA_GC_API int64_t mfa_gc_resolve_windows(mfa_gc *gc);

/* Phase 2: HMM expansion.  neg_scaled_log_probs (float32 [n_tids + 1], may be NULL): Kaldi AddTransitionProbs applied
 * to the result (weight ← weight + table[tid] in float32 for tid > 0), as graph.add_transition_probs does.
 * Totals of the batch go to n_states / n_arcs. */
MFA_GC_API int mfa_gc_finish(mfa_gc *gc, const float *neg_scaled_log_probs, int32_t n_threads, int64_t *n_states, int64_t *n_arcs);

/* Copy the batch out: state_off / arc_base [n_utt + 1] (prefix sums), arc_off [n_states + n_utt] (per utterance S + 1
 * offsets relative to its first arc), arcs as {ilabel, olabel, weight, nextstate} records of 16 bytes, final [n_states]
 * (+inf = not final).  Start state of every graph is 0. */
MFA_GC_API int mfa_gc_fetch(mfa_gc *gc, int64_t *state_off, int64_t *arc_base, int64_t *arc_off, void *arcs, float *final_w);
/* The same copy by n_threads host threads, plus the columns the score-plan builder (mfa_build_score_plans_batch) and the
 * device layout read, so that no pass over the arcs is left to the host language: arc_off32 [n_states + n_utt] (int32 copy
 * of arc_off), arc_next [n_arcs] (next state), arc_pdf [n_arcs] = id2pdf[ilabel] (id2pdf [n_tids + 1], entry 0 unused; NULL:
 * arc_pdf is not written).  Any output pointer may be NULL.  The batch stays fetchable until the next mfa_gc_prepare. */
MFA_GC_API int mfa_gc_fetch_columns(mfa_gc *gc, const int32_t *id2pdf, int32_t n_threads, int64_t *state_off, int64_t *arc_base,
                                    int64_t *arc_off, int32_t *arc_off32, void *arcs, float *final_w, int32_t *arc_next,
                                    int32_t *arc_pdf, int32_t *stats /* [3] or NULL: largest out-degree of a state, smallest input
                                    label, fewest arcs of an utterance — what decides whether the batch can go to the device as it is */);

#ifdef __cplusplus
}
#endif
#endif
