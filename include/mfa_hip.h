/* mfa_hip.h — C ABI of libmfa_hip.so: the MI355X (gfx950) alignment hot path.
 *
 * Drop-in boundary: the reference (Cathoven-AI/Montreal-Forced-Aligner) reaches all arithmetic on
 * this path through the kalpy Python objects; each entry point below names the kalpy call it stands behind.  The
 * Python host module (montreal_forced_aligner_amd/) mirrors those objects and binds these symbols with ctypes —
 * INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *  - Every function returns 0 on success, <0 on error; mfa_last_error(ctx) gives the message.  No exceptions cross
 *    the ABI.  A per-utterance alignment failure is NOT an error: it is reported in status[] (the reference counts
 *    failures and raises only if none succeed — MFA/alignment/mixins.py:305-324).
 *  - Pointers named d_* are DEVICE pointers owned by the caller (torch tensors or mfa_device_alloc); h_* are host
 *    pointers.  The library never frees caller memory.  All work is enqueued on the ctx stream (mfa_set_stream);
 *    nothing synchronises unless the name says so.
 *  - Ragged batches use CSR offsets: frame_off[u]..frame_off[u+1] are the rows of utterance u.
 *  - One ctx per (process, GPU); calls on a ctx are serialised by the caller.
 */
#ifndef MFA_HIP_H_
#define MFA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_API __attribute__((visibility("default")))

typedef struct mfa_ctx mfa_ctx;

/* ---- context ------------------------------------------------------------------------------------------------ */
MFA_API mfa_ctx *mfa_create(int device_id);
MFA_API void mfa_destroy(mfa_ctx *ctx);
MFA_API const char *mfa_last_error(mfa_ctx *ctx);
MFA_API int mfa_version(void);
/* Enqueue on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's default stream, which
 * is what torch uses unless told otherwise).  use_own != 0 switches back to the ctx's private non-blocking stream. */
MFA_API int mfa_set_stream(mfa_ctx *ctx, void *hip_stream, int use_own);
MFA_API int mfa_synchronize(mfa_ctx *ctx);
/* Device-memory helpers for callers without torch. */
MFA_API void *mfa_device_alloc(mfa_ctx *ctx, size_t bytes);
MFA_API int mfa_device_free(mfa_ctx *ctx, void *d_ptr);
MFA_API int mfa_memcpy_h2d(mfa_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
MFA_API int mfa_memcpy_d2h(mfa_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* HIP-event timing of everything enqueued between begin and end on the ctx stream (bench.py's roofline leg). */
MFA_API int mfa_timer_begin(mfa_ctx *ctx);
MFA_API int mfa_timer_end_ms(mfa_ctx *ctx, float *h_ms);
/* Per-kernel accumulated HIP-event times since the last reset.  which: 0 mfcc, 1 cmvn, 2 feats, 3 gmm, 4 viterbi.
 * Enabled with mfa_kernel_timing(ctx, 1) (adds an event pair around each launch). */
MFA_API int mfa_kernel_timing(mfa_ctx *ctx, int enable);
MFA_API int mfa_kernel_time_ms(mfa_ctx *ctx, int which, float *h_ms, int *h_launches);
MFA_API int mfa_kernel_time_reset(mfa_ctx *ctx);

/* ---- MFCC: replaces kalpy.feat.mfcc.MfccComputer(**mfcc_options).compute_mfccs[_for_export]
 *      (MFA/corpus/features.py:193-251, :235; MFA/online/alignment.py:83; options MFA/corpus/features.py:780-820). */
typedef struct {
  float sample_frequency;   /* 16000 */
  float frame_length_ms;    /* 25 */
  float frame_shift_ms;     /* 10 */
  float preemphasis;        /* 0.97 */
  float low_frequency;      /* 20 */
  float high_frequency;     /* 7800 (<=0: relative to Nyquist) */
  float cepstral_lifter;    /* 22 */
  float energy_floor;       /* 0 */
  int32_t num_mel_bins;     /* 23 */
  int32_t num_coefficients; /* 13 */
  int32_t snip_edges;       /* MFA default 0 */
  int32_t remove_dc_offset; /* 1 */
  int32_t use_energy;       /* 0; 1: C0 := the frame's log energy (floored at log(energy_floor) when that is > 0) */
  int32_t raw_energy;       /* 1: energy before pre-emphasis and window; 0: after */
} mfa_mfcc_opts;

MFA_API int mfa_mfcc_configure(mfa_ctx *ctx, const mfa_mfcc_opts *opts);
/* Frames Kaldi extracts from num_samples samples under the configured options (host arithmetic). */
MFA_API int32_t mfa_mfcc_num_frames(mfa_ctx *ctx, int64_t num_samples);
/* d_pcm: int16 samples of all utterances back to back; d_sample_off[n_utt+1]; d_frame_off[n_utt+1] (host computes
 * them with mfa_mfcc_num_frames); d_mfcc: float32 [total_frames][num_coefficients].  max_frames = longest utterance. */
MFA_API int mfa_mfcc_batch(mfa_ctx *ctx, const int16_t *d_pcm, const int64_t *d_sample_off, const int64_t *d_frame_off,
                           int32_t n_utt, int32_t max_frames, float *d_mfcc);

/* Host helper for the call above: utterance u's samples h_src[u][0 .. h_sample_off[u+1] - h_sample_off[u]) copied to
 * h_dst + h_sample_off[u] by n_threads host threads (<= 0: hardware concurrency).  h_dst is the caller's staging buffer
 * (pinned memory, followed by ONE asynchronous host-to-device copy): the reference hands kalpy one Segment per utterance
 * (MFA/corpus/features.py:223-235); a batch of 4 096 ten-second utterances is 1.3 GB of samples. */
MFA_API int mfa_gather_pcm(int32_t n_utt, const int16_t *const *h_src, const int64_t *h_sample_off, int16_t *h_dst,
                           int32_t n_threads);

/* ---- CMVN statistics: replaces CmvnComputer().compute_cmvn_from_features / export_cmvn
 *      (MFA/corpus/acoustic_corpus.py:1315-1367; MFA/online/alignment.py:86-88).
 * d_spk_utt_off[n_spk+1] / d_spk_utt[…]: utterances of each speaker; d_stats: float64 [n_spk][2][dim+1]
 * (row 0: sums + count, row 1: sums of squares — Kaldi's layout).  Deterministic summation order. */
MFA_API int mfa_cmvn_stats(mfa_ctx *ctx, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt, int32_t dim,
                           const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_stats);

/* ---- Final features: replaces FeatureArchive(..., cmvn, deltas | lda_mat, transform) iteration / the explicit chain at
 *      MFA/alignment/multiprocessing.py:1287-1304: ApplyCmvn → compute_deltas | splice_frames+LDA → fMLLR.
 * mode 0: CMVN + Δ+ΔΔ (order 2, window 2): out dim = 3*dim.
 * mode 1: CMVN + splice(±ctx) + LDA d_lda[lda_rows][lda_cols] (+ fMLLR d_fmllr[n_spk][lda_rows][lda_rows+1] if non-NULL).
 * d_utt2spk[n_utt]; d_cmvn may be NULL (no CMVN). */
MFA_API int mfa_feats_batch(mfa_ctx *ctx, const float *d_mfcc, const int64_t *d_frame_off, int32_t n_utt, int32_t max_frames,
                            int32_t dim, const int32_t *d_utt2spk, const double *d_cmvn, int32_t mode, int32_t splice_ctx,
                            const float *d_lda, int32_t lda_rows, int32_t lda_cols, const float *d_fmllr, float *d_out);

/* ---- Acoustic model: replaces GmmAligner.__init__'s model load (+ .boost_silence, applied by the host to gconsts)
 *      (MFA/alignment/multiprocessing.py:814-815).  Host arrays, SoA over all Gaussians:
 * gconsts[G], means_invvars[G][dim], inv_vars[G][dim], pdf_offsets[num_pdfs+1].  Packs them for the MFMA kernel. */
MFA_API int mfa_load_gmm(mfa_ctx *ctx, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets, const float *h_gconsts,
                         const float *h_means_invvars, const float *h_inv_vars);
/* Slot class of a pdf in the packed model (rows it occupies in an MFMA block: 1, 4, 8, 16 or 32); the host must order
 * each utterance's pdf list by descending slot class (mfa_gmm_sort_pdf_list does it). */
MFA_API int32_t mfa_gmm_slot(mfa_ctx *ctx, int32_t pdf);
/* Sort h_pdfs[n] in place into the order the scoring kernels require; h_class_counts[6] receives how many pdfs fall in
 * the classes {32 rows single block, 32 rows multi-block (>32 Gaussians), 16, 8, 4, 1}. */
MFA_API int mfa_gmm_sort_pdf_list(mfa_ctx *ctx, int32_t *h_pdfs, int32_t n, int32_t *h_class_counts);
/* Same, with a per-pdf key (h_first_frame[n], permuted along): inside each class the pdfs are ordered by ascending key.
 * With key = first frame at which the decoder can ask for the pdf (mfa_fst_first_frames + a min over the arcs that emit
 * it), the pdfs a given frame range needs form a PREFIX of every class, which is what mfa_gmm_score_batch's
 * d_pdf_first_frame argument relies on. */
MFA_API int mfa_gmm_sort_pdf_list_keyed(mfa_ctx *ctx, int32_t *h_pdfs, int32_t *h_first_frame, int32_t n,
                                        int32_t *h_class_counts);
/* Host helper: h_depth[s] = the smallest number of arcs on a path from `start` to state s of an epsilon-free graph in
 * CSR form (arc_off[n_states+1], arc_next[n_arcs]); INT32_MAX for unreachable states.  A decoder token can sit on s at
 * frame t only if h_depth[s] <= t (FasterDecoder consumes one frame per arc; kaldi decoder/faster-decoder.cc
 * ProcessEmitting), so an arc leaving s is never scored before frame h_depth[s]. */
MFA_API int mfa_fst_first_frames(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next, int32_t start,
                                 int32_t *h_depth);

/* ---- Acoustic scoring: replaces DecodableAmDiagGmmScaled::LogLikelihood inside GmmAligner.align_utterance and
 *      gmm_compute_likes (MFA/alignment/multiprocessing.py:846-853, :1415).
 * Per utterance u: pdf list d_pdf_list[pdf_off[u]..pdf_off[u+1]) (sorted as above) with d_class_counts[u][6];
 * output d_loglikes + ll_off[u]: float32 [T_u][P_u] row-major (UNSCALED log-likelihoods). */
/* d_pdf_first_frame (may be NULL): int32 parallel to d_pdf_list, ascending inside every class of every utterance
 * (mfa_gmm_sort_pdf_list_keyed).  When given, cell (t, j) is only guaranteed to be written if
 * d_pdf_first_frame[j] <= t rounded up to the end of its 64-frame tile: Kaldi's decodable is evaluated lazily, for the
 * arcs leaving live tokens only, and no token can ask for pdf j before that frame — the skipped cells are never read by
 * mfa_align_batch.  NULL scores every cell (what gmm_compute_likes-style callers want). */
/* Numerics: single-Gaussian pdfs are scored on the float32 matrix pipe as one k-ordered fmaf chain (bit-identical to the
 * oracle's chain, bit-exact end to end).  Pdfs of 2 or more Gaussians go, by default, through the 16-bit matrix pipe with
 * every float32 product formed from split operands (accumulation order differs from the chain):
 *   - two f16 pieces per operand, power-of-two column scales fixed at mfa_load_gmm (3 * 2^-22 per term worst case;
 *     measured <= 1e-6 x max|score of the frame|); a 256-frame tile whose scaled features leave the f16 range is scored
 *     with three bf16 pieces instead.  MFA_GMM_F16=0: bf16 for all.
 *   - pdfs of more than 32 Gaussians are runs of 32-row blocks merged with an online log-sum-exp (same operand splits).
 * Both against a tolerance of 1e-3 on log-likelihoods.  Environment MFA_GMM_BF16=0 keeps every pdf on the float32 pipe. */
MFA_API int mfa_gmm_score_batch(mfa_ctx *ctx, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                int32_t max_frames, const int32_t *d_pdf_list, const int64_t *d_pdf_off,
                                const int32_t *d_class_counts, const int32_t *d_pdf_first_frame, const int64_t *d_ll_off,
                                float *d_loglikes);

/* Debug/profiling aid: when d_trace is non-NULL the scoring kernel records, for every (utterance u, 64-frame sub-tile r)
 * it scores, four uint64 words {start, end (100 MHz wall clock), hardware id (HW_ID | XCC_ID << 32), 32-row blocks
 * walked} at d_trace[((u * tiles) * 4 + r) * 4], tiles = ceil(max_frames / 256).  NULL turns it off. */
MFA_API int mfa_debug_gmm_trace(mfa_ctx *ctx, void *d_trace);
/* Profiling aid, effective only in a library built with -DVIT_STAMPS (tools/viterbi_phases.py): the first-beam decoder
 * launch leaves, per utterance, twelve uint64 shader-clock totals (one per phase of its frame loop) at d_stamps[u * 12]. */
MFA_API int mfa_debug_viterbi_stamps(mfa_ctx *ctx, void *d_stamps);

/* ---- Alignment: replaces GmmAligner.align_utterance(fst, feats) / .export_alignments
 *      (MFA/alignment/multiprocessing.py:846-853, :1311-1315; MFA/online/alignment.py:107) = Kaldi AddTransitionProbs
 *      (done by the host on the arc weights) + AlignUtteranceWrapper + FasterDecoder (beam, min_active 20,
 *      beam_delta 0.5, hash_ratio 2.0) with exactly Kaldi's pruning and tie-breaking order.
 * Graph u (epsilon-free, or with d_state_nemit: see mfa_graph_batch): states state_off[u]..state_off[u+1]; d_arc_off[global_state] .. [global_state+1] index the
 * arc arrays RELATIVE to arc_base[u]; start state d_start[u]; d_final[global_state] (+inf = non-final).
 * Arcs (SoA): d_arc_next (local state), d_arc_weight (graph cost incl. transition probs), d_arc_col (column of the
 * utterance's log-likelihood matrix = position of pdf(tid) in its pdf list), d_arc_ilabel (transition-id),
 * d_arc_olabel (word id). */
typedef struct {
  int32_t n_utt;
  const int64_t *d_state_off; /* [n_utt+1] */
  const int64_t *d_arc_base;  /* [n_utt+1] */
  const int32_t *d_start;     /* [n_utt] */
  const int32_t *d_arc_off;   /* [total_states + n_utt] : per utterance S_u+1 entries, at state_off[u]+u */
  const float *d_final;       /* [total_states] */
  const int32_t *d_arc_next;
  const float *d_arc_weight;
  const int32_t *d_arc_col;
  const int32_t *d_arc_ilabel;
  const int32_t *d_arc_olabel;
  /* Graphs with EPSILON INPUT ARCS (ilabel 0; training graphs compiled by kalpy / Kaldi can hold them) on the wavefront-
   * parallel decoder: NULL for epsilon-free batches; otherwise [total_states] = emitting arcs of every state, whose arcs
   * must then be stored [emitting arcs | epsilon arcs], each kind in its original relative order (FasterDecoder's
   * ProcessEmitting walks the emitting arcs of a state in order and skips the others, ProcessNonemitting the other way round:
   * the interleaving never matters).  d_arc_col of an epsilon arc is ignored.  Requirement (the host checks it and routes
   * anything else to mfa_align_general_batch): at most 64 emitting and at most 64 epsilon arcs per state.  The closure pops
   * its stack one token at a time exactly as Kaldi's does, so neither ties nor negative epsilon weights change its result. */
  const int32_t *d_state_nemit;
} mfa_graph_batch;

typedef struct {
  float beam;            /* 10 */
  float retry_beam;      /* 40; 0 = no retry */
  float acoustic_scale;  /* 0.1 */
  int32_t max_tokens;    /* live-token capacity per utterance (LDS-resident tables), e.g. 1024 */
  int32_t bp_tokens_per_frame; /* back-pointer capacity per utterance = T_u * this, e.g. 512 */
} mfa_align_opts;

/* d_utt_list: which utterances to decode (NULL = all n_utt); outputs (device):
 *   d_ali [total_frames] transition-ids (at frame_off), d_words [total_frames] word ids packed at frame_off[u] with
 *   d_n_words[u] valid entries, d_like[u] = -(graph+acoustic cost)/acoustic_scale, d_frame_like [total_frames] or NULL,
 *   d_status[u]: 0 ok, 1 ok after retry, 2 no final token (failed), 3 token-capacity overflow, 4 back-pointer overflow,
 *                5 unsupported graph (a state with more than 64 arcs), 6 internal consistency check failed,
 *                7 the best path carries more word labels than the utterance has frames — possible only when epsilon input arcs
 *                  carry output labels; d_words holds one entry per frame, so the word sequence cannot be returned (the utterance's
 *                  other outputs are undefined).  Graphs compiled from a lexicon never do this: every word consumes a frame.
 * total_frames = frame_off[n_utt], total_arcs = arc_base[n_utt] (host copies, so the call never synchronises); max_states / max_arcs = largest S_u / A_u
 * in the batch (they bound the token and candidate tables). */
MFA_API int mfa_align_batch(mfa_ctx *ctx, const mfa_graph_batch *graphs, const float *d_loglikes, const int64_t *d_ll_off,
                            const int32_t *d_ll_cols, const int64_t *d_frame_off, int64_t total_frames,
                            int64_t total_arcs, int32_t max_states, int32_t max_arcs, const mfa_align_opts *opts, int32_t *d_ali, int32_t *d_words, int32_t *d_n_words,
                            float *d_like, float *d_frame_like, int32_t *d_status);
/* ---- The same alignment for graphs the wavefront-parallel decoder does not take: EPSILON INPUT ARCS (ilabel 0; training
 * graphs compiled by kalpy / Kaldi can hold them — FstArchive handed to export_alignments,
 * MFA/alignment/multiprocessing.py:846-853) and states with more than 64 arcs.  FasterDecoder exactly as Kaldi runs it,
 * ProcessNonemitting included (LIFO queue, hash-list insertion order), one GPU thread per utterance with every structure in a
 * per-utterance HBM workspace: a correctness path (tens of milliseconds per 10 s utterance, thousands in parallel), results
 * bit-identical to the oracle's.  Graph layout as for mfa_align_batch; d_arc_col of an epsilon arc is ignored.
 * h_frame_off: host copy of d_frame_off (workspace sizing without a synchronisation on device data). */
MFA_API int mfa_align_general_batch(mfa_ctx *ctx, const mfa_graph_batch *graphs, const float *d_loglikes,
                                    const int64_t *d_ll_off, const int32_t *d_ll_cols, const int64_t *d_frame_off,
                                    const int64_t *h_frame_off, int32_t max_states, int32_t max_arcs,
                                    const mfa_align_opts *opts, int32_t *d_ali, int32_t *d_words, int32_t *d_n_words,
                                    float *d_like, float *d_frame_like, int32_t *d_status);
/* Bytes of device workspace mfa_align_batch will hold for a batch shape (so callers can budget HBM). */
MFA_API size_t mfa_align_workspace_bytes(mfa_ctx *ctx, int32_t n_utt, int64_t total_frames, const mfa_align_opts *opts);

/* ---- Alignment from features, acoustic scores evaluated LAZILY: the same replacement as mfa_align_batch, but it takes the
 *      features GmmAligner.align_utterance(fst, feats) takes (MFA/alignment/multiprocessing.py:846-853;
 *      MFA/online/alignment.py:107) and scores only what Kaldi's lazy decodable would be asked for — a superset of it:
 *      the utterance is decoded in windows of `window` frames; before a window is decoded, the pdfs that arcs within
 *      `window` arcs of the live tokens can emit are scored for the window's frames (split-operand MFMA kernels as in
 *      mfa_gmm_score_batch); everything else in the score matrix is left untouched and never read.  Results are those of
 *      mfa_gmm_score_batch + mfa_align_batch bit for bit (same kernels' arithmetic per cell, same decoder).
 * plan: the utterances' pdf lists as for mfa_gmm_score_batch plus two depth keys per pdf and per graph state
 *   d_pdf_first_frame[j] = smallest BFS depth of a source state of the column's arcs
 *   d_pdf_last_depth[j]  = running max, inside the column's class in list order, of the largest such depth
 *   d_state_depth[global state][2] = {BFS depth, smallest BFS depth reachable from the state}
 * — all three come out of mfa_build_score_plan (one call per utterance).
 * d_loglikes [sum T_u * P_u] is caller-provided scratch for the scores (d_ll_off, d_ll_cols as for mfa_align_batch). */
typedef struct {
  const int32_t *d_pdf_list;
  const int64_t *d_pdf_off;
  const int32_t *d_class_counts;
  const int32_t *d_pdf_first_frame;
  const int32_t *d_pdf_last_depth;
  const int32_t *d_state_depth;
  int32_t max_cols;                 /* largest column count of an utterance of the batch (host value) */
  int32_t groups;                   /* 0 or 1: class 0 in one run ordered by first depth; G > 1 (mfa_build_score_plan_grouped):
                                       G runs (pdf id mod G), each ordered by first depth */
  const int32_t *d_group_counts;    /* [n_utt][groups] class-0 columns per run (NULL when groups <= 1) */
} mfa_score_plan;

MFA_API int mfa_align_features_batch(mfa_ctx *ctx, const mfa_graph_batch *graphs, const mfa_score_plan *plan,
                                     const float *d_feats, const int64_t *d_frame_off, int32_t max_frames,
                                     int64_t total_frames, int64_t total_arcs, int32_t max_states, int32_t max_arcs,
                                     const mfa_align_opts *opts, int32_t window, float *d_loglikes, const int64_t *d_ll_off,
                                     const int32_t *d_ll_cols, int32_t *d_ali, int32_t *d_words, int32_t *d_n_words,
                                     float *d_like, float *d_frame_like, int32_t *d_status);
/* Host helper: h_depth[s] = the smallest BFS depth (h_bfs_depth, from mfa_fst_first_frames) among the states reachable
 * from s, s included — computed on the graph's condensation, so self-loops and the small cycles of an ergodic silence
 * topology are fine.  It never decreases along an arc, and it is <= the BFS depth of everything reachable from s: a state
 * whose BFS depth is below h_depth[l] of every live token l can never be visited again.  Returns 0 for a graph that is
 * acyclic apart from self-loops, 1 if larger components were contracted, <0 if malformed. */
MFA_API int mfa_fst_last_depths(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next, int32_t start,
                                const int32_t *h_bfs_depth, int32_t *h_depth);
/* Host helper: the score columns of ONE utterance's graph, in the order the scoring kernels require.
 * In: the graph in CSR form with the pdf id of every arc (h_arc_pdf = pdf of the arc's transition-id) and the slot class
 * of every pdf of the model, h_pdf_class[num_pdfs] in 0..5 = {32 rows single block, 32 rows multi-block, 16, 8, 4, 1}
 * (from mfa_gmm_slot and the pdf's Gaussian count).  Returns 0, -1 for a malformed graph, -2 for a pdf id out of range.
 * A column is a pdf restricted to a cluster of its occurrences: arcs emitting the pdf whose source states' BFS depths lie
 * within cluster_span of the cluster's first (cluster_span <= 0: one column per pdf).  The same phone in two words thus gets
 * two columns, and the lazy-scoring band — a range of graph depths — need not keep it alive in between.
 * Out: h_state_depth[n_states][2] = {BFS depth, mfa_fst_last_depths value}; h_arc_col[n_arcs] = column of every arc;
 * per column (capacity n_arcs each): h_col_pdf, h_col_first = smallest BFS depth of a source, h_col_last = running max
 * (inside the column's slot class, in list order) of the largest BFS depth of a source; h_class_counts[6]; *h_n_cols.
 * Band rule (mfa_align_features_batch): with lo = min over live tokens of h_state_depth[.][1] and hi = max over live
 * tokens of h_state_depth[.][0] + window - 1, the columns a window can ask for are, in every class, the index range
 * [count(h_col_last < lo), count(h_col_first <= hi)). */
MFA_API int mfa_build_score_plan(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next,
                                 const int32_t *h_arc_pdf, int32_t start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                 int32_t cluster_span, int32_t *h_state_depth, int32_t *h_arc_col, int32_t *h_col_pdf,
                                 int32_t *h_col_first, int32_t *h_col_last, int32_t *h_class_counts, int32_t *h_n_cols);
/* The same with the class-0 columns (one 32-row model block per pdf — the bulk of a context-dependent model) laid out in
 * `groups` runs, run g holding the pdfs with id mod groups == g, each run in ascending first depth with its own running
 * max in h_col_last; h_group_counts[groups] = columns per run.  With groups = 8 the lazy scoring kernel gives run g of every
 * utterance to workgroups of one XCD, whose 4 MiB L2 then only ever sees an eighth of the model (MI355X: 8 XCDs; a
 * 5k-pdf model is 51 MB of operands, re-read from the Infinity Cache otherwise); with groups = 16 an XCD serves runs x and
 * x + 8 one after the other (first and second half of the launch).  The band rule holds per run.
 * groups = 1 is mfa_build_score_plan.  Returns -3 for groups outside 1..MFA_PLAN_MAX_GROUPS. */
#define MFA_PLAN_MAX_GROUPS 16
MFA_API int mfa_build_score_plan_grouped(int32_t n_states, const int32_t *h_arc_off, const int32_t *h_arc_next,
                                         const int32_t *h_arc_pdf, int32_t start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                         int32_t cluster_span, int32_t groups, int32_t *h_state_depth, int32_t *h_arc_col,
                                         int32_t *h_col_pdf, int32_t *h_col_first, int32_t *h_col_last, int32_t *h_class_counts,
                                         int32_t *h_group_counts, int32_t *h_n_cols);

/* The score plans of a whole batch, utterances spread over n_threads host threads (the per-utterance call above costs more
 * in its Python caller than in itself once graphs arrive at 20 k per second).  Graphs concatenated as the device layout
 * has them: h_state_off / h_arc_base [n_utt + 1] prefix sums; utterance u's arc offsets (S_u + 1 values, relative to its
 * first arc) at h_arc_off[h_state_off[u] + u]; h_arc_next / h_arc_pdf [total arcs]; h_start [n_utt].  Outputs as above,
 * concatenated: h_state_depth [total states][2], h_arc_col [total arcs]; utterance u's columns at h_col_*[h_arc_base[u] ..
 * + h_n_cols[u]) (capacity = its arcs); h_class_counts [n_utt][6]; h_group_counts [n_utt][groups] (groups > 1);
 * h_n_cols [n_utt].  Returns 0, or the first failing utterance's code with its index in *h_bad_utt. */
MFA_API int mfa_build_score_plans_batch(int32_t n_utt, const int64_t *h_state_off, const int64_t *h_arc_base,
                                        const int32_t *h_arc_off, const int32_t *h_arc_next, const int32_t *h_arc_pdf,
                                        const int32_t *h_start, int32_t num_pdfs, const int32_t *h_pdf_class,
                                        int32_t cluster_span, int32_t groups, int32_t n_threads, int32_t *h_state_depth,
                                        int32_t *h_arc_col, int32_t *h_col_pdf, int32_t *h_col_first, int32_t *h_col_last,
                                        int32_t *h_class_counts, int32_t *h_group_counts, int32_t *h_n_cols,
                                        int32_t *h_bad_utt);

/* ---- fMLLR statistics: replaces the accumulation of CalcFmllrFunction / kalpy FmllrComputer
 *      (MFA/corpus/features.py:506-527; Kaldi FmllrDiagGmmAccs) between the two alignment passes
 *      (MFA/alignment/base.py:510-539).  d_feats [total_frames][dim]: the features the transform will be applied to;
 * d_ali_pdf [total_frames]: pdf-id of the first-pass alignment per frame (<0 = skip); d_weight [total_frames]: frame
 * weights (0 for silence frames: silence_weight 0.0, MFA/corpus/features.py:759-766).  Outputs per speaker, float64,
 * deterministic: d_beta[n_spk], d_K[n_spk][dim][dim+1], d_G[n_spk][dim][dim+1][dim+1].  The per-speaker solve
 * (Kaldi ComputeFmllrMatrixDiagGmmFull) is host-side: montreal_forced_aligner_amd/fmllr.py. */
MFA_API int mfa_fmllr_acc_batch(mfa_ctx *ctx, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                int64_t total_frames, const int32_t *d_ali_pdf, const float *d_weight,
                                const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_beta,
                                double *d_K, double *d_G);

/* The same from the alignment itself: d_ali [total_frames] transition-ids (0 = no alignment for the frame), d_tid2pdf /
 * d_tid_weight [n_tids] host-built tables (pdf of a transition-id; frame weight by its phone: silence_weight for silence
 * phones, 1 otherwise); d_pdf_scratch / d_weight_scratch [total_frames] caller scratch. */
MFA_API int mfa_fmllr_acc_ali_batch(mfa_ctx *ctx, const float *d_feats, const int64_t *d_frame_off, int32_t n_utt,
                                    int64_t total_frames, const int32_t *d_ali, const int32_t *d_tid2pdf,
                                    const float *d_tid_weight, int32_t n_tids, int32_t *d_pdf_scratch, float *d_weight_scratch,
                                    const int32_t *d_spk_utt_off, const int32_t *d_spk_utt, int32_t n_spk, double *d_beta,
                                    double *d_K, double *d_G);
/* Two-model form of the same accumulation — what the reference runs whenever the acoustic model ships final.alimdl
 * (MFA/corpus/features.py:503-511: FmllrComputer(ali_model_path, model_path, ...); MFA/alignment/mixins.py:404-410):
 * the Gaussian posteriors come from the model loaded with mfa_load_gmm (the ALIGNMENT model, evaluated on the
 * speaker-independent features), the statistics are formed with the means and variances given here (the FINAL model;
 * Kaldi gmm-post-to-gpost + FmllrDiagGmmAccs::AccumulateFromPosteriors).  Both models must have the same number of
 * Gaussians in every pdf.  Stays in force until the next mfa_load_gmm, or until called with NULL arrays. */
MFA_API int mfa_fmllr_stats_model(mfa_ctx *ctx, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets,
                                  const float *h_means_invvars, const float *h_inv_vars);

#ifdef __cplusplus
}
#endif
#endif /* MFA_HIP_H_ */
