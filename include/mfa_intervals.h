/* Host-side interval extraction and TextGrid text (C ABI, no GPU, no torch), batched and multi-threaded: the step AFTER the
 * device path.  Replaces, for whole batches of alignments held as arrays, what the reference does per utterance in Python
 * over kalpy objects:
 *   Alignment.generate_ctm(transition_model, phone_table, frame_shift)          = Kaldi SplitToPhones (reordered)
 *   LexiconCompiler.phones_to_pronunciations(words, intervals, text=…)          → HierarchicalCtm
 *   HierarchicalCtm.update_utterance_boundaries(begin, end); fix_unk_words
 *       (AlignmentExtractionFunction, MFA/alignment/multiprocessing.py:1733-1751; MFA/helper.py:772-833)
 *   export_textgrid / Textgrid.save / CtmInterval.to_tg_interval
 *       (MFA/textgrid.py:463-572, :50-161, :115-131; MFA/data.py:2062-2080)
 * montreal_forced_aligner_amd/ctm.py is the specification (one object per interval, in Python); this library produces the
 * same intervals and the same file bytes from the arrays the device hands back (tests/test_intervals_native_cpu.py).
 * Anything irregular — an alignment that is not a sequence of complete phones, phones no combination of the words'
 * pronunciations spells, an interval that collapses after rounding — is reported per utterance / per file with a code and
 * left to the caller (who runs the Python specification on it and gets its exception).
 *
 * All functions return 0 on success, a negative value on error (mfa_iv_last_error).  Buffers belong to the caller.
 */
#ifndef MFA_INTERVALS_H
#define MFA_INTERVALS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_IV_API __attribute__((visibility("default")))

typedef struct mfa_iv mfa_iv;

typedef struct {
  /* transition model (model.TransitionModel tables), index = transition-id, entry 0 unused */
  int32_t n_tids;
  const int32_t *id2state;      /* [n_tids + 1] transition-state */
  const int32_t *id2phone;      /* [n_tids + 1] phone id */
  const int32_t *is_self_loop;  /* [n_tids + 1] */
  const int32_t *is_final;      /* [n_tids + 1] */
  /* lexicon: for word id w the pronunciation variants word_var_off[w] .. word_var_off[w + 1], longest first, duplicates
   * removed (the order ctm.phones_to_pronunciations tries them in); a variant = phone ids AS THEY APPEAR IN ALIGNMENTS
   * (word-position suffixes applied), -1 for a label the phone table does not hold. */
  int32_t n_words;
  const int32_t *word_var_off;  /* [n_words + 1] */
  const int32_t *var_off;       /* [n_variants + 1] into var_phones */
  const int32_t *var_phones;
  int32_t sil_phone;            /* id of the optional-silence phone (bare, no position suffix) */
  int32_t sil_word;             /* word id of the silence word ("<eps>") */
  int32_t oov_word;             /* word id of the out-of-vocabulary word ("<unk>") */
  double frame_shift;           /* seconds per frame (0.01) */
  /* names, UTF-8, concatenated: name i = bytes [off[i], off[i + 1]) */
  int32_t n_phone_names;        /* phone ids 0 .. n_phone_names - 1 */
  const int64_t *phone_name_off;
  const char *phone_names;
  const int64_t *word_name_off; /* [n_words + 1] */
  const char *word_names;
} mfa_iv_config;

MFA_IV_API mfa_iv *mfa_iv_create(const mfa_iv_config *cfg);
MFA_IV_API void mfa_iv_destroy(mfa_iv *iv);
MFA_IV_API const char *mfa_iv_last_error(const mfa_iv *iv);
MFA_IV_API int mfa_iv_version(void);

/* Per-utterance codes in h_err */
#define MFA_IV_OK 0
#define MFA_IV_SKIPPED 1          /* status said the utterance has no alignment */
#define MFA_IV_IRREGULAR 2        /* not a sequence of complete phones (SplitToPhones would report !ok) */
#define MFA_IV_UNSPELLABLE 3      /* the aligned phones cannot be spelt by the aligned words' pronunciations */

/* Alignments of a batch → phone intervals and word items, arrays in, arrays out.
 *   frame_off [n_utt + 1]; ali [total frames] transition-ids; words [total frames]: utterance u's word ids packed at
 *   frame_off[u], n_words[u] of them (the layout mfa_align_*_batch writes); status [n_utt] or NULL (only 0 / 1 are processed).
 * Two calls, because the result is a small fraction of the frames and its size is only known afterwards:
 *   mfa_iv_extract_batch does the work (utterances over n_threads threads; <= 0: hardware concurrency), keeps the result
 *   inside the handle and returns the prefix sums ph_off / it_off [n_utt + 1] and the per-utterance codes h_err [n_utt];
 *   mfa_iv_fetch copies it into caller arrays of exactly ph_off[n_utt] / it_off[n_utt] entries and releases it.
 * Utterance u's phone intervals are entries ph_off[u] .. ph_off[u + 1] of ph_first (frame), ph_len (frames), ph_id (phone id);
 * its word items, in time order, inter-word silence included as items of the silence word, entries it_off[u] .. it_off[u + 1]:
 *   it_word (word id), it_var (index of the matching variant among the word's variants, -1 for a silence item),
 *   it_first / it_count (its phone intervals, indices relative to the utterance's first), it_ref (position among the
 *   utterance's non-silence items = position in the transcript, -1 for a silence item). */
MFA_IV_API int mfa_iv_extract_batch(mfa_iv *iv, int32_t n_utt, const int64_t *frame_off, const int32_t *ali, const int32_t *words,
                                    const int32_t *n_words, const int32_t *status, int32_t n_threads, int64_t *ph_off,
                                    int64_t *it_off, int32_t *h_err);
MFA_IV_API int mfa_iv_fetch(mfa_iv *iv, int32_t *ph_first, int32_t *ph_len, int32_t *ph_id, int32_t *it_word, int32_t *it_var,
                            int32_t *it_first, int32_t *it_count, int32_t *it_ref);

/* The text of one output file per sound file from those arrays (export_textgrid, MFA/textgrid.py:463-572).
 * format: 0 long TextGrid, 1 short TextGrid, 2 json, 3 csv.
 * Files: file f has the tiers of speakers file_spk_off[f] .. file_spk_off[f + 1]; speaker entry s (one per (file, speaker)
 * pair, in first-appearance order) has the name spk_names[spk_name_off[s] .. spk_name_off[s + 1]) and the utterances
 * spk_utt[spk_utt_off[s] .. spk_utt_off[s + 1]) (utterance indices of the extract_batch call, in corpus order).
 * utt_begin [n_utt]: the utterance's begin inside its file (added to every boundary when non-zero); utt_end [n_utt]: its
 * end (the last phone interval is clipped to it) — HierarchicalCtm.update_utterance_boundaries.
 * Out-of-vocabulary items take their transcript spelling (fix_unk_words): relabel k says item with it_ref == relabel_ref[k] of
 * utterance relabel_utt[k] is written as relabel_text[relabel_off[k] .. relabel_off[k + 1]); entries sorted by
 * (utterance, ref).
 * cleanup_silence != 0: items of the silence word (and their phones) are left out, as CorpusAligner.export_textgrids does.
 * Output: the files' bytes back to back in out (capacity out_cap), file f at [out_off[f], out_off[f + 1]); file_err[f] = 0, or
 * 1 when the file needs the Python writer (an interval that is empty after rounding: the reference raises there), 2 when the file
 * has no data (the reference writes nothing).  Returns 0, or -2 with *needed set when out_cap is too small. */
MFA_IV_API int mfa_iv_write_files(mfa_iv *iv, int32_t format, int32_t cleanup_silence, int32_t n_files, const double *file_duration,
                                  const int32_t *file_spk_off, const int64_t *spk_name_off, const char *spk_names,
                                  const int32_t *spk_utt_off, const int32_t *spk_utt, const double *utt_begin,
                                  const double *utt_end, const int64_t *ph_off, const int32_t *ph_first, const int32_t *ph_len,
                                  const int32_t *ph_id, const int64_t *it_off, const int32_t *it_word, const int32_t *it_first,
                                  const int32_t *it_count, const int32_t *it_ref, const int32_t *h_err,
                                  int32_t n_relabel, const int32_t *relabel_utt, const int32_t *relabel_ref,
                                  const int64_t *relabel_off, const char *relabel_text, int32_t n_threads, char *out,
                                  int64_t out_cap, int64_t *out_off, int32_t *file_err, int64_t *needed);

/* Those bytes to disk, one file per entry, by n_threads host threads: file f = out[out_off[f] .. out_off[f + 1]) written to
 * the path paths[path_off[f] .. path_off[f + 1]) (UTF-8); entries with file_err[f] != 0 are skipped.  io_err[f] = errno of a
 * failed open / write / close, 0 otherwise.  Returns the number of files that failed.  (ExportTextGridProcessWorker,
 * MFA/alignment/multiprocessing.py:1865-1960, writes its files from worker processes for the same reason.) */
MFA_IV_API int mfa_iv_save_files(int32_t n_files, const int64_t *path_off, const char *paths, const char *out, const int64_t *out_off,
                                 const int32_t *file_err, int32_t n_threads, int32_t *io_err);

#ifdef __cplusplus
}
#endif
#endif
