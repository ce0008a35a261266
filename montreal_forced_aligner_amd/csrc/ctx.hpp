// Internal context of libmfa_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mfa_hip.h"

// Packed model layout (built by mfa_load_gmm, read by gmm.hip and fmllr.hip).  Rows (Gaussians) are grouped in blocks of
// 32; a block is stored operand-major: for every group m of 8 k-values and half h, the 32 rows' 4-float pieces lie side
// by side —  float offset of (row, logical k = 8m + 2c + h):
//     (row >> 5) · 32·kpad  +  ((2m + h) · 32 + (row & 31)) · 4  +  c
// so the 16-byte A-operand loads of the 32 lanes of a half-wavefront (one row each) read 512 contiguous bytes.
__host__ __device__ inline size_t mfa_packed_offset(int row, int k, int kpad) {
  const int m = k >> 3, o = k & 7, h = o & 1, c = o >> 1;
  return (size_t)(row >> 5) * 32 * kpad + (size_t)(((2 * m + h) * 32 + (row & 31)) * 4 + c);
}

enum { MFA_K_MFCC = 0, MFA_K_CMVN = 1, MFA_K_FEATS = 2, MFA_K_GMM = 3, MFA_K_VITERBI = 4, MFA_K_COUNT = 5 };

struct mfa_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool kernel_timing = false;
  struct Pending { int which; hipEvent_t a, b; };
  std::vector<Pending> pending;
  double k_ms[MFA_K_COUNT] = {0, 0, 0, 0, 0};
  int k_n[MFA_K_COUNT] = {0, 0, 0, 0, 0};
  std::vector<hipEvent_t> event_pool;

  // MFCC tables (device)
  mfa_mfcc_opts mfcc{};
  bool mfcc_ready = false;
  int win = 0, shift = 0, nfft = 0;
  float *d_window = nullptr;   // [16][16][2] window pairs in lane order, see mfcc.hip
  float *d_twiddle = nullptr;  // [2][16][16][2] W256^(i k1), W512^(i + 16 k2) in lane order, see mfcc.hip
  float *d_melw = nullptr;     // [pieces][taps] filterbank piece weights, see mfcc.hip
  int32_t *d_melidx = nullptr;  // [96] first FFT bin per piece + [32] (first piece | pieces << 8) per mel bin
  int n_melw = 0;              // filterbank pieces
  int mel_np_max = 1;
  float *d_dct = nullptr;      // [nceps][nbins] with lifter folded separately
  float *d_lifter = nullptr;   // [nceps]

  // GMM model (device)
  bool gmm_ready = false;
  int dim = 0, kpad = 0, num_pdfs = 0, num_rows = 0;
  float *d_w = nullptr;        // [num_rows][kpad] permuted weights
  void *d_wb = nullptr;        // bf16×3 split of d_w for gmm_bf16_kernel (see gmm.hip), or NULL
  void *d_wh = nullptr;        // f16×2 split of the column-scaled d_w for gmm_split_single_kernel<…, 2>, or NULL
  float *d_gch = nullptr;      // gconsts × gmm_acc_scale
  float *d_fscale = nullptr;   // [kpad] feature column scales of the f16 path
  float gmm_acc_scale = 1.0f;  // S: the f16 path's accumulators are S × the log-likelihood terms (power of two)
  int *d_gmm_redo = nullptr;   // tiles the f16 pass handed to the bf16×3 pass
  int64_t gmm_redo_cap = 0;
  void *d_xsplit = nullptr;    // lazy scoring: f16 hi/lo operands of every 64-frame tile in register layout (gmm_presplit_kernel)
  int *d_xsplit_bad = nullptr; // … and the tile's "a scaled feature left the f16 range" flag
  int64_t xsplit_tiles = 0;
  int32_t *d_col_row0 = nullptr;   // lazy scoring: first packed model row of every score column of the batch (row0[pdf_list[j]])
  int64_t col_row0_cap = 0;
  int32_t *d_band_ranges = nullptr; // [n_utt][11][2]: per window, the band's index range in each run of class 0 and in classes 2..4
  int64_t band_ranges_cap = 0;
  bool xsplit_ready = false;   // d_xsplit holds the operands of the batch mfa_align_features_batch is working on
  float *d_gc = nullptr;       // [num_rows]
  int32_t *d_row0 = nullptr;   // [num_pdfs] first packed row
  int32_t *d_nblk = nullptr;   // [num_pdfs] number of 32-row blocks (slot 32) else 1
  int32_t *d_slot = nullptr;   // [num_pdfs] slot class rows (1,4,8,16,32)
  std::vector<int32_t> h_slot, h_nblk, h_row0, h_ngauss;
  float *d_w_stats = nullptr;  // packed rows of the fMLLR statistics model (two-model form; mfa_fmllr_stats_model) or NULL
  bool has_slot_class[5] = {false, false, false, false, false};   // model has pdfs of slot 32 / 16 / 8 / 4 / 1 rows
  bool has_single32 = false;       // some pdf is one 32-row block (17–32 Gaussians): gmm_split_single_kernel has work
  int max_nblk = 1;                // most 32-row blocks of any pdf
  bool has_multi_block = false;    // some pdf has more than 32 Gaussians (several blocks, merged by gmm_bf16_kernel<…, true>)
  bool all_single_block = false;   // every pdf occupies exactly one 32-row block (no work for the f32 kernel in bf16 mode)
  int32_t *d_nrows = nullptr;  // [num_pdfs] packed rows per pdf (fmllr.hip)
  int *d_gmm_queue = nullptr;  // [8] per-XCD work-item counters of the persistent scoring kernel
  int num_cus = 0;
  void *vit_stamps = nullptr;  // debug: per-utterance phase cycle counters of the decoder (-DVIT_STAMPS builds)
  void *gmm_trace = nullptr;   // debug: per-wavefront timeline records of the scoring kernel (mfa_debug_gmm_trace)

  // Viterbi workspace
  void *d_ws = nullptr;
  size_t ws_bytes = 0;
  void *d_gen_ws = nullptr;    // workspace of the general-graph decoder (viterbi_general.hip)
  size_t gen_ws_bytes = 0;
  int32_t *d_gen_list = nullptr;   // utterances of the general decoder's second tier (full token pool)
  size_t gen_list_cap = 0;

  int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return -1;
  }
};

#define MFA_HIP_CHECK(ctx, expr)                                                                 \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return (ctx)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// Debug aid (environment MFA_DEBUG_SYNC=1): name every launch on stderr, wait for it and report the first failing one.
#include <cstdlib>
inline bool mfa_debug_sync_enabled() {
  static const bool on = [] { const char *e = getenv("MFA_DEBUG_SYNC"); return e && e[0] == '1'; }();
  return on;
}
#define MFA_DEBUG_POINT(ctx, ...)                                                                         \
  do {                                                                                                    \
    if (mfa_debug_sync_enabled()) {                                                                       \
      fprintf(stderr, "[mfa] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr);                       \
      hipError_t _e = hipStreamSynchronize((ctx)->stream);                                                \
      if (_e == hipSuccess) _e = hipGetLastError();                                                       \
      if (_e != hipSuccess) return (ctx)->fail("launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    }                                                                                                     \
  } while (0)

// Scoped per-kernel timing (HIP events on the ctx stream, resolved lazily).
struct KernelTimer {
  mfa_ctx *c; int which; hipEvent_t a = nullptr, b = nullptr;
  KernelTimer(mfa_ctx *ctx, int w) : c(ctx), which(w) {
    if (!c->kernel_timing) return;
    a = get(); b = get();
    (void)hipEventRecord(a, c->stream);
  }
  ~KernelTimer() {
    if (!c->kernel_timing) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({which, a, b});
  }
  hipEvent_t get() {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
  }
};

int mfa_resolve_timers(mfa_ctx *ctx);

// ---- lazy (windowed) scoring: internal interface between the decoder driver (viterbi.hip) and the scoring kernels
// (gmm.hip).  Not part of the ABI.
struct MfaLazyScoring {
  mfa_score_plan plan;
  const float *d_feats;
  int max_frames;
  int window;        // frames per window of the first-beam pass (multiple of 64)
};
struct MfaWindowScore {
  int t_begin, window;          // frames [t_begin, t_begin + window) of every listed utterance
  const int32_t *band;          // [n_utt][2] {min longest-path depth, max BFS depth reachable in the window}; ignored at t_begin 0
  const int32_t *utt_list;      // utterances of this pass (NULL: all) and their count (device scalar, or NULL)
  const int32_t *n_list;
  const int32_t *done;          // per-utterance "finished" word: done[utt * done_stride + done_word] != 0 → skip
  int done_stride, done_word;
  const int32_t *lag;           // per-utterance lag word (or NULL): non-zero → the utterance's window in this call is the PREVIOUS one
  int lag_stride, lag_word;     // (t_begin − window), scored with the proven band whatever hi_slack says
  int cols_per_wave;            // 0: one wavefront walks a sub-tile's whole band; n: one wavefront per n columns of it
  int hi_slack;                 // speculative look-ahead: the band's upper depth bound is lowered by this many arcs (0: the
                                // proven bound); the decoder then checks every score it reads against mfa_band_ranges
};
// Index ranges of the columns mfa_gmm_score_window scored last, per utterance: [n_utt][kMfaRangeSlots][2], relative to the
// first column of the slot's class — slots 0..15: runs of class 0 (only the first `groups` are written), 16..18: classes 2,
// 3, 4, 19: class 1, 20: class 5.  Device memory owned by the context; valid until the next mfa_gmm_score_window call.
constexpr int kMfaRunSlots = MFA_PLAN_MAX_GROUPS;
constexpr int kMfaRangeSlots = kMfaRunSlots + 5;
const int32_t *mfa_band_ranges(mfa_ctx *ctx);
int mfa_gmm_presplit(mfa_ctx *c, const MfaLazyScoring *lazy, const int64_t *d_frame_off, int n_utt, int64_t total_frames);
int mfa_gmm_lazy_supported(mfa_ctx *ctx);   // the loaded model fits the MFMA kernels (dim <= 48)
// Score, for every listed utterance, the (frame, pdf) cells of the window that lie inside the band.  Enqueues on ctx->stream.
int mfa_gmm_score_window(mfa_ctx *ctx, const MfaLazyScoring *lazy, const MfaWindowScore *ws, const int64_t *d_frame_off,
                         int n_utt, const int64_t *d_ll_off, float *d_loglikes);
