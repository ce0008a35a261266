// Context management, memory helpers and timing for libmfa_hip.so.
#include "ctx.hpp"

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

MFA_API int mfa_version(void) { return 1; }

MFA_API mfa_ctx *mfa_create(int device_id) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return nullptr;
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  mfa_ctx *c = new mfa_ctx();
  c->device = device_id;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
  c->stream = c->own_stream;
  if (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
  }
  return c;
}

MFA_API void mfa_destroy(mfa_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  void *ptrs[] = {c->d_window, c->d_twiddle, c->d_melw, c->d_melidx, c->d_dct, c->d_lifter, c->d_w, c->d_gc,
                  c->d_row0, c->d_nblk, c->d_slot, c->d_ws, c->d_nrows, c->d_gmm_queue, c->d_wb, c->d_wh, c->d_gch, c->d_fscale, c->d_gmm_redo, c->d_w_stats, c->d_gen_ws, c->d_gen_list, c->d_xsplit, c->d_xsplit_bad, c->d_col_row0, c->d_band_ranges};
  // teardown: nothing useful can be done with a failure here
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (auto &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->t0);
  (void)hipEventDestroy(c->t1);
  (void)hipStreamDestroy(c->own_stream);
  delete c;
}

MFA_API const char *mfa_last_error(mfa_ctx *c) { return c ? c->err.c_str() : "null context"; }

MFA_API int mfa_set_stream(mfa_ctx *c, void *s, int use_own) {
  if (!c) return -1;
  c->stream = use_own ? c->own_stream : (hipStream_t)s;  // s == NULL is HIP's default (null) stream
  return 0;
}

MFA_API int mfa_synchronize(mfa_ctx *c) {
  MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return 0;
}

MFA_API void *mfa_device_alloc(mfa_ctx *c, size_t bytes) {
  void *p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess) { c->fail("hipSetDevice(%d) failed", c->device); return nullptr; }
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { c->fail("hipMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}

MFA_API int mfa_device_free(mfa_ctx *c, void *p) {
  MFA_HIP_CHECK(c, hipFree(p));
  return 0;
}

MFA_API int mfa_memcpy_h2d(mfa_ctx *c, void *d, const void *h, size_t bytes) {
  MFA_HIP_CHECK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return 0;
}

MFA_API int mfa_memcpy_d2h(mfa_ctx *c, void *h, const void *d, size_t bytes) {
  MFA_HIP_CHECK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  MFA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return 0;
}

MFA_API int mfa_timer_begin(mfa_ctx *c) {
  MFA_HIP_CHECK(c, hipEventRecord(c->t0, c->stream));
  return 0;
}

MFA_API int mfa_timer_end_ms(mfa_ctx *c, float *ms) {
  MFA_HIP_CHECK(c, hipEventRecord(c->t1, c->stream));
  MFA_HIP_CHECK(c, hipEventSynchronize(c->t1));
  MFA_HIP_CHECK(c, hipEventElapsedTime(ms, c->t0, c->t1));
  return 0;
}

MFA_API int mfa_kernel_timing(mfa_ctx *c, int enable) {
  c->kernel_timing = enable != 0;
  return 0;
}

MFA_API int mfa_kernel_time_ms(mfa_ctx *c, int which, float *ms, int *launches) {
  if (which < 0 || which >= MFA_K_COUNT) return c->fail("bad kernel id %d", which);
  if (mfa_resolve_timers(c) != 0) return -1;
  *ms = (float)c->k_ms[which];
  if (launches) *launches = c->k_n[which];
  return 0;
}

MFA_API int mfa_kernel_time_reset(mfa_ctx *c) {
  if (mfa_resolve_timers(c) != 0) return -1;
  for (int i = 0; i < MFA_K_COUNT; i++) { c->k_ms[i] = 0; c->k_n[i] = 0; }
  return 0;
}

// Host helper: the PCM of a batch — one array per utterance on the host — copied back to back into one staging buffer
// (pinned memory, so that a single asynchronous H2D copy follows) by n_threads host threads.  1.3 GB per 4 096 ten-second
// utterances: a single-threaded gather is slower than the device aligns them.
MFA_API int mfa_gather_pcm(int32_t n_utt, const int16_t *const *h_src, const int64_t *h_sample_off, int16_t *h_dst, int32_t n_threads) {
  if (n_utt < 0 || (n_utt > 0 && (!h_src || !h_sample_off || !h_dst))) return -1;
  int t = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  if (t > n_utt) t = n_utt > 0 ? n_utt : 1;
  std::atomic<int32_t> next{0};
  auto work = [&]() {
    for (;;) {
      int32_t u = next.fetch_add(1);
      if (u >= n_utt) break;
      const int64_t n = h_sample_off[u + 1] - h_sample_off[u];
      if (n > 0) memcpy(h_dst + h_sample_off[u], h_src[u], (size_t)n * sizeof(int16_t));
    }
  };
  if (t == 1) { work(); return 0; }
  std::vector<std::thread> th;
  for (int k = 0; k < t; k++) th.emplace_back(work);
  for (auto &x : th) x.join();
  return 0;
}

}  // extern "C"

int mfa_resolve_timers(mfa_ctx *c) {
  for (auto &p : c->pending) {
    MFA_HIP_CHECK(c, hipEventSynchronize(p.b));
    float ms = 0;
    MFA_HIP_CHECK(c, hipEventElapsedTime(&ms, p.a, p.b));
    c->k_ms[p.which] += ms;
    c->k_n[p.which] += 1;
    c->event_pool.push_back(p.a);
    c->event_pool.push_back(p.b);
  }
  c->pending.clear();
  return 0;
}
