// intervals.cpp — host side of the step after the device path: alignments (arrays) → phone intervals → word items →
// TextGrid / json / csv bytes.  C ABI in include/mfa_intervals.h; montreal_forced_aligner_amd/ctm.py is the specification
// this file follows statement for statement where the result depends on it (search order of the word grouping, the
// rounding and snapping rules of the writer, Python's float formatting), and tests/test_intervals_native_cpu.py holds the
// two against each other.  Reference behaviour restated by ctm.py: AlignmentExtractionFunction
// (MFA/alignment/multiprocessing.py:1733-1751), export_textgrid / Textgrid.save (MFA/textgrid.py:463-572, :50-161,
// :115-131), CtmInterval.to_tg_interval (MFA/data.py:2062-2080).  Host code only: g++, threads, no GPU.
#include "../../include/mfa_intervals.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct mfa_iv {
  int32_t n_tids = 0;
  std::vector<int32_t> id2state, id2phone;
  std::vector<uint8_t> self_loop, is_final;
  int32_t n_words = 0;
  std::vector<int32_t> word_var_off, var_off, var_phones;
  int32_t sil_phone = -1, sil_word = 0, oov_word = 1;
  double frame_shift = 0.01;
  int64_t shift_us = 10000;          // frame shift in whole microseconds, 0 when it is not one
  std::vector<std::string> phone_names, word_names;
  std::string err;
  // result of the last mfa_iv_extract_batch, kept until mfa_iv_fetch copies it out: utterances are processed in chunks
  // (one worker thread at a time per chunk), each chunk appends to its own arrays
  struct Chunk {
    std::vector<int32_t> ph_first, ph_len, ph_id, it_word, it_var, it_first, it_count, it_ref;
  };
  std::vector<Chunk> chunks;
  int64_t chunk_size = 1;
  std::vector<int32_t> n_phones, n_items, utt_err;
};

namespace {

int thread_count(int32_t n_threads, int64_t n_items) {
  int t = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  if ((int64_t)t > n_items) t = (int)std::max<int64_t>(1, n_items);
  return t;
}

template <class F>
void parallel_for(int64_t n, int32_t n_threads, F fn) {
  int t = thread_count(n_threads, n);
  if (t <= 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<int64_t> next{0};
  const int64_t chunk = std::max<int64_t>(1, n / (t * 8));
  std::vector<std::thread> th;
  for (int k = 0; k < t; k++)
    th.emplace_back([&]() {
      for (;;) {
        int64_t a = next.fetch_add(chunk);
        if (a >= n) break;
        int64_t b = std::min(n, a + chunk);
        for (int64_t i = a; i < b; i++) fn(i);
      }
    });
  for (auto &x : th) x.join();
}

// ---------------------------------------------------------------------------------------------------------------
// SplitToPhones (reordered graphs), ctm._split_to_phones_loop / Kaldi hmm-utils.cc SplitToPhonesInternal: returns the
// number of phones, or -1 when the alignment is not a sequence of complete phones.
int split_to_phones(const mfa_iv &m, const int32_t *ali, int32_t T, int32_t *first, int32_t *len, int32_t *phone) {
  int n = 0;
  int32_t cur = 0;
  for (int32_t i = 0; i < T; i++) {
    int32_t tid = ali[i];
    if (tid < 1 || tid > m.n_tids) return -1;
    if (m.is_final[tid]) {
      while (i + 1 < T) {
        int32_t nx = ali[i + 1];
        if (nx < 1 || nx > m.n_tids) return -1;
        if (!m.self_loop[nx]) break;
        if (m.id2state[ali[i]] != m.id2state[nx]) return -1;
        i++;
      }
      first[n] = cur; len[n] = i + 1 - cur; phone[n] = m.id2phone[ali[cur]];
      n++;
      cur = i + 1;
    } else if (i + 1 == T) {
      return -1;
    } else {
      int32_t nx = ali[i + 1];
      if (nx < 1 || nx > m.n_tids) return -1;
      if (m.id2state[tid] != m.id2state[nx] && m.id2phone[tid] != m.id2phone[nx]) return -1;
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// Word grouping: ctm.phones_to_pronunciations' search — segment the phone intervals into the aligned word sequence, each
// word spelt by one of its variants (tried in table order), optional-silence intervals in between; a word before "this
// interval is inter-word silence"; depth-first with the dead-state memo.
struct Grouper {
  const mfa_iv &m;
  const int32_t *ph;      // phone ids of the utterance's intervals
  int n;                  // intervals
  const int32_t *words;   // aligned word ids
  int nw;
  std::vector<uint8_t> dead;                 // [(nw + 1) * (n + 1)]
  struct Item { int32_t word, var, first, count, ref; };
  std::vector<Item> out;                     // built back to front

  Grouper(const mfa_iv &mm, const int32_t *p, int nn, const int32_t *w, int nww) : m(mm), ph(p), n(nn), words(w), nw(nww) {
    dead.assign((size_t)(nw + 1) * (n + 1), 0);
  }
  bool solve(int i, int k) {
    uint8_t &d = dead[(size_t)i * (n + 1) + k];
    if (d) return false;
    if (i == nw) {
      for (int j = k; j < n; j++)
        if (ph[j] != m.sil_phone) { d = 1; return false; }
      for (int j = n - 1; j >= k; j--) out.push_back(Item{m.sil_word, -1, j, 1, -1});
      return true;
    }
    int32_t w = words[i];
    if (w >= 0 && w < m.n_words) {
      for (int32_t v = m.word_var_off[w]; v < m.word_var_off[w + 1]; v++) {
        int len = m.var_off[v + 1] - m.var_off[v];
        if (k + len > n) continue;
        const int32_t *vp = m.var_phones.data() + m.var_off[v];
        bool eq = true;
        for (int j = 0; j < len; j++)
          if (ph[k + j] != vp[j]) { eq = false; break; }
        if (!eq) continue;
        if (solve(i + 1, k + len)) { out.push_back(Item{w, v - m.word_var_off[w], k, len, i}); return true; }
      }
    }
    if (k < n && ph[k] == m.sil_phone && solve(i, k + 1)) { out.push_back(Item{m.sil_word, -1, k, 1, -1}); return true; }
    d = 1;
    return false;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Numbers as Python writes them.
// round(x, 6): correctly rounded to six decimals, then the nearest double (float.__round__).
double round6(double x) {
  if (std::fabs(x) < 4.0e9) {          // already the double nearest to k / 10^6?  (every frame time is)
    double s = x * 1e6;
    long long k = std::llround(s);
    if ((double)k / 1e6 == x) return x;
  }
  char buf[64];
  snprintf(buf, sizeof buf, "%.6f", x);
  return strtod(buf, nullptr);
}

// repr(float): shortest digits that round-trip; fixed notation for 1e-4 <= |x| < 1e16, else d.ddde±XX; always a '.0'.
void append_repr(std::string &o, double x) {
  if (x == 0.0) { o += std::signbit(x) ? "-0.0" : "0.0"; return; }
  if (std::isnan(x)) { o += "nan"; return; }
  if (std::isinf(x)) { o += x < 0 ? "-inf" : "inf"; return; }
  char buf[40];
  auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // shortest round-trip
  std::string s(buf, r.ptr);
  size_t epos = s.find('e');
  std::string mant = s.substr(0, epos);
  int exp10 = atoi(s.c_str() + epos + 1);
  bool neg = false;
  if (mant[0] == '-') { neg = true; mant.erase(0, 1); }
  std::string digits;
  for (char c : mant) if (c != '.') digits.push_back(c);
  if (neg) o.push_back('-');
  int decpt = exp10 + 1;                                // value = 0.digits * 10^decpt
  if (decpt <= -4 || decpt > 16) {
    o.push_back(digits[0]);
    if (digits.size() > 1) { o.push_back('.'); o.append(digits, 1, std::string::npos); }
    char e[16];
    snprintf(e, sizeof e, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
    o += e;
    return;
  }
  if (decpt <= 0) {
    o += "0.";
    o.append((size_t)(-decpt), '0');
    o += digits;
  } else if ((size_t)decpt >= digits.size()) {
    o += digits;
    o.append((size_t)decpt - digits.size(), '0');
    o += ".0";
  } else {
    o.append(digits, 0, (size_t)decpt);
    o.push_back('.');
    o.append(digits, (size_t)decpt, std::string::npos);
  }
}

void append_tg_escaped(std::string &o, const std::string &s) {      // label.replace('"', '""')
  for (char c : s) { if (c == '"') o.push_back('"'); o.push_back(c); }
}

void append_json_string(std::string &o, const std::string &s) {     // json.dumps(s, ensure_ascii=False)
  o.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      default:
        if (c < 0x20) { char e[8]; snprintf(e, sizeof e, "\\u%04x", c); o += e; }
        else o.push_back((char)c);
    }
  }
  o.push_back('"');
}

void append_csv_field(std::string &o, const std::string &s) {       // csv excel dialect, QUOTE_MINIMAL
  bool q = s.empty() ? false : false;
  for (char c : s) if (c == ',' || c == '"' || c == '\n' || c == '\r') { q = true; break; }
  if (!q) { o += s; return; }
  o.push_back('"');
  for (char c : s) { if (c == '"') o.push_back('"'); o.push_back(c); }
  o.push_back('"');
}

struct Ival { double begin, end; const std::string *label; };

struct FileJob {
  std::string text;
  int32_t err = 0;
};

}  // namespace

extern "C" {

MFA_IV_API int mfa_iv_version(void) { return 1; }

MFA_IV_API mfa_iv *mfa_iv_create(const mfa_iv_config *c) {
  if (!c || c->n_tids < 0 || c->n_words < 0) return nullptr;
  mfa_iv *m = new mfa_iv();
  m->n_tids = c->n_tids;
  m->id2state.assign(c->id2state, c->id2state + c->n_tids + 1);
  m->id2phone.assign(c->id2phone, c->id2phone + c->n_tids + 1);
  m->self_loop.resize(c->n_tids + 1);
  m->is_final.resize(c->n_tids + 1);
  for (int i = 0; i <= c->n_tids; i++) { m->self_loop[i] = c->is_self_loop[i] != 0; m->is_final[i] = c->is_final[i] != 0; }
  m->n_words = c->n_words;
  m->word_var_off.assign(c->word_var_off, c->word_var_off + c->n_words + 1);
  int32_t nv = m->word_var_off[c->n_words];
  m->var_off.assign(c->var_off, c->var_off + nv + 1);
  m->var_phones.assign(c->var_phones, c->var_phones + m->var_off[nv]);
  m->sil_phone = c->sil_phone; m->sil_word = c->sil_word; m->oov_word = c->oov_word;
  m->frame_shift = c->frame_shift;
  double us = c->frame_shift * 1e6;
  long long k = std::llround(us);
  m->shift_us = (k > 0 && std::fabs(us - (double)k) < 1e-6) ? k : 0;
  m->phone_names.resize(c->n_phone_names);
  for (int i = 0; i < c->n_phone_names; i++)
    m->phone_names[i].assign(c->phone_names + c->phone_name_off[i], c->phone_names + c->phone_name_off[i + 1]);
  m->word_names.resize(c->n_words);
  for (int i = 0; i < c->n_words; i++)
    m->word_names[i].assign(c->word_names + c->word_name_off[i], c->word_names + c->word_name_off[i + 1]);
  return m;
}

MFA_IV_API void mfa_iv_destroy(mfa_iv *iv) { delete iv; }
MFA_IV_API const char *mfa_iv_last_error(const mfa_iv *iv) { return iv ? iv->err.c_str() : "null handle"; }

MFA_IV_API int mfa_iv_extract_batch(mfa_iv *iv, int32_t n_utt, const int64_t *frame_off, const int32_t *ali, const int32_t *words,
                                    const int32_t *n_words, const int32_t *status, int32_t n_threads, int64_t *ph_off,
                                    int64_t *it_off, int32_t *h_err) {
  if (!iv) return -1;
  if (n_utt < 0) { iv->err = "negative utterance count"; return -1; }
  mfa_iv &m = *iv;
  const int t = thread_count(n_threads, n_utt);
  m.chunk_size = std::max<int64_t>(1, (int64_t)n_utt / ((int64_t)t * 8));
  const int64_t n_chunks = n_utt ? (n_utt + m.chunk_size - 1) / m.chunk_size : 0;
  m.chunks.assign((size_t)n_chunks, mfa_iv::Chunk());
  m.n_phones.assign((size_t)n_utt, 0); m.n_items.assign((size_t)n_utt, 0); m.utt_err.assign((size_t)n_utt, 0);
  parallel_for(n_chunks, n_threads, [&](int64_t c) {
    mfa_iv::Chunk &ch = m.chunks[c];
    std::vector<int32_t> first, len, id;
    const int64_t u0 = c * m.chunk_size, u1 = std::min<int64_t>(n_utt, u0 + m.chunk_size);
    for (int64_t u = u0; u < u1; u++) {
      if (status && status[u] != 0 && status[u] != 1) { m.utt_err[u] = MFA_IV_SKIPPED; continue; }
      const int64_t a = frame_off[u];
      const int32_t T = (int32_t)(frame_off[u + 1] - a);
      if ((int32_t)first.size() < T) { first.resize(T); len.resize(T); id.resize(T); }
      int np = T > 0 ? split_to_phones(m, ali + a, T, first.data(), len.data(), id.data()) : 0;
      if (np < 0) { m.utt_err[u] = MFA_IV_IRREGULAR; continue; }
      int nw = n_words[u];
      if (nw < 0 || nw > T) { m.utt_err[u] = MFA_IV_UNSPELLABLE; continue; }
      Grouper g(m, id.data(), np, words + a, nw);
      if (!g.solve(0, 0)) { m.utt_err[u] = MFA_IV_UNSPELLABLE; continue; }
      ch.ph_first.insert(ch.ph_first.end(), first.begin(), first.begin() + np);
      ch.ph_len.insert(ch.ph_len.end(), len.begin(), len.begin() + np);
      ch.ph_id.insert(ch.ph_id.end(), id.begin(), id.begin() + np);
      const int ni = (int)g.out.size();
      for (int j = ni - 1; j >= 0; j--) {
        const Grouper::Item &it = g.out[j];
        ch.it_word.push_back(it.word); ch.it_var.push_back(it.var); ch.it_first.push_back(it.first);
        ch.it_count.push_back(it.count); ch.it_ref.push_back(it.ref);
      }
      m.n_phones[u] = np; m.n_items[u] = ni;
    }
  });
  ph_off[0] = 0; it_off[0] = 0;
  for (int32_t u = 0; u < n_utt; u++) {
    ph_off[u + 1] = ph_off[u] + m.n_phones[u];
    it_off[u + 1] = it_off[u] + m.n_items[u];
    h_err[u] = m.utt_err[u];
  }
  return 0;
}

MFA_IV_API int mfa_iv_fetch(mfa_iv *iv, int32_t *ph_first, int32_t *ph_len, int32_t *ph_id, int32_t *it_word, int32_t *it_var,
                            int32_t *it_first, int32_t *it_count, int32_t *it_ref) {
  if (!iv) return -1;
  size_t p = 0, q = 0;
  for (const mfa_iv::Chunk &ch : iv->chunks) {
    const size_t np = ch.ph_first.size(), ni = ch.it_word.size();
    if (np) {
      memcpy(ph_first + p, ch.ph_first.data(), np * 4); memcpy(ph_len + p, ch.ph_len.data(), np * 4);
      memcpy(ph_id + p, ch.ph_id.data(), np * 4);
    }
    if (ni) {
      memcpy(it_word + q, ch.it_word.data(), ni * 4); memcpy(it_var + q, ch.it_var.data(), ni * 4);
      memcpy(it_first + q, ch.it_first.data(), ni * 4); memcpy(it_count + q, ch.it_count.data(), ni * 4);
      memcpy(it_ref + q, ch.it_ref.data(), ni * 4);
    }
    p += np; q += ni;
  }
  iv->chunks.clear();
  iv->chunks.shrink_to_fit();
  return 0;
}

MFA_IV_API int mfa_iv_write_files(mfa_iv *iv, int32_t format, int32_t cleanup_silence, int32_t n_files, const double *file_duration,
                                  const int32_t *file_spk_off, const int64_t *spk_name_off, const char *spk_names,
                                  const int32_t *spk_utt_off, const int32_t *spk_utt, const double *utt_begin,
                                  const double *utt_end, const int64_t *ph_off, const int32_t *ph_first, const int32_t *ph_len,
                                  const int32_t *ph_id, const int64_t *it_off, const int32_t *it_word, const int32_t *it_first,
                                  const int32_t *it_count, const int32_t *it_ref, const int32_t *h_err,
                                  int32_t n_relabel, const int32_t *relabel_utt, const int32_t *relabel_ref,
                                  const int64_t *relabel_off, const char *relabel_text, int32_t n_threads, char *out,
                                  int64_t out_cap, int64_t *out_off, int32_t *file_err, int64_t *needed) {
  if (!iv) return -1;
  if (format < 0 || format > 3) { iv->err = "unknown output format"; return -1; }
  const mfa_iv &m = *iv;
  const std::string empty;
  const std::string words_name = "words", phones_name = "phones";
  std::vector<FileJob> jobs((size_t)std::max(0, n_files));
  // transcript spellings of out-of-vocabulary items, sorted by (utterance, ref)
  std::vector<std::string> relabels((size_t)std::max(0, n_relabel));
  for (int k = 0; k < n_relabel; k++) relabels[k].assign(relabel_text + relabel_off[k], relabel_text + relabel_off[k + 1]);
  auto find_relabel = [&](int32_t u, int32_t ref) -> const std::string * {
    int lo = 0, hi = n_relabel;
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (relabel_utt[mid] < u || (relabel_utt[mid] == u && relabel_ref[mid] < ref)) lo = mid + 1; else hi = mid;
    }
    if (lo < n_relabel && relabel_utt[lo] == u && relabel_ref[lo] == ref) return &relabels[lo];
    return nullptr;
  };
  auto frame_time = [&](int64_t f) -> double {          // ctm._frame_times: round(f * shift, 6)
    if (m.shift_us > 0 && f * m.shift_us < (1LL << 52)) return (double)(f * m.shift_us) / 1e6;
    return round6((double)f * m.frame_shift);
  };

  parallel_for(n_files, n_threads, [&](int64_t f) {
    FileJob &job = jobs[f];
    const int s0 = file_spk_off[f], s1 = file_spk_off[f + 1];
    const int n_spk = s1 - s0;
    const double duration = round6(file_duration[f]);
    const double snap = m.frame_shift * 2;
    struct Tier { std::string name; std::vector<Ival> iv; };
    std::vector<Tier> tiers;
    std::vector<std::string> spk((size_t)n_spk);
    bool has_data = false;
    for (int s = s0; s < s1; s++) {
      spk[s - s0].assign(spk_names + spk_name_off[s], spk_names + spk_name_off[s + 1]);
      Tier tw, tp;
      tw.name = n_spk > 1 ? spk[s - s0] + " - words" : words_name;
      tp.name = n_spk > 1 ? spk[s - s0] + " - phones" : phones_name;
      for (int q = spk_utt_off[s]; q < spk_utt_off[s + 1]; q++) {
        const int32_t u = spk_utt[q];
        if (h_err[u] != MFA_IV_OK) continue;
        const int64_t a = ph_off[u], b = it_off[u];
        const int np = (int)(ph_off[u + 1] - a), ni = (int)(it_off[u + 1] - b);
        const double off = utt_begin[u];
        // phone times: frame times, shifted by the utterance's begin when it is non-zero, the last one clipped
        // (HierarchicalCtm.update_utterance_boundaries)
        auto pb = [&](int j) { double t = frame_time(ph_first[a + j]); return off != 0.0 ? t + off : t; };
        auto pe = [&](int j) {
          double t = frame_time((int64_t)ph_first[a + j] + ph_len[a + j]);
          if (off != 0.0) t += off;
          if (j == np - 1 && t > utt_end[u]) t = utt_end[u];
          return t;
        };
        for (int i = 0; i < ni; i++) {
          const int32_t w = it_word[b + i];
          if (cleanup_silence && w == m.sil_word) continue;
          const int j0 = it_first[b + i], j1 = j0 + it_count[b + i] - 1;
          const std::string *label = (w >= 0 && w < m.n_words) ? &m.word_names[w] : &empty;
          if (w == m.oov_word && it_ref[b + i] >= 0) {
            const std::string *r = find_relabel(u, it_ref[b + i]);
            if (r) label = r;
          }
          tw.iv.push_back(Ival{pb(j0), pe(j1), label});
          for (int j = j0; j <= j1; j++) {
            int32_t p = ph_id[a + j];
            tp.iv.push_back(Ival{pb(j), pe(j), (p >= 0 && p < (int32_t)m.phone_names.size()) ? &m.phone_names[p] : &empty});
          }
        }
      }
      auto by_begin = [](const Ival &x, const Ival &y) { return x.begin < y.begin; };
      std::stable_sort(tw.iv.begin(), tw.iv.end(), by_begin);
      std::stable_sort(tp.iv.begin(), tp.iv.end(), by_begin);
      has_data = has_data || !tw.iv.empty() || !tp.iv.empty();
      tiers.push_back(std::move(tw));
      tiers.push_back(std::move(tp));
    }
    if (!has_data) { job.err = 2; return; }
    std::string &o = job.text;
    if (format == 3) {                       // csv: Begin, End, Label, Type, Speaker — every interval near the end snapped
      o += "Begin,End,Label,Type,Speaker\r\n";
      for (size_t t = 0; t < tiers.size(); t++) {
        const std::string &speaker = spk[t / 2];
        for (Ival &a : tiers[t].iv) {
          if (duration - a.end < snap) a.end = duration;
          append_repr(o, a.begin); o.push_back(',');
          append_repr(o, a.end); o.push_back(',');
          append_csv_field(o, *a.label); o.push_back(',');
          o += (t % 2 == 0) ? "words" : "phones"; o.push_back(',');
          append_csv_field(o, speaker); o += "\r\n";
        }
      }
      return;
    }
    if (format == 2) {                       // json.dump(js, indent=4, ensure_ascii=False)
      o += "{\n    \"start\": 0,\n    \"end\": ";
      append_repr(o, duration);
      o += ",\n    \"tiers\": {\n";
      for (size_t t = 0; t < tiers.size(); t++) {
        o += "        "; append_json_string(o, tiers[t].name); o += ": {\n            \"type\": \"interval\",\n            \"entries\": [";
        if (tiers[t].iv.empty()) o += "]";
        else {
          o += "\n";
          for (size_t i = 0; i < tiers[t].iv.size(); i++) {
            Ival &a = tiers[t].iv[i];
            if (duration - a.end < snap) a.end = duration;
            o += "                [\n                    "; append_repr(o, a.begin);
            o += ",\n                    "; append_repr(o, a.end);
            o += ",\n                    "; append_json_string(o, *a.label);
            o += "\n                ]";
            o += (i + 1 < tiers[t].iv.size()) ? ",\n" : "\n";
          }
          o += "            ]";
        }
        o += "\n        }";
        o += (t + 1 < tiers.size()) ? ",\n" : "\n";
      }
      o += "    }\n}";
      return;
    }
    // TextGrid (long / short): rounding, snapping of the last interval, overlap clipping, blank filling
    struct Entry { double b, e; const std::string *label; };
    std::vector<std::vector<Entry>> ents(tiers.size());
    for (size_t t = 0; t < tiers.size(); t++) {
      std::vector<Ival> &v = tiers[t].iv;
      std::vector<Entry> &en = ents[t];
      for (size_t i = 0; i < v.size(); i++) {
        Ival a = v[i];
        if (i == v.size() - 1 && duration - a.end < snap) a.end = duration;
        if (a.end < -1 || a.begin == 1000000) { job.err = 1; return; }
        double e = round6(a.end), b = round6(a.begin);
        if (b >= e) { job.err = 1; return; }
        if (i > 0 && en.back().e > b) {
          a.begin = en.back().e;
          b = round6(a.begin);
          if (b >= e) { job.err = 1; return; }
        }
        en.push_back(Entry{b, e, a.label});
      }
      if (!en.empty() && en.back().e > duration) en.back().e = duration;
      // _fill_blanks(entries, duration): blank intervals for gaps larger than 1 ms
      if (!en.empty()) {
        std::vector<Entry> fl;
        fl.reserve(en.size() * 2 + 2);
        if (en[0].b > 0.001) fl.push_back(Entry{0.0, en[0].b, &empty});
        for (size_t i = 0; i < en.size(); i++) {
          if (i > 0 && en[i].b - en[i - 1].e > 0.001) fl.push_back(Entry{en[i - 1].e, en[i].b, &empty});
          fl.push_back(en[i]);
        }
        if (duration - fl.back().e > 0.001) fl.push_back(Entry{fl.back().e, duration, &empty});
        en.swap(fl);
      }
    }
    const bool long_fmt = format == 0;
    std::string dur;
    append_repr(dur, duration);
    o += "File type = \"ooTextFile\"\nObject class = \"TextGrid\"\n\n";
    if (long_fmt) {
      o += "xmin = 0 \nxmax = " + dur + " \ntiers? <exists> \nsize = " + std::to_string(tiers.size()) + " \nitem []: \n";
    } else {
      o += "0\n" + dur + "\n<exists>\n" + std::to_string(tiers.size()) + "\n";
    }
    for (size_t t = 0; t < tiers.size(); t++) {
      const std::vector<Entry> &en = ents[t];
      if (long_fmt) {
        o += "    item [" + std::to_string(t + 1) + "]:\n";
        o += "        class = \"IntervalTier\" \n        name = \"";
        append_tg_escaped(o, tiers[t].name);
        o += "\" \n        xmin = 0 \n        xmax = " + dur + " \n        intervals: size = " + std::to_string(en.size()) + " \n";
      } else {
        o += "\"IntervalTier\"\n\"";
        append_tg_escaped(o, tiers[t].name);
        o += "\"\n0\n" + dur + "\n" + std::to_string(en.size()) + "\n";
      }
      for (size_t i = 0; i < en.size(); i++) {
        if (long_fmt) {
          o += "        intervals [" + std::to_string(i + 1) + "]:\n            xmin = ";
          append_repr(o, en[i].b);
          o += " \n            xmax = ";
          append_repr(o, en[i].e);
          o += " \n            text = \"";
          append_tg_escaped(o, *en[i].label);
          o += "\" \n";
        } else {
          append_repr(o, en[i].b); o.push_back('\n');
          append_repr(o, en[i].e); o += "\n\"";
          append_tg_escaped(o, *en[i].label);
          o += "\"\n";
        }
      }
    }
  });

  int64_t total = 0;
  for (int f = 0; f < n_files; f++) {
    out_off[f] = total;
    file_err[f] = jobs[f].err;
    if (jobs[f].err == 0) total += (int64_t)jobs[f].text.size();
  }
  out_off[n_files] = total;
  if (needed) *needed = total;
  if (total > out_cap) { iv->err = "output buffer too small"; return -2; }
  parallel_for(n_files, n_threads, [&](int64_t f) {
    if (jobs[f].err == 0 && !jobs[f].text.empty()) memcpy(out + out_off[f], jobs[f].text.data(), jobs[f].text.size());
  });
  return 0;
}

// The files themselves, written by n_threads threads: file f = out[out_off[f] .. out_off[f + 1]) to the path
// paths[path_off[f] .. path_off[f + 1]) (UTF-8), skipped when file_err[f] != 0.  io_err[f] receives errno of a failed
// open / write / close (0 = written or skipped).  Returns the number of files that failed.
MFA_IV_API int mfa_iv_save_files(int32_t n_files, const int64_t *path_off, const char *paths, const char *out, const int64_t *out_off,
                                 const int32_t *file_err, int32_t n_threads, int32_t *io_err) {
  std::atomic<int> failed{0};
  parallel_for(n_files, n_threads, [&](int64_t f) {
    io_err[f] = 0;
    if (file_err && file_err[f] != 0) return;
    std::string path(paths + path_off[f], paths + path_off[f + 1]);
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { io_err[f] = errno ? errno : -1; failed.fetch_add(1); return; }
    const size_t n = (size_t)(out_off[f + 1] - out_off[f]);
    bool ok = n == 0 || fwrite(out + out_off[f], 1, n, fp) == n;
    if (fclose(fp) != 0) ok = false;
    if (!ok) { io_err[f] = errno ? errno : -1; failed.fetch_add(1); }
  });
  return failed.load();
}

}  // extern "C"
