// Batched MFCC for gfx950: framing (edge reflection) → DC removal → pre-emphasis → Povey window → 512-point real FFT →
// power spectrum → mel bins → log → DCT-II → lifter.  Replaces kalpy MfccComputer.compute_mfccs (MFA/corpus/features.py:235;
// Kaldi feat/feature-mfcc.cc).
//
// Layout (round 2): a wavefront works on FOUR frames at a time — one 16-lane row per frame — and the 512-point real FFT
// is a 256-point complex FFT done as 16 × 16: lane i of a row holds the 16 packed samples z[i + 16 j] in registers, runs a
// 16-point DFT on them (two radix-4 levels, no memory traffic), applies the twiddle W256^(i·k1), the rows are transposed
// through a padded, conflict-free LDS tile (the only exchange of the whole transform), and a second in-register 16-point
// DFT leaves lane i with Z[i + 16 k2].  Round 1's kernel (one frame per wavefront, four radix-4 Stockham stages through
// LDS) spent as many LDS cycles as VALU cycles, 95 % of them on bank conflicts of its scattered 8-byte stores
// (SQ_LDS_BANK_CONFLICT ≈ SQ_ACTIVE_INST_LDS in the profile): 5.2 ms per 4.1 M frames; the butterflies now run on packed
// float2 arithmetic (v_pk_add/mul/fma_f32) out of registers.
// Row-local steps use the DPP crossbar: the sample before a lane's first one lives in the previous lane (row_shr:1), the
// frame sum is a row reduction, Z[256 − k] lives in lane 16 − i (row_mirror + row_shr:1).
// The mel filterbank reads the power spectrum back from LDS: every triangle is cut into pieces of at most `piece_taps`
// taps, piece p is summed by lane p mod 16 of each row, the pieces of a bin are then added in ascending order (fixed
// order: results are reproducible run to run); lane k of a row finishes cepstral coefficient k.
// Compiled with -ffp-contract=off: the only fused multiply-adds are the explicit ones.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ctx.hpp"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kNfft = 512;
constexpr int kHalf = 256;
#ifndef MFA_MFCC_WAVES
#define MFA_MFCC_WAVES 8
#endif
constexpr int kWavesPerBlock = MFA_MFCC_WAVES;
constexpr int kFramesPerWave = 16;      // four passes of four frames
constexpr int kFramesPerBlock = kWavesPerBlock * kFramesPerWave;
constexpr int kMaxBins = 32;
constexpr int kMaxCeps = 16;            // one lane of a row per cepstral coefficient
constexpr int kMaxDct = 384;            // coefficients × bins
constexpr int kMaxPieces = 96;          // filterbank pieces (16 lanes × at most 6 rounds)
constexpr int kPieceTaps = 8;           // FFT bins per piece (short pieces are zero padded)
constexpr int kMaxMelW = kMaxPieces * kPieceTaps;
constexpr int kRowPad = 17;             // float2 row stride of the transposition tile: 34 dwords ≡ 2 (mod 32)
constexpr int kTileFloats = 4 * 16 * kRowPad * 2;   // one wavefront's LDS tile: 2 176 floats
// row strides of the reused tile are chosen so that the two rows of a 32-lane LDS group land on different banks
constexpr int kPStride = 273;           // power spectrum of one frame: 256 bins + room for a piece's zero-weight tail
constexpr int kPartStride = kMaxPieces + 8;
constexpr int kMelStride = kMaxBins + 8;
constexpr int kPartOff = 4 * kPStride;  // [4][kPartStride] piece sums
constexpr int kMelOff = kPartOff + 4 * kPartStride;   // [4][kMelStride] log mel energies
static_assert(kMelOff + 4 * kMelStride <= kTileFloats, "tile reuse");

struct MfccParams {
  int win, shift, nbins, nceps, snip_edges, remove_dc;
  int n_pieces, n_rounds, np_max;
  int raw_energy;           // use_energy (kEnergy instantiations): C0 is the frame's log energy, before (1) or after (0) pre-emphasis and window
  float log_energy_floor;   // log(energy_floor) or -inf
  float preemph;
  const float *window;      // [16][16][2]: window[2m], window[2m+1] for m = i + 16 j (0 beyond the window)
  const float *tw256;       // [16][16][2]: W256^(i·k1) at [k1][i]
  const float *tw512;       // [16][16][2]: −i · W512^(i + 16 k2) at [k2][i]
  const float *melw;        // [n_pieces][kPieceTaps] (zero padded)
  const int32_t *melinfo;   // [kMaxPieces] first FFT bin of the piece, then [kMaxBins] (first piece | pieces << 8) per bin
  const float *dct;         // [nceps][nbins]
  const float *lifter;      // [nceps]
};

// LDS operations of one wavefront execute in program order; the fences only stop the compiler from moving accesses
// across a hand-over between lanes.
#define WAVE_SYNC()                                              \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// permutations that give every lane a source: no `old` value to keep (and no register copy to set one up)
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// Σ over the 16 lanes of a row, returned in every lane of the row.  The addends are integer-valued and the total stays
// below 2^24, so the order of the additions does not matter.
__device__ __forceinline__ float row_sum_exact(float v) {
  v += dpp_f32<0xB1>(0.0f, v);          // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(0.0f, v);          // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(0.0f, v);         // row_half_mirror
  v += dpp_f32<0x140>(0.0f, v);         // row_mirror
  return v;
}

__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
// a·b (complex): one packed multiply (operand halves picked by op_sel) and two fused multiply-adds
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
  const v2f t = splat(a.y) * (v2f){b.y, b.x};                       // (a.y b.y, a.y b.x)
  float re = fmaf(a.x, b.x, -t.x);
  const float im = fmaf(a.x, b.y, t.y);
  asm("" : "+v"(re));   // keeps the two apart: paired into one v_pk_fma the compiler first builds (−t.x, t.y) with three more instructions
  return (v2f){re, im};
}
// 4-point DFT in place (W4 = −i): a_k ← Σ_n a_n (−i)^(nk).  kZero3: a3 is known to be zero (samples beyond the window).
template <bool kZero3 = false>
__device__ __forceinline__ void dft4(v2f &a0, v2f &a1, v2f &a2, v2f &a3) {
  const v2f s02 = a0 + a2, d02 = a0 - a2;
  const v2f s13 = kZero3 ? a1 : a1 + a3, d13 = kZero3 ? a1 : a1 - a3;
  a0 = s02 + s13; a2 = s02 - s13;
  a1 = (v2f){d02.x + d13.y, d02.y - d13.x};                          // d02 − i·d13
  a3 = (v2f){d02.x - d13.y, d02.y + d13.x};                          // d02 + i·d13
}
// 16-point DFT of a[0..15] (W16 = e^(−2πi/16)); result in natural order.  kLive: a[kLive..15] are known to be zero.
template <int kLive = 16>
__device__ __forceinline__ void dft16(v2f (&a)[16]) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  // n = na + 4 nb, k = kb + 4 ka:  Y[kb + 4 ka] = Σ_na W4^(na ka) · W16^(na kb) · Σ_nb a[na + 4 nb] W4^(nb kb)
#pragma unroll
  for (int na = 0; na < 4; na++) {                                                  // a[na + 4 kb] = inner sum
    if (na + 12 >= kLive && na + 8 < kLive) dft4<true>(a[na], a[na + 4], a[na + 8], a[na + 12]);
    else dft4<false>(a[na], a[na + 4], a[na + 8], a[na + 12]);
  }
  a[1 + 4] = cmul(a[1 + 4], (v2f){c1, -s1});    // W16^1
  a[1 + 8] = cmul(a[1 + 8], (v2f){h, -h});      // W16^2
  a[1 + 12] = cmul(a[1 + 12], (v2f){s1, -c1});  // W16^3
  a[2 + 4] = cmul(a[2 + 4], (v2f){h, -h});      // W16^2
  a[2 + 8] = (v2f){a[2 + 8].y, -a[2 + 8].x};    // W16^4 = −i
  a[2 + 12] = cmul(a[2 + 12], (v2f){-h, -h});   // W16^6
  a[3 + 4] = cmul(a[3 + 4], (v2f){s1, -c1});    // W16^3
  a[3 + 8] = cmul(a[3 + 8], (v2f){-h, -h});     // W16^6
  a[3 + 12] = cmul(a[3 + 12], (v2f){-c1, s1});  // W16^9
#pragma unroll
  for (int kb = 0; kb < 4; kb++) dft4(a[4 * kb], a[4 * kb + 1], a[4 * kb + 2], a[4 * kb + 3]);   // a[4 kb + ka] = Y[kb + 4 ka]
  // natural order: Y[kb + 4 ka] ← a[4 kb + ka] (a register renaming once the loops are unrolled)
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
#pragma unroll
    for (int ka = kb + 1; ka < 4; ka++) { const v2f t = a[4 * kb + ka]; a[4 * kb + ka] = a[4 * ka + kb]; a[4 * ka + kb] = t; }
}

// kJ = ⌈window / 32⌉ packed sample pairs per lane (13 for MFA's 25 ms at 16 kHz)
template <int kJ, bool kEnergy = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, 4) void mfcc_kernel(MfccParams p, const int16_t *__restrict__ pcm,
                                                                        const int64_t *__restrict__ sample_off,
                                                                        const int64_t *__restrict__ frame_off,
                                                                        float *__restrict__ out) {
  __shared__ float s_tile[kWavesPerBlock][kTileFloats];
  __shared__ v2f s_tw256[256];
  __shared__ v2f s_tw512[256];
  __shared__ float s_melw[kMaxMelW];
  __shared__ float s_dct[kMaxDct];
  __shared__ v2f s_wnd[256];
  __shared__ int s_info[kMaxPieces + kMaxBins];
  const int utt = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i = lane & 15;
  const int64_t s0 = sample_off[utt], n = sample_off[utt + 1] - s0;
  const int64_t f0 = frame_off[utt];
  const int T = (int)(frame_off[utt + 1] - f0);
  if ((int)blockIdx.x * kFramesPerBlock >= T) return;  // whole workgroup: nothing to do
  for (int k = threadIdx.x; k < 256; k += 64 * kWavesPerBlock) {
    s_tw256[k] = (v2f){p.tw256[2 * k], p.tw256[2 * k + 1]};
    s_tw512[k] = (v2f){p.tw512[2 * k], p.tw512[2 * k + 1]};
    s_wnd[k] = (v2f){p.window[2 * k], p.window[2 * k + 1]};
  }
  for (int k = threadIdx.x; k < p.n_pieces * kPieceTaps; k += 64 * kWavesPerBlock) s_melw[k] = p.melw[k];
  for (int k = threadIdx.x; k < p.nceps * p.nbins; k += 64 * kWavesPerBlock) s_dct[k] = p.dct[k];
  for (int k = threadIdx.x; k < kMaxPieces + kMaxBins; k += 64 * kWavesPerBlock) s_info[k] = p.melinfo[k];
  float *tile = s_tile[wave];
  for (int k = lane; k < kTileFloats; k += 64) tile[k] = 0.0f;   // nothing non-finite may ever sit under a zero weight
  __syncthreads();
  const int fbase = blockIdx.x * kFramesPerBlock + wave * kFramesPerWave;
  const int nfr = min(kFramesPerWave, T - fbase);
  if (nfr <= 0) return;  // wavefronts are independent from here on
  const int16_t *x = pcm + s0;

  // ---- per-lane constants (the window / twiddle tables are read from LDS in every pass: 60 registers otherwise)
  const float lift_k = i < p.nceps ? p.lifter[i] : 0.0f;
  v2f *tq = (v2f *)tile + q * (16 * kRowPad);     // this row's transposition tile
  float *pq = tile + q * kPStride;                  // … power spectrum
  float *partq = tile + kPartOff + q * kPartStride;  // … filterbank piece sums
  float *melq = tile + kMelOff + q * kMelStride;     // … log mel energies

  // samples of frame f as packed pairs (x[2m] | x[2m+1] << 16), m = i + 16 j
  auto load_frame = [&](int f, unsigned (&raw)[kJ]) {
    const int64_t start = p.snip_edges ? (int64_t)f * p.shift : (int64_t)p.shift * f + p.shift / 2 - p.win / 2;
    const bool fast = start >= 0 && start + 32 * kJ <= n && (((s0 + start) & 1) == 0);
    if (__all(fast)) {          // interior frames (all but the first and last one or two): aligned 4-byte loads
      const unsigned *xs = reinterpret_cast<const unsigned *>(x + start);
#pragma unroll
      for (int j = 0; j < kJ; j++) raw[j] = xs[16 * j + i];
      return;
    }
#pragma unroll
    for (int j = 0; j < kJ; j++) {
      unsigned both = 0;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int s = 2 * (16 * j + i) + e;
        unsigned val = 0;
        if (s < p.win) {
          int64_t si = start + s;
          while (si < 0 || si >= n) si = (si < 0) ? (-si - 1) : (2 * n - 1 - si);  // reflection at the edges
          val = (unsigned short)x[si];
        }
        both |= val << (16 * e);
      }
      raw[j] = both;
    }
  };
  // the frame a row works on (rows past the wavefront's share repeat its last frame and do not store)
  auto frame_of = [&](int it) { return fbase + min(4 * it + q, nfr - 1); };

  unsigned nxt[kJ];
  load_frame(frame_of(0), nxt);
  const int npass = (nfr + 3) >> 2;
  for (int it = 0; it < npass; it++) {
    const int f = frame_of(it);
    const bool live = 4 * it + q < nfr;
    int ti = i;
    asm volatile("" : "+v"(ti));   // table index the compiler cannot see through: the 60 table values are read per pass, not hoisted into registers
    const v2f *wnd = s_wnd + ti, *tw256 = s_tw256 + ti, *tw512 = s_tw512 + ti;
    v2f a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = (v2f){0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < kJ; j++) a[j] = (v2f){(float)(short)(nxt[j] & 0xFFFFu), (float)(short)(nxt[j] >> 16)};
    if (it + 1 < npass) load_frame(frame_of(it + 1), nxt);   // in flight while this pass is processed
    // ---- DC removal (int16-valued samples: the sum is an exact integer < 2^24 in any order)
    const bool ok_e = 2 * (16 * (kJ - 1) + i) < p.win, ok_o = 2 * (16 * (kJ - 1) + i) + 1 < p.win;   // only the last pair can lie beyond the window
    if (!ok_e) a[kJ - 1].x = 0.0f;
    if (!ok_o) a[kJ - 1].y = 0.0f;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < kJ; j++) sum += a[j].x + a[j].y;
    sum = row_sum_exact(sum);
    const float off = p.remove_dc ? (-sum / (float)p.win) : 0.0f;
#pragma unroll
    for (int j = 0; j < kJ; j++) a[j] += splat(off);
    if (!ok_e) a[kJ - 1].x = 0.0f;
    if (!ok_o) a[kJ - 1].y = 0.0f;
    // ---- use_energy: Kaldi's ProcessWindow takes the log energy here when raw_energy (default), after the window otherwise
    // (VecVec over the frame, floored at FLT_EPSILON; the sum's order is this kernel's, the result agrees to float rounding)
    float log_energy = 0.0f;
    auto frame_log_energy = [&]() {
      float e = 0.0f;
#pragma unroll
      for (int j = 0; j < kJ; j++) e = fmaf(a[j].y, a[j].y, fmaf(a[j].x, a[j].x, e));
      e = row_sum_exact(e);
      return __builtin_amdgcn_logf(fmaxf(e, 1.1920928955078125e-07f)) * 0.693147180559945309f;
    };
    if constexpr (kEnergy) { if (p.raw_energy) log_energy = frame_log_energy(); }
    // ---- pre-emphasis + window.  x[2m−1] is the odd sample of the previous lane (row_shr:1); lane 0 takes it from lane 15
    // of the previous register (row_ror:1 → kept where row_shr has no source); sample 0 uses itself (Kaldi)
    {
      float prev_e[kJ];
#pragma unroll
      for (int j = 0; j < kJ; j++) {
        const float up = a[j > 0 ? j - 1 : 0].y;
        const float carry = j == 0 ? a[0].x : dpp_perm<0x121>(up);                            // row_ror:1
        prev_e[j] = dpp_f32<0x111>(carry, a[j].y);                                            // row_shr:1, lane 0 keeps `carry`
      }
#pragma unroll
      for (int j = 0; j < kJ; j++) {
        const v2f prev = (v2f){prev_e[j], a[j].x};
        a[j] = (a[j] - splat(p.preemph) * prev) * wnd[16 * j];
      }
    }
    if constexpr (kEnergy) { if (!p.raw_energy) log_energy = frame_log_energy(); }
    // ---- 256-point complex FFT of z[m] = x[2m] + i x[2m+1]: 16-point DFTs over j, twiddle, transpose, 16-point DFTs over i
    dft16<kJ>(a);
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) a[k1] = cmul(a[k1], tw256[16 * k1]);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) tq[k1 * kRowPad + i] = a[k1];
    WAVE_SYNC();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) a[n2] = tq[i * kRowPad + n2];
    WAVE_SYNC();
    dft16(a);                        // a[k2] = Z[i + 16 k2]
    // ---- real-FFT post-processing → power spectrum.  Z[256 − k] sits in lane 16 − i, register 15 − k2 (lane 0: its own
    // register 16 − k2): row_mirror brings lane 15 − i, row_shr:1 the lane before that one, and lane 0 keeps the `old` value.
    //   2E = Z[k] + conj(Z[N−k]),  2O = −i (Z[k] − conj(Z[N−k])),  2X[k] = 2E + w^k · 2O,  P = |2X|² / 4  (halving is exact)
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
      const v2f zk = a[k2], own = a[(16 - k2) & 15], far = a[15 - k2];
      const float cx = dpp_f32<0x111>(own.x, dpp_perm<0x140>(far.x));
      const float cy = dpp_f32<0x111>(own.y, dpp_perm<0x140>(far.y));
      // 2E = (zk.x + cx, zk.y − cy), D = (zk.x − cx, zk.y + cy), 2O = −i·D: the table holds −i·w^k, so 2X = 2E + (−i w^k)·D
      float ex = zk.x + cx, ey = zk.y - cy, dx = zk.x - cx, dy = zk.y + cy;
      asm("" : "+v"(ex), "+v"(dx));   // (kept scalar: packed, the mixed signs cost three moves per pair)
      const v2f xk = (v2f){ex, ey} + cmul(tw512[16 * k2], (v2f){dx, dy});
      float pw = 0.25f * fmaf(xk.y, xk.y, xk.x * xk.x);
      if (k2 == 0 && i == 0) pw = 0.25f * (xk.x * xk.x);      // Kaldi ComputePowerSpectrum: bin 0 = DC² (the Nyquist bin is never used)
      pq[16 * k2 + i] = pw;
    }
    WAVE_SYNC();
    // ---- mel filterbank: piece (16 r + i) of the table is summed by lane i in round r
    for (int r = 0; r < p.n_rounds; r++) {
      const int piece = 16 * r + i;
      const int pc = min(piece, p.n_pieces - 1);
      const float *wv = s_melw + pc * kPieceTaps;
      const float *pv = pq + s_info[pc];
      float acc = 0.0f;
#pragma unroll
      for (int t = 0; t < kPieceTaps; t++) acc = fmaf(wv[t], pv[t], acc);
      if (piece < p.n_pieces) partq[piece] = acc;
    }
    WAVE_SYNC();
    // ---- pieces of a bin in ascending order, floor at FLT_EPSILON, log (hardware log2: ≲1e-6 absolute on values of
    // 10–25, far inside the 2e-3 the FFT leaves); lane i finishes bins i and i + 16
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int b = i + 16 * half;
      if (b < p.nbins) {
        const int info = s_info[kMaxPieces + b];
        const int first = info & 0xFF, np = info >> 8;
        float e = partq[first];
        for (int r = 1; r < p.np_max; r++) if (r < np) e += partq[first + r];
        melq[b] = __builtin_amdgcn_logf(fmaxf(e, 1.1920928955078125e-07f)) * 0.693147180559945309f;
      }
    }
    WAVE_SYNC();
    // ---- DCT-II row i + lifter
    if (i < p.nceps) {
      const float *d = s_dct + i * p.nbins;
      float acc = 0.0f;
      for (int b = 0; b < p.nbins; b++) acc = fmaf(d[b], melq[b], acc);
      float c_out = acc * lift_k;
      if constexpr (kEnergy) { if (i == 0) c_out = fmaxf(log_energy, p.log_energy_floor); }   // C0 := log energy (MfccComputer::Compute)
      if (live) out[(f0 + f) * p.nceps + i] = c_out;
    }
    WAVE_SYNC();
  }
}

float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

}  // namespace

extern "C" {

MFA_API int mfa_mfcc_configure(mfa_ctx *c, const mfa_mfcc_opts *o) {
  MFA_HIP_CHECK(c, hipSetDevice(c->device));
  int win = (int)(o->sample_frequency * 0.001f * o->frame_length_ms);
  int shift = (int)(o->sample_frequency * 0.001f * o->frame_shift_ms);
  int nfft = 1;
  while (nfft < win) nfft <<= 1;
  if (nfft != kNfft) return c->fail("MFCC kernel supports a 512-point FFT (window %d samples → %d)", win, nfft);
  if (o->num_mel_bins > kMaxBins || o->num_coefficients > kMaxCeps || o->num_coefficients > o->num_mel_bins ||
      o->num_mel_bins * o->num_coefficients > kMaxDct)
    return c->fail("unsupported num_mel_bins/num_coefficients %d/%d", o->num_mel_bins, o->num_coefficients);
  int nb = o->num_mel_bins, nc = o->num_coefficients;
  // window ("povey"), Kaldi FeatureWindowFunction
  std::vector<float> window(win);
  double a = 2.0 * M_PI / (win - 1);
  for (int i = 0; i < win; i++) window[i] = (float)pow(0.5 - 0.5 * cos(a * (double)i), 0.85);
  // lane tables: window pairs at [j][i] for m = i + 16 j; W256^(i·k1) at [k1][i]; W512^(i + 16 k2) at [k2][i]
  std::vector<float> wpairs(16 * 16 * 2, 0.0f);
  for (int j = 0; j < 16; j++)
    for (int i = 0; i < 16; i++)
      for (int e = 0; e < 2; e++) {
        const int sidx = 2 * (16 * j + i) + e;
        wpairs[2 * (16 * j + i) + e] = sidx < win ? window[sidx] : 0.0f;
      }
  std::vector<float> tw(2 * 256 * 2);
  for (int k = 0; k < 16; k++)
    for (int i = 0; i < 16; i++) {
      const double a256 = 2.0 * M_PI * (double)(i * k) / 256.0, a512 = 2.0 * M_PI * (double)(i + 16 * k) / 512.0;
      tw[2 * (16 * k + i)] = (float)cos(a256);
      tw[2 * (16 * k + i) + 1] = (float)(-sin(a256));
      // −i · W512^k = (Im w, −Re w) with w = (cos, −sin): the real-FFT step multiplies D = Z[k] − conj(Z[N−k]) by it
      tw[512 + 2 * (16 * k + i)] = (float)(-sin(a512));
      tw[512 + 2 * (16 * k + i) + 1] = (float)(-cos(a512));
    }
  // Kaldi MelBanks (float arithmetic as in mel-computations.cc)
  float nyquist = 0.5f * o->sample_frequency;
  float low = o->low_frequency, high = o->high_frequency > 0.0f ? o->high_frequency : nyquist + o->high_frequency;
  if (low < 0.0f || low >= nyquist || high <= 0.0f || high > nyquist || high <= low)
    return c->fail("bad mel frequency range [%f, %f]", low, high);
  float fft_bin_width = o->sample_frequency / nfft;
  float mel_low = mel_scale(low), mel_high = mel_scale(high);
  float mel_delta = (mel_high - mel_low) / (nb + 1);
  std::vector<float> melw;
  std::vector<int32_t> first_of(nb), len_of(nb), woff_of(nb);
  for (int bin = 0; bin < nb; bin++) {
    float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta, right = mel_low + (bin + 2) * mel_delta;
    int first = -1, last = -1;
    std::vector<float> w(nfft / 2, 0.0f);
    for (int i = 0; i < nfft / 2; i++) {
      float mel = mel_scale(fft_bin_width * i);
      if (mel > left && mel < right) {
        w[i] = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
        if (first == -1) first = i;
        last = i;
      }
    }
    if (first == -1) return c->fail("mel bin %d is empty (num_mel_bins too large)", bin);
    first_of[bin] = first;
    len_of[bin] = last + 1 - first;
    woff_of[bin] = (int32_t)melw.size();
    melw.insert(melw.end(), w.begin() + first, w.begin() + last + 1);
  }
  // Filterbank plan: every triangle is cut into pieces of kPieceTaps consecutive FFT bins (the last one zero padded);
  // piece p is summed by lane p mod 16 of a row in round p / 16, the pieces of a bin are then added in ascending order.
  const int best_taps = kPieceTaps;
  {
    int np = 0;
    for (int b = 0; b < nb; b++) np += (len_of[b] + best_taps - 1) / best_taps;
    if (np > kMaxPieces) return c->fail("mel filterbank needs %d pieces of %d bins (kernel table holds %d)", np, best_taps, kMaxPieces);
  }
  std::vector<float> piecew;
  std::vector<int32_t> melinfo(kMaxPieces + kMaxBins, 0);
  int n_pieces = 0, np_max = 1;
  for (int b = 0; b < nb; b++) {
    const int k = (len_of[b] + best_taps - 1) / best_taps;
    melinfo[kMaxPieces + b] = n_pieces | (k << 8);
    np_max = std::max(np_max, k);
    for (int q = 0; q < k; q++, n_pieces++) {
      melinfo[n_pieces] = first_of[b] + q * best_taps;
      for (int t = 0; t < best_taps; t++) {
        const int at = q * best_taps + t;
        piecew.push_back(at < len_of[b] ? melw[woff_of[b] + at] : 0.0f);
      }
    }
  }
  std::vector<float> dct((size_t)nc * nb), lifter(nc);
  float norm0 = std::sqrt(1.0f / (float)nb), norm = std::sqrt(2.0f / (float)nb);
  for (int k = 0; k < nc; k++)
    for (int n = 0; n < nb; n++)
      dct[(size_t)k * nb + n] = (k == 0) ? norm0 : (float)(norm * std::cos((double)M_PI / nb * (n + 0.5) * k));
  for (int i = 0; i < nc; i++)
    lifter[i] = o->cepstral_lifter != 0.0f ? (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * i / o->cepstral_lifter)) : 1.0f;

  auto upload = [&](void **dptr, const void *h, size_t bytes) -> int {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    MFA_HIP_CHECK(c, hipMalloc(dptr, bytes));
    MFA_HIP_CHECK(c, hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (upload((void **)&c->d_window, wpairs.data(), wpairs.size() * 4)) return -1;
  if (upload((void **)&c->d_twiddle, tw.data(), tw.size() * 4)) return -1;
  if (upload((void **)&c->d_melw, piecew.data(), piecew.size() * 4)) return -1;
  if (upload((void **)&c->d_melidx, melinfo.data(), melinfo.size() * 4)) return -1;
  c->n_melw = n_pieces; c->mel_np_max = np_max;
  if (upload((void **)&c->d_dct, dct.data(), dct.size() * 4)) return -1;
  if (upload((void **)&c->d_lifter, lifter.data(), lifter.size() * 4)) return -1;
  c->mfcc = *o;
  c->win = win; c->shift = shift; c->nfft = nfft;
  c->mfcc_ready = true;
  return 0;
}

MFA_API int32_t mfa_mfcc_num_frames(mfa_ctx *c, int64_t n) {
  if (!c->mfcc_ready) return -1;
  if (c->mfcc.snip_edges) return n < c->win ? 0 : (int32_t)(1 + (n - c->win) / c->shift);
  return (int32_t)((n + c->shift / 2) / c->shift);
}

MFA_API int mfa_mfcc_batch(mfa_ctx *c, const int16_t *d_pcm, const int64_t *d_sample_off, const int64_t *d_frame_off,
                           int32_t n_utt, int32_t max_frames, float *d_mfcc) {
  if (!c->mfcc_ready) return c->fail("mfa_mfcc_configure has not been called");
  if (n_utt <= 0 || max_frames <= 0) return 0;
  if (n_utt > 65535) return c->fail("at most 65535 utterances per MFCC launch (got %d)", n_utt);
  MfccParams p;
  p.win = c->win; p.shift = c->shift; p.nbins = c->mfcc.num_mel_bins; p.nceps = c->mfcc.num_coefficients;
  p.snip_edges = c->mfcc.snip_edges; p.remove_dc = c->mfcc.remove_dc_offset; p.preemph = c->mfcc.preemphasis;
  p.raw_energy = c->mfcc.raw_energy;
  p.log_energy_floor = c->mfcc.energy_floor > 0.0f ? logf(c->mfcc.energy_floor) : -INFINITY;
  p.window = c->d_window; p.tw256 = c->d_twiddle; p.tw512 = c->d_twiddle + 512;
  p.melw = c->d_melw; p.melinfo = c->d_melidx; p.dct = c->d_dct; p.lifter = c->d_lifter;
  p.n_pieces = c->n_melw; p.n_rounds = (c->n_melw + 15) / 16; p.np_max = c->mel_np_max;
  dim3 grid((max_frames + kFramesPerBlock - 1) / kFramesPerBlock, n_utt);
  KernelTimer kt(c, MFA_K_MFCC);
  const bool energy = c->mfcc.use_energy != 0;       // (its own instantiations: the default kernel carries nothing for it)
  if (c->win <= 32 * 13) {
    if (energy) hipLaunchKernelGGL((mfcc_kernel<13, true>), grid, dim3(64 * kWavesPerBlock), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
    else hipLaunchKernelGGL((mfcc_kernel<13, false>), grid, dim3(64 * kWavesPerBlock), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
  } else {
    if (energy) hipLaunchKernelGGL((mfcc_kernel<16, true>), grid, dim3(64 * kWavesPerBlock), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
    else hipLaunchKernelGGL((mfcc_kernel<16, false>), grid, dim3(64 * kWavesPerBlock), 0, c->stream, p, d_pcm, d_sample_off, d_frame_off, d_mfcc);
  }
  MFA_HIP_CHECK(c, hipGetLastError());
  return 0;
}

}  // extern "C"
