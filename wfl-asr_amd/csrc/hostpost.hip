// Host-side label logic in native code: thresholded tag ids + sub-frame offsets of one clip ->
// segments -> HTK .lab text.  Pure CPU code behind the same C ABI (no device access, no global state): at GPU labeling
// speed (thousands of 30 s clips per second per GPU) the per-frame Python loops of the reference are the bottleneck.
// Semantics follow the reference line by line and are held bit-exact to wfl-asr_amd/postprocess.py (itself pinned by
// fixtures generated from the reference's own functions, tests/golden/postprocess.json):
//   decode_bio_tags          /root/reference/utils.py:10-74
//   median filter            scipy.ndimage.median_filter(ids, size=k) as called at /root/reference/infer.py:170-171, 298-299
//   merge_adjacent_segments  /root/reference/utils.py:148-186
//   save_lab                 /root/reference/utils.py:76-81
// All times are IEEE doubles computed in the reference's order of operations ((idx + offset) * frame_duration, offset =
// the fp32 value widened to double, exactly what Python's float(tensor.item()) yields).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "wfl_asr.h"

namespace {

// scipy "reflect" (half-sample symmetric) boundary: ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...
inline int reflect_index(long i, int n) {
  const long period = 2L * n;
  long r = i % period;
  if (r < 0) r += period;
  return (int)(r >= n ? period - 1 - r : r);
}

}  // namespace

extern "C" {

int32_t wfl_host_median_filter(const int32_t* ids, int32_t n, int32_t size, int32_t* out) {
  if (!ids || !out || n < 0) return -1;
  if (size <= 1 || n == 0) {
    if (n) memcpy(out, ids, sizeof(int32_t) * (size_t)n);
    return 0;
  }
  const int left = size / 2;
  std::vector<int32_t> win((size_t)size);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < size; ++j) win[j] = ids[reflect_index((long)i - left + j, n)];
    std::nth_element(win.begin(), win.begin() + size / 2, win.end());
    out[i] = win[size / 2];
  }
  return 0;
}

int32_t wfl_host_decode_bio(const int32_t* ids, int32_t T, const float* offsets, int32_t n_off, const int32_t* kind,
                            const int32_t* phon, int32_t n_labels, double frame_duration, double* seg_start, double* seg_end,
                            int32_t* seg_ph, int32_t max_segments) {
  if (!ids || !kind || !phon || !seg_start || !seg_end || !seg_ph || T < 0) return -1;
  int n = 0;
  int open_ph = -1, open_idx = -1;
  bool overflow = false, oob = false;
  auto close = [&](int end_idx, bool at_eof) {
    const bool use_off = offsets != nullptr && (!at_eof || (open_idx < n_off && end_idx < n_off));
    double s, e;
    if (use_off && (open_idx >= n_off || end_idx >= n_off)) { oob = true; return; }   // the reference raises IndexError here
    if (use_off) {
      s = ((double)open_idx + (double)offsets[2 * (size_t)open_idx + 0]) * frame_duration;
      e = ((double)end_idx + (double)offsets[2 * (size_t)end_idx + 1]) * frame_duration;
    } else {
      s = ((double)open_idx + 0.5) * frame_duration;
      e = ((double)end_idx + 0.5) * frame_duration;
    }
    if (n < max_segments) {
      seg_start[n] = s; seg_end[n] = e; seg_ph[n] = open_ph;
      ++n;
    } else {
      overflow = true;
    }
  };
  for (int i = 0; i < T; ++i) {
    const int id = ids[i];
    if (id < 0 || id >= n_labels) return -2;
    const int k = kind[id];
    if (k == 0) {                       // "O"
      if (open_ph >= 0) { close(i, false); open_ph = -1; open_idx = -1; }
    } else if (k == 1) {                // "B-x"
      if (open_ph >= 0) close(i, false);
      open_ph = phon[id]; open_idx = i;
    } else if (k == 2) {                // "I-x": continues x, or closes the open run and opens one for x
      const int ph = phon[id];
      if (ph != open_ph) {
        if (open_ph >= 0) close(i, false);
        open_ph = ph; open_idx = i;
      }
    }                                   // any other tag: ignored, like the reference's if/elif chain
  }
  if (open_ph >= 0) close(T - 1, true);
  if (oob) return -4;
  return overflow ? -3 : n;
}

// mode: 0 none, 1 right, 2 left (both extend the earlier segment's end), 3 previous (utils.py:170-183).  In place;
// returns the new count.
int32_t wfl_host_merge_segments(double* start, double* end, int32_t* ph, int32_t n, int32_t mode) {
  if (n <= 0 || mode == 0) return n < 0 ? -1 : n;
  if (!start || !end || !ph) return -1;
  if (mode == 1 || mode == 2) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      if (i > 0 && ph[i] == ph[i - 1] && m > 0) {
        // note: compares with the ORIGINAL predecessor (the input is read before it is overwritten: m <= i)
        end[m - 1] = end[i];
      } else {
        start[m] = start[i]; end[m] = end[i]; ph[m] = ph[i];
        ++m;
      }
    }
    return m;
  }
  if (mode == 3) {
    std::vector<double> s(start, start + n), e(end, end + n);
    std::vector<int32_t> p(ph, ph + n);
    int m = 0;
    for (int i = 0; i < n; ++i) {
      if (i > 1 && p[i - 1] == p[i]) {
        if (m >= 2) {
          // merged[-1] is dropped and merged[-2] becomes (its start, this end, its label)
          --m;
          end[m - 1] = e[i];
        } else {
          start[m] = s[i]; end[m] = e[i]; ph[m] = p[i];
          ++m;
        }
      } else {
        start[m] = s[i]; end[m] = e[i]; ph[m] = p[i];
        ++m;
      }
    }
    return m;
  }
  return -1;
}

// "%d %d %s\n" per segment with int(t * 1e7) truncated toward zero, UTF-8 names; returns the byte count (or the count
// needed, when it exceeds cap: nothing beyond cap is written).
int64_t wfl_host_format_lab(const double* start, const double* end, const int32_t* ph, int32_t n, const char* const* names,
                            int32_t n_names, char* out, int64_t cap) {
  if (n < 0 || (n && (!start || !end || !ph || !names))) return -1;
  int64_t pos = 0;
  char tmp[96];
  for (int i = 0; i < n; ++i) {
    if (ph[i] < 0 || ph[i] >= n_names) return -2;
    const long long a = (long long)(start[i] * 1e7), b = (long long)(end[i] * 1e7);
    const int k = snprintf(tmp, sizeof(tmp), "%lld %lld ", a, b);
    const char* nm = names[ph[i]];
    const size_t ln = strlen(nm);
    if (out && pos + k + (int64_t)ln + 1 <= cap) {
      memcpy(out + pos, tmp, (size_t)k);
      memcpy(out + pos + k, nm, ln);
      out[pos + k + ln] = '\n';
    }
    pos += k + (int64_t)ln + 1;
  }
  return pos;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ audio ingest
// WAV decode + mono mix + peak normalisation of one clip straight into a caller buffer (normally a row of the pinned batch
// that is DMA'd to the GPU), and a threaded loop over a batch of files.  Restates wfl-asr_amd/audio.py's read_wav +
// peak_normalize (/root/reference/infer.py:217-218, 234-235: soundfile.read -> float64, audio / (max|audio| + 1e-8) in
// float64, then float32) operation for operation, so the samples are bit-identical to the Python path; anything the fast
// path does not cover (other sample rates, > 2 channels, clips longer than the row) is reported by status code and goes
// through the Python path.
#include <thread>
#include <atomic>
#include <cmath>

namespace {

struct WavInfo { int tag = 0, ch = 0, sr = 0, bits = 0; const unsigned char* pcm = nullptr; size_t pcm_bytes = 0; };

bool parse_wav(const std::vector<unsigned char>& d, WavInfo& w) {
  if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) || memcmp(d.data() + 8, "WAVE", 4)) return false;
  size_t pos = 12;
  bool have_fmt = false, have_data = false;
  auto u16 = [&](size_t o) { return (unsigned)d[o] | ((unsigned)d[o + 1] << 8); };
  auto u32 = [&](size_t o) { return (unsigned)d[o] | ((unsigned)d[o + 1] << 8) | ((unsigned)d[o + 2] << 16) | ((unsigned)d[o + 3] << 24); };
  while (pos + 8 <= d.size()) {
    const size_t size = u32(pos + 4);
    const size_t body = pos + 8;
    const size_t avail = body <= d.size() ? std::min(size, d.size() - body) : 0;
    if (!memcmp(d.data() + pos, "fmt ", 4) && avail >= 16) {
      w.tag = (int)u16(body); w.ch = (int)u16(body + 2); w.sr = (int)u32(body + 4); w.bits = (int)u16(body + 14);
      if (w.tag == 0xFFFE && avail >= 26) w.tag = (int)u16(body + 24);
      have_fmt = true;
    } else if (!memcmp(d.data() + pos, "data", 4)) {
      w.pcm = d.data() + body; w.pcm_bytes = avail;          // (a later data chunk replaces an earlier one, like the Python loop)
      have_data = true;
    }
    pos += 8 + size + (size & 1);
  }
  return have_fmt && have_data;
}

// Decode a WAV file to float64 mono samples (soundfile.read semantics, 2 channels averaged).
// status: 0 ok; 1 not a WAV / unsupported encoding / non-finite samples; 2 more than 2 channels; 4 cannot open
int decode_mono(const char* path, std::vector<double>& mono, int32_t* sr_out) {
  *sr_out = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return 4;
  std::vector<unsigned char> d;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz > 0) { d.resize((size_t)sz); if (fread(d.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 4; } }
  fclose(f);
  WavInfo w;
  if (!parse_wav(d, w)) return 1;
  *sr_out = w.sr;
  const int bps = w.bits / 8;
  if (!((w.tag == 1 && (w.bits == 8 || w.bits == 16 || w.bits == 24 || w.bits == 32)) || (w.tag == 3 && (w.bits == 32 || w.bits == 64)))) return 1;
  if (w.ch < 1) return 1;
  if (w.ch > 2) return 2;
  const size_t total = w.pcm_bytes / (size_t)bps;             // scalar samples
  const size_t frames = total / (size_t)w.ch;
  auto sample = [&](size_t i) -> double {
    const unsigned char* p = w.pcm + i * (size_t)bps;
    if (w.tag == 1) {
      if (w.bits == 8) return ((double)p[0] - 128.0) / 128.0;
      if (w.bits == 16) { const int16_t v = (int16_t)((unsigned)p[0] | ((unsigned)p[1] << 8)); return (double)v / 32768.0; }
      if (w.bits == 24) { int32_t v = (int32_t)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16)); if (v & 0x800000) v -= 0x1000000; return (double)v / 8388608.0; }
      const int32_t v = (int32_t)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
      return (double)v / 2147483648.0;
    }
    if (w.bits == 32) { float v; memcpy(&v, p, 4); return (double)v; }
    double v; memcpy(&v, p, 8); return v;
  };
  mono.resize(frames);
  if (w.tag == 1 && w.bits == 16) {                             // the common case on its own loop (integers: no NaN to look for)
    const unsigned char* p = w.pcm;
    if (w.ch == 1) {
      for (size_t i = 0; i < frames; ++i) { int16_t v; memcpy(&v, p + 2 * i, 2); mono[i] = (double)v / 32768.0; }
    } else {
      for (size_t i = 0; i < frames; ++i) {
        int16_t v0, v1;
        memcpy(&v0, p + 4 * i, 2); memcpy(&v1, p + 4 * i + 2, 2);
        mono[i] = ((double)v0 / 32768.0 + (double)v1 / 32768.0) / 2.0;
      }
    }
    return 0;
  }
  for (size_t i = 0; i < frames; ++i) {
    // mono of 2 channels = (a + b) / 2, numpy's mean over a length-2 axis
    const double x = w.ch == 1 ? sample(i) : (sample(2 * i) + sample(2 * i + 1)) / 2.0;
    if (!std::isfinite(x)) return 1;                           // NaN: numpy's max would propagate it; +-Inf: the resampler's zero-padded taps
                                                               // would turn it into NaN (Inf * 0.0) in outputs it must not reach -- both go
                                                               // to the Python path (audio.py), which follows numpy / the reference exactly
    mono[i] = x;
  }
  return 0;
}

// x / (max|x| + 1e-8) in float64 (infer.py:234-235, :115), stored as float32 (infer.py:134, 251)
void normalise_to_f32(const double* x, size_t n, float* out) {
  double peak = 0.0;
  for (size_t i = 0; i < n; ++i) { const double a = std::fabs(x[i]); if (a > peak) peak = a; }
  const double den = peak + 1e-8;
  for (size_t i = 0; i < n; ++i) out[i] = (float)(x[i] / den);
}

// Band-limited sinc resampling, Hann window, lowpass_filter_width 6, rolloff 0.99: torchaudio.functional.resample's published
// algorithm, the same arithmetic as wfl-asr_amd/audio.py:resample (parity with torchaudio itself is UNPINNED: the library is
// absent here and the reference holds no resampled fixtures).  float64 throughout.
void resample_f64(const std::vector<double>& x, int orig_freq, int new_freq, std::vector<double>& out) {
  if (orig_freq == new_freq || x.empty()) { out = x; return; }
  int a = orig_freq, b = new_freq;
  while (b) { const int t = a % b; a = b; b = t; }
  const int orig = orig_freq / a, nw = new_freq / a;
  const double lpw = 6.0, rolloff = 0.99;
  const double base = (double)std::min(orig, nw) * rolloff;
  const int width = (int)std::ceil(lpw * orig / base);
  const int klen = 2 * width + orig;
  // Per output phase only the taps inside the window's support are kept (|t| < lowpass_filter_width: about 2 * width of the klen taps
  // -- 35 of 475 for 44.1 -> 16 kHz).  The dense form multiplies the others by cos^2(pi / 2) = 3.7e-33, not 0: dropping them moves a
  // sum by less than 1e-33 of its terms, far below a float64 ulp of any audible sample, and makes the loop 13 times shorter (216 ->
  // 16 ms for a 30 s file at 44.1 kHz, which bounds the end-to-end rate of folders that are not 16 kHz).
  const double scale = base / orig;
  std::vector<int> jlo(nw), cnt(nw);
  int maxcnt = 0;
  for (int i = 0; i < nw; ++i) {
    int lo = klen, hi = -1;
    for (int j = 0; j < klen; ++j) {
      const double t = (-(double)i / nw + (double)(j - width) / orig) * base;
      if (t > -lpw && t < lpw) { if (j < lo) lo = j; hi = j; }
    }
    jlo[i] = hi >= lo ? lo : 0;
    cnt[i] = hi >= lo ? hi - lo + 1 : 0;
    maxcnt = std::max(maxcnt, cnt[i]);
  }
  std::vector<double> kern((size_t)nw * maxcnt, 0.0);
  for (int i = 0; i < nw; ++i)
    for (int q = 0; q < cnt[i]; ++q) {
      const int j = jlo[i] + q;
      double t = (-(double)i / nw + (double)(j - width) / orig) * base;
      t = std::min(std::max(t, -lpw), lpw);
      const double c = std::cos(t * M_PI / lpw / 2.0);
      const double window = c * c;
      t *= M_PI;
      kern[(size_t)i * maxcnt + q] = (t == 0.0 ? 1.0 : std::sin(t) / t) * window * scale;
    }
  const size_t length = x.size();
  const size_t xlen = length + 2 * (size_t)width + orig;       // the zero-padded signal: `width` zeros in front, width + orig behind
  const size_t nfr = (xlen - klen) / orig + 1;
  // Frames whose taps all lie inside x read it in place; the few at either end go through a zero-extended copy of their window
  // (a padded copy of the whole signal costs more than the sums: 10 MB of fresh pages per 30 s at 44.1 kHz).
  const size_t span = (size_t)klen + (size_t)maxcnt;           // what the eight-at-a-time loop below may read from a frame's start
  std::vector<double> edge(span);
  const size_t target = (size_t)std::ceil((double)nw * (double)length / orig);
  out.assign(nfr * nw, 0.0);
  // Each output is one sequential sum (taps in ascending order, as the dense form adds them): a chain of dependent FMAs.  Eight
  // phases run side by side, each over maxcnt taps -- the kernel rows are zero-padded, and x * 0.0 added to a sum leaves it as it is.
  for (size_t f = 0; f < nfr; ++f) {
    const size_t p0 = f * (size_t)orig;                        // frame start in the padded signal = x index + width
    const double* src;
    if (p0 >= (size_t)width && p0 - width + span <= length) src = x.data() + (p0 - width);
    else {
      for (size_t j = 0; j < span; ++j) {
        const size_t pj = p0 + j;
        edge[j] = (pj >= (size_t)width && pj - width < length) ? x[pj - width] : 0.0;
      }
      src = edge.data();
    }
    int i = 0;
    for (; i + 8 <= nw; i += 8) {
      const double* k = kern.data() + (size_t)i * maxcnt;
      const double* s[8];
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s[u] = src + jlo[i + u]; a[u] = 0.0; }
      for (int q = 0; q < maxcnt; ++q) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += s[u][q] * k[(size_t)u * maxcnt + q];
      }
      double* o = &out[f * nw + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) o[u] = a[u];
    }
    for (; i < nw; ++i) {
      const double* k = kern.data() + (size_t)i * maxcnt;
      const double* sj = src + jlo[i];
      const int n = cnt[i];
      double acc = 0.0;
      for (int q = 0; q < n; ++q) acc += sj[q] * k[q];
      out[f * nw + i] = acc;
    }
  }
  if (out.size() > target) out.resize(target);
}

// One file's 16-bit PCM samples as they are (interleaved), for the GPU ingest path (resample.hip).  status: 0 ok; 1 not a 16-bit PCM
// WAV; 2 more than 2 channels; 3 more than cap_samples int16 values (n_frames = the frame count); 4 cannot open
int read_pcm16_one(const char* path, int16_t* out, long cap_samples, int32_t* n_frames, int32_t* channels, int32_t* sr_out) {
  *n_frames = 0; *channels = 0; *sr_out = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return 4;
  // walk the chunk headers (parse_wav's rules: the last `data` chunk counts) and read the samples STRAIGHT into the caller's row -- the
  // row is pinned memory on the product path, so the file's bytes are copied once
  unsigned char h[12];
  if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) { fclose(f); return 1; }
  fseek(f, 0, SEEK_END);
  const long fsz = ftell(f);
  long pos = 12, data_pos = -1;
  size_t data_bytes = 0;
  int tag = 0, ch = 0, sr = 0, bits = 0;
  bool have_fmt = false;
  auto u16 = [](const unsigned char* q) { return (unsigned)q[0] | ((unsigned)q[1] << 8); };
  auto u32 = [](const unsigned char* q) { return (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24); };
  while (pos + 8 <= fsz) {
    unsigned char ck[8];
    if (fseek(f, pos, SEEK_SET) || fread(ck, 1, 8, f) != 8) break;
    const size_t size = u32(ck + 4);
    const long body = pos + 8;
    const size_t avail = body <= fsz ? std::min<size_t>(size, (size_t)(fsz - body)) : 0;
    if (!memcmp(ck, "fmt ", 4) && avail >= 16) {
      unsigned char fm[40];
      const size_t want = std::min<size_t>(avail, sizeof(fm));
      if (fread(fm, 1, want, f) != want) break;
      tag = (int)u16(fm); ch = (int)u16(fm + 2); sr = (int)u32(fm + 4); bits = (int)u16(fm + 14);
      if (tag == 0xFFFE && want >= 26) tag = (int)u16(fm + 24);
      have_fmt = true;
    } else if (!memcmp(ck, "data", 4)) {
      data_pos = body; data_bytes = avail;
    }
    pos += 8 + (long)size + (long)(size & 1);
  }
  if (!have_fmt || data_pos < 0) { fclose(f); return 1; }
  *sr_out = sr;
  if (tag != 1 || bits != 16 || ch < 1) { fclose(f); return 1; }
  if (ch > 2) { fclose(f); return 2; }
  const size_t frames = data_bytes / 2 / (size_t)ch;
  *channels = ch;
  *n_frames = (int32_t)std::min<size_t>(frames, 0x7fffffff);
  if ((long)(frames * (size_t)ch) > cap_samples) { fclose(f); return 3; }
  const size_t nbytes = frames * (size_t)ch * 2;
  const bool ok = fseek(f, data_pos, SEEK_SET) == 0 && fread(out, 1, nbytes, f) == nbytes;
  fclose(f);
  return ok ? 0 : 4;
}

// status: 0 ok; 1 not a WAV / unsupported encoding; 2 more than 2 channels; 3 longer than cap; 4 cannot open
int load_one(const char* path, float* out, long cap, int32_t* n_out, int32_t* sr_out) {
  *n_out = 0;
  std::vector<double> mono;
  const int st = decode_mono(path, mono, sr_out);
  if (st) return st;
  if ((long)mono.size() > cap) { *n_out = (int32_t)std::min<size_t>(mono.size(), 0x7fffffff); return 3; }
  normalise_to_f32(mono.data(), mono.size(), out);
  *n_out = (int32_t)mono.size();
  return 0;
}

}  // namespace

extern "C" {

int32_t wfl_host_load_wav(const char* path, float* out, int64_t cap, int32_t* n_samples, int32_t* sample_rate) {
  if (!path || !out || !n_samples || !sample_rate || cap < 0) return -1;
  return load_one(path, out, (long)cap, n_samples, sample_rate);
}

// rows: out + i * ld floats, each with room for `cap` samples; files are dealt to `threads` workers dynamically.
int32_t wfl_host_load_wavs(const char* const* paths, int32_t n, float* out, int64_t ld, int64_t cap, int32_t* n_samples,
                           int32_t* sample_rates, int32_t* status, int32_t threads) {
  if (n < 0 || (n && (!paths || !out || !n_samples || !sample_rates || !status)) || cap > ld) return -1;
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      status[i] = load_one(paths[i], out + (size_t)i * (size_t)ld, (long)cap, n_samples + i, sample_rates + i);
    }
  };
  if (threads == 1) { work(); return 0; }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work);
  for (auto& t : pool) t.join();
  return 0;
}

// rows: out + i * ld int16 values each (room for cap_samples interleaved samples); dealt to `threads` workers
int32_t wfl_host_read_pcm16(const char* const* paths, int32_t n, int16_t* out, int64_t ld, int64_t cap_samples, int32_t* n_frames,
                            int32_t* channels, int32_t* sample_rates, int32_t* status, int32_t threads) {
  if (n < 0 || (n && (!paths || !out || !n_frames || !channels || !sample_rates || !status)) || cap_samples > ld) return -1;
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      status[i] = read_pcm16_one(paths[i], out + (size_t)i * (size_t)ld, (long)cap_samples, n_frames + i, channels + i, sample_rates + i);
    }
  };
  if (threads == 1) { work(); return 0; }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work);
  for (auto& t : pool) t.join();
  return 0;
}

// The general ingest path of one file (infer.py:217-220, 234-244, 19-28, 114-115): decode, resample to target_sr, peak-normalise
// the whole clip, cut into non-overlapping chunks of chunk_samples when it is longer than that, re-normalise every chunk of a
// cut clip, store rows as float32.  rows: out + r * ld; n_rows / lens report what was written.
// status as wfl_host_load_wav, plus 5 = more than max_rows chunks (nothing written).
int32_t wfl_host_load_wav_chunks(const char* path, int32_t target_sr, int64_t chunk_samples, float* out, int64_t ld, int32_t max_rows,
                                 int32_t* n_rows, int32_t* lens, int32_t* sample_rate) {
  if (!path || !out || !n_rows || !lens || !sample_rate || chunk_samples <= 0 || chunk_samples > ld || target_sr <= 0 || max_rows <= 0)
    return -1;
  *n_rows = 0;
  std::vector<double> mono, rs;
  const int st = decode_mono(path, mono, sample_rate);
  if (st) return st;
  const std::vector<double>* x = &mono;
  if (*sample_rate != target_sr) { resample_f64(mono, *sample_rate, target_sr, rs); x = &rs; }
  const size_t n = x->size();
  // whole-clip peak normalisation (infer.py:234-235; an empty clip stays empty)
  std::vector<double> norm(n);
  {
    double peak = 0.0;
    for (size_t i = 0; i < n; ++i) { const double a = std::fabs((*x)[i]); if (a > peak) peak = a; }
    const double den = peak + 1e-8;
    for (size_t i = 0; i < n; ++i) norm[i] = (*x)[i] / den;
  }
  if (n <= (size_t)chunk_samples) {                        // <= 30 s: one item as is (infer.py:237, 251)
    for (size_t i = 0; i < n; ++i) out[i] = (float)norm[i];
    lens[0] = (int32_t)n;
    *n_rows = 1;
    return 0;
  }
  const size_t rows = (n + (size_t)chunk_samples - 1) / (size_t)chunk_samples;
  if (rows > (size_t)max_rows) { *n_rows = (int32_t)std::min<size_t>(rows, 0x7fffffff); return 5; }
  for (size_t r = 0; r < rows; ++r) {                      // split_audio (infer.py:19-28) + per-segment normalisation (infer.py:115)
    const size_t lo = r * (size_t)chunk_samples, len = std::min((size_t)chunk_samples, n - lo);
    normalise_to_f32(norm.data() + lo, len, out + r * (size_t)ld);
    lens[r] = (int32_t)len;
  }
  *n_rows = (int32_t)rows;
  return 0;
}

}  // extern "C"
