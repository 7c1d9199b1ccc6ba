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
