// Audio ingest on the GPU (/root/reference/infer.py:217-220, 234-235): decoded 16-bit PCM rows ->
// band-limited sinc resampling to the model's rate -> whole-clip peak normalisation -> float32 rows ready for the forward.
//
// The algorithm is torchaudio.functional.resample's published one (Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99), the same
// arithmetic as hostpost.hip:resample_f64 and wfl-asr_amd/audio.py:resample: float64 throughout, one sequential sum per output sample
// over the taps inside the window's support in ascending order, products and sums rounded separately (__dmul_rn / __dadd_rn: no
// contraction into FMAs, which the host build does not use either) -- so a clip resampled here equals the clip resampled by the host
// loader bit for bit (tests/test_gpu_ingest.py).  Parity with torchaudio itself stays UNPINNED (library absent, no fixtures).
// Why on the GPU: a folder of 44.1 kHz files was bound by decode + resample on the host's cores (48 k audio-s/s end to end against
// 111-117 k for 16 kHz files); the host now only copies the file's PCM bytes into a pinned row.
//
// resample_kernel   one thread per output sample; x[idx] is decoded on the fly from the int16 row ((l + r) / 2 for two channels, as
//                   decode_mono does); the float64 result goes to the workspace, the clip's peak to an ordered-integer atomicMax
// normalise_kernel  out = (float)(x / (peak + 1e-8)), the division in float64 (infer.py:235, :251)
#include "common.h"
#include <cmath>
#include <mutex>
#include <vector>
#include "wfl_asr.h"

namespace {

struct ResampleTable {          // device copies of the per-phase taps of one (orig, new) pair
  int orig = 0, nw = 0, width = 0, maxcnt = 0, dev = -1;
  int* jlo = nullptr;           // [nw] first tap of phase i (index into the dense kernel row)
  int* cnt = nullptr;           // [nw] taps inside the window's support
  double* kern = nullptr;       // [nw][maxcnt]
};

struct ResampleArgs {
  const int16_t* pcm; long ld_in;       // [B][ld_in] interleaved samples
  const int* n_in;                      // [B] frames (samples per channel)
  const int* channels;                  // [B] 1 or 2
  int orig, nw, width, maxcnt;
  const int* jlo; const int* cnt; const double* kern;
  double* tmp; long ld_tmp;             // [B][ld_tmp] resampled clip, float64
  unsigned long long* peak;             // [B] bit pattern of max |x| (non-negative doubles order like their bits)
  int out_cap;
};

// (x / 32768.0 and x / 2.0 as multiplications by 2^-15 and 2^-1: exact, hence the same values as decode_mono's divisions, without 35
//  float64 divisions per output sample)
static __device__ __forceinline__ double pcm_sample(const int16_t* row, int ch, long idx) {
  if (ch == 1) return (double)row[idx] * (1.0 / 32768.0);
  return ((double)row[2 * idx] * (1.0 / 32768.0) + (double)row[2 * idx + 1] * (1.0 / 32768.0)) * 0.5;
}

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs p) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const long len = p.n_in[b];
  const int ch = p.channels[b];
  long target = (long)ceil((double)p.nw * (double)len / (double)p.orig);
  if (target > p.out_cap) target = p.out_cap;
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (n < target) {
    const long f = n / p.nw;
    const int i = (int)(n - f * p.nw);
    const long p0 = f * p.orig + p.jlo[i] - p.width;           // x index of the phase's first tap
    const double* k = p.kern + (long)i * p.maxcnt;
    const int16_t* row = p.pcm + (long)b * p.ld_in;
    const int c = p.cnt[i];
    double acc = 0.0;
    for (int q = 0; q < c; ++q) {
      const long idx = p0 + q;
      const double s = (idx >= 0 && idx < len) ? pcm_sample(row, ch, idx) : 0.0;
      acc = __dadd_rn(acc, __dmul_rn(s, k[q]));
    }
    p.tmp[(long)b * p.ld_tmp + n] = acc;
    v = fabs(acc);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (m > 0.0) atomicMax(p.peak + b, (unsigned long long)__double_as_longlong(m));
  }
}

// The same rate in and out: the reference does not resample such a file (torchaudio returns the waveform as it is), so the clip is the
// decoded samples themselves.  (Round 4 routed 16 kHz folders through this -- half the bytes over PCIe, no decode on the host -- and
// measured it SLOWER end to end than the host loader's float rows on a box with a fast host: 105 k against 114 k audio-s/s,
// profiles/round4_e2e_upload_ab.txt; Labeler keeps the host loader for files at the model's rate, the entry point keeps the case.)
__global__ __launch_bounds__(256) void decode_kernel(ResampleArgs p) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  long target = p.n_in[b];
  if (target > p.out_cap) target = p.out_cap;
  const int ch = p.channels[b];
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (n < target) {
    const double x = pcm_sample(p.pcm + (long)b * p.ld_in, ch, n);
    p.tmp[(long)b * p.ld_tmp + n] = x;
    v = fabs(x);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (m > 0.0) atomicMax(p.peak + b, (unsigned long long)__double_as_longlong(m));
  }
}

__global__ __launch_bounds__(256) void normalise_kernel(const double* __restrict__ tmp, long ld_tmp, const unsigned long long* __restrict__ peak,
                                                        const int* __restrict__ n_in, int orig, int nw, int out_cap, float* __restrict__ out,
                                                        long ld_out) {
  const int b = blockIdx.y;
  long target = (long)ceil((double)nw * (double)n_in[b] / (double)orig);
  if (target > out_cap) target = out_cap;
  const double den = __longlong_as_double((long long)peak[b]) + 1e-8;
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < out_cap; n += (long)gridDim.x * 256)
    out[(long)b * ld_out + n] = n < target ? (float)__ddiv_rn(tmp[(long)b * ld_tmp + n], den) : 0.f;   // (the row's tail: zeros, like a padded batch row)
}

__global__ void clear_peaks_kernel(unsigned long long* peak, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) peak[i] = 0ull;
}

std::mutex g_tab_mu;
std::vector<ResampleTable> g_tabs;

// host: the taps of hostpost.hip:resample_f64, phase by phase.  (Returned BY VALUE: the cache is a vector that grows.)
bool table_for(int orig_freq, int new_freq, ResampleTable& out) {
  int a = orig_freq, b = new_freq;
  while (b) { const int t = a % b; a = b; b = t; }
  const int orig = orig_freq / a, nw = new_freq / a;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_tab_mu);
  for (const auto& t : g_tabs) if (t.orig == orig && t.nw == nw && t.dev == dev) { out = t; return true; }
  const double lpw = 6.0, rolloff = 0.99;
  const double base = (double)std::min(orig, nw) * rolloff;
  const int width = (int)std::ceil(lpw * orig / base);
  const int klen = 2 * width + orig;
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
  if (maxcnt == 0) return false;
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
  ResampleTable T;
  T.orig = orig; T.nw = nw; T.width = width; T.maxcnt = maxcnt; T.dev = dev;
  if (hipMalloc(&T.jlo, nw * sizeof(int)) != hipSuccess || hipMalloc(&T.cnt, nw * sizeof(int)) != hipSuccess ||
      hipMalloc(&T.kern, kern.size() * sizeof(double)) != hipSuccess)
    return false;
  if (hipMemcpy(T.jlo, jlo.data(), nw * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(T.cnt, cnt.data(), nw * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(T.kern, kern.data(), kern.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return false;
  g_tabs.push_back(T);
  out = T;
  return true;
}

}  // namespace

extern "C" {

int64_t wfl_resample_workspace_bytes(int32_t B, int32_t out_cap) {
  if (B <= 0 || out_cap <= 0) return -1;
  return (int64_t)B * (((int64_t)out_cap + 31) / 32 * 32) * 8 + 256 + (int64_t)B * 8;
}

int32_t wfl_resample_pcm16(const int16_t* pcm, int64_t ld_in, const int32_t* n_in, const int32_t* channels, int32_t B, int32_t orig_sr,
                           int32_t new_sr, float* out, int64_t ld_out, int32_t out_cap, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  if (!pcm || !n_in || !channels || !out || !workspace || B <= 0 || orig_sr <= 0 || new_sr <= 0 || out_cap <= 0 ||
      ld_out < out_cap || workspace_bytes < wfl_resample_workspace_bytes(B, out_cap))
    return -1;
  ResampleTable Tv;
  const bool same = orig_sr == new_sr;               // decode + normalise only
  if (same) { Tv.orig = 1; Tv.nw = 1; }
  else if (!table_for(orig_sr, new_sr, Tv)) return -2;
  const ResampleTable* T = &Tv;
  hipStream_t s = (hipStream_t)stream;
  const long ld_tmp = ((long)out_cap + 31) / 32 * 32;
  ResampleArgs a{};
  a.pcm = pcm; a.ld_in = ld_in; a.n_in = n_in; a.channels = channels;
  a.orig = T->orig; a.nw = T->nw; a.width = T->width; a.maxcnt = T->maxcnt; a.jlo = T->jlo; a.cnt = T->cnt; a.kern = T->kern;
  a.tmp = (double*)workspace; a.ld_tmp = ld_tmp;
  a.peak = (unsigned long long*)((char*)workspace + (size_t)B * ld_tmp * 8 + 256 - ((size_t)B * ld_tmp * 8) % 256);
  a.out_cap = out_cap;
  hipLaunchKernelGGL(clear_peaks_kernel, dim3((B + 255) / 256), dim3(256), 0, s, a.peak, B);
  if (same) hipLaunchKernelGGL(decode_kernel, dim3((out_cap + 255) / 256, B), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(resample_kernel, dim3((out_cap + 255) / 256, B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(normalise_kernel, dim3(256, B), dim3(256), 0, s, a.tmp, ld_tmp, a.peak, n_in, T->orig, T->nw, out_cap, out, (long)ld_out);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
