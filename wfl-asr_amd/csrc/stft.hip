// The acoustic features of /root/reference/correct_label.py:15-24 on the GPU (the boundary-snapping step
// the reference's notebook runs right after inference): per clip
//   flux[t]    = || |STFT_512(y)|[:, t] - |STFT_512(y)|[:, t-1] ||_2        (librosa.stft(y, n_fft=512, hop_length=160), flux[0] = 0)
//   mfcc[c][t] = DCT-II_ortho( 10 log10(max(1e-10, mel_128 . |STFT_2048(y)|^2)) floored at max - 80 )[c],  c < 13
//                (librosa.feature.mfcc(y, sr, n_mfcc=13, hop_length=160): n_fft 2048, 128 Slaney mel bands, top_db 80)
// Both STFTs are centred with zero padding (librosa >= 0.10) and use the periodic Hann window.  The reference computes them with
// librosa 0.11, which is not available here and for which the reference holds no fixtures: the arithmetic is restated from librosa's
// documented definitions (wfl-asr_amd/correct_label.py holds the same restatement in numpy; tests/test_gpu_correct_label.py holds this
// file to it) and its parity with librosa itself is UNPINNED.
//
// stft_kernel<NFFT>   the log-mel front-end's scheme (logmel.hip) for any n_fft: one workgroup = 32 frames of one clip, the
//                     (31 hop + n_fft)-sample segment staged once in LDS (skewed by one word per hop so that the 32 frames of an MFMA
//                     operand read fall into 32 banks), the real DFT as a GEMM against Hann-folded cos | -sin tables on the exact-fp32
//                     MFMA (v_mfma_f32_32x32x2_f32), magnitude or power out.  fp32 like librosa's own float32 STFT.
// flux_kernel, mel_db_kernel, mfcc_kernel   one wave per frame / one thread per (frame, band) / per (frame, coefficient): HBM-bound,
//                     a few MB per clip.
// The mel filter bank and the DCT matrix come from the caller (the numpy restatement builds them once), so both paths share them.
#include "common.h"
#include <cmath>
#include <mutex>
#include <vector>
#include "wfl_asr.h"

namespace {

constexpr int SHOP = 160;        // hop_length of both STFTs (correct_label.py:15)
constexpr int SFT = 32;          // frames per workgroup

struct StftArgs {
  const float* wav; long ldw;    // [B][ldw]
  const int* lens;               // [B] samples per clip or null (= L)
  int L, B, n_frames;            // frames per clip = 1 + L / hop (the batch-wide count; a shorter clip's frames beyond 1 + len / hop are zero)
  const float* Wc; const float* Ws;   // [n_fft][nb_pad] window-folded cos / -sin
  int nbins, nb_pad;
  float* out;                    // [B][n_frames][nbins]
  int power;                     // 1: |X|^2, 0: |X|
};

template <int NFFT>
__global__ __launch_bounds__(256) void stft_kernel(StftArgs p) {
  constexpr int SEG = (SFT - 1) * SHOP + NFFT;
  constexpr int SEG_LDS = SEG + SEG / SHOP + 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* seg = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, f0 = blockIdx.x * SFT;
  const int len = p.lens ? min(p.lens[b], p.L) : p.L;
  const int nfr = 1 + len / SHOP;
  const float* w = p.wav + (long)b * p.ldw;
  // centre = True, zero padding: frame f covers samples f * hop - n_fft / 2 .. + n_fft
  const long s0 = (long)f0 * SHOP - NFFT / 2;
  for (int j = tid; j < SEG; j += 256) {
    const long i = s0 + j;
    seg[j + j / SHOP] = (i >= 0 && i < len) ? w[i] : 0.f;
  }
  __syncthreads();
  const int r = lane & 31, kh = lane >> 5;
  const int ntile = p.nb_pad / 32;
  for (int ct = wid; ct < ntile; ct += 4) {
    f32x16 re, im;
#pragma unroll
    for (int e = 0; e < 16; ++e) { re[e] = 0.f; im[e] = 0.f; }
    const float* fa = seg + (SHOP + 1) * r;                        // frame r (skewed: + n + n / hop)
    const float* bc = p.Wc + (long)kh * p.nb_pad + ct * 32 + r;
    const float* bs = p.Ws + (long)kh * p.nb_pad + ct * 32 + r;
#pragma unroll 8
    for (int n = 0; n < NFFT; n += 2) {
      const int nn = n + kh;
      const float x = fa[nn + nn / SHOP];
      const float c = bc[(long)n * p.nb_pad], sn = bs[(long)n * p.nb_pad];
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(x, c, re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(x, sn, im, 0, 0, 0);
    }
    // D[row][col]: col = lane & 31 (bin), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (frame of the block)
    const int bin = ct * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int fr = f0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (fr < p.n_frames && bin < p.nbins) {
        const float pw = re[e] * re[e] + im[e] * im[e];
        p.out[((long)b * p.n_frames + fr) * p.nbins + bin] = fr < nfr ? (p.power ? pw : sqrtf(pw)) : 0.f;
      }
    }
  }
}

// flux[b][t] = sqrt(sum_k (S[t][k] - S[t-1][k])^2), flux[b][0] = 0 (np.pad(flux, (1,)) then cut to the frame count).  A clip shorter
// than the batch (lens) has 1 + len / hop frames: the frames behind them are zero in S, and so is the flux there -- in particular at the
// first such frame, where the difference to the clip's last real frame would otherwise show up as a boundary that does not exist.
__global__ __launch_bounds__(256) void flux_kernel(const float* __restrict__ S, int B, int F, int nbins, const int* __restrict__ lens,
                                                   float* __restrict__ flux) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= (long)B * F) return;
  const int t = (int)(i % F);
  const int nf = lens ? 1 + lens[i / F] / SHOP : F;
  float acc = 0.f;
  if (t > 0 && t < nf) {
    const float* a = S + i * nbins;
    const float* p = a - nbins;
    for (int k = lane; k < nbins; k += 64) { const float d = a[k] - p[k]; acc = fmaf(d, d, acc); }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) flux[i] = sqrtf(acc);
}

static __device__ __forceinline__ unsigned f2ord_(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static __device__ __forceinline__ float ord2f_(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// logmel[b][t][m] = 10 log10(max(1e-10, sum_k mel_w[m][k] P[t][k])); per-clip maximum into an ordered-uint atomicMax (power_to_db's top_db)
__global__ __launch_bounds__(256) void mel_db_kernel(const float* __restrict__ P, int B, int F, int nbins, const float* __restrict__ mel_w,
                                                     int n_mels, const int* __restrict__ lens, int L, float* __restrict__ logmel,
                                                     unsigned* __restrict__ clipmax) {
  const long total = (long)B * F * n_mels;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float lv = -INFINITY;
  int b = 0;
  if (i < total) {
    const int m = (int)(i % n_mels);
    const long bt = i / n_mels;
    b = (int)(bt / F);
    const int t = (int)(bt - (long)b * F);
    const float* pr = P + bt * nbins;
    const float* wr = mel_w + (long)m * nbins;
    float acc = 0.f;
    for (int k = 0; k < nbins; ++k) acc = fmaf(wr[k], pr[k], acc);
    lv = 10.0f * log10f(fmaxf(acc, 1e-10f));
    logmel[i] = lv;
    const int len = lens ? min(lens[b], L) : L;
    if (t >= 1 + len / SHOP) lv = -INFINITY;                      // frames behind a shorter clip do not take part in its maximum
  }
  // (a workgroup may straddle two clips: one atomic per lane that holds a value is fine at this size)
  if (lv > -INFINITY) atomicMax(clipmax + b, f2ord_(lv));
}

// mfcc[b][c][t] = sum_m dct[c][m] * max(logmel[b][t][m], max_b - 80)
__global__ __launch_bounds__(256) void mfcc_kernel(const float* __restrict__ logmel, const unsigned* __restrict__ clipmax, int B, int F,
                                                   int n_mels, const float* __restrict__ dct, int n_mfcc, float* __restrict__ mfcc) {
  const long total = (long)B * F * n_mfcc;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % n_mfcc);
  const long bt = i / n_mfcc;
  const int b = (int)(bt / F), t = (int)(bt - (long)b * F);
  const float floorv = ord2f_(clipmax[b]) - 80.0f;
  const float* lr = logmel + bt * n_mels;
  const float* dr = dct + (long)c * n_mels;
  float acc = 0.f;
  for (int m = 0; m < n_mels; ++m) acc = fmaf(dr[m], fmaxf(lr[m], floorv), acc);
  mfcc[((long)b * n_mfcc + c) * F + t] = acc;
}

__global__ void clear_u32_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

struct Twiddle { int nfft = 0, dev = -1, nbins = 0, nb_pad = 0; float* Wc = nullptr; float* Ws = nullptr; };
std::mutex g_tw_mu;
std::vector<Twiddle> g_tw;

// (returned BY VALUE: the cache is a vector that grows, a pointer into it would dangle after the next table is added)
bool twiddles(int nfft, Twiddle& out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_tw_mu);
  for (const auto& t : g_tw) if (t.nfft == nfft && t.dev == dev) { out = t; return true; }
  Twiddle T;
  T.nfft = nfft; T.dev = dev; T.nbins = nfft / 2 + 1; T.nb_pad = (T.nbins + 31) / 32 * 32;
  std::vector<float> wc((size_t)nfft * T.nb_pad, 0.f), ws((size_t)nfft * T.nb_pad, 0.f);
  for (int n = 0; n < nfft; ++n) {
    const double hann = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)nfft);        // periodic Hann (scipy get_window("hann", N))
    for (int k = 0; k < T.nbins; ++k) {
      const double ang = 2.0 * M_PI * (double)(((long)n * k) % nfft) / (double)nfft;
      wc[(size_t)n * T.nb_pad + k] = (float)(hann * std::cos(ang));
      ws[(size_t)n * T.nb_pad + k] = (float)(-hann * std::sin(ang));
    }
  }
  if (hipMalloc(&T.Wc, wc.size() * 4) != hipSuccess || hipMalloc(&T.Ws, ws.size() * 4) != hipSuccess) return false;
  if (hipMemcpy(T.Wc, wc.data(), wc.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(T.Ws, ws.data(), ws.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
    return false;
  g_tw.push_back(T);
  out = T;
  return true;
}

template <int NFFT>
int launch_stft(StftArgs a, hipStream_t s) {
  constexpr int SEG = (SFT - 1) * SHOP + NFFT;
  constexpr int lds = (SEG + SEG / SHOP + 2 + 3) / 4 * 4 * 4;
  auto k = stft_kernel<NFFT>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  hipLaunchKernelGGL(k, dim3((a.n_frames + SFT - 1) / SFT, a.B), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" {

int64_t wfl_boundary_workspace_bytes(int32_t B, int32_t L) {
  if (B <= 0 || L <= 0) return -1;
  const int64_t F = 1 + L / SHOP;
  return (int64_t)B * F * (257 + 1025 + 128) * 4 + 1024 + (int64_t)B * 4;
}

int32_t wfl_boundary_features(const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L, const float* mel_w,
                              const float* dct, float* flux, float* mfcc, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!wav || !mel_w || !dct || !flux || !mfcc || !workspace || B <= 0 || L <= 0 || ldw < L ||
      workspace_bytes < wfl_boundary_workspace_bytes(B, L))
    return -1;
  hipStream_t s = (hipStream_t)stream;
  Twiddle t512, t2048;
  if (!twiddles(512, t512) || !twiddles(2048, t2048)) return -2;
  const int F = 1 + L / SHOP;
  float* S = (float*)workspace;                                  // [B][F][257] magnitude
  float* P = S + (size_t)B * F * 257;                            // [B][F][1025] power
  float* LM = P + (size_t)B * F * 1025;                          // [B][F][128] log-mel (dB)
  unsigned* cmax = (unsigned*)(((uintptr_t)(LM + (size_t)B * F * 128) + 255) / 256 * 256);
  StftArgs a{};
  a.wav = wav; a.ldw = ldw; a.lens = lens; a.L = L; a.B = B; a.n_frames = F;
  a.Wc = t512.Wc; a.Ws = t512.Ws; a.nbins = 257; a.nb_pad = t512.nb_pad; a.out = S; a.power = 0;
  if (int r = launch_stft<512>(a, s)) return r;
  a.Wc = t2048.Wc; a.Ws = t2048.Ws; a.nbins = 1025; a.nb_pad = t2048.nb_pad; a.out = P; a.power = 1;
  if (int r = launch_stft<2048>(a, s)) return r;
  const long rows = (long)B * F;
  hipLaunchKernelGGL(flux_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, B, F, 257, lens, flux);
  hipLaunchKernelGGL(clear_u32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, cmax, B);
  hipLaunchKernelGGL(mel_db_kernel, dim3((unsigned)((rows * 128 + 255) / 256)), dim3(256), 0, s, P, B, F, 1025, mel_w, 128, lens, L, LM, cmax);
  hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)((rows * 13 + 255) / 256)), dim3(256), 0, s, LM, cmax, B, F, 128, dct, 13, mfcc);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
