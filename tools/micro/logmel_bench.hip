// Diagnostic (not product): phase breakdown of logmel_power_kernel from in-kernel 100 MHz stamps.
//   hipcc -O3 --offload-arch=gfx950 -DWFL_LOGMEL_STAMPS -I wfl-asr_amd/csrc tools/micro/logmel_bench.hip -o tools/micro/logmel_bench
#include "../../wfl-asr_amd/csrc/logmel.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void bench_fill_kernel(int* dst, long n, int v) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) dst[i] = v; }
int wfl_launch_fill_i32(int* dst, long n, int value, hipStream_t s) { hipLaunchKernelGGL(bench_fill_kernel, dim3(64), dim3(256), 0, s, dst, n, value); return 0; }

int main() {
  const int B = 16, L = 480000, nfr = 3000, nm = 80;
  std::vector<float> wav((size_t)B * L);
  for (size_t i = 0; i < wav.size(); ++i) wav[i] = 0.5f * sinf(0.01f * (float)(i % 7919)) + 0.1f * sinf(0.37f * (float)i);
  std::vector<float> wc(200 * 224), ws(200 * 224);
  for (size_t i = 0; i < wc.size(); ++i) { wc[i] = cosf(0.001f * i); ws[i] = sinf(0.001f * i); }
  std::vector<int> lo(nm), cnt(nm);
  const int maxw = 32;
  std::vector<float> mw((size_t)nm * maxw, 0.01f);
  for (int m = 0; m < nm; ++m) { lo[m] = m * 2; cnt[m] = 4 + m / 4; }
  float *dw, *dwc, *dws, *dmw, *raw; int *dlo, *dcnt; unsigned* cm; unsigned long long* st; bf16_t* out;
  hipMalloc(&dw, wav.size() * 4); hipMemcpy(dw, wav.data(), wav.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dwc, wc.size() * 4); hipMemcpy(dwc, wc.data(), wc.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dws, ws.size() * 4); hipMemcpy(dws, ws.data(), ws.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dmw, mw.size() * 4); hipMemcpy(dmw, mw.data(), mw.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dlo, nm * 4); hipMemcpy(dlo, lo.data(), nm * 4, hipMemcpyHostToDevice);
  hipMalloc(&dcnt, nm * 4); hipMemcpy(dcnt, cnt.data(), nm * 4, hipMemcpyHostToDevice);
  hipMalloc(&raw, (size_t)B * nfr * nm * 4); hipMalloc(&cm, B * 4);
  const int nblk = ((nfr + FT - 1) / FT) * B;
  hipMalloc(&st, (size_t)nblk * 8 * 8); hipMemset(st, 0, (size_t)nblk * 64);
  hipMalloc(&out, (size_t)(8 + B * 3040 + 64) * nm * 2);
  LogmelArgs a{};
  a.wav = dw; a.ldw = L; a.lens = nullptr; a.L = L; a.B = B; a.n_samples = L; a.n_frames = nfr; a.n_mels = nm;
  a.Wc = dwc; a.Ws = dws; a.mel_lo = dlo; a.mel_cnt = dcnt; a.mel_w = dmw; a.mel_maxw = maxw; a.raw = raw; a.clipmax = cm; a.stamps = nullptr;
  for (int i = 0; i < 3; ++i) wfl_launch_logmel(a, out, nm, 8, 3040, nullptr, 0, nullptr);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) wfl_launch_logmel(a, out, nm, 8, 3040, nullptr, 0, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  a.stamps = st;
  wfl_launch_logmel(a, out, nm, 8, 3040, nullptr, 0, nullptr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nblk * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[4] = {"stage signal", "MFMA loop (block's slowest wave incl. pw store)", "barrier", "mel projection + stores"};
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 4]); }
  printf("logmel (power + finish) %.1f us per launch; %d blocks; first block start -> last block end %.1f us\n", ms * 100.f, nblk, (t1 - t0) / 100.0);
  for (int k = 0; k < 4; ++k) {
    std::vector<double> d(nblk);
    for (int b = 0; b < nblk; ++b) d[b] = (double)(h[b * 8 + k + 1] - h[b * 8 + k]) / 100.0;
    std::sort(d.begin(), d.end());
    printf("  %-50s median %7.2f us   p90 %7.2f   max %7.2f\n", names[k], d[nblk / 2], d[nblk * 9 / 10], d[nblk - 1]);
  }
  std::vector<double> tot(nblk);
  for (int b = 0; b < nblk; ++b) tot[b] = (double)(h[b * 8 + 4] - h[b * 8]) / 100.0;
  std::sort(tot.begin(), tot.end());
  printf("  block total median %.2f us (x %.2f rounds at 3 blocks/CU = %.1f us)\n", tot[nblk / 2], nblk / 768.0, tot[nblk / 2] * nblk / 768.0);
  return 0;
}
