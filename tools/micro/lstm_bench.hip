// Diagnostic (not product): time per recurrence step of lstm_kernel and, with -DWFL_LSTM_STAMPS, where a step goes.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 [-DWFL_LSTM_STAMPS] [-DWFL_LSTM_NOWAIT] -I wfl-asr_amd/csrc \
//         tools/micro/lstm_bench.hip -o tools/micro/lstm_bench
// usage: lstm_bench [H=256] [B=16] [T=1500]
#include "../../wfl-asr_amd/csrc/lstm.hip"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <vector>

__global__ void fill_i32_kernel(int* dst, long n, int value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = value;
}
int wfl_launch_fill_i32(int* dst, long n, int value, hipStream_t s) {
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n, value);
  return 0;
}

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 256, B = argc > 2 ? atoi(argv[2]) : 16, T = argc > 3 ? atoi(argv[3]) : 1500;
  const int P = T + 20, lead = 16, d = 2 * H;
  const long R = lead + (long)B * P + 256;
  std::vector<float> gx((size_t)R * 8 * H);
  unsigned x = 12345;
  for (auto& v : gx) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 2.f - 1.f; }
  std::vector<unsigned short> whh((size_t)2 * 4 * H * H);
  for (auto& v : whh) { x = x * 1664525u + 1013904223u; const float f = (((x >> 8) & 0xffff) / 65536.0f * 2.f - 1.f) / 16.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  float* dgx; bf16_t *dw, *dout; void* ex; unsigned* err; unsigned long long* st;
  hipMalloc(&dgx, gx.size() * 4); hipMemcpy(dgx, gx.data(), gx.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dw, whh.size() * 2); hipMemcpy(dw, whh.data(), whh.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&dout, (size_t)R * d * 2); hipMemset(dout, 0, (size_t)R * d * 2);
  hipMalloc(&ex, wfl_lstm_exchange_bytes(H, B));
  hipMalloc(&err, 256); hipMemset(err, 0, 256);
  hipMalloc(&st, 32 * 8 * 8); hipMemset(st, 0, 32 * 8 * 8);
  LstmArgs a{};
  a.gx = dgx; a.ldgx = 8 * H; a.whh = dw; a.out = dout; a.ldo = d; a.lead = lead; a.B = B; a.T = T; a.P = P; a.H = H;
  a.U = wfl_lstm_units_per_wg(H); a.error = err;
#ifdef WFL_LSTM_STAMPS
  a.stamps = st;
  a.team_shift = getenv("LSTM_TEAM_SHIFT") ? atoi(getenv("LSTM_TEAM_SHIFT")) : 0;
#endif
  for (int i = 0; i < 2; ++i) { int r = wfl_launch_lstm(a, ex, 0); if (r) { printf("launch failed %d\n", r); return 1; } }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = 4;
  for (int i = 0; i < reps; ++i) wfl_launch_lstm(a, ex, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
  printf("H %d B %d T %d U %d G %d: %.3f ms per launch = %.3f us per step (error word %u)\n", H, B, T, a.U, H / a.U, ms / reps, 1e3 * ms / reps / T, e);
  {   // the roll call of the last launch: which XCD every slice of every (direction, group) team ran on
    const int groups = (B + 15) / 16, G = H / a.U;
    std::vector<unsigned long long> roll((size_t)2 * groups * 64);
    hipMemcpy(roll.data(), (unsigned long long*)ex + 2L * groups * 4 * 16 * (H / 2), roll.size() * 8, hipMemcpyDeviceToHost);
    int together = 0;
    for (int t = 0; t < 2 * groups; ++t) {
      bool same = true;
      for (int sl = 1; sl < G; ++sl) same = same && (unsigned)roll[(size_t)t * 64 + sl] == (unsigned)roll[(size_t)t * 64];
      together += same;
      if (getenv("LSTM_BENCH_ROLL")) {
        printf("  team dir %d group %d: XCC", t / groups, t % groups);
        for (int sl = 0; sl < G; ++sl) printf(" %u", (unsigned)roll[(size_t)t * 64 + sl] - 1u);
        printf("\n");
      }
    }
    printf("  teams on one XCD: %d of %d\n", together, 2 * groups);
  }
#ifdef WFL_LSTM_STAMPS
  std::vector<unsigned long long> h(32 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  const char* nm[5] = {"poll (loads until every tag matches)", "strip tags -> LDS + barrier", "fragment reads + MFMA", "cell update + publish + out store", "loop back (gx refill, next step start)"};
  for (int k = 0; k < 5; ++k) {
    std::vector<double> v;
    for (int s = 0; s < 31; ++s) {
      const unsigned long long a0 = h[s * 8 + k], a1 = k < 4 ? h[s * 8 + k + 1] : h[(s + 1) * 8];
      v.push_back((double)(a1 - a0) / 100.0);
    }
    std::sort(v.begin(), v.end());
    printf("  %-45s median %6.2f us  min %6.2f  max %6.2f\n", nm[k], v[v.size() / 2], v[0], v.back());
  }
#endif
  return 0;
}
