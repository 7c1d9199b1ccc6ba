// Diagnostic (not product): DESIGN.md section 8 "next (1)" put to the test -- the 192 x 256 x 32 bf16 GEMM tile computed by FOUR waves
// (one per SIMD, up to 512 registers each, 96 x 128 outputs per wave: 192 accumulator registers) instead of eight in two ping-pong groups,
// with the fragments double-buffered in registers: the ds_reads of step s + 1 are issued among the 48 MFMAs of step s, one s_barrier per
// step (ring hand-over only), 14 fragment reads per 48 MFMAs instead of 10 per 24.  Same LDS image (64-byte rows, 16-byte chunk swizzle),
// same LDS-DMA staging (28 KiB per step, seven 1 KiB pieces per wave) and ring as gemm256.hip / gemm_stream.hip, so its K loop is
// comparable with theirs (tools/gemm_lab.py on the same box).  One tile per block, plain bf16 epilogue from registers.
//   usage: gemm4w [N] [K] [NS]      (M = 16 x 1520 rows; prints wall time, K-loop time per 32-deep step, K-loop clock, a spot check)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define BK 32
#define XB (192 * BK * 2)          // frame tile bytes
#define STB (XB + 256 * BK * 2)    // stage bytes (28 KiB)

static __device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }
static __device__ __forceinline__ void glds(const bf16_t* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

template <int NS, bool NOSTAGE, bool NOMMA>
__global__ __launch_bounds__(256) void gemm4w_kernel(const bf16_t* A, long lda, const bf16_t* W, int M, int N, int K, bf16_t* C, long ldc,
                                                     unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int tiles_n = N / 256;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 192, n0 = tn * 256;
  const int wm = (wid >> 1) * 96, wn = (wid & 1) * 128;
  if (tid == 0) stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();

  // staging: the stage's 28 pieces of 1 KiB (12 frame row groups, then 16 weight row groups), seven per wave
  const bf16_t* src[7];
  int dst[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int pc = wid * 7 + i;
    const int rloc = lane >> 2;
    if (pc < 12) {
      const int row = pc * 16 + rloc;
      int am = m0 + row;
      am = am < M ? am : M - 1;
      src[i] = A + (long)am * lda + ((lane & 3) ^ swz(row)) * 8;
      dst[i] = pc * 1024;
    } else {
      const int row = (pc - 12) * 16 + rloc;
      src[i] = W + (long)(n0 + row) * K + ((lane & 3) ^ swz(row)) * 8;
      dst[i] = XB + (pc - 12) * 1024;
    }
  }
  const int nk = K / BK;
  int iss = 0, islot = 0;
  auto stage = [&]() __attribute__((always_inline)) {
    if (iss >= nk) return;
    char* base = smem + islot * STB;
#pragma unroll
    for (int i = 0; i < 7; ++i) glds(src[i] + iss * BK, base + dst[i]);
    ++iss;
    islot = islot + 1 == NS ? 0 : islot + 1;
  };
  auto wait_stage = [&](int need) __attribute__((always_inline)) {     // this wave's pieces of stage `need` have landed
    if (need >= nk) return;
    const int y = iss - need - 1;
    if (y >= 3) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if (y == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (y == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int frag = c * 64 + ((g ^ swz(c)) << 4);
  const int x_off = wm * 64 + frag, w_off = XB + wn * 64 + frag;

  f32x4 acc[6][8];
#pragma unroll
  for (int u = 0; u < 6; ++u)
#pragma unroll
    for (int v = 0; v < 8; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 X0[6], X1[6], W0[8], W1[8];
#define RD(XD, WD, slot)                                                                                              \
  _Pragma("unroll") for (int v_ = 0; v_ < 8; ++v_) WD[v_] = *(const bf16x8*)(smem + (slot) * STB + w_off + v_ * 1024);   \
  _Pragma("unroll") for (int u_ = 0; u_ < 6; ++u_) XD[u_] = *(const bf16x8*)(smem + (slot) * STB + x_off + u_ * 1024)
#define MM(XS, WS, u0, u1)                                                                                            \
  _Pragma("unroll") for (int u_ = (u0); u_ < (u1); ++u_)                                                             \
    _Pragma("unroll") for (int v_ = 0; v_ < 8; ++v_)                                                                 \
      if (NOMMA) { asm volatile("" :: "v"(XS[u_]), "v"(WS[v_])); }                                                    \
      else acc[u_][v_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WS[v_], XS[u_], acc[u_][v_], 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
  // one step: MFMAs of step s on (XS, WS); the fragments of step s + 1 are read into (XD, WD) behind the first row group's MFMAs
  // hand-over in front of it: stage s + 1 visible to all waves, the slot read during step s - 1 (stage s) free -> stage s + NS - 1
#define STEP(XS, WS, XD, WD, sidx, rslot)                                                                             \
  wait_stage((sidx) + 1);                                                                                            \
  __builtin_amdgcn_s_barrier();                                                                                      \
  SB();                                                                                                              \
  if (!NOSTAGE) stage(); else { if (iss < nk) { ++iss; islot = islot + 1 == NS ? 0 : islot + 1; } }                  \
  SB();                                                                                                              \
  MM(XS, WS, 0, 1); SB();                                                                                            \
  RD(XD, WD, rslot); SB();                                                                                           \
  MM(XS, WS, 1, 6); SB()

#pragma unroll
  for (int t = 0; t < NS - 1; ++t) stage();
  wait_stage(0);
  __builtin_amdgcn_s_barrier();
  RD(X0, W0, 0);
  unsigned long long t1 = 0, c1 = 0;
  if (tid == 0) { t1 = __builtin_amdgcn_s_memrealtime(); c1 = __builtin_amdgcn_s_memtime(); }
  int s1 = 1;                                       // slot of stage s + 1
  for (int s = 0; s < nk; s += 2) {
    const int s2 = s1 + 1 == NS ? 0 : s1 + 1;
    STEP(X0, W0, X1, W1, s, s1);
    STEP(X1, W1, X0, W0, s + 1, s2);
    s1 = s2 + 1 == NS ? 0 : s2 + 1;
  }
  if (tid == 0) {
    stamps[blockIdx.x * 8 + 1] = t1;
    stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
    stamps[blockIdx.x * 8 + 5] = c1;
    stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
  }
  // epilogue: lane (g, c) holds channels 4g .. 4g + 3 of frame c of every 16 x 16 tile
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int m = m0 + wm + u * 16 + c;
    if (m < M) {
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc[u][v][e];
        *(bf16x4*)(C + (long)m * ldc + n0 + wn + v * 16 + 4 * g) = o;
      }
    }
  }
  if (tid == 0) stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
}

template <int NS, bool NOSTAGE, bool NOMMA>
static void run(const char* name, const bf16_t* A, const bf16_t* W, bf16_t* C, unsigned long long* stamps, int M, int N, int K,
                const std::vector<float>& hA, const std::vector<float>& hW) {
  const int tiles = ((M + 191) / 192) * (N / 256);
  const int lds = NS * STB;
  auto k = gemm4w_kernel<NS, NOSTAGE, NOMMA>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("attr failed\n"); return; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, (long)K, W, M, N, K, C, (long)N, stamps);
  hipDeviceSynchronize();
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, (long)K, W, M, N, K, C, (long)N, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms / 20); sum += ms / 20;
  }
  std::vector<unsigned long long> st(tiles * 8);
  hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> kl, clk, tot;
  for (int b = 0; b < tiles; ++b) {
    kl.push_back((st[b * 8 + 2] - st[b * 8 + 1]) / 100.0);
    clk.push_back((double)(st[b * 8 + 6] - st[b * 8 + 5]) / ((st[b * 8 + 2] - st[b * 8 + 1]) * 10.0));
    tot.push_back((st[b * 8 + 4] - st[b * 8 + 0]) / 100.0);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  // spot check (only meaningful for the full build)
  double maxerr = 0;
  if (!NOSTAGE && !NOMMA) {
    std::vector<bf16_t> hC((size_t)M * N);
    hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
    for (int t = 0; t < 400; ++t) {
      const int m = (int)((1009L * t + 7) % M), n = (int)((617L * t + 3) % N);
      double ref = 0;
      for (int kk = 0; kk < K; ++kk) ref += (double)hA[(size_t)m * K + kk] * hW[(size_t)n * K + kk];
      maxerr = fmax(maxerr, fabs((double)(float)hC[(size_t)m * N + n] - ref) / (fabs(ref) + 1.0));
    }
  }
  const double us = sum / 5 * 1000, fl = 2.0 * M * N * K;
  printf("  %-10s NS=%d  %7.1f us (min %7.1f) %6.0f TF | blocks %4d | K loop %7.2f us = %.3f us per 32-deep step | K-loop clock %.2f GHz | block total %6.2f us | spot err %.2g\n",
         name, NS, us, best * 1000, fl / us / 1e6, tiles, med(kl), med(kl) / (K / BK), med(clk), med(tot), maxerr);
}

#include <algorithm>
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512, K = argc > 2 ? atoi(argv[2]) : 2048;
  const int M = 16 * 1520;
  if (N % 256 || K % 64) { printf("N %% 256, K %% 64\n"); return 1; }
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<bf16_t> bA(hA.size()), bW(hW.size());
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (size_t i = 0; i < hA.size(); ++i) { bA[i] = (bf16_t)rnd(); hA[i] = (float)bA[i]; }
  for (size_t i = 0; i < hW.size(); ++i) { bW[i] = (bf16_t)(rnd() * 0.1f); hW[i] = (float)bW[i]; }
  bf16_t *A, *W, *C;
  unsigned long long* stamps;
  hipMalloc(&A, bA.size() * 2 + 4096); hipMalloc(&W, bW.size() * 2 + 4096); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&stamps, 8192 * 8 * 8);
  hipMemcpy(A, bA.data(), bA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, bW.data(), bW.size() * 2, hipMemcpyHostToDevice);
  hipMemset(stamps, 0, 8192 * 8 * 8);
  printf("--- four-wave tile, M=%d N=%d K=%d\n", M, N, K);
  run<4, false, false>("stock", A, W, C, stamps, M, N, K, hA, hW);
  run<5, false, false>("stock", A, W, C, stamps, M, N, K, hA, hW);
  run<5, true, false>("nostage", A, W, C, stamps, M, N, K, hA, hW);
  run<5, false, true>("nomma", A, W, C, stamps, M, N, K, hA, hW);
  return 0;
}
