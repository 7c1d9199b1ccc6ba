// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (gfx950): which k a lane's 32 bytes are, which lane's scale byte
// multiplies them, and the instruction's issue rate.  (round 4; tools/micro/build.sh; run on the GPU box, prints a report)
//   test 1  A = B = 1.0 everywhere, scale_a = 2^(c % 4) for lane (c, g) -> D[i][j] must be 128 * 2^(i % 4): the A scale of lane (i, *)
//           applies to row i, 32 k's per lane
//   test 2  A = 1.0 only in byte t of lane group g (all rows), B = 1.0, scale_a = 2^g' for lane group g': D = 2^(g of the bytes) for
//           every (g, t) iff a lane's scale applies to exactly that lane's own 32 bytes
//   test 3  the same for the B side
//   test 4  A (lane g, byte t) = distinct small integers, B nonzero only at (lane g2, byte t2): D != 0 iff (g, t) pairs with the
//           same (g2, t2): operands contract lane-for-lane, byte-for-byte
//   test 5  op_sel: scale bytes 1..3 of the scale register
//   test 6  cycles per instruction, back to back, one wave per SIMD and two
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void mfma_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* d, int opsel) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  if (opsel == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 1, sa[l], 1, sb[l]);
  if (opsel == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 2, sa[l], 2, sb[l]);
  if (opsel == 3) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 3, sa[l], 3, sb[l]);
  d[l] = acc;
}

template <int NACC>
__global__ void mfma_rate(const i32x8* a, const i32x8* b, f32x4* d, long long* cyc, int iters) {
  const int l = threadIdx.x & 63;
  i32x8 av = a[l], bv = b[l];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int s = 127;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[i], 0, 0, 0, s, 0, s);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 r = acc[0];
  for (int i = 1; i < NACC; ++i) r += acc[i];
  d[threadIdx.x] = r;
  if (l == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void mfma_rate_nonscaled(const long* a, const long* b, f32x4* d, long long* cyc, int iters) {
  const int l = threadIdx.x & 63;
  long av = a[l], bv = b[l];
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av, bv, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 r = acc[0];
  for (int i = 1; i < 8; ++i) r += acc[i];
  d[threadIdx.x] = r;
  if (l == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static const unsigned char ONE = 0x38;      // e4m3 1.0

int main() {
  unsigned char *dA, *dB; int *dsa, *dsb; float* dD; long long* dc;
  CK(hipMalloc(&dA, 64 * 32)); CK(hipMalloc(&dB, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dD, 64 * 16 * 64));
  CK(hipMalloc(&dc, 8 * 4096));
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  std::vector<int> sa(64), sb(64);
  std::vector<float> D(64 * 4);
  auto run = [&](int opsel) -> int {
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, (const i32x8*)dA, (const i32x8*)dB, dsa, dsb, (f32x4*)dD, opsel);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 64 * 16, hipMemcpyDeviceToHost));
    return 0;
  };
  auto Dij = [&](int i, int j) { return D[(j + 16 * (i / 4)) * 4 + (i % 4)]; };   // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
  // ---- test 1
  std::fill(A.begin(), A.end(), ONE); std::fill(B.begin(), B.end(), ONE);
  for (int l = 0; l < 64; ++l) { sa[l] = 127 + ((l & 15) % 4); sb[l] = 127; }
  if (run(0)) return 1;
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (Dij(i, j) != 128.f * (1 << (i % 4))) ++bad;
  printf("test1 (row scale from lane (i, *), C/D map): %s  D[0][0]=%g D[1][0]=%g D[3][5]=%g\n", bad ? "FAIL" : "ok", Dij(0, 0), Dij(1, 0), Dij(3, 5));
  // ---- test 2 / 3: own-lane scale
  for (int side = 0; side < 2; ++side) {
    bad = 0;
    for (int g = 0; g < 4; ++g) for (int t = 0; t < 32; ++t) {
      std::vector<unsigned char>& X = side ? B : A; std::vector<unsigned char>& Y = side ? A : B;
      std::fill(X.begin(), X.end(), 0); std::fill(Y.begin(), Y.end(), ONE);
      for (int c = 0; c < 16; ++c) X[(g * 16 + c) * 32 + t] = ONE;
      for (int l = 0; l < 64; ++l) { (side ? sb : sa)[l] = 127 + (l >> 4); (side ? sa : sb)[l] = 127; }
      if (run(0)) return 1;
      const float want = (float)(1 << g);
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (Dij(i, j) != want) { if (bad < 4) printf("  side %d g %d t %d: D[%d][%d] = %g, want %g\n", side, g, t, i, j, Dij(i, j), want); ++bad; }
    }
    printf("test%d (%s: a lane's scale multiplies its own 32 bytes): %s\n", 2 + side, side ? "B" : "A", bad ? "FAIL" : "ok");
  }
  // ---- test 4: contraction pairs (g, t) with (g, t)
  bad = 0;
  for (int g = 0; g < 4; ++g) for (int t = 0; t < 32; t += 5) {
    std::fill(B.begin(), B.end(), 0);
    for (int c = 0; c < 16; ++c) B[(g * 16 + c) * 32 + t] = ONE;
    for (int l = 0; l < 64; ++l) for (int u = 0; u < 32; ++u) A[l * 32 + u] = ((l >> 4) == g && u == t) ? 0x40 /* 2.0 */ : ONE;
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 127;
    if (run(0)) return 1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (Dij(i, j) != 2.f) ++bad;
  }
  printf("test4 (lane-for-lane, byte-for-byte contraction): %s\n", bad ? "FAIL" : "ok");
  // ---- test 5: op_sel picks the scale byte
  std::fill(A.begin(), A.end(), ONE); std::fill(B.begin(), B.end(), ONE);
  bad = 0;
  for (int os = 0; os < 4; ++os) {
    for (int l = 0; l < 64; ++l) { sa[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24); sb[l] = 127 | (126 << 8) | (127 << 16) | (125 << 24); }
    if (run(os)) return 1;
    const float want[4] = {128.f, 128.f * 2 / 2, 128.f * 4, 128.f * 8 / 4};
    if (Dij(2, 3) != want[os]) { ++bad; printf("  op_sel %d: D = %g, want %g\n", os, Dij(2, 3), want[os]); }
  }
  printf("test5 (op_sel = scale byte index): %s\n", bad ? "FAIL" : "ok");
  // ---- test 6: rate
  for (int waves = 1; waves <= 2; ++waves) {
    const int iters = 4000;
    hipLaunchKernelGGL(mfma_rate<8>, dim3(256), dim3(256 * waves), 0, 0, (const i32x8*)dA, (const i32x8*)dB, (f32x4*)dD, dc, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> c(256 * 4 * waves);
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : c) m += (double)v; m /= c.size();
    printf("test6 scaled 16x16x128 fp8, %d wave(s)/SIMD: %.1f cycles (s_memtime ticks) per MFMA per wave\n", waves, m / (iters * 8.0));
    hipLaunchKernelGGL(mfma_rate_nonscaled, dim3(256), dim3(256 * waves), 0, 0, (const long*)dA, (const long*)dB, (f32x4*)dD, dc, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    m = 0; for (auto v : c) m += (double)v; m /= c.size();
    printf("test6 non-scaled 16x16x32 fp8, %d wave(s)/SIMD: %.1f ticks per MFMA per wave\n", waves, m / (iters * 8.0));
  }
  return 0;
}
