// Diagnostic (not product): is the SOURCE register of a transcendental VALU instruction (v_exp_f32 ...) safe to overwrite by the
// next VALU instruction while another wave on the same SIMD keeps the transcendental unit busy?
// Found in round 3 behind the "conv0 beside attention" corruption: conv0_group_kernel<true> contains
//     v_exp_f32 v90, v0
//     v_fma_f32 v0, |v87|, s40, v85        <- overwrites the exp's source
// and produced wrong values in lanes 48..63 of isolated rows whenever attention workgroups (v_exp_f32-heavy softmax) shared its SIMDs.
// Victim: per iteration  x -> [v_exp_f32 r, x ; DIST x s_nop 0 ; v_mov_b32 x, other]  with the sequence pinned by inline asm; r is checked
// against the same instruction run without the overwrite.  Neighbour (second stream): a kernel that issues v_exp_f32 back to back.
// The neighbour asks for 36 KiB of LDS (like attn_kernel<64, 2, true, false>) so that it cannot fill a CU's wave slots by itself.
// usage: trans_war_probe [neighbour 0|1] [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { unsigned blk, tid, it, dist, xbits, other, exp, got, exp_of_other, pad[3]; };

template <int DIST>
static __device__ __forceinline__ void exp_then_overwrite(float& r, float& x, float other) {
  if (DIST == 0) asm volatile("v_exp_f32 %0, %1\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  if (DIST == 1) asm volatile("v_exp_f32 %0, %1\n\ts_nop 0\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  if (DIST == 2) asm volatile("v_exp_f32 %0, %1\n\ts_nop 1\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  if (DIST == 4) asm volatile("v_exp_f32 %0, %1\n\ts_nop 3\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  if (DIST == 8) asm volatile("v_exp_f32 %0, %1\n\ts_nop 7\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  if (DIST == 16) asm volatile("v_exp_f32 %0, %1\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %1, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
  // 100: the overwriting instruction READS the exp's result (hardware interlock on the result), as in "consume before reuse"
  if (DIST == 100) asm volatile("v_exp_f32 %0, %1\n\ts_nop 0\n\tv_add_f32 %1, %0, %2\n\ts_nop 4" : "=&v"(r), "+v"(x) : "v"(other));
}

template <int DIST>
__global__ __launch_bounds__(256) void victim(int iters, Rec* recs, unsigned* nrec, int maxrec, unsigned* checked) {
  const unsigned tid = threadIdx.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float x0 = -0.001f * (float)((tid * 7 + it * 13 + blockIdx.x) & 4095);       // exp2 argument in (-4.1, 0]
    const float other = -7.0f - 0.001f * (float)((tid + it) & 1023);
    float x = x0, r, ref;
    asm volatile("v_exp_f32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "=&v"(ref) : "v"(x0));          // the reference: nothing touches the source
    exp_then_overwrite<DIST>(r, x, other);
    if (__float_as_uint(r) != __float_as_uint(ref)) {
      ++bad;
      const unsigned slot = atomicAdd(nrec, 1u);
      if ((int)slot < maxrec) {
        float eo;
        asm volatile("v_exp_f32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "=&v"(eo) : "v"(other));
        Rec q{};
        q.blk = blockIdx.x; q.tid = tid; q.it = it; q.dist = DIST; q.xbits = __float_as_uint(x0); q.other = __float_as_uint(other);
        q.exp = __float_as_uint(ref); q.got = __float_as_uint(r); q.exp_of_other = __float_as_uint(eo);
        recs[slot] = q;
      }
    }
    if (DIST != 100 && __float_as_uint(x) != __float_as_uint(other)) atomicAdd(nrec + 1, 1u);   // (the overwrite itself must have happened)
  }
  if (tid == 0) atomicAdd(checked, (unsigned)iters);
}

__global__ __launch_bounds__(256) void exp_hammer(int iters, float* sink) {
  extern __shared__ char hammer_lds[];                  // (36 KiB requested at launch, untouched: at most four of these blocks per CU,
                                                        //  which leaves wave slots on every SIMD for the victim's blocks)
  float a = -0.5f - 0.001f * threadIdx.x, b = -0.25f, c = -0.125f, d = -0.0625f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\ts_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      a -= 1.5f; b -= 1.25f; c -= 1.125f; d -= 1.0625f;
    }
  }
  if (a + b + c + d == 123.456f) sink[0] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int DIST>
static int run(bool nb, int iters, hipStream_t s0, hipStream_t s1, Rec* recs, unsigned* nrec, unsigned* checked, float* sink) {
  const int maxrec = 1024;
  CK(hipMemset(nrec, 0, 8));
  CK(hipMemset(checked, 0, 4));
  for (int rep = 0; rep < 20; ++rep) {
    if (nb) hipLaunchKernelGGL(exp_hammer, dim3(4096), dim3(256), 36864, s1, 1000, sink);
    hipLaunchKernelGGL(victim<DIST>, dim3(1024), dim3(256), 0, s0, iters, recs, nrec, maxrec, checked);
    CK(hipGetLastError());
  }
  CK(hipDeviceSynchronize());
  unsigned n[2], ck;
  CK(hipMemcpy(n, nrec, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&ck, checked, 4, hipMemcpyDeviceToHost));
  std::vector<Rec> r(n[0] < (unsigned)maxrec ? n[0] : maxrec);
  if (!r.empty()) CK(hipMemcpy(r.data(), recs, sizeof(Rec) * r.size(), hipMemcpyDeviceToHost));
  unsigned lanes[4] = {0, 0, 0, 0}, as_other = 0;
  for (auto& q : r) { ++lanes[(q.tid & 63) >> 4]; as_other += q.got == q.exp_of_other; }
  printf("distance %3d, exp-heavy neighbour %d: %u wrong results of %.3g checked (overwrite missing: %u); of the first %zu: lanes 0-15 %u, 16-31 %u, 32-47 %u, 48-63 %u; "
         "equal to exp2(the overwriting value): %u\n", DIST, (int)nb, n[0], (double)ck * 256.0, n[1], r.size(), lanes[0], lanes[1], lanes[2], lanes[3], as_other);
  for (size_t i = 0; i < r.size() && i < 4; ++i)
    printf("    blk %u tid %u it %u: x %08x other %08x expected %08x got %08x exp2(other) %08x\n", r[i].blk, r[i].tid, r[i].it, r[i].xbits, r[i].other,
           r[i].exp, r[i].got, r[i].exp_of_other);
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const int only_nb = argc > 1 ? atoi(argv[1]) : -1;
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  Rec* recs; unsigned *nrec, *checked; float* sink;
  CK(hipMalloc(&recs, sizeof(Rec) * 1024)); CK(hipMalloc(&nrec, 8)); CK(hipMalloc(&checked, 4)); CK(hipMalloc(&sink, 64));
  for (int nb = 0; nb < 2; ++nb) {
    if (only_nb >= 0 && nb != only_nb) continue;
    if (run<0>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<1>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<2>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<4>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<8>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<16>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
    if (run<100>(nb, iters, s0, s1, recs, nrec, checked, sink)) return 1;
  }
  return 0;
}
