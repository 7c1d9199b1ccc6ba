// "model.precision: high" (round 3): the epilogue of a split-precision GEMM.
//
// In this mode every Linear / Conv1d of the forward runs as THREE bf16 MFMA passes over split operands,
//     A W^T  ~=  A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T          (x_hi = bf16(x), x_lo = bf16(x - x_hi): 16 significant bits each way),
// summed in fp32 by the fp32-output / accumulate form of the GEMM kernels (the split-precision classifier of round 2, for every layer:
// model.hip, Runner::gemm_precise), and every activation is carried as a bf16 pair hi + lo (the low halves live in a twin of the
// workspace).  tests/study_quant.py's table says what that removes: the bf16 rounding of weights (0.155 / 13 flips of 1500 on cfg2)
// and of GEMM inputs (0.094 / 10) -- what is left is the attention's bf16 q, k, v, P (0.006 / 1).  ~3x the GEMM cost; for callers
// who need the reference's `.lab` (/root/reference/infer.py:86-96, 293-307).
//
// precise_finish_kernel turns the fp32 sums into the layer's output: bias (+ per-clip bias), GLU, activation, positional table,
// residual (hi + lo) -- the operations and their order of the GEMM kernels' own epilogues (common.h, GemmArgs) -- and writes the result
// as hi AND lo.  HBM-bound element-wise work.
#include "common.h"

// (struct PreciseFinishArgs: common.h -- one definition for the kernel and for model.hip)

__global__ __launch_bounds__(256) void precise_finish_kernel(PreciseFinishArgs p) {
  const int chunks = p.n_out / 8;                                   // 8 output columns per thread
  const long total = (long)p.B * p.T * chunks;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    const long bt = i / chunks;
    const int b = (int)(bt / p.T), t = (int)(bt - (long)b * p.T);
    if (p.clip_T && t >= p.clip_T[b]) continue;
    const int n0 = ch * 8;
    const float* ar = p.acc + ((long)b * p.P + t) * p.ld_acc;
    float v[8];
    if (p.glu) {
      // output column n = 16 j + w  <-  a = acc[32 j + w], gate = acc[32 j + 16 + w]
      const int j = n0 >> 4, w0 = n0 & 15;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = ar[32 * j + w0 + e], g = ar[32 * j + 16 + w0 + e];
        if (p.bias) { a += p.bias[32 * j + w0 + e]; g += p.bias[32 * j + 16 + w0 + e]; }
        v[e] = a * sigmoidf_(g);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = ar[n0 + e];
        if (p.bias) x += p.bias[n0 + e];
        if (p.clip_bias) x += p.clip_bias[(long)p.clip_idx[b] * p.clip_ld + n0 + e];
        v[e] = x;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e];
      if (p.act == WFL_ACT_GELU) x = gelu_erf(x);
      else if (p.act == WFL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == WFL_ACT_SIGMOID) x = sigmoidf_(x);
      if (p.pos) x += bf2f(p.pos[(long)t * p.ldpos + n0 + e]);
      if (p.pos_lo) x += bf2f(p.pos_lo[(long)t * p.ldpos + n0 + e]);
      v[e] = x;
    }
    const long orow = p.c_lead + (long)b * p.c_pitch + t;
    if (p.res) {
      const bf16x8 rh = *(const bf16x8*)(p.res + orow * p.ldres + n0);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf2f(rh[e]) + p.alpha * v[e];
      if (p.res_lo) {
        const bf16x8 rl = *(const bf16x8*)(p.res_lo + orow * p.ldres + n0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(rl[e]);
      }
    }
    bf16x8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) { oh[e] = f2bf(v[e]); ol[e] = f2bf(v[e] - bf2f(oh[e])); }
    *(bf16x8*)(p.out + orow * p.ldc + n0) = oh;
    if (p.out_lo) *(bf16x8*)(p.out_lo + orow * p.ldc + n0) = ol;
  }
}

int wfl_launch_precise_finish(const PreciseFinishArgs& a, hipStream_t s) {
  if (a.n_out % 8 || a.ldc % 8 || (a.res && a.ldres % 8) || a.B <= 0 || a.T <= 0) return -1;
  const long total = (long)a.B * a.T * (a.n_out / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(precise_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
