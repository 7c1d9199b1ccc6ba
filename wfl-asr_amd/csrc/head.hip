// Per-frame tag decision and sub-frame boundary offsets (HBM-bound, one wave per frame).
//   tag decision: /root/reference/infer.py:86-96 suppress_low_confidence (softmax -> max prob, argmax;
//                 prob < threshold => "O") fused with model.py:196-198 decode_predictions
//   offsets:      /root/reference/model.py:139-141 Conv1d(d -> 2, k=1) + Sigmoid on the GELU'd k=3 conv output
#include "common.h"

// (struct TagArgs: common.h -- one definition for the kernel and for model.hip)

__global__ __launch_bounds__(256) void tag_decide_kernel(TagArgs p) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0 && p.status_dst) *p.status_dst = p.status_src ? (int)*p.status_src : 0;
  if (r >= p.rows) return;
  if (p.clip_T) {
    const int b = (int)(r / p.Tmax), t = (int)(r - (long)b * p.Tmax);
    if (t >= p.clip_T[b]) {
      if (lane == 0) {
        if (p.logits) { p.maxprob[r] = 0.f; p.ids[r] = p.o_id; if (p.argmax) p.argmax[r] = p.o_id; }
        if (p.offsets) { p.offsets[r * 2] = 0.f; p.offsets[r * 2 + 1] = 0.f; }
      }
      return;
    }
  }
  if (p.logits) {
    const float* lp = p.logits + r * p.ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < p.C; c += 64) {
      const float v = lp[c];
      if (v > best) { best = v; bi = c; }
    }
    // wave arg-max, ties -> lowest index (torch.max / argmax return the first maximal element)
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const float ov = __shfl_xor(best, s);
      const int oi = __shfl_xor(bi, s);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    float se = 0.f;
    for (int c = lane; c < p.C; c += 64) se += __expf(lp[c] - best);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) se += __shfl_xor(se, s);
    if (lane == 0) {
      const float mp = 1.0f / se;
      p.maxprob[r] = mp;
      if (p.argmax) p.argmax[r] = bi;
      p.ids[r] = mp < p.threshold ? p.o_id : bi;
    }
  }
  if (p.offsets) {
    const int b = (int)(r / p.T), t = (int)(r - (long)b * p.T);
    const bf16_t* hp = p.hid + (p.lead + (long)b * p.P + t) * p.ldh;
    float a0 = 0.f, a1 = 0.f;
    for (int c0 = lane * 8; c0 < p.d; c0 += 512) {
      const bf16x8 hv = *(const bf16x8*)(hp + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = bf2f(hv[e]);
        a0 += x * p.w2[c0 + e];
        a1 += x * p.w2[p.d + c0 + e];
      }
      if (p.hid_lo) {
        const bf16x8 lv = *(const bf16x8*)(p.hid_lo + (hp - p.hid) + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = bf2f(lv[e]);
          a0 += x * p.w2[c0 + e];
          a1 += x * p.w2[p.d + c0 + e];
        }
      }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { a0 += __shfl_xor(a0, s); a1 += __shfl_xor(a1, s); }
    if (lane == 0) {
      p.offsets[r * 2] = sigmoidf_(a0 + p.b2[0]);
      p.offsets[r * 2 + 1] = sigmoidf_(a1 + p.b2[1]);
    }
  }
}

int wfl_launch_tag_decide(const TagArgs& a, hipStream_t s) {
  if (a.rows <= 0) return -1;
  if (a.offsets && (a.d % 8 || a.ldh % 8)) return -1;
  hipLaunchKernelGGL(tag_decide_kernel, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// dst = (init ? 0 : dst) + alpha * src  — multi-language averaging of logits / offsets
// (/root/reference/infer.py:146-156, 266-276: one forward per language id, mean of logits and of offsets)
__global__ __launch_bounds__(256) void axpy_kernel(float* dst, const float* src, long n, float alpha, int init) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dst[i] = (init ? 0.f : dst[i]) + alpha * src[i];
}

int wfl_launch_axpy(float* dst, const float* src, long n, float alpha, int init, hipStream_t s) {
  if (n <= 0) return -1;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, n, alpha, init);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
