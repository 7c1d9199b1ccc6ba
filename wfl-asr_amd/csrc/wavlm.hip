// WavLM-specific kernels (the rest of the WavLM encoder reuses gemm / attention / layernorm):
//   * waveform statistics for Wav2Vec2FeatureExtractor's do_normalize   (HF feature_extraction_wav2vec2.py:78-97)
//   * conv layer 0 of the feature encoder (Cin = 1, k = 10, stride 5) fused with its normalisation + GELU:
//       "group" checkpoints (base / base-plus): GroupNorm(C groups) = per-channel statistics over ALL time steps
//          of a clip -> pass 1 leaves sum / sum-of-squares per (clip, time block, channel), a reduce kernel adds the blocks up in
//          a fixed order (no atomics: bit-identical from run to run) WITHOUT storing the 98 MB/clip activation, pass 2 recomputes the 10-tap conv (10 MACs) and writes GELU(GN(y)) once;
//       "layer" checkpoints (large): conv + bias -> LayerNorm over channels -> GELU, one wave per time step.
//     (HF modeling_wavlm.py:675-744, 772-782)
//   * channel regrouping for the grouped positional conv (HF modeling_wavlm.py:48-90): [rows][d] -> [groups][rows][64]
//     so each group's k=128 conv becomes a contiguous-tap GEMM
//   * the gated relative-position-bias gate (HF modeling_wavlm.py:167-180)
#include "common.h"

// ---------------------------------------------------------------------------------------------- waveform statistics
// One workgroup per clip, every partial sum added in a FIXED order (no atomics: the result is bit-identical from run to run)
__global__ __launch_bounds__(1024) void wav_stats_kernel(const float* __restrict__ wav, long ldw, int L, double* __restrict__ stats,
                                                         const int* __restrict__ lens) {
  __shared__ double ps[16], pq[16];
  const int b = blockIdx.x;
  if (lens) L = lens[b];
  const float* w = wav + (long)b * ldw;
  double s = 0.0, q = 0.0;
  for (long i = threadIdx.x; i < L; i += 1024) {
    const double v = w[i];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pq[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < 16; ++i) { ts += ps[i]; tq += pq[i]; }
    stats[2 * b] = ts;
    stats[2 * b + 1] = tq;
  }
}

// (struct Conv0Args: common.h -- one definition for the kernel and for model.hip)

static __device__ __forceinline__ void wav_norm(const Conv0Args& p, int b, float& mean, float& rstd) {
  mean = 0.f; rstd = 1.f;
  if (p.wstats) {
    const int Lb = p.lens ? p.lens[b] : p.L;
    const double m = p.wstats[2 * b] / Lb;
    const double var = p.wstats[2 * b + 1] / Lb - m * m;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-7));
  }
}

#define C0_TT 512          // time steps per workgroup
// Frames leave LDS four at a time: frames 4m .. 4m+3 start at sample 20 m = byte 80 m, so seven 16-byte-aligned ds_read_b128 cover the
// 25 samples they need (round 2 read every frame's ten samples on their own).
//
// THIS TRANSLATION UNIT IS COMPILED WITH -fno-slp-vectorize (build.py).  Round 2 saw conv0's second pass produce wrong rows now and then
// while another forward's attention workgroups shared its CUs, and padded the kernel's LDS request so that it had its CUs to itself.
// Round 3 found the cause (tools/micro/conv0_probe.hip, profiles/round3_conv0_probe.txt): hipcc's SLP vectoriser
// had packed the two channels of a thread into v_pk_fma_f32 and, for the broadcast sample operand, emitted the form with
// op_sel:[0,1,0] -- the LOW lane takes the HIGH half of src1.  On gfx950 a packed-f32 instruction of that form (v_pk_fma / v_pk_mul /
// v_pk_add with op_sel[1] = 1) returns a low-half result computed as if that operand were zero, in lanes 48-63, about 3.5e-4 of the
// time, WHILE ANOTHER WAVE ON THE SAME SIMD ISSUES MFMA INSTRUCTIONS; never without such a neighbour, and never for the other operand
// selections (src0, src2, op_sel_hi, v_pk_mov_b32) -- each checked with pinned instruction sequences.  conv0 was the only kernel of the
// library that contained the form, and the only one ever seen wrong.  Without the SLP vectoriser no packed f32 instruction is left in
// this file; build.py additionally refuses any object of the library whose device code contains the form.
template <bool APPLY>
__global__ __launch_bounds__(256) void conv0_group_kernel(Conv0Args p) {
  __shared__ __attribute__((aligned(16))) float xs[C0_TT * 5 + 16];
  const int b = blockIdx.y, t0 = blockIdx.x * C0_TT;
  const int Lb = p.lens ? p.lens[b] : p.L;                      // this clip's samples and level-0 frames
  const int T0b = p.lens ? (Lb >= 10 ? (Lb - 10) / 5 + 1 : 0) : p.T0;
  const int nt = max(0, min(C0_TT, T0b - t0));                  // (0: a block beyond a shorter clip's end -- zero partial sums, no output)
  float mean, rstd;
  wav_norm(p, b, mean, rstd);
  const float* w = p.wav + (long)b * p.ldw;
  // (the whole array is filled: a group of four frames reads up to three frames past nt, whose results are dropped)
  for (int i = threadIdx.x; i < C0_TT * 5 + 16; i += 256) {
    const long s = (long)t0 * 5 + i;
    xs[i] = (i < nt * 5 + 5 && s < Lb) ? (w[s] - mean) * rstd : 0.f;
  }
  __syncthreads();
  for (int c0 = threadIdx.x * 2; c0 < p.C; c0 += 512) {
    float wa[10], wb[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) { wa[j] = p.w[c0 * 10 + j]; wb[j] = p.w[(c0 + 1) * 10 + j]; }
    float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;               // !APPLY: this time block's partial sums
    float sca = 0.f, scb = 0.f, sha = 0.f, shb = 0.f;           // APPLY: the channel's GroupNorm as y * sc + sh
    bf16_t* op = nullptr;
    if (APPLY) {
      const double* st = p.cstats + ((long)b * p.C + c0) * 2;
      const double ma = st[0] / T0b, mb = st[2] / T0b;
      const double va = st[1] / T0b - ma * ma, vb = st[3] / T0b - mb * mb;
      sca = (float)(1.0 / sqrt((va > 0 ? va : 0) + 1e-5)) * p.gamma[c0];
      scb = (float)(1.0 / sqrt((vb > 0 ? vb : 0) + 1e-5)) * p.gamma[c0 + 1];
      sha = p.beta[c0] - (float)ma * sca; shb = p.beta[c0 + 1] - (float)mb * scb;
      op = p.out + (p.lead + (long)b * p.P + t0) * p.C + c0;
    }
    for (int m = 0; m * 4 < nt; ++m) {
      float x[28];
#pragma unroll
      for (int q = 0; q < 7; ++q) *(f32x4*)(x + 4 * q) = *(const f32x4*)(xs + 20 * m + 4 * q);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int t = m * 4 + f;
        if (t >= nt) break;
        float ya = 0.f, yb = 0.f;                                // (same FMA order as ever: tap 0 first)
#pragma unroll
        for (int j = 0; j < 10; ++j) { ya = fmaf(wa[j], x[5 * f + j], ya); yb = fmaf(wb[j], x[5 * f + j], yb); }
        if (!APPLY) {
          sa += ya; qa += ya * ya; sb += yb; qb += yb * yb;
        } else {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
          bf16x2 o;
          const float ga = gelu_erf(fmaf(ya, sca, sha)), gb = gelu_erf(fmaf(yb, scb, shb));
          o[0] = f2bf(ga);
          o[1] = f2bf(gb);
          *(bf16x2*)(op + (long)t * p.C) = o;
          if (p.out_lo) {
            bf16x2 ol;
            ol[0] = f2bf(ga - bf2f(o[0]));
            ol[1] = f2bf(gb - bf2f(o[1]));
            *(bf16x2*)(p.out_lo + (op - p.out) + (long)t * p.C) = ol;
          }
        }
      }
    }
    if (!APPLY) {
      // partial sums of this time block; conv0_stats_reduce_kernel adds the blocks up in a fixed order (deterministic)
      float* st = p.cpart + (((long)b * gridDim.x + blockIdx.x) * p.C + c0) * 2;
      *(f32x4*)st = (f32x4){sa, qa, sb, qb};
    }
  }
}

__global__ __launch_bounds__(256) void conv0_stats_reduce_kernel(const float* __restrict__ part, int nblk, int C, double* __restrict__ cstats) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < nblk; ++k) {
    const float2 v = *(const float2*)(part + (((long)b * nblk + k) * C + c) * 2);
    s += (double)v.x;
    q += (double)v.y;
  }
  cstats[((long)b * C + c) * 2] = s;
  cstats[((long)b * C + c) * 2 + 1] = q;
}

// "layer" mode: one wave per time step, 8 channels per lane (C <= 512), LayerNorm over channels, GELU
__global__ __launch_bounds__(256) void conv0_layer_kernel(Conv0Args p) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform time step: scalar sample loads)
  const int b = blockIdx.y;
  float mean, rstd;
  wav_norm(p, b, mean, rstd);
  const int c0 = lane * 8;
  const bool act = c0 < p.C;
  float wt[8][10], bs[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = act ? c0 + e : 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) wt[e][j] = p.w[c * 10 + j];
    bs[e] = p.bias ? p.bias[c] : 0.f;
    gm[e] = p.gamma[c];
    bt[e] = p.beta[c];
  }
  const float* w = p.wav + (long)b * p.ldw;
  const int Lb = p.lens ? p.lens[b] : p.L;
  const int T0b = p.lens ? (Lb >= 10 ? (Lb - 10) / 5 + 1 : 0) : p.T0;
  // The samples of a time step are loaded one iteration ahead: the chain load -> conv -> two wave reductions -> GELU -> store is
  // latency-bound (round 3: 3.44 -> 2.56 ms at 64 x 10 s, WavLM-large).  NT = 2 independent steps per iteration measured 2.88 ms.
  constexpr int NT = 1;
  const int stride = gridDim.x * 4;
  auto samples = [&](int t, float* x) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const long s = (long)t * 5 + j;
      x[j] = (t < T0b && s < Lb) ? w[s] : mean;              // ((mean - mean) * rstd = 0 behind the clip's end, as before)
    }
  };
  float xn[NT][10];
#pragma unroll
  for (int k = 0; k < NT; ++k) samples(blockIdx.x * 4 + wid + k * stride, xn[k]);
  for (int tb = blockIdx.x * 4 + wid; tb < T0b; tb += NT * stride) {
    float x[NT][10], y[NT][8], sum[NT], sq[NT], mu[NT], rs[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
#pragma unroll
      for (int j = 0; j < 10; ++j) x[k][j] = (xn[k][j] - mean) * rstd;
      samples(tb + (NT + k) * stride, xn[k]);
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      sum[k] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = bs[e];
#pragma unroll
        for (int j = 0; j < 10; ++j) a = fmaf(wt[e][j], x[k][j], a);
        y[k][e] = act ? a : 0.f;
        sum[k] += y[k][e];
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
      for (int k = 0; k < NT; ++k) sum[k] += __shfl_xor(sum[k], o);
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      mu[k] = sum[k] / p.C;
      sq[k] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = act ? y[k][e] - mu[k] : 0.f; sq[k] += d * d; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
      for (int k = 0; k < NT; ++k) sq[k] += __shfl_xor(sq[k], o);
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int t = tb + k * stride;
      rs[k] = rsqrtf(sq[k] / p.C + 1e-5f);
      if (act && t < T0b) {
        bf16x8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = gelu_erf((y[k][e] - mu[k]) * rs[k] * gm[e] + bt[e]);
          o[e] = f2bf(v);
          ol[e] = f2bf(v - bf2f(o[e]));
        }
        *(bf16x8*)(p.out + (p.lead + (long)b * p.P + t) * p.C + c0) = o;
        if (p.out_lo) *(bf16x8*)(p.out_lo + (p.lead + (long)b * p.P + t) * p.C + c0) = ol;
      }
    }
  }
}

int wfl_launch_wav_stats(const float* wav, long ldw, int B, int L, double* stats, hipStream_t s, const int* lens) {
  hipLaunchKernelGGL(wav_stats_kernel, dim3(B), dim3(1024), 0, s, wav, ldw, L, stats, lens);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wfl_launch_conv0(const Conv0Args& a, int group_norm, hipStream_t s) {
  if (a.C % 8 || a.C > 512 || a.T0 <= 0) return -1;
  if (group_norm) {
    if (!a.cpart) return -1;
    dim3 grid((a.T0 + C0_TT - 1) / C0_TT, a.B);
    // (no LDS padding any more: see the note at conv0_group_kernel)
    hipLaunchKernelGGL(conv0_group_kernel<false>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(conv0_stats_reduce_kernel, dim3((a.C + 255) / 256, a.B), dim3(256), 0, s, a.cpart, (int)grid.x, a.C, a.cstats);
    hipLaunchKernelGGL(conv0_group_kernel<true>, grid, dim3(256), 0, s, a);
  } else {
    int bx = (a.T0 + 3) / 4;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(conv0_layer_kernel, dim3(bx, a.B), dim3(256), 0, s, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------- per-clip frame counts
// lens[b] samples -> frames of clip b at every level of a conv stack without padding: t <- (t - kernel) / stride + 1 (0 when shorter
// than the kernel); out[level][b].  One level with kernel 0 / stride hop gives the mel front-end's 1 + len / hop ... (len - 0) / hop + 1.
struct ClipFramesArgs { const int* lens; int B, L, n; int kernel[8], stride[8]; int min_len; int* out; };
__global__ void clip_frames_kernel(ClipFramesArgs p) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= p.B) return;
  long t = min(max(p.lens[b], 0), p.L);
  if (t < p.min_len) t = -1;                        // too short for the front-end: no frames at any level
  for (int i = 0; i < p.n; ++i) {
    t = t >= p.kernel[i] && t >= 0 ? (t - p.kernel[i]) / p.stride[i] + 1 : 0;
    p.out[i * p.B + b] = (int)t;
  }
}
int wfl_launch_clip_frames(const int* lens, int B, int L, int n, const int* kernel, const int* stride, int min_len, int* out, hipStream_t s) {
  if (n <= 0 || n > 8 || B <= 0) return -1;
  ClipFramesArgs a{};
  a.lens = lens; a.B = B; a.L = L; a.n = n; a.min_len = min_len; a.out = out;
  for (int i = 0; i < n; ++i) { a.kernel[i] = kernel[i]; a.stride[i] = stride[i]; }
  hipLaunchKernelGGL(clip_frames_kernel, dim3((B + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------- positional-conv regroup
// x [R rows][d] (frame rows) -> xg [groups][R][64]: channels g*cpg .. +cpg of every valid frame, zero elsewhere
// (padding channels, halo rows): each group's k-tap conv then reads k*64 contiguous elements per frame.
__global__ __launch_bounds__(256) void regroup_kernel(const bf16_t* __restrict__ x, int d, int groups, int cpg, long R, long lead,
                                                      int B, int P, int T, bf16_t* __restrict__ xg, const int* __restrict__ clip_T) {
  const long total = (long)groups * R * 8;     // 16-byte chunks
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i & 7);
    const long gr = i >> 3;
    const long r = gr % R;
    const int g = (int)(gr / R);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
    const long k = r - lead;
    if (k >= 0 && k < (long)B * P && (k % P) < (clip_T ? clip_T[k / P] : T)) {
      const int c0 = ch * 8;
      if (c0 + 8 <= cpg) v = *(const bf16x8*)(x + r * d + g * cpg + c0);
      else if (c0 < cpg) {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c0 + e < cpg) v[e] = x[r * d + g * cpg + c0 + e];
      }
    }
    *(bf16x8*)(xg + (gr << 6) + ch * 8) = v;
  }
}

int wfl_launch_regroup(const bf16_t* x, int d, int groups, int cpg, long R, long lead, int B, int P, int T, bf16_t* xg, hipStream_t s,
                       const int* clip_T) {
  if (cpg > 64 || cpg % 8 || (cpg * groups) != d) return -1;
  const long total = (long)groups * R * 8;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, d, groups, cpg, R, lead, B, P, T, xg, clip_T);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------- rel-pos gate
// gate[b][h][t] = ga * (gb * const_h - 1) + 2,  (ga, gb) = sigmoid( (W8 x_head + b8).view(2, 4).sum(-1) )
__global__ __launch_bounds__(256) void relpos_gate_kernel(const bf16_t* __restrict__ x, long ldx, long lead, int B, int P, int T,
                                                          int heads, int hd, const float* __restrict__ w8,
                                                          const float* __restrict__ b8, const float* __restrict__ cst,
                                                          float* __restrict__ gate, const bf16_t* __restrict__ x_lo) {
  // the 8 x hd projection every thread applies: staged in LDS once per workgroup (it used to be re-read from global memory, 8 hd
  // uniform loads per thread), transposed to [k][8] so that one k costs two broadcast ds_read_b128
  __shared__ __attribute__((aligned(16))) float sw[128 * 8];
  for (int i = threadIdx.x; i < 8 * hd; i += 256) sw[(i % hd) * 8 + i / hd] = w8[i];
  __syncthreads();
  // NP positions (b, t, head) per thread.  55.6 us per call at 64 x 499 frames x 16 heads with the projection read from global memory,
  // 48.7 from LDS; NP = 4 (one pair of broadcast reads serving four frames) measured 76-80 us: too few waves left to hide the loads
  constexpr int NP = 1;
  const long total = (long)B * T * heads;
  const long S = (long)gridDim.x * 256;             // positions i0, i0 + S, ...: neighbouring lanes read neighbouring head slices
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += S * NP) {
    const bf16_t* xp[NP];
    long go[NP];
    int hh[NP];
    float acc[NP][8];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const long i = i0 + q * S < total ? i0 + q * S : total - 1;  // (spare positions behind the end: computed, not stored)
      const int h = (int)(i % heads);
      const long bt = i / heads;
      const int t = (int)(bt % T), b = (int)(bt / T);
      xp[q] = x + (lead + (long)b * P + t) * ldx + h * hd;
      go[q] = ((long)b * heads + h) * T + t;
      hh[q] = h;
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[q][o] = b8[o];
    }
    for (int k = 0; k < hd; k += 8) {
      bf16x8 v[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) v[q] = *(const bf16x8*)(xp[q] + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x4 wa = *(const f32x4*)(sw + (k + e) * 8), wb = *(const f32x4*)(sw + (k + e) * 8 + 4);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const float xv = bf2f(v[q][e]);
#pragma unroll
          for (int o = 0; o < 4; ++o) { acc[q][o] = fmaf(wa[o], xv, acc[q][o]); acc[q][4 + o] = fmaf(wb[o], xv, acc[q][4 + o]); }
        }
      }
      if (x_lo) {                                   // (precision high: the hidden states' low halves)
#pragma unroll
        for (int q = 0; q < NP; ++q) v[q] = *(const bf16x8*)(x_lo + (xp[q] - x) + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f32x4 wa = *(const f32x4*)(sw + (k + e) * 8), wb = *(const f32x4*)(sw + (k + e) * 8 + 4);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const float xv = bf2f(v[q][e]);
#pragma unroll
            for (int o = 0; o < 4; ++o) { acc[q][o] = fmaf(wa[o], xv, acc[q][o]); acc[q][4 + o] = fmaf(wb[o], xv, acc[q][4 + o]); }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (i0 + q * S >= total) break;
      const float ga = sigmoidf_(acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3]);
      const float gb = sigmoidf_(acc[q][4] + acc[q][5] + acc[q][6] + acc[q][7]);
      gate[go[q]] = ga * (gb * cst[hh[q]] - 1.0f) + 2.0f;
    }
  }
}

int wfl_launch_relpos_gate(const bf16_t* x, long ldx, long lead, int B, int P, int T, int heads, int hd, const float* w8,
                           const float* b8, const float* cst, float* gate, hipStream_t s, const bf16_t* x_lo) {
  if (hd % 8 || ldx % 8 || hd > 128) return -1;
  const long total = (long)B * T * heads;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relpos_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, lead, B, P, T, heads, hd, w8, b8, cst, gate, x_lo);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// table[h][delta + T - 1] = rel_emb[bucket(delta)][h]   for delta = key - query in [-(T-1), T-1]
__global__ __launch_bounds__(256) void relpos_table_kernel(const float* __restrict__ rel_emb, const int* __restrict__ bod, int max_t,
                                                           int heads, int T, float* __restrict__ table) {
  const int n = 2 * T - 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * heads; i += gridDim.x * 256) {
    const int h = i / n, j = i - h * n;
    const int delta = j - (T - 1);
    table[i] = rel_emb[bod[delta + max_t - 1] * heads + h];
  }
}

int wfl_launch_relpos_table(const float* rel_emb, const int* bucket_of_delta, int max_t, int heads, int T, float* table, hipStream_t s) {
  if (T > max_t) return -1;
  const int n = (2 * T - 1) * heads;
  hipLaunchKernelGGL(relpos_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rel_emb, bucket_of_delta, max_t, heads, T, table);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
