// LayerNorm over channels of bf16 frame rows (HBM-bound: one read + one write of [B, T, C]), and the halo
// zeroing that keeps the frame-row layout invariant (common.h).  Replaces nn.LayerNorm calls at HF
// modeling_whisper.py:391,399,642 and /root/reference/model.py:10,27-28.
#include "common.h"

// One wave per frame row, 8 channels (16 bytes) per lane per step, statistics in fp32 (mean, then centred
// variance, as torch does), biased variance, eps inside the rsqrt.
template <int NCH, bool GELU>   // 16-byte chunks per lane (C <= NCH * 512); GELU: exact-erf GELU after the affine
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y,
                                                        long ldy, const float* __restrict__ gam,
                                                        const float* __restrict__ bet, float eps, long lead, int B, int P,
                                                        int T, int C, const bf16_t* __restrict__ x_lo, bf16_t* __restrict__ y_lo,
                                                        int n_div, const int* __restrict__ clip_T) {
  // clip_T: [B] valid frames per clip or null (rows t >= clip_T[b] are left alone)
  // x_lo / y_lo: the low halves of a residual-stream tensor carried as hi + lo (common.h, GemmArgs::res_lo); both optional
  // n_div: channels the statistics are taken over; < C when the row carries zero padding columns (gamma = beta = 0 there: the
  //        `encoder_type: none` head, whose width 80 lives in 128 columns -- model.hip, pad_head_state)
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // index over B*T valid rows
  if (r >= (long)B * T) return;
  const int b = (int)(r / T), t = (int)(r - (long)b * T);
  if (clip_T && t >= clip_T[b]) return;
  const long row = lead + (long)b * P + t;
  const bf16_t* xp = x + row * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
      const bf16x8 a = *(const bf16x8*)(xp + c0);
      if (x_lo) {
        const bf16x8 al = *(const bf16x8*)(x_lo + row * ldx + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(a[e]) + bf2f(al[e]); sum += v[i][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(a[e]); sum += v[i][e]; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
  const float mean = sum / (float)n_div;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) sq += __shfl_xor(sq, s);
  if (n_div != C) sq -= (float)(C - n_div) * mean * mean;       // the padding columns' (0 - mean)^2
  const float rstd = rsqrtf(fmaxf(sq, 0.f) / (float)n_div + eps);
  bf16_t* yp = y + row * ldy;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
      const f32x4 g0 = *(const f32x4*)(gam + c0), g1 = *(const f32x4*)(gam + c0 + 4);
      const f32x4 b0 = *(const f32x4*)(bet + c0), b1 = *(const f32x4*)(bet + c0 + 4);
      bf16x8 o, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = (v[i][e] - mean) * rstd * g0[e] + b0[e];
        float y1 = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
        if (GELU) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); }
        o[e] = f2bf(y0);
        o[4 + e] = f2bf(y1);
        ol[e] = f2bf(y0 - bf2f(o[e]));
        ol[4 + e] = f2bf(y1 - bf2f(o[4 + e]));
      }
      *(bf16x8*)(yp + c0) = o;
      if (y_lo) *(bf16x8*)(y_lo + row * ldy + c0) = ol;
    }
  }
}

template <bool GELU>
static int launch_ln(const bf16_t* x, long ldx, bf16_t* y, long ldy, const float* g, const float* b, float eps, long lead, int B,
                     int P, int T, int C, hipStream_t s, const bf16_t* x_lo, bf16_t* y_lo, int n_div, const int* clip_T) {
  const long rows = (long)B * T;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (C <= 512)
    hipLaunchKernelGGL((layernorm_kernel<1, GELU>), grid, dim3(256), 0, s, x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, x_lo, y_lo, n_div, clip_T);
  else if (C <= 1024)
    hipLaunchKernelGGL((layernorm_kernel<2, GELU>), grid, dim3(256), 0, s, x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, x_lo, y_lo, n_div, clip_T);
  else
    hipLaunchKernelGGL((layernorm_kernel<4, GELU>), grid, dim3(256), 0, s, x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, x_lo, y_lo, n_div, clip_T);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wfl_launch_layernorm_act(const bf16_t* x, long ldx, bf16_t* y, long ldy, const float* g, const float* b, float eps, long lead,
                             int B, int P, int T, int C, int gelu, hipStream_t s, const bf16_t* x_lo, bf16_t* y_lo, int n_div,
                             const int* clip_T) {
  if (C % 8 || ldx % 8 || ldy % 8 || C > 2048 || n_div < 0 || n_div > C) return -1;
  if (n_div == 0) n_div = C;
  return gelu ? launch_ln<true>(x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, s, x_lo, y_lo, n_div, clip_T)
              : launch_ln<false>(x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, s, x_lo, y_lo, n_div, clip_T);
}

int wfl_launch_layernorm(const bf16_t* x, long ldx, bf16_t* y, long ldy, const float* g, const float* b, float eps,
                         long lead, int B, int P, int T, int C, hipStream_t s) {
  return wfl_launch_layernorm_act(x, ldx, y, ldy, g, b, eps, lead, B, P, T, C, 0, s, nullptr, nullptr, 0, nullptr);
}

// ---------------------------------------------------------------------------------------------- e4m3 rows (BASELINE configs[4], round 3)
// The fp8 x fp8 GEMMs (gemm_stream.hip, A8) take their frame operand as OCP e4m3 bytes with one fp32 scale per row (row maximum -> 448):
//   layernorm_fp8_kernel   LayerNorm of a residual-stream row (hi + lo) written as e4m3 + scale    (HF modeling_whisper.py:384, 399)
//   quant_rows_fp8_kernel  bf16 rows (the attention context) -> e4m3 + scale
// One wave per row, 8 channels per lane per step; scale[row index of the buffer].
// ylo (round 4, the e4m3 PAIR of gemm_mx.hip): what the e4m3 rounding left behind, 16 x, as e4m3 again -- lo = e4m3(16 (x / s - hi))
static __device__ __forceinline__ void store_row_fp8(float (*v)[8], int nch, int C, int lane, unsigned char* yp, float* scale_out,
                                                     unsigned char* ylo = nullptr) {
  float mx = 0.f;
  for (int i = 0; i < nch; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[i][e]));
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
  const float scale = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / scale;
  for (int i = 0; i < nch; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
      float q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = fminf(fmaxf(v[i][e] * inv, -448.f), 448.f);
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
      *(uint2*)(yp + c0) = make_uint2((unsigned)w0, (unsigned)w1);
      if (ylo) {
        typedef __attribute__((ext_vector_type(2))) float f32x2_;
        const f32x2_ h0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), h1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
        const f32x2_ h2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), h3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
        const float hv[8] = {h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
        int l0 = 0, l1 = 0;
        l0 = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (q[0] - hv[0]), 16.f * (q[1] - hv[1]), l0, false);
        l0 = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (q[2] - hv[2]), 16.f * (q[3] - hv[3]), l0, true);
        l1 = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (q[4] - hv[4]), 16.f * (q[5] - hv[5]), l1, false);
        l1 = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (q[6] - hv[6]), 16.f * (q[7] - hv[7]), l1, true);
        *(uint2*)(ylo + c0) = make_uint2((unsigned)l0, (unsigned)l1);
      }
    }
  }
  if (lane == 0) *scale_out = scale;
}

template <int NCH, bool NORM>
__global__ __launch_bounds__(256) void rows_fp8_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ x_lo,
                                                       const float* __restrict__ gam, const float* __restrict__ bet, float eps, long lead,
                                                       int B, int P, int T, int C, unsigned char* __restrict__ y8, long ldy8,
                                                       float* __restrict__ scale, unsigned char* __restrict__ y8_lo) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (long)B * T) return;
  const int b = (int)(r / T), t = (int)(r - (long)b * T);
  const long row = lead + (long)b * P + t;
  const bf16_t* xp = x + row * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    if (c0 < C) {
      const bf16x8 a = *(const bf16x8*)(xp + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = bf2f(a[e]);
      if (x_lo) {
        const bf16x8 al = *(const bf16x8*)(x_lo + row * ldx + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] += bf2f(al[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  if (NORM) {                                   // (the same arithmetic as layernorm_kernel: mean, then centred variance)
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (i * 64 + lane) * 8;
      if (c0 < C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
      }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sq += __shfl_xor(sq, s);
    const float rstd = rsqrtf(fmaxf(sq, 0.f) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (i * 64 + lane) * 8;
      if (c0 < C) {
        const f32x4 g0 = *(const f32x4*)(gam + c0), g1 = *(const f32x4*)(gam + c0 + 4);
        const f32x4 b0 = *(const f32x4*)(bet + c0), b1 = *(const f32x4*)(bet + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
          v[i][4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
        }
      }
    }
  }
  store_row_fp8(v, NCH, C, lane, y8 + row * ldy8, scale + row, y8_lo ? y8_lo + row * ldy8 : nullptr);
}

// g == null: plain quantisation of the bf16 rows (x_lo is still added when given); else LayerNorm(g, b, eps) first
int wfl_launch_rows_fp8(const bf16_t* x, long ldx, const bf16_t* x_lo, const float* g, const float* b, float eps, long lead, int B, int P,
                        int T, int C, unsigned char* y8, long ldy8, float* scale, hipStream_t s, unsigned char* y8_lo) {
  if (C % 8 || ldx % 8 || ldy8 % 8 || C > 2048 || !y8 || !scale) return -1;
  const long rows = (long)B * T;
  const dim3 grid((unsigned)((rows + 3) / 4));
#define WFL_ROWS8(NCH)                                                                                                                  \
  do {                                                                                                                                  \
    if (g) hipLaunchKernelGGL((rows_fp8_kernel<NCH, true>), grid, dim3(256), 0, s, x, ldx, x_lo, g, b, eps, lead, B, P, T, C, y8, ldy8, scale, y8_lo);   \
    else hipLaunchKernelGGL((rows_fp8_kernel<NCH, false>), grid, dim3(256), 0, s, x, ldx, x_lo, g, b, eps, lead, B, P, T, C, y8, ldy8, scale, y8_lo);    \
  } while (0)
  if (C <= 512) WFL_ROWS8(1); else if (C <= 1024) WFL_ROWS8(2); else WFL_ROWS8(4);
#undef WFL_ROWS8
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Zero every row that is not a valid frame: [0, lead), each clip's [T, P), and `tail_rows` rows behind the last
// clip.  ld_bytes = bytes per row (multiple of 16).
__global__ __launch_bounds__(256) void zero_halo_kernel(char* buf, long ld_bytes, long lead, int B, int P, int T,
                                                        long tail_rows) {
  const int halo = P - T;
  const long nrows = lead + (long)B * halo + tail_rows - halo;   // last clip's halo is part of the tail
  const long chunks_per_row = ld_bytes >> 4;
  const long total = nrows * chunks_per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long hr = i / chunks_per_row, ch = i - hr * chunks_per_row;
    long row;
    if (hr < lead) row = hr;
    else {
      const long k = hr - lead;
      const long b = k / halo;
      if (b < B - 1) row = lead + b * P + T + (k - b * halo);
      else row = lead + (long)(B - 1) * P + T + (k - (long)(B - 1) * halo);
    }
    *(uint4*)(buf + row * ld_bytes + ch * 16) = make_uint4(0, 0, 0, 0);
  }
}

int wfl_launch_zero_halo(bf16_t* buf, long ld_bytes, long lead, int B, int P, int T, long tail_rows, hipStream_t s) {
  if (ld_bytes % 16 || P <= T || tail_rows < P - T) return -1;
  const long nrows = lead + (long)B * (P - T) + tail_rows - (P - T);
  const long total = nrows * (ld_bytes >> 4);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_halo_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (char*)buf, ld_bytes, lead, B, P, T, tail_rows);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// bf16 frame rows -> compact fp32 [B][T][C] (parity-test output of the encoder's hidden states)
// (split, shift): compact channel c lives in row column c + (c >= split ? shift : 0) -- the padded head layout of model.hip's
// pad_head_state; split = C, shift = 0 for every other model
__global__ __launch_bounds__(256) void rows_to_f32_kernel(const bf16_t* __restrict__ x, long ldx, long lead, int B, int P, int T,
                                                          int C, float* __restrict__ out, int split, int shift,
                                                          const bf16_t* __restrict__ x_lo) {
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long o = (lead + (long)b * P + t) * ldx + c + (c >= split ? shift : 0);
    out[i] = x_lo ? bf2f(x[o]) + bf2f(x_lo[o]) : bf2f(x[o]);              // (precision high: hi + lo)
  }
}

int wfl_launch_rows_to_f32(const bf16_t* x, long ldx, long lead, int B, int P, int T, int C, float* out, hipStream_t s, int split,
                           int shift, const bf16_t* x_lo) {
  const long total = (long)B * T * C;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rows_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, lead, B, P, T, C, out, split, shift, x_lo);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// compact fp32 [B][T][C] -> bf16 frame rows (wfl_head: the caller's encoder output)
__global__ __launch_bounds__(256) void f32_to_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ x, long ldx, long lead,
                                                          int B, int P, int T, int C, int split, int shift, bf16_t* __restrict__ x_lo) {
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long o = (lead + (long)b * P + t) * ldx + c + (c >= split ? shift : 0);
    const bf16_t h = f2bf(in[i]);
    x[o] = h;
    if (x_lo) x_lo[o] = f2bf(in[i] - bf2f(h));              // (precision high: the value's low half)
  }
}

int wfl_launch_f32_to_rows(const float* in, bf16_t* x, long ldx, long lead, int B, int P, int T, int C, hipStream_t s, int split,
                           int shift, bf16_t* x_lo) {
  const long total = (long)B * T * C;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f32_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, x, ldx, lead, B, P, T, C, split, shift, x_lo);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ void fill_i32_kernel(int* dst, long n, int value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = value;
}

int wfl_launch_fill_i32(int* dst, long n, int value, hipStream_t s) {
  if (n <= 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n, value);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

int wfl_launch_copy16(void* dst, const void* src, long bytes, hipStream_t s) {
  if (bytes <= 0) return 0;
  if (bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16) return -1;
  const long n = bytes / 16;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// One launch for all frame-row buffers of a forward (same clip geometry per entry, different widths / pitches).
// (struct ZeroMulti: common.h -- one definition for the kernel and for model.hip)

__global__ __launch_bounds__(256) void zero_halo_multi_kernel(ZeroMulti z) {
  const int k = blockIdx.y;
  if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0 && z.err_word) *z.err_word = 0u;
  if (k >= z.n) return;
  char* buf = z.buf[k];
  const long ld_bytes = z.ld_bytes[k], lead = z.lead[k], tail_rows = z.tail_rows[k];
  const int P = z.P[k], T = z.T[k], B = z.B;
  const int halo = P - T;
  const long nrows = lead + (long)B * halo + tail_rows - halo;
  const long chunks_per_row = ld_bytes >> 4;
  const long total = nrows * chunks_per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long hr = i / chunks_per_row, ch = i - hr * chunks_per_row;
    long row;
    if (hr < lead) row = hr;
    else {
      const long kk = hr - lead;
      const long b = kk / halo;
      if (b < B - 1) row = lead + b * P + T + (kk - b * halo);
      else row = lead + (long)(B - 1) * P + T + (kk - (long)(B - 1) * halo);
    }
    *(uint4*)(buf + row * ld_bytes + ch * 16) = make_uint4(0, 0, 0, 0);
  }
}

int wfl_launch_zero_halo_multi(const ZeroMulti& z, hipStream_t s) {
  if (z.n <= 0 || z.n > 10) return -1;
  for (int k = 0; k < z.n; ++k)
    if (z.ld_bytes[k] % 16 || z.P[k] <= z.T[k] || z.tail_rows[k] < z.P[k] - z.T[k]) return -1;
  hipLaunchKernelGGL(zero_halo_multi_kernel, dim3(256, z.n), dim3(256), 0, s, z);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
