// Flash-style attention for the Conformer block's LARGE heads: head_dim = d / conformer_heads = 384 (Whisper-small),
// 512 (WavLM-large), 640 (Whisper-large) with the reference's default conformer_heads = 2 (/root/reference/model.py:26,
// config.yaml:28).  Same algorithm and operand layouts as attention.hip (S^T = K.Q^T, P kept in registers as the B
// operand of O^T = V^T.P^T, K tile XOR-swizzled, row-major V tile read through ds_read_b64_tr_b16), but the register budget is spent on the output
// accumulators (head_dim / 4 VGPRs per lane), so each wave owns ONE 16-query tile, Q fragments are re-read from L2
// for every key tile instead of living in registers, and head_dim 640 uses 32-key tiles to fit K and V in LDS.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4b_t;
typedef __attribute__((address_space(3))) s16x4b_t* lds_s16x4b_t;
static __device__ __forceinline__ bf16x4 ds_read_tr_big(const char* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4b_t)p));
}

template <int HD, int KTL>
__global__ __launch_bounds__(256) void attn_big_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CPR = HD / 8;              // 16-byte chunks per K row (multiple of 16)
  constexpr int VP = HD * 2 + 32;          // V tile row pitch in bytes (see attention.hip)
  constexpr int KS = HD / 32, DT = HD / 16, NKK = KTL / 16, NS2 = KTL / 32;
  char* Ks = smem;
  char* Vs = smem + KTL * HD * 2;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int nqb = (p.T + 63) / 64;
  const int nblk = nqb * p.heads * p.B;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int qb = bid % nqb, bh = bid / nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int T = p.clip_T ? p.clip_T[b] : p.T;      // (a batch of clips of different lengths: this block's clip, attention.hip)
  if (qb * 64 >= T) return;
  const int q0 = qb * 64 + wid * 16;
  const long row0 = p.lead + (long)b * p.P;
  const bf16_t* Kg = p.QK + p.d + h * HD;
  const bf16_t* Vg = p.V + h * HD;
  int qrow = q0 + c;
  qrow = qrow < T ? qrow : T - 1;
  const bf16_t* qp = p.QK + (row0 + qrow) * p.ldqk + h * HD + g * 8;

  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;
  const int ntiles = (T + KTL - 1) / KTL;

  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
#pragma unroll 4
    for (int ch = tid; ch < KTL * CPR; ch += 256) {
      const int r = ch / CPR, cc = ch - r * CPR;
      *(bf16x8*)(Ks + r * (HD * 2) + ((cc ^ (r & 15)) << 4)) = *(const bf16x8*)(Kg + (row0 + kt * KTL + r) * p.ldqk + cc * 8);
    }
#pragma unroll 4
    for (int ch = tid; ch < KTL * CPR; ch += 256) {
      const int r = ch / CPR, cc = ch - r * CPR;
      *(bf16x8*)(Vs + r * VP + cc * 16) = *(const bf16x8*)(Vg + (row0 + kt * KTL + r) * p.ldv + cc * 8);
    }
    __syncthreads();

    f32x4 st[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) st[kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 qf = *(const bf16x8*)(qp + ks * 32);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int r = kk * 16 + c;
        const bf16x8 kf = *(const bf16x8*)(Ks + r * (HD * 2) + (((ks * 4 + g) ^ (r & 15)) << 4));
        st[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, st[kk], 0, 0, 0);
      }
    }
    if (kt * KTL + KTL > T) {
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kt * KTL + kk * 16 + g * 4 + e >= T) st[kk][e] = -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, st[kk][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    float ps = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __builtin_amdgcn_exp2f(st[kk][e] - mnew);
        st[kk][e] = pv;
        ps += pv;
      }
    lrun = lrun * alpha + ps;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
    bf16x8 pf[NS2];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[s2][e] = f2bf(st[2 * s2][e]);
        pf[s2][4 + e] = f2bf(st[2 * s2 + 1][e]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int s2 = 0; s2 < NS2; ++s2) {
        const char* vp = Vs + (32 * s2 + 4 * g + (c >> 2)) * VP + (dt * 16 + 4 * (c & 3)) * 2;
        const bf16x4 lo = ds_read_tr_big(vp);
        const bf16x4 hi = ds_read_tr_big(vp + 16 * VP);
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s2], o[dt], 0, 0, 0);
      }
    }
  }

  float l = lrun;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  const int q = q0 + c;
  if (q < T) {
    bf16_t* op = p.O + (row0 + q) * p.ldo + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[dt][e] * inv);
      *(bf16x4*)(op + dt * 16) = ov;
      if (p.O_lo) {                                   // (precision: high) what the rounding left behind
        bf16x4 ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) ol[e] = f2bf(o[dt][e] * inv - bf2f(ov[e]));
        *(bf16x4*)(p.O_lo + (op - p.O) + dt * 16) = ol;
      }
    }
  }
}

template <int HD, int KTL>
static int launch_big(const AttnArgs& a, hipStream_t s) {
  constexpr int lds = KTL * HD * 2 + KTL * (HD * 2 + 32);
  auto k = attn_big_kernel<HD, KTL>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  dim3 grid(((a.T + 63) / 64) * a.heads * a.B);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// head_dim 384 / 512 / 640 (no relative-position bias); returns -4 for anything else
int wfl_launch_attention_big(const AttnArgs& a, hipStream_t s) {
  if (a.bias) return -4;
  switch (a.d / a.heads) {
    case 384: return launch_big<384, 64>(a, s);
    case 512: return launch_big<512, 64>(a, s);
    case 640: return launch_big<640, 32>(a, s);
  }
  return -4;
}
