// Flash attention for head_dim 64 as an EIGHT-wave workgroup whose two wave groups alternate between the matrix pipe and everything
// else (round 4).  Same arithmetic, operand layouts and output as attn_kernel<64, 2> (attention.hip: S^T = K . Q^T with MFMA 16x16x32,
// deferred-reference online softmax, row sums on the matrix pipe, V^T fragments by ds_read_b64_tr_b16, K / V tiles staged by LDS-DMA) --
// it replaces the same reference code, HF modeling_whisper.py:215-238 / SDPA as called from /root/reference/model.py:155-156 -- but the
// loop is cut into two phases per key tile, separated by workgroup barriers, and the groups run one phase apart:
//
//     compute phase (C) of tile t:   row sums + O^T += V^T(t-1) . P^T(t-1)   (20 MFMAs)      then   S^T(t) = K(t) . Q^T   (16 MFMAs),
//                                    with the LDS-DMA of tile t + 3 among them and the K(t + 1) fragment reads behind them
//     other phase   (L) of tile t:   softmax numerators P(t) from S^T(t) (exp2, max, rescale, pack: the vector pipe), V^T(t) fragments
//
// While group A (waves 0-3) issues MFMAs, group B (waves 4-7, on the same four SIMDs) runs its softmax and its LDS reads, and vice versa:
// the matrix pipe never waits for an exp2 or a ds_read of its own wave, which is what held attn_kernel<64, 2> at 44 % MFMA-busy
// (matrix pipe, softmax arithmetic, fragment reads and tile staging each worth 14-21 us of 96, overlapped only by
// chance between three co-resident workgroups).  One workgroup per CU, 256 queries, 32 per wave; a key tile is staged ONCE for 256
// queries instead of once per 128.  Eight tile buffers (128 KiB), three tiles in flight.
#include "common.h"
#include <cstdlib>

#define PKT 64
#define PP_NB 8       // tile buffers (128 KiB: one workgroup per CU either way); a buffer is refilled five tiles after its last read
#define PP_D 3        // tiles in flight ahead of the one being computed
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short pp_s16x4_t;
typedef __attribute__((address_space(3))) pp_s16x4_t* pp_lds_s16x4_t;
typedef __attribute__((address_space(1))) const void* pp_gptr_t;
typedef __attribute__((address_space(3))) void* pp_lptr_t;
static __device__ __forceinline__ bf16x4 pp_read_tr(const char* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((pp_lds_s16x4_t)p));
}
static __device__ __forceinline__ int pp_kswz(int row) { return (row >> 1) & 7; }      // attention.hip: k_swz<64>
static __device__ __forceinline__ int pp_vswz(int row) { return (row >> 1) & 3; }      // attention.hip: v_swz<64>
#define pp_mfma(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
template <int N>
static __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool OUT8>
__global__ __launch_bounds__(512) void attn_pp_kernel(AttnArgs p) {
  constexpr int HD = 64, QT = 2, KS = 2, DT = 4;
  constexpr int KIMG = PKT * HD * 2;               // 8 KiB: K image, then the V image
  constexpr int TILE = 2 * KIMG;
  constexpr int NQW = 8 * QT * 16;                 // 256 queries per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int g = lane >> 4, c = lane & 15;
  // XCD-aware order (attention.hip): blocks i and i + 8 share an XCD; every XCD gets a contiguous run of (clip, head, query block)
  const int nqb = (p.T + NQW - 1) / NQW;
  const int nblk = nqb * p.heads * p.B;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int qb = bid % nqb;
  const int bh = bid / nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int T = p.clip_T ? p.clip_T[b] : p.T;
  if (qb * NQW >= T) return;                        // (block-uniform: before any barrier)
  const int q0 = qb * NQW + wid * (QT * 16);
  const long row0 = p.lead + (long)b * p.P;
  const bf16_t* Kg = p.QK + p.d + h * HD;
  const bf16_t* Vg = p.V + h * HD;

  // ---- Q fragments (B operand): lane -> query frame q0 + 16 qt + c (clamped: a wave beyond the clip's end computes on its last frame
  //      and stores nothing), channels 32 ks + 8 g .. + 8
  bf16x8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + c;
    q = q < T ? q : T - 1;
    const bf16_t* qp = p.QK + (row0 + q) * p.ldqk + h * HD + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = *(const bf16x8*)(qp + ks * 32);
  }

  constexpr float RESCALE_THR = 8.0f;
  f32x4 o[QT][DT], osum[QT];
  float negm[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    negm[qt] = 0.f;
    osum[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = f2bf(c == 0 ? 1.0f : 0.0f);

  const int ntiles = (T + PKT - 1) / PKT;
  // ---- LDS-DMA staging: one 1 KiB piece of the K image and one of the V image per wave and tile; lane l fetches the 16-byte chunk that
  //      belongs at its LDS position under the K / V swizzles
  int kdoff, vdoff;
  {
    const int byte = wid * 1024 + lane * 16;
    const int r = byte / (HD * 2), pos = (byte % (HD * 2)) >> 4;
    kdoff = r * (int)p.ldqk + ((pos ^ pp_kswz(r)) << 3);
    vdoff = r * (int)p.ldv + (((((pos >> 1) ^ pp_vswz(r)) << 1) | (pos & 1)) << 3);
  }
  int issued = 0;                                   // tiles this wave has issued its two pieces of
  auto dma_tile = [&](int kt) __attribute__((always_inline)) {
    const bf16_t* kb = Kg + (row0 + (long)kt * PKT) * p.ldqk;
    const bf16_t* vb = Vg + (row0 + (long)kt * PKT) * p.ldv;
    char* dst = smem + (kt & (PP_NB - 1)) * TILE + wid * 1024;
    __builtin_amdgcn_global_load_lds((pp_gptr_t)(kb + kdoff), (pp_lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((pp_gptr_t)(vb + vdoff), (pp_lptr_t)(dst + KIMG), 16, 0, 0);
    ++issued;
  };
  auto wait_tile = [&](int kt) __attribute__((always_inline)) {      // this wave's pieces of tile kt have landed
    const int younger = issued - kt - 1;
    if (younger >= 2) pp_wait_vm<4>(); else if (younger == 1) pp_wait_vm<2>(); else pp_wait_vm<0>();
  };

  bf16x8 kf[KS][4], vf[DT][2], pf[QT][2];
  auto read_k = [&](int kt) __attribute__((always_inline)) {
    const char* Ks = smem + (kt & (PP_NB - 1)) * TILE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int r = kk * 16 + c;
        kf[ks][kk] = *(const bf16x8*)(Ks + r * (HD * 2) + (((ks * 4 + g) ^ pp_kswz(r)) << 4));
      }
  };
  auto read_v = [&](int kt) __attribute__((always_inline)) {
    const char* Vs = smem + (kt & (PP_NB - 1)) * TILE + KIMG;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int vr = 32 * s2 + 4 * g + (c >> 2);        // (row vr + 16 has the same window swizzle)
        const char* vp = Vs + vr * (HD * 2) + ((dt ^ pp_vswz(vr)) * 16 + 4 * (c & 3)) * 2;
        const bf16x4 lo = pp_read_tr(vp);
        const bf16x4 hi = pp_read_tr(vp + 16 * (HD * 2));
        vf[dt][s2] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  };

  // ---- prologue: three tiles in flight, tile 0 landed, its K fragments in registers; group B one barrier behind
#pragma unroll
  for (int t = 0; t < PP_D; ++t)
    if (t < ntiles) dma_tile(t);
  wait_tile(ntiles > 1 ? 1 : 0);                    // tiles 0 and 1: K(1) is read behind the first compute phase's MFMAs
  __builtin_amdgcn_s_barrier();
  read_k(0);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) pf[qt][s2] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) vf[dt][s2] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0)
  if (grp && PP_ABL != 6) __builtin_amdgcn_s_barrier();            // group B runs one phase behind

#ifndef PP_ABL
#define PP_ABL 0      // diagnostic builds (tools/build_attn_pp_ablations.sh; wrong results, same launch): 1 no softmax arithmetic, 2 no context / row-sum
#endif                //   MFMAs, 3 no score MFMAs, 4 fragments read once, 5 no DMA inside the loop, 6 no barriers inside the loop
#define PSB() __builtin_amdgcn_sched_barrier(0)
  f32x4 st[QT][4];
  for (int t = 0; t <= ntiles; ++t) {
    // ================= compute phase: the matrix pipe is this group's
    PSB();
#if PP_ABL != 2
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) osum[qt] = pp_mfma(ones, pf[qt][s2], osum[qt]);      // row sums of P(t - 1) (zero at t = 0)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = pp_mfma(vf[dt][s2], pf[qt][s2], o[qt][dt]);
#else
    asm volatile("" :: "v"(vf[0][0]), "v"(pf[0][0]));
    osum[0][0] += 1.f;
#endif
    if (t < ntiles) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) st[qt][kk] = (f32x4){negm[qt], negm[qt], negm[qt], negm[qt]};
#if PP_ABL != 3
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) st[qt][kk] = pp_mfma(kf[ks][kk], qf[qt][ks], st[qt][kk]);
#else
      asm volatile("" :: "v"(kf[0][0]), "v"(kf[1][3]));
#endif
#if PP_ABL == 5
      if (t + PP_D < ntiles) ++issued;
#else
      if (t + PP_D < ntiles) dma_tile(t + PP_D);
#endif     // (an LDS-DMA instruction issued among MFMAs costs a third of one issued among ds_reads)
      if (t + 1 < ntiles && (PP_ABL != 4 || t == 0)) read_k(t + 1);             // behind the MFMAs that read the old fragments; tile t + 1 landed two barriers ago
    }
    PSB();
    if (t == ntiles) {                               // the last tile's context is in: group B is one barrier ahead in count, A closes it
      if (!grp && PP_ABL != 6) __builtin_amdgcn_s_barrier();
      break;
    }
    // K(x) is first read by group A behind the MFMAs of C(x - 1): tile x has to be in before the barrier in front of that phase, which
    // group B reaches at the end of ITS C(x - 2) and group A at the end of its L(x - 2)
    if (grp && t + 2 < ntiles) wait_tile(t + 2);
    if (PP_ABL != 6) __builtin_amdgcn_s_barrier();
    PSB();
    // ================= the other phase: softmax of tile t, V fragments for the next compute phase
    if (t * PKT + PKT > T) {                         // last tile: keys >= T do not exist
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t * PKT + kk * 16 + g * 4 + e >= T) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) st[qt][kk][e] = -INFINITY;
          }
    }
    {
      float mx[QT];
      bool over = t == 0;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float m = st[qt][0][0];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e) m = fmaxf(m, st[qt][kk][e]);
        mx[qt] = m;
        over = over || m > RESCALE_THR;
      }
      if (__builtin_amdgcn_ballot_w64(over) != 0) {      // wave-uniform and rare after the first tiles
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float m = mx[qt];
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          const float d = t == 0 ? m : fmaxf(m, 0.f);    // new reference - old reference (raise only; tile 0 takes the tile's maximum)
          const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[qt][kk][e] -= d;
          negm[qt] -= d;
          osum[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 tt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if PP_ABL == 1
          tt[e] = __builtin_bit_cast(bf16x2, st[qt][2 * s2][e])[1];
          tt[4 + e] = __builtin_bit_cast(bf16x2, st[qt][2 * s2 + 1][e])[1];
#else
          tt[e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2][e]));
          tt[4 + e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2 + 1][e]));
#endif
        }
        pf[qt][s2] = tt;
      }
    if (PP_ABL != 4 || t == 0) read_v(t);                                       // (no lgkmcnt wait in front of the barrier: a buffer is refilled five tiles from now)
    if (!grp && t + 2 < ntiles) wait_tile(t + 2);
    if (PP_ABL != 6) __builtin_amdgcn_s_barrier();
  }
#undef PSB

  // ---- normalise and store: lane holds channels h * HD + 16 dt + 4 g + e of query frame q0 + 16 qt + c
  float amax8 = 0.f;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = __shfl(osum[qt][0], c);          // row sum of query c lives in lane (g = 0, c), register 0
    const float inv = 1.0f / l;
    const int q = q0 + qt * 16 + c;
    if (OUT8) {
      if (q < T) {
        unsigned char* op8 = p.O8 + (row0 + q) * p.ldo8 + h * HD + g * 4;
        const float sc = inv * p.o8_scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = o[qt][dt][e] * sc;
            amax8 = fmaxf(amax8, fabsf(y));
            x[e] = fminf(fmaxf(y, -448.f), 448.f);
          }
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
          *(int*)(op8 + dt * 16) = w;
          if (p.O8_lo) {
            typedef __attribute__((ext_vector_type(2))) float f32x2_;
            const f32x2_ h0 = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), h1 = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
            int wl = 0;
            wl = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (x[0] - h0[0]), 16.f * (x[1] - h0[1]), wl, false);
            wl = __builtin_amdgcn_cvt_pk_fp8_f32(16.f * (x[2] - h1[0]), 16.f * (x[3] - h1[1]), wl, true);
            *(int*)(p.O8_lo + (op8 - p.O8) + dt * 16) = wl;
          }
        }
      }
      continue;
    }
    if (q < T) {
      bf16_t* op = p.O + (row0 + q) * p.ldo + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[qt][dt][e] * inv);
        *(bf16x4*)(op + dt * 16) = ov;
        if (p.O_lo) {
          bf16x4 ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) ol[e] = f2bf(o[qt][dt][e] * inv - bf2f(ov[e]));
          *(bf16x4*)(p.O_lo + (op - p.O) + dt * 16) = ol;
        }
      }
    }
  }
  if (OUT8 && p.err) {
    if (__builtin_amdgcn_ballot_w64(amax8 > 448.f) && lane == 0) atomicOr(p.err, 2u);
  }
}

template <bool OUT8>
static int launch_pp(const AttnArgs& a, hipStream_t s) {
  constexpr int lds = PP_NB * 2 * PKT * 64 * 2;
  auto k = attn_pp_kernel<OUT8>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  const int nqb = (a.T + 255) / 256;
  hipLaunchKernelGGL(k, dim3(nqb * a.heads * a.B), dim3(512), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Returns 1 when this kernel does not take the launch (attention.hip's kernels do).  head_dim 64, no relative-position bias, bf16
// operands, clips of at least two key tiles; WFL_ATTN_PP=0 keeps attn_kernel<64, 2> (A/B runs).
int wfl_launch_attention_pp(const AttnArgs& a, hipStream_t s) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("WFL_ATTN_PP"); on = e ? atoi(e) : 1; }
  if (!on) return 1;
  if (a.heads <= 0 || a.d % a.heads || a.d / a.heads != 64 || a.bias || a.gate || a.QK_lo || a.V_lo) return 1;
  if (a.T < 2 * PKT) return 1;
  if (a.O8) {
    if (a.ldo8 % 4) return -4;
    return launch_pp<true>(a, s);
  }
  return launch_pp<false>(a, s);
}
