// bf16 MFMA GEMM for gfx950 with fused epilogues — the one kernel family behind every Linear / Conv1d of the
// WFL-ASR forward (q/k/v/out projections, FFN, lang_proj, pointwise convs, the dense k=31 Conformer conv, the
// Whisper stem, dilated convs, classifier, offset head).  Replaces the ATen/HF calls at
// /root/reference/model.py:26-38,98,131,135,137-142 and HF modeling_whisper.py:309-354,391-407,618-619.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2x2, each wave a 64x64 sub-tile = 4x4 MFMA
// 16x16x32 bf16 accumulators.  Operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction, no VGPR round trip) into two stages; the LDS image is lane-linear, so the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and again on
// the read (chunk' = chunk ^ ((row >> 1) & 7) on 128-byte rows).
//
// MFMA roles are transposed (A operand = weight rows, B operand = frame rows), so each lane ends up with 4
// consecutive output channels of one frame and stores 8 bytes; blocks in the V range of a packed q|k|v
// projection swap the roles back and write V transposed ([channel][frame]) for the attention kernel's P.V.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)   // A tile + W tile
#define LDS_BYTES (2 * STAGE_BYTES)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __forceinline__ void glds16(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int ACT, bool GLU, bool OUTF32, bool VT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
  // tiles ordered so that consecutive tiles share the same 128-row A panel.
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;

  // ---- staging addresses.  Wave w issues 4 A and 4 W loads per K tile; load i covers tile rows
  // (4w+i)*8 .. +8, lane l -> row (l>>3), physical 16-byte chunk (l&7), logical chunk = phys ^ ((row>>1)&7).
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wid * 4 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    a_src[i] = p.A + (long)am * p.lda + chunk * 8;
    w_src[i] = p.W + (long)(n0 + row) * p.K + chunk * 8;
  }
  const int nk = p.K / BK;

  auto stage = [&](int buf, int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / p.cin;
    const long koff = (long)tap * p.tap_stride + (k0 - tap * p.cin);
    char* base = smem + buf * STAGE_BYTES + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + koff, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + k0, base + BM * BK * 2 + i * 1024);
  };

  // ---- fragment read offsets (bytes inside a 128x64 tile): row = sub*16 + (lane&15), chunk = 4s + (lane>>4)
  int frag_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
    frag_off[s] = (lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool swap_roles = VT && p.Vt != nullptr && n0 >= p.vt_n0;   // block-uniform

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* At = smem + (kt & 1) * STAGE_BYTES + wm * 128;
    const char* Wt = smem + (kt & 1) * STAGE_BYTES + BM * BK * 2 + wn * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xa[i] = *(const bf16x8*)(At + i * 2048 + frag_off[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8*)(Wt + j * 2048 + frag_off[s]);
      if (VT && swap_roles) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], wb[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  if (VT && swap_roles) {
    // acc[i][j][e]: frame m = m0+wm+16i+4*(lane>>4)+e, channel n = n0+wn+16j+(lane&15); 4 consecutive frames/lane
    const int nv = p.N - p.vt_n0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm + i * 16 + (lane >> 4) * 4;
      if (m >= p.M) continue;
      const int b = m / p.P, t = m - b * p.P;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 16 + (lane & 15);
        if (n >= p.n_valid) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf((t + e < p.T) ? acc[i][j][e] + bv : 0.f);
        *(bf16x4*)(p.Vt + ((long)b * nv + (n - p.vt_n0)) * p.P + t) = o;
      }
    }
    return;
  }

  // acc[i][j][e]: frame m = m0+wm+16i+(lane&15), channel n = n0+wn+16j+4*(lane>>4)+e
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    const int b = m / p.P, t = m - b * p.P;
    if (t >= p.T) continue;
    const long orow = p.c_lead + (long)b * p.c_pitch + t;
    const float* cb = p.clip_bias ? p.clip_bias + (long)p.clip_idx[b] * p.clip_ld : nullptr;
    if (GLU) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int n = n0 + wn + jp * 32 + (lane >> 4) * 4;        // 'a' rows; gates at n + 16
        const int oc = (n0 + wn) / 2 + jp * 16 + (lane >> 4) * 4;  // output channel
        if (n >= p.n_valid) continue;
        f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) { ba = *(const f32x4*)(p.bias + n); bg = *(const f32x4*)(p.bias + n + 16); }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc[i][2 * jp][e] + ba[e];
          const float g = acc[i][2 * jp + 1][e] + bg[e];
          o[e] = f2bf(a * sigmoidf_(g));
        }
        *(bf16x4*)((bf16_t*)p.C + orow * p.ldc + oc) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 16 + (lane >> 4) * 4;
        if (n >= p.n_valid) continue;
        f32x4 v = acc[i][j];
        if (p.bias) { const f32x4 bb = *(const f32x4*)(p.bias + n); v += bb; }
        if (cb) { const f32x4 bb = *(const f32x4*)(cb + n); v += bb; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<ACT>(v[e]);
        if (p.pos) {
          const bf16x4 pp = *(const bf16x4*)(p.pos + (long)t * p.ldpos + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(pp[e]);
        }
        if (p.res) {
          const bf16x4 rr = *(const bf16x4*)(p.res + orow * p.ldres + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = bf2f(rr[e]) + p.alpha * v[e];
        }
        if (OUTF32) {
          float* o = (float*)p.C + orow * p.ldc + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.n_valid) o[e] = v[e];
        } else if (n + 4 <= p.n_valid) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
          *(bf16x4*)((bf16_t*)p.C + orow * p.ldc + n) = o;
        } else {
          bf16_t* o = (bf16_t*)p.C + orow * p.ldc + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.n_valid) o[e] = f2bf(v[e]);
        }
      }
    }
  }
}

template <int ACT, bool GLU, bool OUTF32, bool VT>
static int launch_t(const GemmArgs& a, hipStream_t s) {
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  auto k = gemm_bf16_kernel<ACT, GLU, OUTF32, VT>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(tiles), dim3(256), LDS_BYTES, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wfl_launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.N % BN || a.K <= 0 || a.K % BK || a.cin <= 0 || a.cin % BK || a.P <= 0 || a.P % 4 ||
      a.ldc % 4 || (a.res && a.ldres % 4) || (a.pos && a.ldpos % 4) || (a.Vt && a.vt_n0 % BN))
    return -1;
  if (a.glu) {
    if (a.out_f32 || a.act != WFL_ACT_NONE || a.Vt) return -1;
    return launch_t<WFL_ACT_NONE, true, false, false>(a, s);
  }
  if (a.Vt) {
    if (a.out_f32 || a.act != WFL_ACT_NONE) return -1;
    return launch_t<WFL_ACT_NONE, false, false, true>(a, s);
  }
  if (a.out_f32) {
    if (a.act == WFL_ACT_NONE) return launch_t<WFL_ACT_NONE, false, true, false>(a, s);
    if (a.act == WFL_ACT_SIGMOID) return launch_t<WFL_ACT_SIGMOID, false, true, false>(a, s);
    return -1;
  }
  switch (a.act) {
    case WFL_ACT_NONE: return launch_t<WFL_ACT_NONE, false, false, false>(a, s);
    case WFL_ACT_GELU: return launch_t<WFL_ACT_GELU, false, false, false>(a, s);
    case WFL_ACT_RELU: return launch_t<WFL_ACT_RELU, false, false, false>(a, s);
    case WFL_ACT_SIGMOID: return launch_t<WFL_ACT_SIGMOID, false, false, false>(a, s);
  }
  return -1;
}
