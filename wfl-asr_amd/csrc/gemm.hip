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
// consecutive output channels of one frame and stores 8 bytes.
#include "common.h"
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)   // A tile + W tile
#ifndef NSTAGE
#define NSTAGE 4
#endif
static_assert(NSTAGE >= 3 && NSTAGE <= 4, "the mid-iteration prefetch needs a ring of 3 or 4 K tiles");
#define EPI_BYTES (128 * 132 * 4)            // fp32 epilogue tile, reuses the operand ring
#define LDS_BYTES (NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifdef WFL_GEMM_STAMPS
// Diagnostic build only (tools/gemm_diag.py): per-block 100 MHz timestamps of the phases, written to a side buffer.
#define STAMP(k) do { if (tid == 0 && p.stamps) p.stamps[(long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

static __device__ __forceinline__ void glds16(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int ACT, bool GLU, bool OUTF32>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  STAMP(0);
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
    long toff = (long)tap * p.tap_stride;
    if (p.tap_wrap > 0) {                           // three segments of tap_wrap taps (GemmArgs::tap_wrap)
      const int sg = tap / p.tap_wrap;
      toff = (long)(tap - sg * p.tap_wrap) * p.tap_stride + (sg == 1 ? p.seg_off : 0);
    }
    const long koff = toff + (k0 - tap * p.cin);
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

  // bias for this lane's 4x4 channels (normal roles), fetched before the K loop so its latency is off the epilogue
  f32x4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bj[j] = p.bias ? *(const f32x4*)(p.bias + n0 + wn + j * 16 + (lane >> 4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- main loop: NSTAGE-deep LDS ring of K tiles + software-pipelined fragment reads (one wave per SIMD has no
  // partner wave to hide LDS latency behind, so the reads of the next half tile are issued under this half's MFMAs).
  //
  //   iteration kt, F0 = fragments of (tile kt, k 0..31) already in flight:
  //     read F1 = (kt, k 32..63) | 16 MFMA on F0
  //     wait "tile kt+1 landed" (counted vmcnt: every wave issues 8 LDS-DMA loads per tile, later tiles stay in
  //     flight) ; raw s_barrier ; prefetch tile kt+NSTAGE-1 into the slot of tile kt-1 (the barrier proves every
  //     wave finished kt-1)
  //     read F0 = (kt+1, k 0..31) | 16 MFMA on F1
  //
  // acc[u][v] = mfma(FA[v], FB[u]); FA rows come from the weight tile and FB rows from the frame tile (D[n][m])
  const int fa_off = BM * BK * 2 + wn * 128;
  const int fb_off = wm * 128;
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
#define LOAD_FRAGS(FA, FB, slot, s)                                                              \
  do {                                                                                           \
    const char* ta_ = smem + (slot) * STAGE_BYTES + fa_off + frag_off[s];                        \
    const char* tb_ = smem + (slot) * STAGE_BYTES + fb_off + frag_off[s];                        \
    _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_) FA[v_] = *(const bf16x8*)(ta_ + v_ * 2048); \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) FB[u_] = *(const bf16x8*)(tb_ + u_ * 2048); \
  } while (0)
#define MMA(FA, FB, lo, hi)                                                                      \
  _Pragma("unroll") for (int q_ = (lo); q_ < (hi); ++q_)                                         \
    acc[q_ >> 2][q_ & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[q_ & 3], FB[q_ >> 2], acc[q_ >> 2][q_ & 3], 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)   /* lgkmcnt(0) only; a builtin so hipcc's wait model sees it */
#define WAIT_TILES(later)                                                        \
  do {                                                                           \
    if ((later) >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          \
    else if ((later) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
  } while (0)

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nk) stage(t, t);
  {
    const int later = (nk - 1 < NSTAGE - 2) ? nk - 1 : NSTAGE - 2;   // tiles issued after tile 0
    WAIT_TILES(later);
    __builtin_amdgcn_s_barrier();
    STAMP(1);
    LOAD_FRAGS(fa0, fb0, 0, 0);
  }
  for (int kt = 0; kt < nk; ++kt) {
    // One straight-line body for every iteration (a peeled last iteration made hipcc rotate the 64 accumulator
    // registers through v_accvgpr_mov every trip): on the last trip the wait/barrier are trivially satisfied and
    // the F0 read fetches a stale ring slot that nothing consumes.
    LGKM0();                                  // F0 (issued half an iteration ago) has arrived
    LOAD_FRAGS(fa1, fb1, kt % NSTAGE, 1);
    SB();
    MMA(fa0, fb0, 0, 16);
    SB();
    {
      const int last = (kt + NSTAGE - 2 < nk - 1) ? kt + NSTAGE - 2 : nk - 1;   // newest tile issued so far
      WAIT_TILES(last - (kt + 1));
      __builtin_amdgcn_s_barrier();
      if (kt + NSTAGE - 1 < nk) stage((kt + NSTAGE - 1) % NSTAGE, kt + NSTAGE - 1);
    }
    LGKM0();                                  // F1 has arrived
    LOAD_FRAGS(fa0, fb0, (kt + 1) % NSTAGE, 0);
    SB();
    MMA(fa1, fb1, 0, 16);
    SB();
  }
#undef LOAD_FRAGS
#undef MMA
#undef SB
#undef LGKM0
#undef WAIT_TILES

  // ------------------------------------------------------------------ epilogue
  // The accumulators leave through LDS so that HBM sees whole rows: every lane owns 4 channels x 1 frame per MFMA
  // tile, which as direct stores is 16 scattered 8-byte stores per lane (store-issue bound, 32-byte segments).
  // Stage 1 applies bias / per-clip bias / activation / GLU in fp32 and parks the 128x128 tile in LDS
  // (row pitch 132 floats: the 8 lanes of a ds_write_b128 group hit 8 distinct 4-bank slots); stage 2 re-reads it
  // row-wise, adds the positional table and the residual with 16-byte loads and writes 16 bytes per lane.
  constexpr int NC = GLU ? 64 : 128;          // staged columns
  constexpr int EP = NC + 4;                  // pitch in floats
  float* stg = (float*)smem;
  STAMP(2);
  __syncthreads();                            // all waves are done with the operand stages
#ifdef WFL_GEMM_STAMPS
  if (tid == 0 && p.stamps) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    p.stamps[(long)blockIdx.x * 8 + 6] = xcc;
    p.stamps[(long)blockIdx.x * 8 + 7] = hwid;
  }
#endif
  // stage 1: acc[i][j][e] is frame m = wm+16i+c, channel n = wn+16j+4g+e
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ml = wm + i * 16 + (lane & 15);
    const float* cb = nullptr;
    if (p.clip_bias) {
      int m = m0 + ml;
      m = m < p.M ? m : p.M - 1;
      cb = p.clip_bias + (long)p.clip_idx[m / p.P] * p.clip_ld;
    }
    if (GLU) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const f32x4 ba = bj[2 * jp], bg = bj[2 * jp + 1];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][2 * jp][e] + ba[e]) * sigmoidf_(acc[i][2 * jp + 1][e] + bg[e]);
        *(f32x4*)(stg + ml * EP + wn / 2 + jp * 16 + (lane >> 4) * 4) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nl = wn + j * 16 + (lane >> 4) * 4;
        f32x4 v = acc[i][j] + bj[j];
        if (cb) { const f32x4 bb = *(const f32x4*)(cb + n0 + nl); v += bb; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<ACT>(v[e]);
        *(f32x4*)(stg + ml * EP + nl) = v;
      }
    }
  }
  __syncthreads();
  STAMP(3);

  // stage 2: thread -> 8 consecutive channels of one frame per pass.  All global loads (residual, positional table)
  // of the 128/RPP passes are issued before the first use so their latencies overlap instead of adding up.
  constexpr int CP = NC / 8;                  // 8-channel chunks per row
  constexpr int RPP = 256 / CP;               // rows per pass
  constexpr int NP = 128 / RPP;
  const int cidx = tid % CP;
  const int nb = (GLU ? n0 / 2 : n0) + cidx * 8;
  const int nvalid = GLU ? p.n_valid / 2 : p.n_valid;
  if (nb >= nvalid) return;
  long orow[NP];
  int tt[NP];
  bool ok[NP];
  {
    const int m = m0 + tid / CP;
    int b = m / p.P, t = m - b * p.P;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      ok[pass] = (m + pass * RPP < p.M) && t < p.T;
      tt[pass] = t;
      orow[pass] = p.c_lead + (long)b * p.c_pitch + t;
      t += RPP;
      if (p.P >= RPP) { if (t >= p.P) { t -= p.P; ++b; } }
      else while (t >= p.P) { t -= p.P; ++b; }    // (a pitch below RPP = 32 rows -- clips of at most 8 frames -- wraps more than once)
    }
  }
  if (p.clip_T) {                              // ragged batches: a clip's own frame count (behind a scalar branch: common.h, GemmArgs::clip_T)
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
      if (ok[pass] && tt[pass] >= p.clip_T[(orow[pass] - p.c_lead) / p.c_pitch]) ok[pass] = false;
  }
  // Unconditional loads (a branch per load would make hipcc wait vmcnt(0) after each): rows that are not stored
  // are halo / tail rows of the residual's frame-row buffer, which exist; the positional row is clamped.
  bf16x8 rr[NP], rl[NP], pp[NP];
  if (p.res) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) rr[pass] = *(const bf16x8*)(p.res + orow[pass] * p.ldres + nb);
    if (p.res_lo) {                                   // residual stream carried as hi + lo (common.h)
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) rl[pass] = *(const bf16x8*)(p.res_lo + orow[pass] * p.ldres + nb);
    }
  }
  if (p.pos) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
      pp[pass] = *(const bf16x8*)(p.pos + (long)(tt[pass] < p.T ? tt[pass] : p.T - 1) * p.ldpos + nb);
  }
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    if (!ok[pass]) continue;
    const int r = pass * RPP + tid / CP;
    float v[8];
    {
      const f32x4 v0 = *(const f32x4*)(stg + r * EP + cidx * 8);
      const f32x4 v1 = *(const f32x4*)(stg + r * EP + cidx * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
    }
    if (p.pos) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(pp[pass][e]);
    }
    if (p.res) {
      if (p.res_lo) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf2f(rr[pass][e]) + bf2f(rl[pass][e])) + p.alpha * v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(rr[pass][e]) + p.alpha * v[e];
      }
    }
    if (OUTF32) {
      float* o = (float*)p.C + orow[pass] * p.ldc + nb;
      if (p.acc_f32) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (nb + e < nvalid) v[e] += o[e];
      }
      if (nb + 8 <= nvalid && (p.ldc & 3) == 0) {
        *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (nb + e < nvalid) o[e] = v[e];
      }
    } else if (nb + 8 <= nvalid) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      *(bf16x8*)((bf16_t*)p.C + orow[pass] * p.ldc + nb) = o;
      if (p.c_lo) {
        bf16x8 ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) ol[e] = f2bf(v[e] - bf2f(o[e]));
        *(bf16x8*)(p.c_lo + orow[pass] * p.ldc + nb) = ol;
      }
    } else {
      bf16_t* o = (bf16_t*)p.C + orow[pass] * p.ldc + nb;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (nb + e < nvalid) {
          o[e] = f2bf(v[e]);
          if (p.c_lo) p.c_lo[orow[pass] * p.ldc + nb + e] = f2bf(v[e] - bf2f(o[e]));
        }
    }
  }
  STAMP(4);
}

template <int ACT, bool GLU, bool OUTF32>
static int launch_t(const GemmArgs& a, hipStream_t s) {
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  auto k = gemm_bf16_kernel<ACT, GLU, OUTF32>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
  }
  g_wfl_gemm_kernel_id = 4;
  hipLaunchKernelGGL(k, dim3(tiles), dim3(256), LDS_BYTES, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int g_wfl_gemm_kernel_id = 0;

int wfl_launch_gemm256(const GemmArgs& a, hipStream_t s);   // gemm256.hip; returns 1 when it does not take the shape
int wfl_launch_gemm_stream(const GemmArgs& a, hipStream_t s);   // gemm_stream.hip; likewise
bool wfl_gemm256_tri_takes(const GemmArgs& a);                  // gemm256.hip: a three-segment launch its slice-by-slice walk takes
bool wfl_gemm_stream_conv_takes(const GemmArgs& a);             // gemm_stream.hip: a launch its tap-stationary conv mode takes

static int tile_pref() {
  static int pref = -1;
  if (pref < 0) {
    const char* e = getenv("WFL_GEMM_TILE");      // "128" forces the 128x128 kernel (A/B runs, tests)
    pref = (e && atoi(e) == 128) ? 128 : 256;
  }
  return pref;
}

int wfl_launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.N % BN || a.K <= 0 || a.K % BK || a.cin <= 0 || a.cin % BK || a.P <= 0 || a.P % 8 ||
      (!a.out_f32 && a.ldc % 8) || (a.res && a.ldres % 8) || (a.pos && a.ldpos % 8))
    return -1;
  if (a.a8 >= 2) {                                  // fp8 x fp8 on the block-scaled MFMA (gemm_mx.hip): nothing else reads these operands
    const int r = wfl_launch_gemm_mx(a, s);
    return r == 1 ? -1 : r;
  }
  if (tile_pref() == 256) {
    // "model.precision: high" launches (three segments): the slice-by-slice walk of gemm256.hip stages two thirds of the bytes of the
    // segment-major walk; the tap-stationary conv mode (a third of them again) keeps the dense multi-tap convs.  WFL_TRI_RES=0 leaves the
    // residual launches with the streaming kernel (A/B runs).
    if (a.tap_wrap > 0 && wfl_gemm256_tri_takes(a) && !wfl_gemm_stream_conv_takes(a)) {
      static int tri_res = -1;
      if (tri_res < 0) { const char* e = getenv("WFL_TRI_RES"); tri_res = e ? atoi(e) : 1; }
      if (!a.res || tri_res) {
        const int r3 = wfl_launch_gemm256(a, s);
        if (r3 != 1) return r3;
      }
    }
    int r = wfl_launch_gemm_stream(a, s);
    if (r != 1) return r;
    if (a.ln_s || a.w8_scale) return -1;            // only the streaming kernel folds a LayerNorm / reads e4m3 weights
    r = wfl_launch_gemm256(a, s);
    if (r != 1) return r;
  }
  if (a.ln_s || a.w8_scale) return -1;
  if (a.glu) {
    if (a.out_f32 || a.act != WFL_ACT_NONE) return -1;
    return launch_t<WFL_ACT_NONE, true, false>(a, s);
  }
  if (a.out_f32) {
    if (a.act == WFL_ACT_NONE) return launch_t<WFL_ACT_NONE, false, true>(a, s);
    if (a.act == WFL_ACT_SIGMOID) return launch_t<WFL_ACT_SIGMOID, false, true>(a, s);
    return -1;
  }
  switch (a.act) {
    case WFL_ACT_NONE: return launch_t<WFL_ACT_NONE, false, false>(a, s);
    case WFL_ACT_GELU: return launch_t<WFL_ACT_GELU, false, false>(a, s);
    case WFL_ACT_RELU: return launch_t<WFL_ACT_RELU, false, false>(a, s);
    case WFL_ACT_SIGMOID: return launch_t<WFL_ACT_SIGMOID, false, false>(a, s);
  }
  return -1;
}
