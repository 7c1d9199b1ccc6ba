// 256 x 256 x 32 variant of the bf16 MFMA GEMM (same contract and epilogues as gemm.hip).
//
// Why: tools/gemm_diag.py shows the 128x128 tile is bound by L2 -> LDS bandwidth per CU (~50-70 GB/s): each K step
// moves 32 KiB for 2.1 MFLOP.  A 256x256 block tile moves the same 32 KiB per 32-deep K step for 4.2 MFLOP, i.e. twice
// the arithmetic intensity (128 FLOP/B), and with M = B*pitch ~ 24k rows its tile counts (190 per 512 output columns)
// fill the 256 CUs in whole rounds.
//
// 512 threads = 8 waves (2 per SIMD) in 2(M) x 4(N); each wave owns 128 frames x 64 channels = 8 x 4 MFMA 16x16x32
// accumulators (128 VGPRs).  Operand tiles arrive by global_load_lds_dwordx4 into a 4-deep ring of 32 KiB stages
// (counted vmcnt, raw s_barrier).  Rows are 64 bytes (32 bf16), so four rows share a 256-byte bank row; the 16-byte
// chunk swizzle chunk' = chunk ^ ((-(row >> 2)) & 3) makes every ds_read_b128 lane group hit 16 distinct slots.
// K loop = two-group ping-pong (see the comment at the loop): measured 10-12 % faster than running all eight waves in
// lock step, and tools/gemm_lab.py's ablations show what is left: MFMA alone 0.40 us per 32-deep step, operand DMA alone
// 0.50 us (the L2 -> LDS limit of ~65 GB/s per CU), both together ~0.61-0.65 us.
// The epilogue leaves through LDS in two 128-row halves (fp32, pitch 260) exactly like gemm.hip's.
#include "common.h"
#include <cstdlib>

#define BM2 256
#define BN2 256
#define BK2 32
#define ST2 (2 * 256 * BK2 * 2)          // one stage: frame tile + weight tile = 32 KiB
#define NST2 4
#define EP2 260                           // epilogue pitch (floats)
#define LDS2 (256 * 132 * 4)              // 135,168 B (>= 128 * EP2 * 4 epilogue staging >= NST2 * ST2 operand ring)

typedef __attribute__((address_space(1))) const void* gptr2_t;
typedef __attribute__((address_space(3))) void* lptr2_t;

static __device__ __forceinline__ void glds16b(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((gptr2_t)g, (lptr2_t)l, 16, 0, 0);
}
static __device__ __forceinline__ int swz2(int row) { return (-(row >> 2)) & 3; }

#ifdef WFL_GEMM_STAMPS
// (slots 0-4: 100 MHz wall clock at the phase boundaries; slots 5 / 6: the core-clock counter at stamps 1 / 2 -- the clock the chip holds inside the
//  K loop is their difference over the wall time between the two, MI355X_MICROARCH.md "DVFS give-back" item 6)
#define STAMP2(k) do { if (tid == 0 && p.stamps) { p.stamps[(long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 1) p.stamps[(long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 2) p.stamps[(long)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP2(k) do { } while (0)
#endif

// TRI ("model.precision: high", GemmArgs::tap_wrap; round 4): the launch's K holds the three segments [A_hi W_hi | A_lo W_hi | A_hi W_lo].
//   Walked segment by segment that is 3 K / 32 stages of DMA for 3 K / 32 steps of MFMAs -- and the K loop is bound by the L2 -> LDS
//   bytes, so the exact-label mode paid three times the default's staging.  TRI walks K SLICE by slice instead: slice j's four tiles
//   arrive as two stages, 2j = (A_hi, W_hi) and 2j + 1 = (A_lo, W_lo), and three MFMA steps run off them,
//       3j: A_hi W_hi (slot 2j)      3j + 1: A_lo (slot 2j + 1) W_hi (fragments kept)      3j + 2: A_hi (slot 2j) W_lo (slot 2j + 1)
//   i.e. two stages of DMA per three steps of MFMAs (the loop turns MFMA-bound) and 3 MT + 8 fragment reads per slice instead of 3 MT + 12.
//   Same products, same fp32 accumulators; the order of the additions differs from the segment-major walk (both are exact-label forms).
template <int ACT, bool GLU, bool OUTF32, int MT, bool TRI = false>   // MT = 16-frame MFMA tiles per wave: 8 -> 256-row block, 6 -> 192
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
  static_assert(!TRI || !OUTF32, "three-segment launches write bf16 pairs");
  constexpr int BMV = MT * 32;              // rows of the block tile that are computed (the staged tile is always 256 rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  STAMP2(0);
  const int tiles_n = p.N / BN2;
  const int tiles_m = (p.M + BMV - 1) / BMV;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BMV, n0 = tile_n * BN2;
  const int wm = (wid >> 2) * (MT * 16), wn = (wid & 3) * 64;

  // staging: wave w issues loads i = 0,1 for each operand; load covers tile rows (2w+i)*16 .. +16,
  // lane l -> row (l >> 2), physical chunk (l & 3), logical chunk = phys ^ swz2(row)
  // The 192-row block stages only the 12 frame row groups it computes: waves 0-3 load two each (rows 0..127), waves 4-7
  // one each (rows 128..191) -- 28 KiB per K step instead of 32 (the K loop is bound by L2 -> LDS bytes).
  const bool two_x = MT == 8 || wid < 4;
  const int xg0 = (MT == 8 || wid < 4) ? wid * 2 : 8 + (wid - 4);
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int wrow = (wid * 2 + i) * 16 + (lane >> 2);
    const int xrow = (xg0 + i) * 16 + (lane >> 2);
    int am = m0 + xrow;
    am = am < p.M ? am : p.M - 1;
    a_src[i] = p.A + (long)am * p.lda + ((lane & 3) ^ swz2(xrow)) * 8;
    w_src[i] = p.W + (long)(n0 + wrow) * p.K + ((lane & 3) ^ swz2(wrow)) * 8;
  }
  const int nk = p.K / BK2;
  auto stage = [&](int buf, int kt) {
    const int k0 = kt * BK2;
    const int tap = k0 / p.cin;
    long toff = (long)tap * p.tap_stride;
    if (p.tap_wrap > 0) {                           // three segments of tap_wrap taps (GemmArgs::tap_wrap)
      const int sg = tap / p.tap_wrap;
      toff = (long)(tap - sg * p.tap_wrap) * p.tap_stride + (sg == 1 ? p.seg_off : 0);
    }
    const long koff = toff + (k0 - tap * p.cin);
    char* base = smem + buf * ST2;
    glds16b(a_src[0] + koff, base + xg0 * 1024);
    if (two_x) glds16b(a_src[1] + koff, base + xg0 * 1024 + 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16b(w_src[i] + k0, base + 256 * BK2 * 2 + wid * 2048 + i * 1024);
  };

  const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ swz2(lane & 15)) << 4);
  const int x_off = wm * 64 + frag_off;                       // frame tile, + u * 1024
  const int w_off = 256 * BK2 * 2 + wn * 64 + frag_off;       // weight tile, + v * 1024

  f32x4 acc[MT][4];
#pragma unroll
  for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 bj[4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
    bj[v] = p.bias ? *(const f32x4*)(p.bias + n0 + wn + v * 16 + (lane >> 4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};

#define SB2() __builtin_amdgcn_sched_barrier(0)
#define LGKM2() __builtin_amdgcn_s_waitcnt(0xC07F)
#define WAIT2(later)                                                          \
  do {                                                                        \
    if (two_x) {                                                              \
      if ((later) >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      \
      else if ((later) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
    } else {                                                                  \
      if ((later) >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      \
      else if ((later) == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
    }                                                                         \
  } while (0)
#define LDW(dst, slot) _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_) dst[v_] = *(const bf16x8*)(smem + (slot) * ST2 + w_off + v_ * 1024)
#define LDX(dst, slot, u0, cnt) _Pragma("unroll") for (int u_ = 0; u_ < (cnt); ++u_) dst[u_] = *(const bf16x8*)(smem + (slot) * ST2 + x_off + ((u0) + u_) * 1024)
#ifdef WFL_ABL_NOMMA      // diagnostic: keep the fragment reads alive, issue no MFMA
#define MMA2(fw, fx, u0, cnt)                                                 \
  _Pragma("unroll") for (int u_ = 0; u_ < (cnt); ++u_) asm volatile("" :: "v"(fx[u_]));   \
  _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_) asm volatile("" :: "v"(fw[v_]))
#else
#define MMA2(fw, fx, u0, cnt)                                                 \
  _Pragma("unroll") for (int u_ = 0; u_ < (cnt); ++u_)                        \
    _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_)                          \
      acc[(u0) + u_][v_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[v_], fx[u_], acc[(u0) + u_][v_], 0, 0, 0)
#endif

  // Ping-pong K loop: the two wave groups (waves 0-3 / 4-7, one wave of each per SIMD) run the same program one barrier
  // apart, so that while one group issues its LDS reads + LDS-DMA for a K step ("L slot") the other owns the matrix
  // pipe ("C slot").  Barrier b(j) ends slot j-1:
  //     group 0:      L(0) b1 C(0) b2 L(1) b3 C(1) ...          group 1:   b1 L(0) b2 C(0) b3 L(1) ...
  // Stage kt+1 becomes readable at b(2kt+2): group 0 retires its own DMA for it at the end of C(kt), group 1 at the end
  // of L(kt), both just before that barrier.  Stage kt+3 reuses the ring slot of stage kt-1, whose last reads (group 1's
  // L(kt-1)) are drained (lgkmcnt(0)) before b(2kt); it is issued after b(2kt) (group 0) / b(2kt+1) (group 1).
  const int grp = wid >> 2;
  bf16x8 fw[4], fx[MT];
  if constexpr (TRI) {
    // Ring of NS3 = 5 slots of exactly one stage (frame tile of MT * 32 rows, then the weight tile), stage q in slot q % 5, three stages
    // in flight.  Stage 2j + 3 is issued in step 3j over the slot of stage 2j - 2 (last read in step 3j - 1), stage 2j + 4 in step
    // 3j + 1 over that of 2j - 1 (last read in step 3j - 1 too): the plain loop's "a slot is rewritten no sooner than one step after
    // its last read".  Waits: step 3j retires stage 2j + 1, step 3j + 2 stage 2j + 2 (two younger stages stay in flight), step 3j + 1
    // none -- a stage has four to five steps to land.
    // What bounds it (tools/tri_stamps.py, profiles/round4_tri_stamps_*.txt): a slice takes 1.65-2.0 us; 1.2-1.5 us with the DMA removed,
    // which is the matrix pipe itself at the clock the chip holds under this load (72 MFMAs x 2 waves per SIMD x 16 cycles = 2304 cycles:
    // 1.44 us at 1.6 GHz, MI355X_MICROARCH.md "DVFS give-back").  Two other forms measured the same within a box's noise and were not kept:
    // four slots with two stages in flight, and all eight waves on one register-pipelined program (two fragment sets rotating, two
    // barriers per slice, every fragment read once: 1.74-1.83 us per slice; its 256-row form spills).
    constexpr int XB3 = MT * 32 * BK2 * 2;            // frame tile bytes
    constexpr int ST3 = XB3 + 256 * BK2 * 2;          // stage bytes: 28 KiB (192 rows) / 32 KiB (256 rows)
    constexpr int NS3 = 5;
    const int kseg = p.K / 3;
    const int nkh = kseg / BK2;
    const int nstage = 2 * nkh;
    int iss = 0;                                      // stages issued so far
    int islot = 0;                                    // slot of the next stage to issue (= iss % NS3)
    int sk0 = 0, skin = 0;                            // slice of the next stage to issue: its k, its position inside its tap
    long stoff = 0;                                   //   and its tap's offset (no division in the loop)
    auto stage3 = [&]() __attribute__((always_inline)) {
      if (iss >= nstage) return;
      const int h = iss & 1;
      const long koff = stoff + skin + (h ? p.seg_off : 0);
      const int wk = sk0 + (h ? 2 * kseg : 0);
      char* base = smem + islot * ST3;
      glds16b(a_src[0] + koff, base + xg0 * 1024);
      if (two_x) glds16b(a_src[1] + koff, base + xg0 * 1024 + 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16b(w_src[i] + wk, base + XB3 + wid * 2048 + i * 1024);
      ++iss;
      islot = islot + 1 == NS3 ? 0 : islot + 1;
      if (h) {
        sk0 += BK2; skin += BK2;
        if (skin == p.cin) { skin = 0; stoff += p.tap_stride; }
      }
    };
    // wait until stage `need` has landed (this wave's pieces): iss - need - 1 younger stages stay in flight
    auto wait3 = [&](int need) __attribute__((always_inline)) {
      if (need >= nstage) return;
      const int younger = iss - need - 1;
      WAIT2(younger);
    };
    const int x3 = wm * 64 + frag_off, w3 = XB3 + wn * 64 + frag_off;
#define LDW3(slot) _Pragma("unroll") for (int v_ = 0; v_ < 4; ++v_) fw[v_] = *(const bf16x8*)(smem + (slot) * ST3 + w3 + v_ * 1024)
#define LDX3(slot) _Pragma("unroll") for (int u_ = 0; u_ < MT; ++u_) fx[u_] = *(const bf16x8*)(smem + (slot) * ST3 + x3 + u_ * 1024)
    stage3();
    stage3();
    stage3();
    wait3(0);
    __builtin_amdgcn_s_barrier();
    STAMP2(1);
    if (grp) __builtin_amdgcn_s_barrier();
    int sh = 0;                                       // slot of stage 2j; stage 2j + 1 sits in the next one
    for (int j = 0; j < nkh; ++j) {
      const int sl = sh + 1 == NS3 ? 0 : sh + 1;
      // ---- step 3j: A_hi W_hi
      LDW3(sh);
      LDX3(sh);
#ifndef WFL_ABL_NOSTAGE
      stage3();
#endif
      if (grp) wait3(2 * j + 1);
      LGKM2();
      __builtin_amdgcn_s_barrier();
      SB2();
      MMA2(fw, fx, 0, MT);
      SB2();
      if (!grp) wait3(2 * j + 1);
      __builtin_amdgcn_s_barrier();
      SB2();
      // ---- step 3j + 1: A_lo W_hi (the weight fragments stay)
      LDX3(sl);
#ifndef WFL_ABL_NOSTAGE
      stage3();
#endif
      LGKM2();
      __builtin_amdgcn_s_barrier();
      SB2();
      MMA2(fw, fx, 0, MT);
      SB2();
      __builtin_amdgcn_s_barrier();
      SB2();
      // ---- step 3j + 2: A_hi W_lo
      LDW3(sl);
      LDX3(sh);
      if (grp) wait3(2 * j + 2);
      LGKM2();
      __builtin_amdgcn_s_barrier();
      SB2();
      MMA2(fw, fx, 0, MT);
      SB2();
      if (!grp) wait3(2 * j + 2);
      if (!(grp && j == nkh - 1)) __builtin_amdgcn_s_barrier();
      SB2();
      sh = sl + 1 == NS3 ? 0 : sl + 1;
    }
#undef LDW3
#undef LDX3
  } else {
#pragma unroll
  for (int t = 0; t < NST2 - 1; ++t)
    if (t < nk) stage(t, t);
  {
    const int later = (nk - 1 < NST2 - 2) ? nk - 1 : NST2 - 2;
    WAIT2(later);
    __builtin_amdgcn_s_barrier();
    STAMP2(1);
    if (grp) __builtin_amdgcn_s_barrier();
  }
#define WAITNEXT()                                                                         \
  do {                                                                                     \
    if (kt + 1 < nk) { const int yl = (kt + 3 < nk ? kt + 3 : nk - 1) - (kt + 1); WAIT2(yl); } \
  } while (0)
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt % NST2;
    // ---- L slot
    LDW(fw, slot);
    LDX(fx, slot, 0, MT);
#ifndef WFL_ABL_NOSTAGE
    if (kt + NST2 - 1 < nk) stage((kt + NST2 - 1) % NST2, kt + NST2 - 1);
#endif
    if (grp) WAITNEXT();
    LGKM2();
    __builtin_amdgcn_s_barrier();
    SB2();
    // ---- C slot
    MMA2(fw, fx, 0, MT);
    SB2();
    if (!grp) WAITNEXT();
    if (!(grp && kt == nk - 1)) __builtin_amdgcn_s_barrier();
    SB2();
  }
  }
#undef WAITNEXT
#undef SB2
#undef LGKM2
#undef WAIT2
#undef LDW
#undef LDX
#undef MMA2

  // ------------------------------------------------------------------ epilogue: two half-blocks through LDS
  constexpr int NC = GLU ? 128 : 256;
  constexpr int EP = NC + 4;
  constexpr int CP = NC / 8;                  // 8-channel chunks per row
  constexpr int RPP = 512 / CP;               // rows per pass
  constexpr int HR = MT * 16;                 // rows per half
  constexpr int NP = (HR + RPP - 1) / RPP;
  float* stg = (float*)smem;
  STAMP2(2);
#ifdef WFL_GEMM_STAMPS
  if (tid == 0 && p.stamps) p.stamps[(long)blockIdx.x * 8 + 7] = blockIdx.x;
#endif
  const int cidx = tid % CP;
  const int nb = (GLU ? n0 / 2 : n0) + cidx * 8;
  const int nvalid = GLU ? p.n_valid / 2 : p.n_valid;

  // bias / per-clip bias / activation on ALL eight waves at once (the staging below runs one half at a time)
  if (!GLU) {
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const float* cb = nullptr;
      if (p.clip_bias) {
        int m = m0 + wm + u * 16 + (lane & 15);
        m = m < p.M ? m : p.M - 1;
        cb = p.clip_bias + (long)p.clip_idx[m / p.P] * p.clip_ld;
      }
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        f32x4 v = acc[u][v4] + bj[v4];
        if (cb) { const f32x4 bb = *(const f32x4*)(cb + n0 + wn + v4 * 16 + (lane >> 4) * 4); v += bb; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<ACT>(v[e]);
        acc[u][v4] = v;
      }
    }
  }

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                          // operand ring / previous half fully consumed
    if ((wid >> 2) == half) {
      // acc[u][v][e]: frame ml = 16u + c, channel nl = wn + 16v + 4g + e
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        const int ml = u * 16 + (lane & 15);
        if (GLU) {
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const f32x4 ba = bj[2 * jp], bg = bj[2 * jp + 1];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[u][2 * jp][e] + ba[e]) * sigmoidf_(acc[u][2 * jp + 1][e] + bg[e]);
            *(f32x4*)(stg + ml * EP + wn / 2 + jp * 16 + (lane >> 4) * 4) = v;
          }
        } else {
#pragma unroll
          for (int v4 = 0; v4 < 4; ++v4) *(f32x4*)(stg + ml * EP + wn + v4 * 16 + (lane >> 4) * 4) = acc[u][v4];
        }
      }
    }
    __syncthreads();
    if (half == 0) STAMP2(3);
    if (nb < nvalid) {
      long orow[NP];
      int tt[NP];
      bool ok[NP];
      {
        const int m = m0 + half * HR + tid / CP;
        int b = m / p.P, t = m - b * p.P;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
          ok[pass] = (m + pass * RPP < p.M) && t < p.T && (tid / CP + pass * RPP < HR);
          tt[pass] = t;
          orow[pass] = p.c_lead + (long)b * p.c_pitch + t;
          t += RPP;
          if (p.P >= RPP) { if (t >= p.P) { t -= p.P; ++b; } }
          else while (t >= p.P) { t -= p.P; ++b; }    // (clips of at most 8 frames: a pitch below RPP wraps more than once)
        }
      }
      if (p.clip_T) {                          // ragged batches: a clip's own frame count (behind a scalar branch)
#pragma unroll
        for (int pass = 0; pass < NP; ++pass)
          if (ok[pass] && tt[pass] >= p.clip_T[(orow[pass] - p.c_lead) / p.c_pitch]) ok[pass] = false;
      }
      // residual rows are prefetched four passes at a time (all accumulators of the other half are still live, so
      // registers are tight); the positional row (one launch per forward) is read in place
      constexpr int GP = NP < 4 ? NP : 4;
#pragma unroll
      for (int g0 = 0; g0 < NP; g0 += GP) {
        bf16x8 rr[GP], rl[GP];
        if (p.res) {
#pragma unroll
          for (int q = 0; q < GP; ++q)
            if (g0 + q < NP) rr[q] = *(const bf16x8*)(p.res + orow[g0 + q] * p.ldres + nb);
          if (p.res_lo) {                             // residual stream carried as hi + lo (common.h)
#pragma unroll
            for (int q = 0; q < GP; ++q)
              if (g0 + q < NP) rl[q] = *(const bf16x8*)(p.res_lo + orow[g0 + q] * p.ldres + nb);
          }
        }
#pragma unroll
        for (int q = 0; q < GP; ++q) {
          const int pass = g0 + q;
          if (pass >= NP) continue;               // NP = 6 for the 192-row block
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
            const bf16x8 pp = *(const bf16x8*)(p.pos + (long)tt[pass] * p.ldpos + nb);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(pp[e]);
            if (TRI && p.pos_lo) {                   // the table as a pair (precise.hip's order: hi, then lo)
              const bf16x8 pl = *(const bf16x8*)(p.pos_lo + (long)tt[pass] * p.ldpos + nb);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += bf2f(pl[e]);
            }
          }
          if (p.res) {
            if (p.res_lo) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (bf2f(rr[q][e]) + bf2f(rl[q][e])) + p.alpha * v[e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = bf2f(rr[q][e]) + p.alpha * v[e];
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
      }
    }
  }
  STAMP2(4);
}

template <int ACT, bool GLU, bool OUTF32, int MT, bool TRI = false>
static int launch256_mt(const GemmArgs& a, hipStream_t s) {
  constexpr int BMV = MT * 32;
  const int tiles = ((a.M + BMV - 1) / BMV) * (a.N / BN2);
  constexpr int ring3 = 5 * (MT * 32 * BK2 * 2 + 256 * BK2 * 2);          // TRI: five one-stage slots (140 / 160 KiB)
  constexpr int lds = TRI && ring3 > LDS2 ? ring3 : LDS2;
  auto k = gemm256_kernel<ACT, GLU, OUTF32, MT, TRI>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  g_wfl_gemm_kernel_id = MT == 6 ? 2 : 3;
  hipLaunchKernelGGL(k, dim3(tiles), dim3(512), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Block height: 256 or 192 rows, whichever fills the 256 CUs in fewer / fuller rounds (one block per CU).  Cost model:
// rounds * (rows + fixed per-block overhead worth ~96 rows).
template <int ACT, bool GLU, bool OUTF32, bool TRI = false>
static int launch256_t(const GemmArgs& a, hipStream_t s) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("WFL_GEMM_BM"); forced = e ? atoi(e) : 0; }
  auto cost = [&](int bm) {
    const long tiles = (long)((a.M + bm - 1) / bm) * (a.N / BN2);
    return ((tiles + 255) / 256) * (long)(bm + 96);
  };
  const bool use192 = forced == 192 || (forced != 256 && cost(192) < cost(256));
  return use192 ? launch256_mt<ACT, GLU, OUTF32, 6, TRI>(a, s) : launch256_mt<ACT, GLU, OUTF32, 8, TRI>(a, s);
}

// Three-segment launches ("model.precision: high", GemmArgs::tap_wrap) this file walks slice by slice (template TRI).  WFL_TRI=0: the
// segment-major walk everywhere (A/B runs).
bool wfl_gemm256_tri_takes(const GemmArgs& a) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("WFL_TRI"); on = e ? atoi(e) : 1; }
  if (!on || a.tap_wrap <= 0 || a.out_f32 || a.K % 3) return false;
  const int kseg = a.K / 3;
  if (kseg % BK2 || a.cin % BK2 || (long)a.tap_wrap * a.cin != kseg) return false;
  // (no lower bound on M: the slice-by-slice walk adds its products up in another order than the segment-major one, and a clip labelled
  //  alone has to equal the same clip inside a batch bit for bit -- the walk is chosen by shape, never by batch size)
  if (a.N % BN2) return false;
  if (a.glu) return a.act == WFL_ACT_NONE;
  return a.act == WFL_ACT_NONE || a.act == WFL_ACT_GELU || a.act == WFL_ACT_RELU;
}

// Returns 1 when this kernel does not take the shape (caller falls back to the 128x128 kernel).
int wfl_launch_gemm256(const GemmArgs& a, hipStream_t s) {
  if (a.N % BN2 || a.K % BK2 || a.cin % BK2) return 1;
  if (wfl_gemm256_tri_takes(a)) {
    if (a.glu) return launch256_t<WFL_ACT_NONE, true, false, true>(a, s);
    switch (a.act) {
      case WFL_ACT_NONE: return launch256_t<WFL_ACT_NONE, false, false, true>(a, s);
      case WFL_ACT_GELU: return launch256_t<WFL_ACT_GELU, false, false, true>(a, s);
      case WFL_ACT_RELU: return launch256_t<WFL_ACT_RELU, false, false, true>(a, s);
    }
  }
  if (a.M < 2048) return 1;
  if (a.glu) return launch256_t<WFL_ACT_NONE, true, false>(a, s);
  if (a.out_f32) {
    if (a.act == WFL_ACT_NONE) return launch256_t<WFL_ACT_NONE, false, true>(a, s);
    return 1;
  }
  switch (a.act) {
    case WFL_ACT_NONE: return launch256_t<WFL_ACT_NONE, false, false>(a, s);
    case WFL_ACT_GELU: return launch256_t<WFL_ACT_GELU, false, false>(a, s);
    case WFL_ACT_RELU: return launch256_t<WFL_ACT_RELU, false, false>(a, s);
  }
  return 1;
}
