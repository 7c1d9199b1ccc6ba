// Persistent ("streaming") 192 x 256 x 32 bf16 MFMA GEMM for gfx950 -- the main kernel behind the Linear / Conv1d layers of
// the WFL-ASR forward whose output is bf16 with a plain epilogue (bias, GELU/ReLU, residual, optionally a LayerNorm
// folded in front).  Same GemmArgs contract as gemm.hip / gemm256.hip, which keep the GLU, fp32-output, positional-table
// and per-clip-bias launches.  Replaces /root/reference/model.py:18-19, 26, 31-37, 131, 140 and HF
// modeling_whisper.py:309-354, 391-407 (q/k/v/out projections, FFN), HF modeling_whisper.py:384-385/401-402 (the
// pre-LayerNorms, when folded).
//
// What differs from gemm256.hip (whose tile, ring, swizzle and ping-pong K loop it shares):
//   * one workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ... and keeps ONE operand stream going across tile
//     boundaries: the LDS-DMA for the first three K steps of the next tile is issued during the last three steps of
//     the current one, so a tile never pays the ~2 us pipeline fill again;
//   * the epilogue never touches LDS (the ring belongs to the next tile by then): the weight-tile rows each lane feeds
//     to the MFMA are permuted so that a lane ends up with two runs of 8 consecutive output channels of one frame, i.e.
//     two 16-byte stores per 16-frame tile, 64 contiguous bytes per frame per store instruction.  The stores are not
//     waited for: they drain under the next tile's K loop (vmcnt counts them, in order, so the first two waits of a new
//     tile allow for them; from the third on they must have retired).  Stores of rows / columns that must not be written
//     (halo frames, n >= n_valid) go to a scratch line instead of being branched around, which keeps the count exact;
//   * each wave group runs the previous tile's epilogue at the start of its first L slot of the next tile, i.e. while the
//     other group owns the matrix pipe;
//   * LayerNorm folding (GemmArgs::ln_s != null):  LN(x) W^T + b = rstd * (x W'^T - mean * s) + b'  with W' = gamma o W,
//     s_n = sum_k W'_nk, b' = b + W beta (packed at load time).  The row statistics either come from the GEMM that produced
//     x (template STATS on its residual epilogue: per row and 256-column tile, sum and sum of squares of the bf16 values it
//     stores; the four waves' partials meet in LDS and the last arriver -- an LDS ticket, no barrier -- adds them in a fixed
//     order and writes them; the consumer, LNF = 2, reads them in its epilogue), or are summed inside the consumer from the
//     fragments its MFMAs consume (LNF = 1, v_dot2c_f32_bf16 in the C slots; measured slower, kept for experiments);
//   * a tap-stationary mode for dense multi-tap convolutions (template CONV, see the kernel).
// The K loop itself is bound by the L2 -> LDS operand DMA (28 KiB per 32-deep step of a 192 x 256 tile; ~65 GB/s per CU) next
// to 24 MFMAs per wave: tools/gemm_lab.py's ablations give 0.40 us per step for the MFMAs alone, 0.45-0.50 us for the DMA
// alone and 0.59-0.64 us for the real loop.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define SBK 32
#ifndef SNST
#define SNST 4         // ring depth (stages): SNST - 1 stages in flight ahead of the one being computed.  The code is written for 3 .. 6;
#endif                 // measured (tools/gemm_lab.py r4=-DSNST=4 r5=-DSNST=5): 5 stages change nothing (0.611 vs 0.612 us per K step at
                       // K = 2048), so the 0.12 us the operand DMA adds to a K step is not latency a deeper ring could hide; neither is
                       // it the place the LDS-DMA instructions are issued from (WFL_DMA_IN_C = 1, 2: same time).  What is left is the
                       // issue cost of the 28 one-KiB LDS-DMA instructions themselves: the per-CU L2 -> LDS ceiling.

#ifndef SNCU
#define SNCU 256       // persistent workgroups per launch (lab: -DSNCU=128 lets two streams' launches run side by side: same step time, round 4)
#endif

typedef __attribute__((address_space(1))) const void* sgptr_t;
typedef __attribute__((address_space(3))) void* slptr_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

static __device__ __forceinline__ void sglds(const bf16_t* g, char* l) {
  __builtin_amdgcn_global_load_lds((sgptr_t)g, (slptr_t)l, 16, 0, 0);
}
static __device__ __forceinline__ int sswz(int row) { return (-(row >> 2)) & 3; }
// Swizzle of the CONV extended frame tile: its fragment reads start at ANY row (one per tap), and 2 * ((row >> 2) & 1) is one of
// the eight 16-byte-chunk XORs that keep 16 consecutive rows from any start conflict-free for ds_read_b128 (exhaustive search
// over the functions of (row >> 2) & 3; the aligned-only sswz above costs 2-way conflicts on 3 taps of 4: +0.10 us per K step).
static __device__ __forceinline__ int xswz(int row) { return ((row >> 2) & 1) << 1; }
template <int N>
static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef WFL_GEMM_STAMPS
// (slots 0-4: 100 MHz wall clock at the phase boundaries; slots 5 / 6: the core-clock counter at stamps 1 / 2 -- the clock the chip holds inside the
//  K loop is their difference over the wall time between the two, MI355X_MICROARCH.md "DVFS give-back" item 6)
#define SSTAMP(k) do { if (tid == 0 && p.stamps) { p.stamps[(long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 1) p.stamps[(long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 2) p.stamps[(long)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif

// LNF: 0 no LayerNorm, 1 folded with statistics summed in the kernel, 2 folded with statistics read from p.stats_in.
// STATS: the (residual) epilogue also writes p.stats_out for the next LayerNorm-folded consumer.
// CONV: dense multi-tap Conv1d (taps one input row apart, K = taps x cin): K is walked channel-chunk-major / tap-minor and the
//   frame operand of a whole chunk -- the tile's 192 rows plus the taps - 1 rows of overlap, 32 channels -- is staged ONCE and
//   re-read at a one-row offset per tap, so a K step moves 16 KiB (the weight tile) instead of 28: the k = 31 Conformer conv
//   (a quarter of the model's FLOPs) leaves the L2 -> LDS bound and becomes MFMA-bound.
// W8: the weight operand is OCP e4m3 (one byte per element, per-output-channel fp32 scale p.w8_scale): the weight tile of a K
//   step is 8 KiB instead of 16 (one LDS-DMA piece per wave instead of two: 20 KiB per step instead of 28 -- the K loop is bound
//   by exactly these bytes), fragments are read as 8 bytes and converted to bf16 in registers (v_cvt_pk_f32_fp8 +
//   v_cvt_pk_bf16_f32, exact), the MFMA stays bf16 x bf16, the scale multiplies the accumulator in the epilogue.
// A8: BOTH operands are e4m3 (GemmArgs::a8).  A row of 64 e4m3 values is as long as a row of 32 bf16 values, so the kernel stages,
//   swizzles, counts and reads exactly as for bf16 -- the launcher presents the byte matrices as bf16 matrices of half the width -- and
//   a 16-byte fragment holds the k's of TWO v_mfma_f32_16x16x32_fp8_fp8 (its low and its high 8 bytes; frame and weight fragments
//   split the same way, so the two instructions contract matching k's): a step moves the same 28 KiB for twice the FLOPs and holds 48
//   MFMAs per wave between its two barriers instead of 24 -- the loop is bound by the matrix pipe, not by the operand DMA.  (Round 3's
//   first form -- 32-byte rows, half-size stages -- kept the 24-MFMA step and its barrier overhead: 0.60 us per 32 k's against 0.79
//   for W8 and ~0.45 now.)  The scales (per output channel, per frame row) multiply the accumulator in the epilogue.
//   OUT8: the epilogue writes e4m3 (GemmArgs::c8) instead of bf16 -- fc1's GELU output, the next GEMM's fp8 operand.
template <int ACT, int MT, bool RES, int LNF, bool STATS, bool CONV, bool W8 = false, bool A8 = false, bool OUT8 = false>
__global__ __launch_bounds__(512) void gemm_stream_kernel(GemmArgs p) {
  static_assert(!CONV || (MT == 6 && !RES && LNF == 0 && !STATS), "conv mode: 192-row tiles, plain epilogue");
  static_assert(!W8 || (!CONV && MT == 6), "fp8 weights: plain 192-row mode");
  static_assert(!A8 || (!W8 && !CONV && MT == 6 && LNF == 0 && !STATS), "fp8 x fp8: the plain 192-row mode, no LayerNorm folding");
  static_assert(!OUT8 || (A8 && !RES), "fp8 output: the fp8 x fp8 kernel, no residual");
  constexpr int BMV = MT * 32;                    // frame rows per tile
#ifdef WFL_LAB_STB32
  constexpr int STB = 512 * SBK * 2;
  constexpr int WOFF = 256 * SBK * 2;
#else
  constexpr int STB = CONV ? 256 * SBK * 2 : BMV * SBK * 2 + 256 * SBK * (W8 ? 1 : 2);   // stage bytes: frame tile then weight tile
  constexpr int WOFF = CONV ? 0 : BMV * SBK * 2;
#endif
  constexpr int AEXT = 224 * SBK * 2;             // CONV: extended frame tile (BMV + up to 32 taps - 1 rows), two of them
  constexpr int AOFF = SNST * STB;
  constexpr bool LOUT = RES || CONV;                  // epilogues that can write a low half (CONV: "model.precision: high" convs, c_lo without a residual)
  constexpr int NSTORE = (LOUT ? 4 : 2) * MT + (STATS ? MT : 0);   // epilogue stores per wave (never branched around; a residual
                                                                  // launch always stores the hi and the lo half)
  constexpr int SROW = 4;                             // float2 slots per row of the statistics buffer (one per 256-column tile)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stat_lds = (float*)(smem + SNST * STB + (CONV ? 2 * AEXT : 0));  // LNF == 1: [2 groups][MT*16 rows][2]; STATS: [2][4 waves][MT*16][2] + 2 counters
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3;
  const int g = lane >> 4, c = lane & 15;
  const int G = gridDim.x;

  SSTAMP(0);
  const int tiles_n = p.N / 256;
  const int tiles_m = (p.M + BMV - 1) / BMV;
  const int ntiles = tiles_m * tiles_n;
  const int nk = p.K / SBK;
  auto tile_of = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {   // XCD-aware order: v and v + 8 share an XCD; contiguous run per XCD
    const int q = ntiles >> 3, r = ntiles & 7, x = v & 7, i = v >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const int tm = bid / tiles_n;
    m0 = tm * BMV;
    n0 = (bid - tm * tiles_n) * 256;
  };
  const int wm = grp * (MT * 16), wn = wq * 64;

  // ---- operand stream (prefetch cursor).  Frame row groups: MT = 8: waves load two each; MT = 6: waves 0-3 two, 4-7 one.
  // (CONV: the 14 row groups of the extended tile: waves 0-5 two each, 6-7 one)
  const bool two_x = CONV ? wid < 6 : (MT == 8 || wid < 4);
  const int xg0 = CONV ? (wid < 6 ? wid * 2 : 12 + (wid - 6)) : ((MT == 8 || wid < 4) ? wid * 2 : 8 + (wid - 4));
  // CONV with GemmArgs::tap_wrap ("model.precision: high"): K holds three segments [A_hi W_hi | A_lo W_hi | A_hi W_lo] of tap_wrap taps each; the
  // channel-chunk counter then runs over 3 x (cin / SBK) chunks, chunk c of segment sg reading A at (sg == 1 ? seg_off : 0) + c * SBK and W at
  // sg * tap_wrap * cin + tap * cin + c * SBK (the K order inside the launch: segment, chunk, tap)
  const int ntaps = CONV ? (p.tap_wrap > 0 ? p.tap_wrap : p.K / p.cin) : 1;
  const int nchunk = CONV ? p.cin / SBK : 1;
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
  const char* w8_src = nullptr;                     // W8: this lane's 16 source bytes of the wave's ONE weight piece (32 rows x 32 bytes)
  int pv = blockIdx.x, pkt = 0, issued = 0;
  int islot = 0;                                    // ring slot of the next stage to issue (= issued % SNST)
  int ptap_k = 0;                                   // position inside the current tap (conv GEMMs), elements
  long pbase = 0;                                   // tap * tap_stride
  int ptap_i = 0, pseg = 0;                         // GemmArgs::tap_wrap: tap inside the segment, segment
  auto next_tap = [&]() __attribute__((always_inline)) {
    ptap_k = 0;
    if (p.tap_wrap > 0 && ++ptap_i == p.tap_wrap) { ptap_i = 0; ++pseg; pbase = pseg == 1 ? p.seg_off : 0; }
    else pbase += p.tap_stride;
  };
  int pcc = 0, ptap = 0, pccg = 0;                  // CONV: channel chunk / tap being issued; chunks issued so far (buffer parity)
  long pa_off = 0, pw_off = 0;                      // CONV: element offsets of the chunk being issued into A and into a W row
  auto set_src = [&](int v) __attribute__((always_inline)) {
    int m0, n0;
    tile_of(v, m0, n0);
    int lane = tid & 63;                             // opaque copy: the lane-dependent parts of the four source pointers are recomputed per
    asm volatile("" : "+v"(lane));                   //   tile (a few VALU ops) instead of living in registers -- and spilling -- across the K loops
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wrow = (wid * 2 + i) * 16 + (lane >> 2);
      const int xrow = (xg0 + i) * 16 + (lane >> 2);
      int am = m0 + xrow;
      const int am_max = CONV ? p.M - 1 + ntaps - 1 : p.M - 1;     // (conv: row m + tap of the caller's buffer, as in gemm.hip)
      am = am < am_max ? am : am_max;
      a_src[i] = p.A + (long)am * p.lda + ((lane & 3) ^ (CONV ? xswz(xrow) : sswz(xrow))) * 8;
      w_src[i] = p.W + (long)(n0 + wrow) * p.K + ((lane & 3) ^ sswz(wrow)) * 8;
    }
    if (W8) {
      // fp8 weight tile in LDS: 256 rows x 32 bytes; 8 rows = one 256-byte bank row = 32 slots of 8 bytes, slot (r & 7) * 4 + q for
      // the q-th 8 k's of row r, XORed with ((r >> 3) & 1) << 1 | ((r >> 4) & 1) << 4 so that the 32 lanes of a ds_read_b64 group
      // (rows 8 j + i, i, j < 4) hit 32 different slots.  The DMA writes lane l's 16 bytes to 16-byte unit l & 15 of bank row
      // Gr = 4 wid + (l >> 4); it therefore FETCHES the logical unit (l & 15) ^ (Gr & 1 | (Gr >> 1 & 1) << 3).
      const int Gr = wid * 4 + (lane >> 4);
      const int ul = (lane & 15) ^ ((Gr & 1) | (((Gr >> 1) & 1) << 3));
      w8_src = (const char*)p.W + (long)(n0 + 8 * Gr + (ul >> 1)) * p.K + (ul & 1) * 16;
    }
  };
  set_src(pv);
  auto issue_stage = [&]() __attribute__((always_inline)) {                        // the DMA of stage (pv, pkt); advances the position inside the tile
    if (CONV) {
      char* base = smem + islot * STB;
      islot = islot + 1 == SNST ? 0 : islot + 1;
      if (ptap == 0) {                               // first tap of a channel chunk: its extended frame tile rides along
        char* ab = smem + AOFF + (pccg & 1) * AEXT;
        sglds(a_src[0] + pa_off, ab + xg0 * 1024);
        if (two_x) sglds(a_src[1] + pa_off, ab + xg0 * 1024 + 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) sglds(w_src[i] + ptap * p.cin + pw_off, base + wid * 2048 + i * 1024);
      ++issued;
      ++pkt;
      if (++ptap == ntaps) {
        ptap = 0; ++pcc; ++pccg;
        if (pcc == nchunk) { pcc = 0; ++pseg; }      // (segments only with tap_wrap; a plain conv ends its tile here)
        pa_off = (pseg == 1 ? p.seg_off : 0) + (long)pcc * SBK;
        pw_off = (long)pseg * ntaps * p.cin + (long)pcc * SBK;
      }
      return;
    }
    const long koff = pbase + ptap_k;
    char* base = smem + islot * STB;
    islot = islot + 1 == SNST ? 0 : islot + 1;
    sglds(a_src[0] + koff, base + xg0 * 1024);
    if (two_x) sglds(a_src[1] + koff, base + xg0 * 1024 + 1024);
    if (W8) {
      sglds((const bf16_t*)(w8_src + pkt * SBK), base + WOFF + wid * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) sglds(w_src[i] + pkt * SBK, base + WOFF + wid * 2048 + i * 1024);
    }
    ++issued;
    ++pkt;
    ptap_k += SBK;
    if (ptap_k == p.cin) next_tap();
  };
  // Steady-state variant: the frame pieces (and the weight pieces that are not deferred) now, the NDC deferred weight pieces from
  // inside the MFMA block of the same step (issue_deferred).  An LDS-DMA instruction issued right behind a burst of ds_reads costs
  // the wave 100-185 cycles, one issued between MFMAs about 60 (MI355X_MICROARCH.md, cycle constants): with all four in the L slot
  // the L slot (~750 cycles) was twice the C slot (24 MFMAs = 384) and set the K step.
  int d_pkt = 0;
  char* d_base = nullptr;
  auto issue_stage_split = [&](auto ndc_c) __attribute__((always_inline)) {
    constexpr int NDC = decltype(ndc_c)::value;
    const long koff = pbase + ptap_k;
    char* base = smem + islot * STB;
    islot = islot + 1 == SNST ? 0 : islot + 1;
    sglds(a_src[0] + koff, base + xg0 * 1024);
    if (two_x) sglds(a_src[1] + koff, base + xg0 * 1024 + 1024);
    if (W8) {
      if (NDC == 0) sglds((const bf16_t*)(w8_src + pkt * SBK), base + WOFF + wid * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2 - NDC; ++i) sglds(w_src[i] + pkt * SBK, base + WOFF + wid * 2048 + i * 1024);
    }
    d_pkt = pkt; d_base = base;
    ++issued;
    ++pkt;
    ptap_k += SBK;
    if (ptap_k == p.cin) next_tap();
  };
  auto issue_deferred = [&](int j, auto ndc_c) __attribute__((always_inline)) {     // j-th deferred weight piece (j < NDC)
    constexpr int NDC = decltype(ndc_c)::value;
    if (W8) sglds((const bf16_t*)(w8_src + d_pkt * SBK), d_base + WOFF + wid * 1024);
    else sglds(w_src[2 - NDC + j] + d_pkt * SBK, d_base + WOFF + wid * 2048 + (2 - NDC + j) * 1024);
  };
  auto prefetch_one = [&]() __attribute__((always_inline)) {                       // issue the next stage of the stream, if there is one
    if (pv >= ntiles) return;
    issue_stage();
    if (pkt == nk) {
      pkt = 0; ptap_k = 0; pbase = 0; pcc = 0; ptap = 0; ptap_i = 0; pseg = 0; pa_off = 0; pw_off = 0;
      pv += G;
      if (pv < ntiles) set_src(pv);
    }
  };
  // wait until the stage of global step `need` has landed (this wave's part), given `issued` stages so far and whether an
  // epilogue's stores were issued after it.  Younger stages: issued - need - 1 in {0, 1, 2}.
  auto wait_stage = [&](int need, bool with_stores) __attribute__((always_inline)) {
    const int younger = issued - need - 1;
    if (younger < 0) return;                        // no such stage (end of the stream)
    // DMA pieces per wave and stage: CONV two weight pieces (the extended frame tile's pieces, one stage in `taps`, are not counted:
    // a smaller count only waits for more, and they are issued taps - 3 steps before their first read); else two / one frame
    // pieces + the weight pieces
    constexpr int L2X = CONV ? 2 : (W8 ? 3 : 4), L1X = CONV ? 2 : (W8 ? 2 : 3);
    static_assert(SNST >= 3 && SNST <= 6, "ring depth");
#define WFL_WAITY(L)                                                                              \
    do {                                                                                          \
      if (SNST >= 6 && younger >= 4) { if (with_stores) wait_vm<4 * (L) + NSTORE>(); else wait_vm<4 * (L)>(); }       \
      else if (SNST >= 5 && younger >= 3) { if (with_stores) wait_vm<3 * (L) + NSTORE>(); else wait_vm<3 * (L)>(); }  \
      else if (younger >= 2) { if (with_stores) wait_vm<2 * (L) + NSTORE>(); else wait_vm<2 * (L)>(); }              \
      else if (younger == 1) { if (with_stores) wait_vm<(L) + NSTORE>(); else wait_vm<(L)>(); }                      \
      else { if (with_stores) wait_vm<NSTORE>(); else wait_vm<0>(); }                                                \
    } while (0)
    if (CONV || two_x) WFL_WAITY(L2X); else WFL_WAITY(L1X);
#undef WFL_WAITY
  };

  // ---- fragment addressing.  Weight-tile row feeding MFMA row i of column tile v is channel
  //   ch(v, i) = 32 (v >> 1) + 8 (i >> 2) + 4 (v & 1) + (i & 3)
  // so that lane (g, c) holds channels 8g .. 8g+7 (v = 0, 1) and 32 + 8g .. +7 (v = 2, 3) of frame c.
  int w_off[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
#ifdef WFL_LAB_NOPERM
    const int r = wn + 16 * v + c;
#else
    const int r = wn + 32 * (v >> 1) + 8 * (c >> 2) + 4 * (v & 1) + (c & 3);
#endif
    w_off[v] = W8 ? WOFF + (r >> 3) * 256 + ((((r & 7) * 4 + g) ^ (((r >> 3) & 1) << 1) ^ (((r >> 4) & 1) << 4)) << 3)
                  : WOFF + r * 64 + ((g ^ sswz(r)) << 4);
  }
  const int x_off = wm * 64 + c * 64 + ((g ^ sswz(c)) << 4);      // + u * 1024 (16 rows; sswz(16u + c) == sswz(c))

  f32x4 acc[MT][4];
#pragma unroll
  for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // LayerNorm statistics partials of this wave's share of the 16-frame tiles (u = wq and wq + 4)
  float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;   // (named, not an array: a runtime index would send them to scratch)

  // ---- epilogue of tile (m0, n0): registers -> HBM
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    const int nb = n0 + wn + 8 * g;                 // first channel of this lane's first run; second run at +32
    // Opaque bases (round 4): left to itself hipcc computes the MT per-tile row indices and statistics-table addresses once, in front of
    // the K loop, and keeps them in registers across it -- registers this kernel does not have, so they spill, and a spill's reload inside
    // the epilogue is an `s_waitcnt vmcnt(0)`: a wait for the operand DMA in flight AND for every store issued so far, in the one place
    // built around never waiting for a store.  Computed here from values the compiler cannot see through, they cost one add each.
    int mrow0 = m0 + wm + c;
    int sidx_w = (((grp * 4 + wq) * (MT * 16)) + c) * 2, sidx_g = ((grp * 4 * (MT * 16)) + c) * 2, sidx_l = (grp * (MT * 16) + c) * 2;
    asm volatile("" : "+v"(mrow0), "+v"(sidx_w), "+v"(sidx_g), "+v"(sidx_l));
    f32x4 bj[4], sj[4], cj[(W8 || A8) ? 4 : 1];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#ifdef WFL_ABL_EPI_NOLOAD       // (diagnostic, tools/gemm_lab.py: the epilogue without its operand loads -- what waiting for them behind the DMA costs)
        bj[2 * h + q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (LNF) sj[2 * h + q] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (W8 || A8) cj[2 * h + q] = (f32x4){1.f, 1.f, 1.f, 1.f};
#else
        bj[2 * h + q] = p.bias ? *(const f32x4*)(p.bias + nb + 32 * h + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (LNF) sj[2 * h + q] = *(const f32x4*)(p.ln_s + nb + 32 * h + 4 * q);
        if (W8 || A8) cj[2 * h + q] = *(const f32x4*)(p.w8_scale + nb + 32 * h + 4 * q);
#endif
      }
    float sa[A8 ? MT : 1];                          // A8: the frame rows' scales
    if (A8) {
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        int m = mrow0 + 16 * u;
        m = m < p.M ? m : p.M - 1;
        sa[u] = p.a8_scale ? p.a8_scale[p.a8_lead + m] : p.a8_static;
      }
    }
    int P_ = p.P, cpitch = p.c_pitch;               // opaque too: 1 / P would otherwise be kept across the K loop
    asm volatile("" : "+s"(P_), "+s"(cpitch));
    const float invP = 1.0f / (float)P_;
    int orow[MT];                                   // output row index, or -1 for rows that are not stored
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int m = mrow0 + 16 * u;
      int b = (int)((float)m * invP);
      int t = m - b * P_;
      if (t < 0) { t += P_; --b; }
      if (t >= P_) { t -= P_; ++b; }
      orow[u] = (m < p.M && t < p.T) ? (int)p.c_lead + b * cpitch + t : -1;
    }
    if (p.clip_T) {                                 // ragged batches: a clip's own frame count (a scalar branch around six loads: inside
#pragma unroll                                      // the expression above they cost the LayerNorm-folded launches 2 us each)
      for (int u = 0; u < MT; ++u)
        if (orow[u] >= 0) {
          const int m = mrow0 + 16 * u;             // (clip and frame once more from the row index: an integer division by c_pitch
          int b = (int)((float)m * invP);           //  would keep its magic reciprocal in a register across the K loop)
          int t = m - b * P_;
          if (t < 0) { t += P_; --b; }
          if (t >= P_) { t -= P_; ++b; }
          if (t >= p.clip_T[b]) orow[u] = -1;
        }
    }
    // residual hi + lo halves: a ring of three 16-frame tiles in flight (all six at once would not fit the register file next
    // to the accumulators); tile u + 3 is requested as soon as tile u has been consumed
    constexpr int RING = 3;
    bf16x8 rr[RES ? RING : 1][2], rl[RES ? RING : 1][2];
    const bf16_t* res_lo = p.res_lo ? p.res_lo : p.res;     // no low half: read the high one again and scale it away
    const float lo_scale = p.res_lo ? 1.f : 0.f;
    auto load_res = [&](int u, int slot) __attribute__((always_inline)) {
      const long r = orow[u] >= 0 ? orow[u] : p.c_lead;     // any valid row: the value is never stored
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#ifdef WFL_ABL_EPI_NOLOAD
        rr[slot][h] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; rl[slot][h] = rr[slot][h]; (void)r;
#else
        rr[slot][h] = *(const bf16x8*)(p.res + r * p.ldres + nb + 32 * h);
        rl[slot][h] = *(const bf16x8*)(res_lo + r * p.ldres + nb + 32 * h);
#endif
      }
    };
    if (RES) {
#pragma unroll
      for (int u = 0; u < RING && u < MT; ++u) load_res(u, u);
    }
    float mu[LNF ? MT : 1], rs[LNF ? MT : 1];
    if (LNF == 1) {
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        const float2 s = *(const float2*)(stat_lds + sidx_l + u * 32);
        mu[u] = s.x;
        rs[u] = s.y;
      }
    }
    if (LNF == 2) {
      // the producer left (sum, sum of squares) per 256-column tile of every row, SROW slots per row; slots are added in a
      // fixed order, the unused ones masked (all loads are issued before the first use)
      f32x4 sp[MT][2];
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        int m = mrow0 + 16 * u;
        m = m < p.M ? m : p.M - 1;
        const f32x4* q = (const f32x4*)(p.stats_in + (p.stats_lead + m) * (2 * SROW));
        sp[u][0] = q[0];
        sp[u][1] = q[1];
      }
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        float a1 = sp[u][0][0], a2 = sp[u][0][1];
        if (p.stats_nsl > 1) { a1 += sp[u][0][2]; a2 += sp[u][0][3]; }
        if (p.stats_nsl > 2) { a1 += sp[u][1][0]; a2 += sp[u][1][1]; }
        if (p.stats_nsl > 3) { a1 += sp[u][1][2]; a2 += sp[u][1][3]; }
        const float mean = a1 / (float)p.K;
        mu[u] = mean;
        rs[u] = rsqrtf(fmaxf(a2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
      }
    }
    char* trash = (char*)p.trash + lane * 16;
#ifdef WFL_ABL_EPI_NOSTORE      // diagnostic build (wrong results): every output store goes to the scratch line -- what the stores cost the launch
#define WFL_KEEP(k) false      //   (tools/ab_lib.py nostore gemm_stream.hip -DWFL_ABL_EPI_NOSTORE: cfg2 128.3 k -> 154.0 k audio-s/s, profiles/round4_cfg2_store_cost.txt)
#else
#define WFL_KEEP(k) (k)
#endif
    float t1 = 0.f, t2 = 0.f;
    float amax8 = 0.f;                               // OUT8: largest stored |x| * scale (above 448 it did not fit e4m3)
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      __builtin_amdgcn_sched_barrier(0);            // one 16-frame tile at a time: keeps the register footprint bounded
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[u][2 * h + q][e];
            if (W8 || A8) v *= cj[2 * h + q][e];
            if (A8) v *= sa[u];
            if (LNF) v = (v - mu[u] * sj[2 * h + q][e]) * rs[u];
            v = apply_act<ACT>(v + bj[2 * h + q][e]);
            if (RES) v = (bf2f(rr[u % RING][h][4 * q + e]) + lo_scale * bf2f(rl[u % RING][h][4 * q + e])) + p.alpha * v;
            x[4 * q + e] = v;
          }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(x[e]);
        const bool keep = orow[u] >= 0 && nb + 32 * h < p.n_valid;
        if (OUT8) {                                  // e4m3 (saturating at +-448) instead of bf16: 8 bytes per run of 8 channels
          int w0 = 0, w1 = 0;
          const float keepf = keep ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float y = x[e] * p.c8_inv_scale;
            amax8 = fmaxf(amax8, fabsf(y) * keepf);
            x[e] = fminf(fmaxf(y, -448.f), 448.f);
          }
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], w1, true);
          char* d8 = (char*)(p.c8 + (long)orow[u] * p.ldc8 + nb + 32 * h);
          d8 = WFL_KEEP(keep) ? d8 : trash;
          *(uint2*)d8 = make_uint2((unsigned)w0, (unsigned)w1);
          continue;
        }
        char* dst = (char*)((bf16_t*)p.C + (long)orow[u] * p.ldc + nb + 32 * h);
        dst = WFL_KEEP(keep) ? dst : trash;
        *(bf16x8*)dst = o;
        if (LOUT) {                                  // low half: what the bf16 rounding of the sum left behind
          bf16x8 ol;
#pragma unroll
          for (int e = 0; e < 8; ++e) ol[e] = f2bf(x[e] - bf2f(o[e]));
          char* dl = (char*)(p.c_lo + (long)orow[u] * p.ldc + nb + 32 * h);
          dl = WFL_KEEP(keep && p.c_lo) ? dl : trash;
          *(bf16x8*)dl = ol;
        }
        if (STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float r = bf2f(o[e]); t1 += r; t2 = fmaf(r, r, t2); }
        }
      }
      if (STATS) {                                   // this wave's 64 columns of frame c: add the four lane groups; the
        t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);   // partial goes to this wave's LDS slot
        t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
        if (g == 0) *(float2*)(stat_lds + sidx_w + u * 32) = float2{t1, t2};
        t1 = t2 = 0.f;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (RES && u + RING < MT) load_res(u + RING, u % RING);
    }
    if (OUT8 && p.err) {                             // (only elements that were stored are judged: halo rows and padding columns are not)
      if (__builtin_amdgcn_ballot_w64(amax8 > 448.f) && lane == 0) atomicOr(p.err, 2u);
    }
    if (STATS) {
      // The last of the group's four waves to get here adds the four partials of every row in a fixed order (bit-exact
      // whatever the arrival order) and writes (sum, sum of squares) for this 256-column tile; the others send their MT
      // stores to the scratch line so every wave's store count stays the same.  No barrier: LDS executes the group's
      // writes, the counter atomics and the reads in issue order, and a wave drains its writes before it takes a ticket.
      unsigned* cnt = (unsigned*)(stat_lds + 8 * (MT * 16) * 2) + grp;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(cnt, 1u);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      const bool last = ticket == 3;
      if (last && lane == 0) *cnt = 0;
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          const float2 v = *(const float2*)(stat_lds + sidx_g + w4 * (MT * 32) + u * 32);
          a1 += v.x; a2 += v.y;
        }
        float* sd = p.stats_out + ((long)orow[u] * SROW + (n0 >> 8)) * 2;
        sd = (last && orow[u] >= 0 && g == 0) ? sd : (float*)trash;
        *(float2*)sd = float2{a1, a2};
      }
    }
  };

  if (STATS && tid < 2) ((unsigned*)(stat_lds + 8 * (MT * 16) * 2))[tid] = 0;   // the groups' arrival counters
  // ---- prologue: three stages in flight, stage 0 landed, group 1 one barrier behind
#pragma unroll
  for (int t = 0; t < SNST - 1; ++t) prefetch_one();
  wait_stage(0, false);
  __builtin_amdgcn_s_barrier();
  SSTAMP(1);
  if (grp) __builtin_amdgcn_s_barrier();

#define SSB() __builtin_amdgcn_sched_barrier(0)
  bf16x8 fw[4], fx[MT];
#ifndef WFL_W8_CVT_IN_C
#define WFL_W8_CVT_IN_C 1   // W8: the e4m3 -> bf16 conversion of a step's weight fragments in the C slot (each right in front of its first MFMA) instead of the L slot
#endif
  uint2 fraw[W8 ? 4 : 1];
  int s = 0;                                         // global K-step counter
  int rslot = 0;                                     // its ring slot (= s % SNST)
  int ctap = 0, ccg = 0;                              // CONV: tap of the step being computed; chunks finished (buffer parity)
  auto read_frags = [&]() __attribute__((always_inline)) {
    const char* sb = smem + rslot * STB;
    if (W8 && WFL_W8_CVT_IN_C) {
#pragma unroll
      for (int v = 0; v < 4; ++v) fraw[v] = *(const uint2*)(sb + w_off[v]);     // converted behind the barrier, among the MFMAs (mma_all)
    } else if (W8) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const uint2 raw = *(const uint2*)(sb + w_off[v]);
        // e4m3 -> bf16, two values per instruction (v_cvt_scalef32_pk_bf16_fp8, scale 1: exact -- round 4; rounds 2-3 went through fp32
        // with v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32, twice the instructions in the L slot that bounds this kernel's K step)
        const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, true);
        const bf16x2 c2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, false), d2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, true);
        fw[v] = (bf16x8){a[0], a[1], b[0], b[1], c2[0], c2[1], d2[0], d2[1]};
      }
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) fw[v] = *(const bf16x8*)(sb + w_off[v]);
    }
    if (CONV) {
      const char* ab = smem + AOFF + (ccg & 1) * AEXT;
      const int r0 = wm + c + ctap;                  // row of the extended tile; + 16u never changes the swizzle
      const int xb = r0 * 64 + ((g ^ xswz(r0)) << 4);
#pragma unroll
      for (int u = 0; u < MT; ++u) fx[u] = *(const bf16x8*)(ab + xb + u * 1024);
      if (++ctap == ntaps) { ctap = 0; ++ccg; }
      return;
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) fx[u] = *(const bf16x8*)(sb + x_off + u * 1024);
  };
  // The MFMAs of one K step; with a folded LayerNorm, wave wq of a group also sums the frame fragments u = wq and wq + 4
  // (v_dot2c_f32_bf16: sum and sum of squares), issued right behind that tile's MFMAs so they run in the MFMAs' shadow.
  auto mma_all = [&](auto ndc_c) __attribute__((always_inline)) {
    constexpr int NDC = decltype(ndc_c)::value;      // deferred weight pieces to issue between the MFMAs (0: none)
    const bf16x2 one2 = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
    for (int u = 0; u < MT; ++u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (W8 && WFL_W8_CVT_IN_C && u == 0) {
          const uint2 raw = fraw[v];
          const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, true);
          const bf16x2 c2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, false), d2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, true);
          fw[v] = (bf16x8){a[0], a[1], b[0], b[1], c2[0], c2[1], d2[0], d2[1]};
        }
        if (A8) {                                    // a 16-byte fragment = the k's of two fp8 MFMAs (low half, high half)
          typedef __attribute__((ext_vector_type(2))) long i64x2;
          const i64x2 a = __builtin_bit_cast(i64x2, fw[v]), b = __builtin_bit_cast(i64x2, fx[u]);
          acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[0], b[0], acc[u][v], 0, 0, 0);
          acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[1], b[1], acc[u][v], 0, 0, 0);
        } else acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[v], fx[u], acc[u][v], 0, 0, 0);
      }
      if (NDC >= 1 && u == 1) { __builtin_amdgcn_sched_barrier(0); issue_deferred(0, ndc_c); __builtin_amdgcn_sched_barrier(0); }
      if (NDC >= 2 && u == 3) { __builtin_amdgcn_sched_barrier(0); issue_deferred(1, ndc_c); __builtin_amdgcn_sched_barrier(0); }
      if (LNF == 1) {
        if ((u & 3) == wq) {                         // wave-uniform
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x2 xv = {fx[u][2 * j], fx[u][2 * j + 1]};
            if (u < 4) {
              sa1 = __builtin_amdgcn_fdot2_f32_bf16(xv, one2, sa1, false);
              sa2 = __builtin_amdgcn_fdot2_f32_bf16(xv, xv, sa2, false);
            } else {
              sb1 = __builtin_amdgcn_fdot2_f32_bf16(xv, one2, sb1, false);
              sb2 = __builtin_amdgcn_fdot2_f32_bf16(xv, xv, sb2, false);
            }
          }
        }
      }
    }
  };
  auto ln_publish = [&]() __attribute__((always_inline)) {   // mean / rstd of this wave's frames -> its group's LDS table
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float a1 = i ? sb1 : sa1, a2 = i ? sb2 : sa2;
      a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
      a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
      const float mean = a1 / (float)p.K;
      const float var = fmaxf(a2 / (float)p.K - mean * mean, 0.f);
      const int u = wq + 4 * i;
      if (u < MT && g == 0) {
        float2 o = {mean, rsqrtf(var + p.ln_eps)};
        *(float2*)(stat_lds + (grp * (MT * 16) + u * 16 + c) * 2) = o;
      }
    }
    sa1 = sa2 = sb1 = sb2 = 0.f;
  };
  int pm0 = 0, pn0 = 0;
  bool have_prev = false;
  for (int tv = blockIdx.x; tv < ntiles; tv += G) {
    int m0, n0;
    tile_of(tv, m0, n0);
    const bool last_tile = tv + G >= ntiles;
    // One K step, general form: used for the first SNST - 2 steps of a tile (the previous tile's epilogue and its stores sit in
    // the vmcnt queue: the stages issued before them are 0 .. SNST - 2) and the last SNST (the stream crosses into the next tile,
    // or ends).
    auto step_general = [&](int kt, auto first_c) __attribute__((always_inline)) {
      if (decltype(first_c)::value && have_prev) epilogue(pm0, pn0);
      read_frags();
      prefetch_one();
      if (grp) wait_stage(s + 1, have_prev && kt < SNST - 2);
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      SSB();
      mma_all(std::integral_constant<int, 0>{});
      if (LNF == 1 && kt == nk - 1) ln_publish();
      SSB();
      if (!grp) wait_stage(s + 1, have_prev && kt < SNST - 2);
      if (LNF == 1) __builtin_amdgcn_s_waitcnt(0xC07F);   // the statistics are in LDS before the barrier
      if (LNF == 1 || !(grp && last_tile && kt == nk - 1)) __builtin_amdgcn_s_barrier();
      SSB();
      ++s;
      rslot = rslot + 1 == SNST ? 0 : rslot + 1;
    };
    // Steady state, kt in [SNST - 2, nk - SNST - 1]: the stage being issued (kt + SNST - 1) and the ones awaited next lie inside
    // this tile and nothing but operand DMA is in the queue, so the waits are constants: SNST - 2 stages stay in flight behind the
    // awaited one.  One copy per wave group (no per-step branches).
    auto steps_steady = [&](auto grp_c) __attribute__((always_inline)) {
      constexpr bool GRP1 = decltype(grp_c)::value;
      constexpr int NL = CONV ? 2 : ((MT == 8 || !GRP1) ? (W8 ? 3 : 4) : (W8 ? 2 : 3));
#ifndef WFL_DMA_IN_C
#define WFL_DMA_IN_C 0     // (measured neutral at 1 and 2: tools/gemm_lab.py)
#endif
      constexpr int NDC = CONV ? 0 : (W8 ? (WFL_DMA_IN_C > 1 ? 1 : WFL_DMA_IN_C) : WFL_DMA_IN_C);   // weight pieces issued inside the C slot
      const std::integral_constant<int, NDC> ndc_c{};
      for (int kt = SNST - 2; kt <= nk - SNST - 1; ++kt) {
#ifndef WFL_ABL_NOLDS          // diagnostic builds (tools/gemm_lab.py): no fragment reads / no operand DMA in the steady loop
        read_frags();
#endif
#ifndef WFL_ABL_NOSTAGE
        if (CONV) issue_stage(); else issue_stage_split(ndc_c);
#else
        ++issued; ++pkt; if (CONV) { if (++ptap == ntaps) { ptap = 0; ++pcc; ++pccg; } }
#endif
        // group 1 waits before the barrier: stage kt+2 is out in full, stage kt+3 only with its L-slot pieces so far
        if (GRP1) wait_vm<(SNST - 2) * NL - NDC>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        SSB();
        mma_all(ndc_c);
        SSB();
        if (!GRP1) wait_vm<(SNST - 2) * NL>();
        __builtin_amdgcn_s_barrier();
        SSB();
        ++s;
        rslot = rslot + 1 == SNST ? 0 : rslot + 1;
      }
    };
    step_general(0, std::true_type{});
#pragma unroll
    for (int kt = 1; kt < SNST - 2; ++kt) step_general(kt, std::false_type{});
    if (grp) steps_steady(std::true_type{}); else steps_steady(std::false_type{});
    for (int kt = nk - SNST > SNST - 2 ? nk - SNST : SNST - 2; kt < nk; ++kt) step_general(kt, std::false_type{});
    pm0 = m0; pn0 = n0;
    have_prev = true;
  }
  SSTAMP(2);
  SSTAMP(3);
  if (have_prev) epilogue(pm0, pn0);
  if (LNF == 1 && !grp) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier
  SSTAMP(4);
#undef SSB
}

template <int ACT, int MT, bool RES, int LNF, bool STATS, bool CONV = false, bool W8 = false, bool A8 = false, bool OUT8 = false>
static int launch_stream(const GemmArgs& a, hipStream_t s) {
  constexpr int BMV = MT * 32;
#ifdef WFL_LAB_STB32
  constexpr int lds = SNST * 512 * SBK * 2 + 8 * (MT * 16) * 2 * 4 + 64;
#else
  constexpr int lds = (CONV ? SNST * 256 * SBK * 2 + 2 * 224 * SBK * 2 : SNST * (BMV * SBK * 2 + 256 * SBK * (W8 ? 1 : 2))) + 8 * (MT * 16) * 2 * 4 + 64;
#endif
  const int tiles = ((a.M + BMV - 1) / BMV) * (a.N / 256);
  auto k = gemm_stream_kernel<ACT, MT, RES, LNF, STATS, CONV, W8, A8, OUT8>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  g_wfl_gemm_kernel_id = CONV ? 6 : ((W8 || A8) ? 7 : (MT == 6 ? 1 : 5));     // (7: both fp8 forms)
  hipLaunchKernelGGL(k, dim3(tiles < SNCU ? tiles : SNCU), dim3(512), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Block height.  192 rows everywhere: with tiles dealt round-robin to persistent workgroups the 256-row block only wins when
// it saves a whole round, and hipcc (ROCm 7.2) cannot hold its 128 accumulators + 48 fragment registers + the epilogue
// state in 256 VGPRs without spilling into the K loop (measured: fc1 124 us with the spilling 256-row build, 74 us with
// 192 rows).  WFL_GEMM_BM=256 still selects it for experiments.
template <int ACT, bool RES, int LNF, bool STATS = false>
static int launch_stream_mt(const GemmArgs& a, hipStream_t s) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("WFL_GEMM_BM"); forced = e ? atoi(e) : 0; }
#ifdef WFL_STREAM_MT8
  if (forced == 256) return launch_stream<ACT, 8, RES, LNF, STATS>(a, s);
#endif
  return launch_stream<ACT, 6, RES, LNF, STATS>(a, s);
}

// Dense multi-tap convolutions take the tap-stationary mode (template CONV): taps exactly one input row apart, at most 32
// of them, whole 32-channel chunks, plain epilogue.  (Chosen by shape only -- never by batch size -- because its K order, and so
// its rounding, differs from the other kernels'.)
static bool wfl_gemm_stream_conv(const GemmArgs& a) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("WFL_GEMM_NO_CONV"); off = e && atoi(e) ? 1 : 0; }
  if (off) return false;
  // (at least SNST - 1 taps: a chunk's extended frame tile is issued SNST - 1 stages before its first tap and overwrites the buffer of
  //  the chunk before last, whose last tap must have been read by then)
  const int taps = a.tap_wrap > 0 ? a.tap_wrap : a.K / a.cin;          // (tap_wrap: three segments of that many taps, K = 3 taps cin)
  if (a.tap_wrap > 0 && a.K != 3 * a.tap_wrap * a.cin) return false;
  return a.cin < a.K && a.K % a.cin == 0 && taps <= 32 && taps >= SNST - 1 && a.tap_stride == a.lda && a.cin % SBK == 0 &&
         !a.res && !a.ln_s && !a.stats_out;
}

bool wfl_gemm_stream_conv_takes(const GemmArgs& a);
// The launches this kernel takes (everything else stays with gemm256 / gemm).
bool wfl_gemm_stream_takes(const GemmArgs& a) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("WFL_GEMM_NO_STREAM"); off = e && atoi(e) ? 1 : 0; }
  if (off) return false;
#ifdef WFL_LAB_NOSTREAM
  return false;
#endif
  if (a.glu || a.out_f32 || a.pos || a.clip_bias) return false;
  if (a.a8 >= 2) return false;                     // gemm_mx.hip's operands
  if (a.tap_wrap > 0 && (a.ln_s || a.w8_scale || a.a8)) return false;
  if (a.c_lo && !a.res && !wfl_gemm_stream_conv(a)) return false;      // a low half without a residual: only the conv mode's epilogue writes one
  if (a.w8_scale && (a.cin < a.K || a.K % 64)) return false;
  if (a.a8 && (!a.w8_scale || a.ln_s || a.stats_out || a.stats_in || a.K % 128 || a.lda % 16 || a.K / 64 < 8 || (a.c8 && (a.res || a.ldc8 % 8))))
    return false;
  if (a.c8 && !a.a8) return false;                 // only the residual epilogue here keeps a low half (gemm256 / gemm do it for any)
  if (a.N % 256 || a.K % SBK || a.cin % SBK || a.K / SBK < 8 || a.n_valid % 8) return false;
  // (no lower bound on M: a LayerNorm folded through the producer's statistics must not depend on the batch size -- a clip
  // labelled alone has to equal the same clip inside a batch bit for bit)
  if (a.stats_out && (!a.res || a.N / 256 > 4)) return false;          // 4 tile slots per row of the statistics buffer
  if (a.stats_in && (!a.ln_s || a.stats_nsl <= 0 || a.stats_nsl > 4)) return false;
  if (a.res && a.act != WFL_ACT_NONE) return false;
  if (a.ln_s && (a.res || a.cin < a.K || (a.act != WFL_ACT_NONE && a.act != WFL_ACT_GELU))) return false;
  if (a.act == WFL_ACT_SIGMOID) return false;
  return true;
}


bool wfl_gemm_stream_conv_takes(const GemmArgs& a) { return wfl_gemm_stream_takes(a) && wfl_gemm_stream_conv(a); }

// Returns 1 when this kernel does not take the launch (caller falls back to gemm256 / gemm).
int wfl_launch_gemm_stream(const GemmArgs& a, hipStream_t s) {
  if (!wfl_gemm_stream_takes(a)) return 1;
  static void* trash[32] = {nullptr};        // per device: the scratch line must live where the kernel runs
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 31;
  if (!trash[dev]) {
    if (hipMalloc(&trash[dev], 4096) != hipSuccess) return -2;
  }
  GemmArgs g = a;
  g.trash = trash[dev];
  if (g.a8) {
    // fp8 x fp8 (BASELINE configs[4]): q|k|v (plain), out_proj / fc2 (residual hi + lo), fc1 (GELU, e4m3 out).  The kernel sees both
    // byte matrices as bf16 matrices of half the width (the note at the kernel): K, cin and lda in 2-byte units.
    g.K /= 2; g.cin = g.K; g.lda /= 2;
    if (g.c8) return g.act == WFL_ACT_GELU ? launch_stream<WFL_ACT_GELU, 6, false, 0, false, false, false, true, true>(g, s) : -1;
    if (g.res) return g.act == WFL_ACT_NONE ? launch_stream<WFL_ACT_NONE, 6, true, 0, false, false, false, true>(g, s) : -1;
    if (g.act == WFL_ACT_NONE) return launch_stream<WFL_ACT_NONE, 6, false, 0, false, false, false, true>(g, s);
    return -1;
  }
  if (g.w8_scale) {
    // fp8-weight launches: the Whisper encoder's LayerNorm-folded projections (statistics from the producer) and its residual
    // GEMMs (statistics emitted), i.e. everything between the stem and the final LayerNorm
    if (g.ln_s && g.stats_in) {
      if (g.act == WFL_ACT_NONE) return launch_stream<WFL_ACT_NONE, 6, false, 2, false, false, true>(g, s);
      if (g.act == WFL_ACT_GELU) return launch_stream<WFL_ACT_GELU, 6, false, 2, false, false, true>(g, s);
      return -1;
    }
    if (g.res && !g.ln_s && g.act == WFL_ACT_NONE)
      return g.stats_out ? launch_stream<WFL_ACT_NONE, 6, true, 0, true, false, true>(g, s)
                         : launch_stream<WFL_ACT_NONE, 6, true, 0, false, false, true>(g, s);
    if (!g.res && !g.ln_s) {
      if (g.act == WFL_ACT_NONE) return launch_stream<WFL_ACT_NONE, 6, false, 0, false, false, true>(g, s);
      if (g.act == WFL_ACT_GELU) return launch_stream<WFL_ACT_GELU, 6, false, 0, false, false, true>(g, s);
    }
    return -1;                                         // no other kernel reads e4m3 weights
  }
  if (g.ln_s) {
    if (g.stats_in) {
      switch (g.act) {
        case WFL_ACT_NONE: return launch_stream_mt<WFL_ACT_NONE, false, 2>(g, s);
        case WFL_ACT_GELU: return launch_stream_mt<WFL_ACT_GELU, false, 2>(g, s);
      }
      return 1;
    }
    switch (g.act) {
      case WFL_ACT_NONE: return launch_stream_mt<WFL_ACT_NONE, false, 1>(g, s);
      case WFL_ACT_GELU: return launch_stream_mt<WFL_ACT_GELU, false, 1>(g, s);
    }
    return 1;
  }
  if (wfl_gemm_stream_conv(g)) {
    switch (g.act) {
      case WFL_ACT_NONE: return launch_stream<WFL_ACT_NONE, 6, false, 0, false, true>(g, s);
      case WFL_ACT_GELU: return launch_stream<WFL_ACT_GELU, 6, false, 0, false, true>(g, s);
      case WFL_ACT_RELU: return launch_stream<WFL_ACT_RELU, 6, false, 0, false, true>(g, s);
    }
    return 1;
  }
  if (g.res) return g.stats_out ? launch_stream_mt<WFL_ACT_NONE, true, 0, true>(g, s) : launch_stream_mt<WFL_ACT_NONE, true, 0>(g, s);
  switch (g.act) {
    case WFL_ACT_NONE: return launch_stream_mt<WFL_ACT_NONE, false, 0>(g, s);
    case WFL_ACT_GELU: return launch_stream_mt<WFL_ACT_GELU, false, 0>(g, s);
    case WFL_ACT_RELU: return launch_stream_mt<WFL_ACT_RELU, false, 0>(g, s);
  }
  return 1;
}
