// Bidirectional LSTM recurrence for gfx950 (the only op of the path that is sequential in time).
// Replaces nn.LSTM at /root/reference/model.py:104-111, 182-183: gates i,f,g,o, hidden H = d/2 per direction,
// zero initial state, both biases, outputs concat(fwd, bwd); padded frames do not exist here (every clip in a
// batch has the same T).
//
// The input projection x W_ih^T + b_ih + b_hh of BOTH directions is one big MFMA GEMM (gemm256.hip) into fp32 `gx`,
// with the gate rows reordered unit-major / gate-minor so that one float4 holds (i,f,g,o) of one hidden unit.
// This kernel only runs the recurrence  gates_t = gx_t + W_hh h_{t-1}  as a persistent launch:
//
//   grid = (G slices of the hidden units) x (groups of 16 clips) x (2 directions), 320 threads = 4 compute waves + 1 loader wave,
//   one WG per CU.
//   * W_hh lives in REGISTERS for all T steps: WG (slice, group, dir) owns the 4U x H slice of U = 32 units, each wave
//     the MFMA A-operand fragments of its two 16-row tiles (4 units x 4 gates each): 2 x H/32 bf16x8 per lane
//     (64 VGPRs at H = 256, 160 at H = 640).  Round 1 kept a 64-unit slice in LDS and re-read 128 KiB of it per step
//     (0.22 us of LDS time on the critical path, and 133 KiB of LDS per WG).
//   * A wave's tiles come in pairs holding the even / odd units of 8 consecutive units, so lane (g, c) ends up with
//     the four gates of units 8p + 2g and 8p + 2g + 1 of clip c: the cell update is lane-local (cell state in
//     registers for all T steps) and the two new hidden values are one 4-byte run of the output row.
//   * Per step the G slice WGs of a (group, dir) exchange h through global memory as self-validating 8-byte GRANULES
//     {2 x bf16 h, 32-bit tag = step + 1} (MI355X_MICROARCH.md, hand-off price list: granule = one naturally aligned 8-byte
//     {data, tag} written by ONE sc1 store; observed untorn): the producer's lanes store their granule straight from
//     registers (write-through, no LDS staging, no drain, no counter), consumers poll the granules themselves with sc1
//     16-byte loads (two granules each) until every tag shows the step they wait for, strip the tags into a
//     fragment-ordered LDS image (one ds_write_b128 per K step and lane), pass ONE workgroup barrier and read their MFMA B
//     fragments back.  Round 1's hand-off (1 KiB sc1 stores, vmcnt(0) drain, agent-scope counter add, one polling lane,
//     barrier, sc1 fragment loads by every wave) cost ~5.8 us per step; this one 0.98 us (H = 256).  A wave's granule stores
//     cover whole 128-byte lines (image layout [block of 8 units][clip][4 granules]: one store instruction = 512 contiguous
//     bytes); with 32-byte pieces of four different lines per instruction the same step took 2.44 us.
//     Ping-pong (two granule images, two LDS images) is WAR-safe: nobody can publish step s+2 before every WG consumed
//     step s (it needs all of s+1).  Tags of an earlier launch are cleared by a fill kernel in front of every launch.
//   * The launch is a 1-D grid laid out so that, under the observed round-robin workgroup placement, all slices of a (group,
//     direction) team land on ONE XCD.  The team verifies that with a roll call at kernel start (each slice publishes its XCC id);
//     when it holds, the per-step granules are stored with the default cache policy -- they stay in the XCD's L2 and the consumers'
//     sc1 loads hit them there -- instead of written through to memory and fetched back: poll 0.60 -> 0.32 us, step 1.76 -> 1.15 us
//     (H = 256, before the loader wave).  When it does not hold the team keeps the write-through stores, which are correct under any placement.
//   * gx (the input projection's pre-activations, 8 KiB per workgroup and step, from HBM) is streamed by a FIFTH wave: LDS-DMA into
//     an eight-slot LDS ring, six steps ahead; the compute waves start a step from one ds_read_b128 per tile.  Loaded by the compute
//     waves themselves (a register ring refilled behind each poll) those HBM loads sat in front of the NEXT step's poll in the
//     wave's in-order vmcnt queue: +0.17 us per step at 16 clips, +0.43 at 64.  Now 0.98 us per step (16 clips), 1.24 (64).
//   A poll that gives up ORs bit 0 into the forward's error word (wfl_asr.h: status) instead of hanging the GPU, and the
//   launch finishes without further waiting.  Launches are cut so that one launch never needs more than 128 resident
//   workgroups: two such launches (two batches in flight on two streams) always fit the 256 CUs together.
#include "common.h"
#include <mutex>
#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

static __device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
static __device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

#ifdef WFL_LSTM_STAMPS
#define LSTAMP(k) do { if (p.stamps && tl == 0 && slice == 0 && tid == 0 && s >= 64 && s < 96) \
    p.stamps[(s - 64) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif
#define LSTM_SC1 16            // buffer cache policy: sc1 (agent scope: served / written at the memory side of L2)
#define LSTM_SPIN_LIMIT 3000000u
#ifndef LSTM_STAGGER
#define LSTM_STAGGER 4           // s_sleep units (64 clocks) between two poll attempts
#endif

typedef __attribute__((address_space(1))) const void* lstm_gptr_t;
typedef __attribute__((address_space(3))) void* lstm_lptr_t;
#define LSTM_DL 6              // steps of gx the loader wave keeps in flight
#define LSTM_NR 8              // slots of the gx ring in LDS (a power of two, >= LSTM_DL + 2)
#define LSTM_THREADS 320       // four compute waves + the loader wave
#ifndef LSTM_GX_AUX
#define LSTM_GX_AUX 0          // cache policy of the gx stream (2 = nt)
#endif

// RAG: clips of different lengths (LstmArgs::clip_T).  Its own instantiation, because a per-lane frame index turns the scalar row
// arithmetic of the loader wave and of the output store into 64-bit vector arithmetic on the step's critical path: with clips of one
// length (every Whisper batch) that cost 3.85 against 3.25 ms per 16-clip forward (A/B on one box, default head).
// SPLIT ("model.precision: high", LstmArgs::whh_lo / out_lo): W_hh and h as bf16 pairs hi + lo -- the low halves of W_hh in registers
// beside the high ones, h_t published as two granules (a second image behind the first), gates += W_hi h_hi + W_lo h_hi + W_hi h_lo.
// SELFGX (round 4): no loader wave -- 256 threads, ONE wave per SIMD, so a wave may hold up to 512 registers: the split-precision recurrence
// of hidden sizes above 256 (Whisper-small's 384, WavLM-large's 512, Whisper-large's 640), whose W_hh slice needs H / 2 registers per wave
// for its two halves.  Every compute wave streams the gx of its own tiles itself (LDS-DMA into its own part of the ring, LSTM_DL steps
// ahead, issued right BEHIND a step's poll so that only the next step's poll has to retire it): the +0.2 .. 0.4 us per step the loader
// wave was introduced to remove is the price of the exact mode here.  No barrier is involved: a wave reads back only what it fetched.
template <int H, int MAXT, bool RAG, bool SPLIT = false, bool SELFGX = false>     // MAXT = MFMA tiles per wave (even: tiles come in even / odd unit pairs)
__global__ __launch_bounds__(SELFGX ? 256 : LSTM_THREADS) void lstm_kernel(LstmArgs p) {
  static_assert(MAXT % 2 == 0, "tiles are paired");
  static_assert(!SPLIT || H <= 256 || SELFGX, "split precision: both halves of the W_hh slice stay in registers (above 256: one wave per SIMD)");
  constexpr int KS = H / 32;                     // K steps of 32 hidden units
  constexpr int NP = MAXT / 2;                   // tile pairs per wave
  __shared__ bf16x8 hfrag[2][KS][64];            // h_{t-1} as MFMA B fragments: [parity][K step][lane]
  __shared__ bf16x8 hfrag_lo[SPLIT ? 2 : 1][SPLIT ? KS : 1][64];
  // gx ring: [slot][compute wave][tile][lane] float4 = the (i, f, g, o) pre-activations a lane starts a step from, brought in by the
  // loader wave's LDS-DMA (dynamic shared memory: LSTM_NR x 4 x MAXT KiB)
  extern __shared__ __attribute__((aligned(16))) char lstm_dyn[];
  f32x4* gxl = (f32x4*)lstm_dyn;
  __shared__ int dflag;
  __shared__ int xcd_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  // 1-D grid laid out for the observed round-robin placement (blocks b and b + 8 share an XCD): all G slices of a (group, direction)
  // team get linear ids congruent mod 8.  Placement is NOT a contract, so the team checks it (roll call below) before relying on it.
  const int U = p.U, G = p.G;
#ifdef WFL_LSTM_STAMPS
  const int tl = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / G) - p.team_shift;
  if (tl < 0) return;
#else
  const int tl = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / G);       // team of this launch
#endif
  const int slice = (blockIdx.x >> 3) % G;
  if (tl >= 2 * p.ngroups_launch) return;
  const int grp = p.grp0 + (tl >> 1), dir = tl & 1;
  const int npair = U >> 3;
  if (tid == 0) dflag = 0;

  // ---- W_hh fragments -> registers (once).  MFMA A row a = lane & 15 of tile (pair pp, parity e) is gate a & 3 of unit
  // 8 pp + 2 (a >> 2) + e; the lane supplies k = 32 ks + 8 g .. + 8.
  bf16x8 w[MAXT][KS], w_lo[SPLIT ? MAXT : 1][SPLIT ? KS : 1];
#pragma unroll
  for (int i = 0; i < MAXT && wid < 4; ++i) {
    // (a wave whose pair index is beyond the slice -- only with the narrow slices of the test hook -- runs the same code on
    // the last valid pair's data and stores nothing: no per-tile control flow inside the step)
    const int pp = wid + 4 * (i >> 1) < npair ? wid + 4 * (i >> 1) : npair - 1, e = i & 1;
    const int ul = 8 * pp + 2 * (c >> 2) + e;
    const bf16_t* wsrc = p.whh + (((long)(dir * G + slice) * 4 * U) + 4 * ul + (c & 3)) * H + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w[i][ks] = *(const bf16x8*)(wsrc + ks * 32);
      // Opaque to the compiler from here on: an invariant global load is "rematerialisable", and hipcc then re-loads all
      // 32 KiB of a wave's fragments from L2 in every step (measured: 2.1 us per step) instead of keeping them in registers.
      asm volatile("" : "+v"(w[i][ks]));
      if (SPLIT) {
        w_lo[i][ks] = *(const bf16x8*)(p.whh_lo + (wsrc - p.whh) + ks * 32);
        asm volatile("" : "+v"(w_lo[i][ks]));
      }
    }
  }

  const int clip = grp * 16 + c;
  const int clip_rd = clip < p.B ? clip : p.B - 1;
  const long gran_per_img = 16L * (H / 2);                                    // granules per parity image
  unsigned long long* hx = p.hx + ((long)(dir * p.ngroups + grp) * 4) * gran_per_img;     // [hi, lo][2 parity] images of this team
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, (int)(4 * gran_per_img * 8), 0x00020000);
  constexpr unsigned LO_IMG = (unsigned)(2 * 16L * (H / 2) * 8);                            // byte offset of the low halves' images

  float cstate[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) cstate[i] = 0.f;
  bool dead = false;                             // a poll gave up: stop waiting, finish the launch

  // ---- roll call: every slice publishes the XCD it runs on (granule {xcc + 1, tag 1}, sc1 = always visible); when the whole team
  // sits on ONE XCD its members share that XCD's L2, and the per-step granules can be stored with the default policy (the line
  // stays in L2, the consumers' sc1 loads -- which bypass only L1 -- hit it there) instead of written through to memory and fetched
  // back from there by every consumer: the hand-off's round trip drops from the fabric's to L2's.  Otherwise (or if the roll call
  // times out) the team keeps the write-through stores, which are correct under any placement.
  bool same_xcd = false;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    unsigned long long* roll = p.roll + ((long)(dir * p.ngroups + grp)) * 64;        // up to 64 slices per team
    const __amdgpu_buffer_rsrc_t roll_rsrc = __builtin_amdgcn_make_buffer_rsrc(roll, 0, 64 * 8, 0x00020000);
    if (tid == 0) {
      const u32x2 gr = {xcc + 1u, 1u};
      __builtin_amdgcn_raw_buffer_store_b64(gr, roll_rsrc, (unsigned)(slice * 8), 0, LSTM_SC1);
    }
    if (wid == 0) {
      unsigned spins = 0;
      bool all_here = false, all_same = false;
      for (;;) {
        const bool mine = lane < G;
        u32x2 v = {0u, 0u};
        if (mine) v = __builtin_amdgcn_raw_buffer_load_b64(roll_rsrc, (unsigned)(lane * 8), 0, LSTM_SC1);
        all_here = __all(!mine || v[1] == 1u);
        all_same = __all(!mine || v[0] == xcc + 1u);
        if (all_here) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > LSTM_SPIN_LIMIT) {
          if (lane == 0) { atomicOr(p.error, 1u); dflag = 1; }
          break;
        }
      }
      if (lane == 0) xcd_flag = (all_here && all_same) ? 1 : 0;
    }
    __syncthreads();
    same_xcd = xcd_flag != 0;
#ifdef WFL_LSTM_NO_XCD
    same_xcd = false;
#endif
    if (dflag) dead = true;
  }

  // ---- the input projection's pre-activations gx: wave 4 streams them from HBM into the LDS ring with LDS-DMA, LSTM_DL steps ahead.
  // They used to be loaded by the compute waves themselves (a register ring refilled after each poll), but a wave's vector-memory
  // operations retire in issue order: the NEXT step's granule poll then sat behind HBM loads that take longer than a step (1-2 us
  // against 0.7), +0.17 us per step at 16 clips and +0.43 at 64 (tools/micro/lstm_bench.hip, -DWFL_LSTM_NO_GX).  The loader wave has
  // its own vmcnt; it joins the compute waves' one barrier per step, in front of which it has seen step s + 1's data land.
  const float* gx_lane = p.gx + (p.lead + (long)clip_rd * p.P) * p.ldgx + dir * 4 * H + 4 * (slice * U + 2 * g);
  const int Tc = RAG ? p.clip_T[clip_rd] : p.T;          // frames of this lane's clip (LstmArgs::clip_T)
  auto issue_own_gx = [&](int s) __attribute__((always_inline)) {       // SELFGX: this wave's own tiles of step s
    int t = dir == 0 ? s : Tc - 1 - s;
    if (RAG) t = s < Tc ? t : 0;
    const float* gp = gx_lane + (long)t * p.ldgx;
    char* dst = lstm_dyn + (long)(s & (LSTM_NR - 1)) * (4 * MAXT * 1024);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int pp = wid + 4 * (i >> 1) < npair ? wid + 4 * (i >> 1) : npair - 1;
      __builtin_amdgcn_global_load_lds((lstm_gptr_t)(gp + 4 * (8 * pp + (i & 1))), (lstm_lptr_t)(dst + (wid * MAXT + i) * 1024), 16, 0, LSTM_GX_AUX);
    }
  };
  if (SELFGX) {
    for (int s = 0; s < LSTM_DL && s < p.T; ++s) issue_own_gx(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (!SELFGX && wid == 4) {
    auto issue = [&](int s) __attribute__((always_inline)) {
      int t = dir == 0 ? s : Tc - 1 - s;             // (this lane's clip: its backward direction starts at its own last frame)
      if (RAG) t = s < Tc ? t : 0;                   // steps beyond the clip's length: any valid row, the result is never stored
      const float* gp = gx_lane + (long)t * p.ldgx;
      char* dst = lstm_dyn + (long)(s & (LSTM_NR - 1)) * (4 * MAXT * 1024);
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          const int pp = w4 + 4 * (i >> 1) < npair ? w4 + 4 * (i >> 1) : npair - 1;
#ifndef WFL_LSTM_NO_GX
          __builtin_amdgcn_global_load_lds((lstm_gptr_t)(gp + 4 * (8 * pp + (i & 1))), (lstm_lptr_t)(dst + (w4 * MAXT + i) * 1024), 16, 0, LSTM_GX_AUX);
#endif
        }
    };
    for (int s = 0; s < LSTM_DL && s < p.T; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // (the compute waves' barrier in front of step 0)
    for (int s = 1; s < p.T; ++s) {
      // in front of barrier s: steps <= s + 1 have landed; steps s + 2 .. s + LSTM_DL - 1 may still be in flight
      if (s + LSTM_DL - 1 < p.T) {
        issue(s + LSTM_DL - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LSTM_DL - 2) * 4 * MAXT) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    return;
  }
  __syncthreads();

  auto step = [&](int s) __attribute__((always_inline)) {
    const int t = dir == 0 ? s : Tc - 1 - s;               // (per lane: the clip's own frame, LstmArgs::clip_T)
    f32x4 acc[MAXT];
    bf16x8 hf[KS], hfl[SPLIT ? KS : 1];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc[i] = gxl[(((s & (LSTM_NR - 1)) * 4 + wid) * MAXT + i) * 64 + lane];
    LSTAMP(0);

    if (s > 0) {
      // ---- gather h_{s-1}: this wave's K steps (ks = wid, wid + 4, ...), two 16-byte loads = four granules per K step
      const int par = (s - 1) & 1;
      const unsigned want = (unsigned)s;            // tag of step s-1
      constexpr int NK = (KS + 3) / 4;
      if (!dead) {
        // Two poll attempts in flight, issued half a round trip apart (a failed attempt costs a whole ~0.6 us round trip; the
        // loads return in issue order, so attempt n + 1 is already on its way when attempt n is examined).
        u32x4 ga0[NK], gb0[NK], ga1[NK], gb1[NK];
        u32x4 la0[SPLIT ? NK : 1], lb0[SPLIT ? NK : 1];      // SPLIT: the low halves' granules (the single-attempt poll only)
        auto issue = [&](u32x4* ga, u32x4* gb) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < NK; ++j) {
            const int ks = wid + 4 * j;
            if (ks < KS) {
              const unsigned off = (unsigned)((((par * (H / 8) + 4 * ks + g) * 16) + c) * 32);
              ga[j] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off, 0, LSTM_SC1);
              gb[j] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off + 16, 0, LSTM_SC1);
              if (SPLIT) {
                la0[j] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off + LO_IMG, 0, LSTM_SC1);
                lb0[j] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off + LO_IMG + 16, 0, LSTM_SC1);
              }
            }
          }
        };
        auto landed = [&](const u32x4* ga, const u32x4* gb) __attribute__((always_inline)) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < NK; ++j) {
            const int ks = wid + 4 * j;
            if (ks < KS) ok = ok && ga[j][1] == want && ga[j][3] == want && gb[j][1] == want && gb[j][3] == want;
            if (SPLIT && ks < KS) ok = ok && la0[j][1] == want && la0[j][3] == want && lb0[j][1] == want && lb0[j][3] == want;
          }
#ifdef WFL_LSTM_NOWAIT
          return true;
#endif
          return (bool)__all(ok);
        };
        auto stage_lds = [&](const u32x4* ga, const u32x4* gb) __attribute__((always_inline)) {   // strip the tags
#pragma unroll
          for (int j = 0; j < NK; ++j) {
            const int ks = wid + 4 * j;
            if (ks < KS) {
              u32x4 d = {ga[j][0], ga[j][2], gb[j][0], gb[j][2]};
              bf16x8 hv;
              __builtin_memcpy(&hv, &d, 16);
              hfrag[par][ks][lane] = hv;
              if (SPLIT) {
                u32x4 dl = {la0[j][0], la0[j][2], lb0[j][0], lb0[j][2]};
                bf16x8 hl;
                __builtin_memcpy(&hl, &dl, 16);
                hfrag_lo[par][ks][lane] = hl;
              }
            }
          }
        };
        unsigned spins = 0;
#ifdef WFL_LSTM_POLL2
        issue(ga0, gb0);
        for (;;) {
          __builtin_amdgcn_s_sleep(LSTM_STAGGER);
          issue(ga1, gb1);
          if (landed(ga0, gb0)) { LSTAMP(1); stage_lds(ga0, gb0); break; }
          __builtin_amdgcn_s_sleep(LSTM_STAGGER);
          issue(ga0, gb0);
          if (landed(ga1, gb1)) { LSTAMP(1); stage_lds(ga1, gb1); break; }
          if ((spins += 2) > LSTM_SPIN_LIMIT) {
            if (lane == 0) { atomicOr(p.error, 1u); dflag = 1; }
            break;
          }
        }
#else
        for (;;) {
          issue(ga0, gb0);
          if (landed(ga0, gb0)) { LSTAMP(1); stage_lds(ga0, gb0); break; }
          __builtin_amdgcn_s_sleep(1);
          if (++spins > LSTM_SPIN_LIMIT) {
            if (lane == 0) { atomicOr(p.error, 1u); dflag = 1; }
            break;
          }
        }
#endif
      }
      // SELFGX: the gx of step s + LSTM_DL, behind this step's poll (whose wait retired everything older, the DMA of steps <= s + LSTM_DL - 1
      // included) and in front of the next one's
      if (SELFGX && s + LSTM_DL < p.T) issue_own_gx(s + LSTM_DL);
      __syncthreads();
      LSTAMP(2);
      if (dflag) dead = true;
      // ---- gates += W_slice . h_{s-1}: all K-step fragments first, then one tile pair at a time, so that the cell update of a
      // pair (vector pipe) can run beside the next pair's MFMAs (matrix pipe)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) hf[ks] = hfrag[par][ks][lane];
      if (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hfl[ks] = hfrag_lo[par][ks][lane];
      }
    } else {
      if (SELFGX && LSTM_DL < p.T) issue_own_gx(LSTM_DL);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) hf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};      // h_{-1} = 0
      if (SPLIT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hfl[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
#ifdef WFL_LSTM_KMAJOR
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < MAXT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][ks], hf[ks], acc[i], 0, 0, 0);
#endif
    LSTAMP(3);
    // ---- cell update: acc[i] = (i, f, g, o) pre-activations of unit 8 pp + 2 g + (i & 1), clip c
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
      const int pp = wid + 4 * ip;
#ifndef WFL_LSTM_KMAJOR
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc[2 * ip + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2 * ip + e][ks], hf[ks], acc[2 * ip + e], 0, 0, 0);
          if (SPLIT) {
            acc[2 * ip + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[2 * ip + e][ks], hf[ks], acc[2 * ip + e], 0, 0, 0);
            acc[2 * ip + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2 * ip + e][ks], hfl[ks], acc[2 * ip + e], 0, 0, 0);
          }
        }
#endif
      bf16_t hb[2], hbl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = 2 * ip + e;
        const float ig = sigm(acc[i][0]), fg = sigm(acc[i][1]), gg = tanh_(acc[i][2]), og = sigm(acc[i][3]);
        cstate[i] = fg * cstate[i] + ig * gg;
        const float hv = og * tanh_(cstate[i]);
        hb[e] = f2bf(hv);
        hbl[e] = f2bf(hv - bf2f(hb[e]));
      }
      unsigned bits, bits_lo = 0;
      __builtin_memcpy(&bits, hb, 4);
      if (SPLIT) __builtin_memcpy(&bits_lo, hbl, 4);
      const int u0 = slice * U + 8 * pp + 2 * g;                   // first of this lane's two consecutive units
      if (pp < npair) {
        if (s + 1 < p.T) {                                         // publish: one granule, write-through
          const unsigned off = (unsigned)((((((s & 1) * (H / 8) + (u0 >> 3)) * 16) + c) * 4 + g) * 8);
          const u32x2 gr = {bits, (unsigned)(s + 1)};
          if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(gr, hx_rsrc, off, 0, 0);        // stays in the team's L2
          else __builtin_amdgcn_raw_buffer_store_b64(gr, hx_rsrc, off, 0, LSTM_SC1);          // write-through: any placement
          if (SPLIT) {
            const u32x2 gl = {bits_lo, (unsigned)(s + 1)};
            if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(gl, hx_rsrc, off + LO_IMG, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(gl, hx_rsrc, off + LO_IMG, 0, LSTM_SC1);
          }
        }
#ifndef WFL_LSTM_NO_OUT       // (diagnostic builds: tools/micro/lstm_bench.hip)
        if (clip < p.B && (!RAG || s < Tc)) {
          *(unsigned*)(p.out + (p.lead + (long)clip * p.P + t) * p.ldo + dir * H + u0) = bits;
          if (SPLIT) *(unsigned*)(p.out_lo + (p.lead + (long)clip * p.P + t) * p.ldo + dir * H + u0) = bits_lo;
        }
#endif
      }
    }
    LSTAMP(4);
  };

  for (int s = 0; s < p.T; ++s) step(s);
}

template <int H, int MAXT, bool RAG, bool SPLIT = false, bool SELFGX = false>
static int launch_lstm_r(const LstmArgs& a, int groups, hipStream_t s) {
  const int teams = 2 * groups;
  constexpr int lds = LSTM_NR * 4 * MAXT * 1024;       // the gx ring (dynamic; the fragment images are static shared memory)
  constexpr int NTHREADS = SELFGX ? 256 : LSTM_THREADS;
  auto k = lstm_kernel<H, MAXT, RAG, SPLIT, SELFGX>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
#ifdef WFL_LSTM_STAMPS
  hipLaunchKernelGGL(k, dim3(8 * a.G * ((teams + a.team_shift + 7) / 8)), dim3(NTHREADS), lds, s, a);
#else
  hipLaunchKernelGGL(k, dim3(8 * a.G * ((teams + 7) / 8)), dim3(NTHREADS), lds, s, a);
#endif
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int H, int MAXT>
static int launch_lstm_t(const LstmArgs& a, int groups, hipStream_t s) {
  if (a.whh_lo && a.out_lo) {
    if constexpr (H <= 256)
      return a.clip_T ? launch_lstm_r<H, MAXT, true, true>(a, groups, s) : launch_lstm_r<H, MAXT, false, true>(a, groups, s);
    else       // both halves of the slice no longer fit 256 registers: four waves, one per SIMD, no loader wave
      return a.clip_T ? launch_lstm_r<H, MAXT, true, true, true>(a, groups, s) : launch_lstm_r<H, MAXT, false, true, true>(a, groups, s);
  }
  return a.clip_T ? launch_lstm_r<H, MAXT, true>(a, groups, s) : launch_lstm_r<H, MAXT, false>(a, groups, s);
}

bool wfl_lstm_split_precision_supported(int H) { return H <= 640; }

// Units per WG: 32 (four waves x one pair of 16-row tiles) when it divides H, else the largest multiple of 8 below that does.
// Measured at H = 256, 16 clips (tools/micro/lstm_bench.hip): 64 units per WG (4 WGs per direction) 2.14 us per step, 32 units
// (8 WGs) 1.76 us, 16 units 1.77 us -- the step is hand-off latency plus one WG's own MFMA + cell update, so halving the slice
// pays until the gather's fan-in takes it back.
int wfl_lstm_units_per_wg(int H) {
  if (const char* e = getenv("WFL_LSTM_UNITS")) {        // test hook: force a slice width (more WGs per direction)
    const int U = atoi(e);
    if (U >= 8 && U % 8 == 0 && U <= 32 && U <= H && H % U == 0) return U;
  }
  int best = 0;
  for (int U = 8; U <= 32 && U <= H; U += 8)
    if (H % U == 0) best = U;
  return best;
}

long wfl_lstm_exchange_bytes(int H, int B) {
  const long groups = (B + 15) / 16;
  // granule images [2 dir][groups][2 halves][2 parity][H/8][16][4] x 8 bytes (the low halves' images: precision high only), then the
  // roll-call granules [2 dir][groups][64] x 8 bytes
  return 2 * groups * 4 * 16 * (long)(H / 2) * 8 + 2 * groups * 64 * 8 + 64;
}

int wfl_launch_lstm(LstmArgs a, void* exchange, hipStream_t s) {
  if (a.U <= 0 || a.U % 8 || a.U > 32 || a.H % a.U || a.H % 32 || a.ldgx % 4 || a.ldo % 8 || a.T <= 0 || a.B <= 0) return -1;
  if (!a.error) return -1;                 // the forward's error word (ORed into, never cleared here)
  a.G = a.H / a.U;
  const int groups = (a.B + 15) / 16;
  if (2 * a.G > 128) return -5;
  a.hx = (unsigned long long*)exchange;
  a.roll = a.hx + 2L * groups * 4 * 16 * (a.H / 2);
  a.ngroups = groups;
  // tags of an earlier launch must not validate: clear the granule images and the roll call (a kernel, not a memset node: common.h)
  if (wfl_launch_fill_i32((int*)exchange, 2L * groups * 4 * 16 * (a.H / 2) * 2 + 2L * groups * 64 * 2, 0, s)) return -3;
  // Residency contract.  A team (the G workgroups of one direction of one group of 16 clips) makes progress only while ALL its
  // workgroups are resident -- one per CU, ~133 KiB of LDS each -- and launches are plain, so nothing checks that.  Two rules keep a
  // team from ever waiting on a CU that another waiting team holds:
  //   (1) a launch has at most 128 workgroups (whole teams), half the chip;
  //   (2) at most TWO recurrence launches of one device are in flight, whatever the number of batches in flight: launch n waits for the
  //       event recorded behind launch n - 2 (any stream, any model) before it starts.  Two launches are at most 256 workgroups = the
  //       chip's CUs, so both become fully resident as soon as the other streams' GEMM / attention / element-wise workgroups -- which
  //       finish without waiting for anybody -- have retired (at worst one persistent GEMM launch later); a third could leave all three
  //       partially resident, each holding CUs the others wait for.
  // Round 2 relied on (1) alone ("two launches always fit the chip"), which three or more batches in flight -- the default for this
  // head -- broke: partially resident teams of several launches held each other's CUs until other streams' kernels drained
  // (the cfg3 sweep that fell from 15.2 k to 5.0 k audio-s/s at four in flight).  Overlap is not lost: a batch's
  // recurrence still runs under the other batches' GEMMs and under ONE other recurrence.
  // (Not while the stream is being captured into a graph: an event of another stream would pull that stream into the capture.  Graph
  //  replay keeps at most two slots in flight, for which (1) is enough.)
  static std::mutex chain_mu;
  static hipEvent_t chain_ev[32][2] = {{nullptr, nullptr}};
  static unsigned chain_n[32] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 31;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  const bool chained = cap == hipStreamCaptureStatusNone;
  std::unique_lock<std::mutex> chain_lock(chain_mu, std::defer_lock);
  if (chained) {
    chain_lock.lock();                     // (wait -> launches -> record must not interleave with another host thread's)
    hipEvent_t& ev = chain_ev[dev][chain_n[dev] & 1];
    if (!ev) {
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return -2;
    } else if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return -3;
  }
  const int per = 128 / (2 * a.G) > 0 ? 128 / (2 * a.G) : 1;
  for (int g0 = 0; g0 < groups; g0 += per) {
    a.grp0 = g0;
    const int n = groups - g0 < per ? groups - g0 : per;
    a.ngroups_launch = n;
    int r;
    switch (a.H) {
      case 32: r = launch_lstm_t<32, 2>(a, n, s); break;
      case 64: r = launch_lstm_t<64, 2>(a, n, s); break;       // (encoder_type none: 80 mel bins in 128 columns)
      case 128: r = launch_lstm_t<128, 2>(a, n, s); break;
      case 192: r = launch_lstm_t<192, 2>(a, n, s); break;     // (Whisper-tiny)
      case 256: r = launch_lstm_t<256, 2>(a, n, s); break;
      case 384: r = launch_lstm_t<384, 2>(a, n, s); break;
      case 512: r = launch_lstm_t<512, 2>(a, n, s); break;
      case 640: r = launch_lstm_t<640, 2>(a, n, s); break;
      default: return -4;
    }
    if (r) return r;
  }
  if (chained) {
    if (hipEventRecord(chain_ev[dev][chain_n[dev] & 1], s) != hipSuccess) return -3;
    ++chain_n[dev];
  }
  return 0;
}
