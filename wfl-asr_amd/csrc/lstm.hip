// Bidirectional LSTM recurrence for gfx950 (the only op of the path that is sequential in time).
// Replaces nn.LSTM at /root/reference/model.py:104-111, 182-183: gates i,f,g,o, hidden H = d/2 per direction,
// zero initial state, both biases, outputs concat(fwd, bwd); padded frames do not exist here (every clip in a
// batch has the same T).
//
// The input projection x W_ih^T + b_ih + b_hh of BOTH directions is one big MFMA GEMM (gemm.hip) into fp32 `gx`,
// with the gate rows reordered unit-major / gate-minor so that one float4 holds (i,f,g,o) of one hidden unit.
// This kernel only runs the recurrence  gates_t = gx_t + W_hh h_{t-1}  as a persistent launch:
//
//   grid = (G slices of the hidden units) x (groups of 16 clips) x (2 directions), 256 threads, one WG per CU.
//   WG (slice, group, dir) keeps its 4U x H slice of W_hh in LDS for all T steps (rows ordered unit-major /
//   gate-minor: an MFMA 16x16x32 tile = 4 units x 4 gates, so every lane ends up with the four gates of ONE unit
//   of ONE clip and the cell update is lane-local; the cell state lives in registers for all T steps).
//   Per step the G slice WGs of a (group, dir) exchange their U new hidden values through a ping-pong buffer in
//   global memory with the write-through hand-off of cdna_hip_programming.md Guideline 16 (R1, counter form):
//   wave 0 stores the slice as whole 1 KiB chunks with sc1 stores, drains them (vmcnt(0)), then ONE lane adds 1 to a
//   monotonic agent-scope counter; consumers poll that counter with sc1 loads (one lane, s_sleep, bounded),
//   pass a workgroup barrier and read h_{t-1} with sc1 16-byte loads straight into MFMA B-operand fragments.
//   Ping-pong is WAR-safe: nobody can publish step s+2 before every WG consumed step s (it needs all of s+1).
//   A spin that gives up sets a sticky error word (checked by the host) instead of hanging the GPU.
#include "common.h"
#include <cstdlib>

struct LstmArgs {
  const float* gx; long ldgx;      // frame rows, fp32: col = dir*4H + 4*unit + gate
  const bf16_t* whh;               // [2][G][4U][H] bf16, slice rows = 4*u_local + gate
  bf16_t* out; long ldo;           // frame rows: col = dir*H + unit
  long lead;
  int B, T, P, H, U, G;
  bf16_t* hx;                      // exchange: [2 dir][groups][2 parity][G][16][U]
  unsigned* counters;              // [2 dir][groups], zeroed by the launcher
  unsigned* error;                 // sticky time-out word
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

static __device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
static __device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <int H>
static __device__ __forceinline__ int w_swz(int row) {
  constexpr int CPR = H / 8;
  if (CPR >= 16) return row & 15;
  if (CPR == 8) return (row >> 1) & 7;
  return (row >> 2) & 3;
}

template <int H, int MAXT>   // MAXT = MFMA tiles (4 units each) per wave
__global__ __launch_bounds__(256) void lstm_kernel(LstmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = H / 32;
  constexpr int CPR = H / 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int slice = blockIdx.x, grp = blockIdx.y, dir = blockIdx.z;
  const int U = p.U, G = p.G;
  const int ntile = U / 4;                       // tiles of this WG
  char* Ws = smem;                               // [4U][H] bf16, swizzled 16-byte chunks
  bf16_t* hst = (bf16_t*)(smem + 4 * U * H * 2);  // [16][U] staging of the new hidden values
  int* dflag = (int*)(smem + 4 * U * H * 2 + 16 * U * 2);   // a spin gave up (block-wide)
  if (tid == 0) *dflag = 0;

  // ---- W_hh slice -> LDS (once)
  {
    const bf16_t* wsrc = p.whh + ((long)(dir * G + slice) * 4 * U) * H;
    const int nchunk = 4 * U * CPR;
    for (int i = tid; i < nchunk; i += 256) {
      const int r = i / CPR, cc = i % CPR;
      *(bf16x8*)(Ws + r * (H * 2) + ((cc ^ w_swz<H>(r)) << 4)) = *(const bf16x8*)(wsrc + (long)r * H + cc * 8);
    }
  }
  __syncthreads();

  const int clip = grp * 16 + c;
  const int clip_rd = clip < p.B ? clip : p.B - 1;
  const long ngroups = gridDim.y;
  bf16_t* hx = p.hx + ((long)(dir * ngroups + grp) * 2) * G * 16 * U;       // [parity][G][16][U]
  unsigned* counter = p.counters + dir * ngroups + grp;
  const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * G * 16 * U * 2, 0x00020000);

  float cstate[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) cstate[i] = 0.f;
  bool dead = false;                             // a spin gave up: stop waiting, finish the launch

  // gx of step 0
  f32x4 gxv[MAXT];
  auto load_gx = [&](int t) {
    const float* gp = p.gx + (p.lead + (long)clip_rd * p.P + t) * p.ldgx + dir * 4 * H;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int tile = wid + 4 * i;
      if (tile < ntile) gxv[i] = *(const f32x4*)(gp + 4 * (slice * U + tile * 4 + g));
    }
  };
  load_gx(dir == 0 ? 0 : p.T - 1);

  for (int s = 0; s < p.T; ++s) {
    const int t = dir == 0 ? s : p.T - 1 - s;
    f32x4 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) acc[i] = gxv[i];
    if (s + 1 < p.T) load_gx(dir == 0 ? s + 1 : p.T - 2 - s);     // prefetch: independent of the recurrence

    if (s > 0) {
      // ---- wait until all G slices of step s-1 are published
      if (tid == 0 && !dead) {
        const unsigned target = (unsigned)G * (unsigned)s;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 1000000u) { atomicOr(p.error, 1u); *dflag = 1; break; }
        }
      }
      __syncthreads();
      if (*dflag) dead = true;
      // ---- gates += W_slice . h_{s-1}   (h read with sc1 loads straight into B fragments)
      const int par = (s - 1) & 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + 8 * g;
        const int sl = k / U, within = k - sl * U;
        const unsigned off = (unsigned)((((par * G + sl) * 16 + c) * U + within) * 2);
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off, 0, 16);
        bf16x8 hf;
        __builtin_memcpy(&hf, &raw, 16);
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
          const int tile = wid + 4 * i;
          if (tile < ntile) {
            const int r = tile * 16 + c;
            const bf16x8 wf = *(const bf16x8*)(Ws + r * (H * 2) + (((ks * 4 + g) ^ w_swz<H>(r)) << 4));
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf, acc[i], 0, 0, 0);
          }
        }
      }
    }

    // ---- cell update: acc[i] = (i, f, g, o) pre-activations of unit slice*U + tile*4 + g, clip c
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int tile = wid + 4 * i;
      if (tile < ntile) {
        const float ig = sigm(acc[i][0]), fg = sigm(acc[i][1]), gg = tanh_(acc[i][2]), og = sigm(acc[i][3]);
        cstate[i] = fg * cstate[i] + ig * gg;
        const float h = og * tanh_(cstate[i]);
        const bf16_t hb = f2bf(h);
        const int ul = tile * 4 + g;
        hst[c * U + ul] = hb;
      }
    }
    __syncthreads();
    // ---- publish: wave 0 writes the [16][U] slice as whole 1 KiB chunks, write-through, then signals once
    if (wid == 0 && s + 1 < p.T) {
      const int par = s & 1;
      const int nbytes = 16 * U * 2;
      const unsigned base = (unsigned)(((par * G + slice) * 16) * U * 2);
      for (int o = lane * 16; o < nbytes; o += 1024) {
        u32x4 v;
        __builtin_memcpy(&v, (const char*)hst + o, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, hx_rsrc, base + o, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // waves 1-3: the same tile to the layer's output frame rows, 16 bytes per lane
    if (wid > 0) {
      const int cpr = U / 8;                   // 16-byte chunks per clip
      for (int ch = tid - 64; ch < 16 * cpr; ch += 192) {
        const int cl = ch / cpr, part = ch - cl * cpr;
        const int clipw = grp * 16 + cl;
        if (clipw < p.B)
          *(bf16x8*)(p.out + (p.lead + (long)clipw * p.P + t) * p.ldo + dir * H + slice * U + part * 8) =
              *(const bf16x8*)(hst + cl * U + part * 8);
      }
    }
    // the next step's barrier (after the poll) also protects hst against the next cell update
  }
}

template <int H, int MAXT>
static int launch_lstm_t(const LstmArgs& a, int groups, hipStream_t s) {
  const int lds = 4 * a.U * H * 2 + 16 * a.U * 2 + 16;
  auto k = lstm_kernel<H, MAXT>;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(k, dim3(a.G, groups, 2), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Units per WG: the largest multiple of 8 dividing H whose 4U x H bf16 slice fits ~128 KiB of LDS (<= 64 units)
int wfl_lstm_units_per_wg(int H) {
  if (const char* e = getenv("WFL_LSTM_UNITS")) {        // test hook: force a slice width (more WGs per direction)
    const int U = atoi(e);
    if (U >= 8 && U % 8 == 0 && U <= H && H % U == 0 && 4L * U * H * 2 + 16 * U * 2 <= 132 * 1024) return U;
  }
  int best = 0;
  for (int U = 8; U <= 64 && U <= H; U += 8)
    if (H % U == 0 && 4L * U * H * 2 + 16 * U * 2 <= 132 * 1024) best = U;
  return best;
}

long wfl_lstm_exchange_bytes(int H, int B) {
  const long groups = (B + 15) / 16;
  return 2 * groups * 2 * 16 * (long)H * 2 + 2 * groups * 4 + 64;     // hx + counters + error word
}

int wfl_launch_lstm(LstmArgs a, void* exchange, hipStream_t s) {
  if (a.U <= 0 || a.U % 8 || a.H % a.U || a.H % 32 || a.ldgx % 4 || a.ldo % 8 || a.T <= 0 || a.B <= 0) return -1;
  a.G = a.H / a.U;
  const int groups = (a.B + 15) / 16;
  if ((long)a.G * groups * 2 > 256) return -5;         // every WG must be resident (one per CU): split the batch
  const long hx_bytes = 2L * groups * 2 * 16 * a.H * 2;
  a.hx = (bf16_t*)exchange;
  a.counters = (unsigned*)((char*)exchange + hx_bytes);
  if (!a.error) return -1;                 // the forward's error word (ORed into, never cleared here)
  if (wfl_launch_fill_i32((int*)a.counters, 2 * groups, 0, s)) return -3;   // (a kernel, not a memset node: common.h)
  const int maxt = (a.U / 4 + 3) / 4;
  switch (a.H) {
    case 32: return maxt <= 2 ? launch_lstm_t<32, 2>(a, groups, s) : -4;
    case 256: return launch_lstm_t<256, 4>(a, groups, s);
    case 384: return launch_lstm_t<384, 4>(a, groups, s);
    case 512: return launch_lstm_t<512, 4>(a, groups, s);
    case 640: return launch_lstm_t<640, 4>(a, groups, s);
  }
  return -4;
}
