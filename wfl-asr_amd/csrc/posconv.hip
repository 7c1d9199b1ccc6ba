// WavLM's positional convolution on gfx950: x + GELU(grouped Conv1d(d, d, k = 128, padding 64, groups 16)(x)), last output step
// dropped.  Replaces WavLMPositionalConvEmbedding (HF modeling_wavlm.py:48-90) as called from WavLMEncoder(StableLayerNorm).forward
// (:1040-1043), i.e. the first thing /root/reference/model.py:161 runs on the projected features.
//
// A group is a 64-channel-wide convolution (48 valid channels for WavLM-base): per frame 2 x 64 x 8192 FLOPs over a 128-tap window
// of the regrouped rows xg[group][row][64] (wavlm.hip, regroup_kernel).  As one GEMM per group (rounds 1-2: N = 64 on the 128-wide
// tile, K = 8192 re-staged per frame row) it ran at 0.08 of the MFMA roof: the frame operand is a Toeplitz matrix and was fetched
// 128 times.  Here it is tap-stationary, like gemm_stream.hip's conv mode but with the whole window resident:
//   one workgroup (4 waves) = 256 frames x one group's 64 output channels;
//   the 256 + 127 input rows are staged ONCE (two 32-channel halves, 64-byte rows, the any-start-row swizzle xswz of gemm_stream.hip)
//   and re-read at a one-row offset per tap, so a K step (one tap, 32 channels) moves only its 4 KiB weight tile (LDS-DMA, four-stage
//   ring, one piece per wave) for 16 MFMAs per wave: the loop is bound by the matrix pipe and the fragment reads, not by staging;
//   two workgroups share a CU (64 KiB of LDS each), one's staging / epilogue under the other's K loop;
//   the epilogue adds bias, GELU, the residual's hi + lo halves and writes the hi + lo output rows (common.h, GemmArgs::res_lo).
// K order: channel half major, tap minor (the per-group GEMM summed tap major): same products, another fp32 summation order.
#include "common.h"

typedef __attribute__((address_space(1))) const void* pc_gptr_t;
typedef __attribute__((address_space(3))) void* pc_lptr_t;

#define PC_BM 256                 // frames per workgroup
#define PC_MAXTAPS 128
#define PC_EXT_ROWS 384           // PC_BM + PC_MAXTAPS - 1 rounded up to whole 16-row DMA pieces
#define PC_EXT_BYTES (PC_EXT_ROWS * 64)
#define PC_WST_BYTES 4096         // one K step's weight tile: 64 channels x 32 k
#define PC_NST 4                  // weight ring depth
#define PC_LDS (2 * PC_EXT_BYTES + PC_NST * PC_WST_BYTES)

// (struct PosConvArgs: common.h -- one definition for the kernel and for model.hip)

static __device__ __forceinline__ int pc_sswz(int row) { return (-(row >> 2)) & 3; }              // aligned 16-row reads (weights)
static __device__ __forceinline__ int pc_xswz(int row) { return ((row >> 2) & 1) << 1; }          // 16-row reads from ANY start row

__global__ __launch_bounds__(256, 2) void posconv_kernel(PosConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ext = smem;                                   // [2 halves][PC_EXT_ROWS][64 bytes]
  char* wst = smem + 2 * PC_EXT_BYTES;                // [PC_NST][64 channels][64 bytes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int gi = blockIdx.y;
  const long m0 = (long)blockIdx.x * PC_BM;
  const int taps = p.taps, ns = 2 * taps;             // K steps: [half][tap]
  const bf16_t* xg = p.xg + (long)gi * p.R * 64;
  const bf16_t* wg = p.w[gi];

  // ---- the window: rows lead + m0 - taps/2 .. + PC_BM + taps - 2 of the group, both channel halves (24 pieces of 16 rows each)
  const long row_first = p.lead + m0 - taps / 2;
  for (int q = wid; q < 2 * (PC_EXT_ROWS / 16); q += 4) {
    const int h = q / (PC_EXT_ROWS / 16), piece = q - h * (PC_EXT_ROWS / 16);
    const int r = piece * 16 + (lane >> 2);
    long src = row_first + r;
    src = src < 0 ? 0 : (src > p.R - 1 ? p.R - 1 : src);                       // (rows beyond the window's last tap are never multiplied)
    const bf16_t* sp = xg + src * 64 + h * 32 + ((lane & 3) ^ pc_xswz(r)) * 8;
    __builtin_amdgcn_global_load_lds((pc_gptr_t)sp, (pc_lptr_t)(ext + h * PC_EXT_BYTES + piece * 1024), 16, 0, 0);
  }
  // ---- weight stream: K step s = (half h = s / taps, tap j = s % taps): rows 16 wid .. + 15 of the 64-channel tile are this wave's piece
  const int wrow = 16 * wid + (lane >> 2);
  const bf16_t* wsrc = wg + (long)wrow * p.ldw + ((lane & 3) ^ pc_sswz(wrow)) * 8;
  auto issue_w = [&](int s) __attribute__((always_inline)) {
    const int h = s >= taps ? 1 : 0, j = s - h * taps;
    __builtin_amdgcn_global_load_lds((pc_gptr_t)(wsrc + j * 64 + h * 32), (pc_lptr_t)(wst + (s & (PC_NST - 1)) * PC_WST_BYTES + wid * 1024), 16, 0, 0);
  };
  for (int s = 0; s < PC_NST - 1; ++s) issue_w(s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  f32x4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int w_off[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = 16 * v + c;
    w_off[v] = r * 64 + ((g ^ pc_sswz(r)) << 4);
  }
  const int wm = wid * 64;

  for (int s = 0; s < ns; ++s) {
    // stage s has landed (this wave's piece; the barrier makes it everybody's) and stage s - 1 has been read by every wave
    if (s + PC_NST - 2 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PC_NST - 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + PC_NST - 1 < ns) issue_w(s + PC_NST - 1);
    const int h = s >= taps ? 1 : 0, j = s - h * taps;
    const char* ws = wst + (s & (PC_NST - 1)) * PC_WST_BYTES;
    const char* xs = ext + h * PC_EXT_BYTES;
    bf16x8 fw[4], fx[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) fw[v] = *(const bf16x8*)(ws + w_off[v]);
    const int r0 = wm + c + j;                         // + 16 u never changes the swizzle
    const int xb = r0 * 64 + ((g ^ pc_xswz(r0)) << 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) fx[u] = *(const bf16x8*)(xs + xb + u * 1024);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[v], fx[u], acc[u][v], 0, 0, 0);
  }

  // ---- epilogue: lane (g, c) holds channels 16 v + 4 g .. + 3 of frame m0 + wm + 16 u + c
  const float* bias = p.bias[gi];
  f32x4 bj[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int n0 = 16 * v + 4 * g;
    bj[v] = n0 + 4 <= p.cpg ? *(const f32x4*)(bias + n0) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const long nrows = (long)p.B * p.P;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long m = m0 + wm + 16 * u + c;
    const int t = (int)(m % p.P);
    if (m >= nrows || t >= p.T) continue;
    if (p.clip_T && t >= p.clip_T[m / p.P]) continue;
    const long row = p.lead + m;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int n0 = 16 * v + 4 * g;
      if (n0 + 4 > p.cpg) continue;
      const long off = row * p.ld + (long)gi * p.cpg + n0;
      const bf16x4 rh = *(const bf16x4*)(p.res + off);
      bf16x4 rl = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
      if (p.res_lo) rl = *(const bf16x4*)(p.res_lo + off);
      bf16x4 oh, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = (bf2f(rh[e]) + bf2f(rl[e])) + gelu_erf(acc[u][v][e] + bj[v][e]);
        oh[e] = f2bf(y);
        ol[e] = f2bf(y - bf2f(oh[e]));
      }
      *(bf16x4*)(p.out + off) = oh;
      if (p.out_lo) *(bf16x4*)(p.out_lo + off) = ol;
    }
  }
}

// Returns 1 when the shape is not this kernel's (caller falls back to one GEMM per group).
int wfl_launch_posconv(const PosConvArgs& a, hipStream_t s) {
  if (a.groups <= 0 || a.groups > 16 || a.cpg <= 0 || a.cpg > 64 || a.cpg % 8 || a.taps < 4 || a.taps > PC_MAXTAPS || a.taps % 2 ||
      a.ldw < (long)a.taps * 64 || a.ld % 4 || (a.cpg * a.groups) % 4)
    return 1;
  if (a.lead < a.taps / 2) return 1;                   // the first window starts inside the buffer's leading halo
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)posconv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS) != hipSuccess) return -2;
  }
  const long nrows = (long)a.B * a.P;
  dim3 grid((unsigned)((nrows + PC_BM - 1) / PC_BM), (unsigned)a.groups);
  hipLaunchKernelGGL(posconv_kernel, grid, dim3(256), PC_LDS, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
