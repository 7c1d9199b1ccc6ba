// fp8 x fp8 GEMM on the block-scaled MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands) -- the encoder GEMMs of an
// fp8-weight model (BASELINE configs[4]: q|k|v, out_proj, fc1, fc2 of every Whisper layer; replaces HF modeling_whisper.py:309-354,
// 391-407 as called from /root/reference/model.py:155-156).  Round 4.  Same GemmArgs contract, persistent tile walk, XCD-aware tile
// order, two-group ping-pong K loop and register epilogue as gemm_stream.hip; what differs is the operand staging and the MFMA.
//
// The instruction, as measured on the device (tools/micro/mx_probe.hip, profiles/round4_mx_probe.txt): 32.6 cycles per MFMA with two
// waves per SIMD against 12.9 for the non-scaled v_mfma_f32_16x16x32_fp8_fp8 -- 1.58 x its FLOPs per cycle.  A lane (c, g) of a 16-row
// tile supplies 32 bytes of row c: bytes 0..15 are k's 16 g .. 16 g + 15 of the instruction's FIRST 64 k's, bytes 16..31 the same k's
// of its SECOND 64 (two K = 64 halves, each laid out like the non-scaled instruction); operands contract lane for lane, byte for byte;
// the scale byte of lane (c, b) multiplies row c's k's 32 b .. 32 b + 31, i.e. lanes g = 0, 1 scale the first half's two blocks and
// lanes g = 2, 3 the second half's.  Two forms:
//
//   SINGLE (GemmArgs::a8 == 2): A is one e4m3 plane, a stage holds 128 k's (128-byte rows; a lane takes chunks 2 g, 2 g + 1 of both
//          operands -- with every scale byte 2^0 any consistent assignment of bytes to k's gives the same sum).  160-row tiles: frame
//          tile 20 KiB + weight tile 32 KiB per stage, three stages.  The per-row / per-channel fp32 scales multiply the accumulator in
//          the epilogue.  Three mantissa bits on the activations: the opt-in `model.activation_dtype: fp8`.
//   PAIR   (GemmArgs::a8 == 3): A is TWO e4m3 planes, hi = e4m3(x / s) and lo = e4m3(16 (x / s - hi)): eight significant bits, what a
//          bf16 operand carries.  A stage holds 64 k's: LDS row r of the frame tile is [hi 64 B | lo 64 B]; a lane takes chunk g of the hi
//          half as its first 16 bytes and chunk g of the lo half as its second 16, the weight fragment is the same 16 bytes twice, and the
//          frame operand's scale bytes are 2^0 in lanes g = 0, 1 and 2^-4 in lanes g = 2, 3.  ONE MFMA therefore adds (hi + lo / 16) . w
//          over 64 k's: the second pass costs no second weight fragment, no second accumulator and no extra instruction.  192-row tiles,
//          40 KiB per stage, three stages.  (tests/study_fp8.py: pairs sit 0.06 / 0.009 from the reference on the fp8 checkpoint, bf16
//          activations 0.08 / 0.012, single e4m3 activations 1.1 / 0.19.)
//
// LDS images are filled by LDS-DMA (1 KiB per wave instruction, lane-linear), so the swizzle lives in the SOURCE address.  128-byte
// rows keep 16-byte chunk ch of row r at chunk ch ^ f(r): SINGLE f(r) = bit 1 of r | bit 3 of r << 2 (conflict-free for the reads of
// chunks 2 g + j both from the frame tile's rows c and from the weight tile's permuted rows), PAIR f(r) = r & 6 (chunks g and 4 + g) --
// found by exhaustive search over the XOR-linear functions of the row bits against the four 16-lane groups a ds_read_b128 is serviced
// in; the PAIR form's 64-byte weight rows use gemm_stream.hip's swizzle.
// K loop: the two wave groups (rows 0..BMV/2, BMV/2..BMV) run one barrier apart -- while one issues its MFMAs (C slot) the other reads
// its fragments and issues the DMA of stage s + 2 (L slot); a tile's epilogue runs in the first L slot of the next tile, under the other
// group's MFMAs.
#include "common.h"
#include <cstdlib>
#include <cstring>

typedef __attribute__((address_space(1))) const void* mgptr_t;
typedef __attribute__((address_space(3))) void* mlptr_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define MXNCU 256
#ifndef MX_PRE_ISSUE
#define MX_PRE_ISSUE 1     // at a tile boundary the next stage is issued in front of the epilogue's stores
#endif
#ifndef MX_FULL_LINE_ST
#define MX_FULL_LINE_ST 1  // bf16 outputs: rows c and c + 8 trade halves so that a store instruction writes whole 128-byte lines
#endif
#ifndef MX_STAGGER_CYC
#define MX_STAGGER_CYC 0   // s_memtime ticks (100 MHz on gfx950: 10 ns each)
#endif
#ifndef MX_EPI_G1_EARLY
#define MX_EPI_G1_EARLY 1  // group 1's epilogue right behind its tile's last MFMAs (kernels without a residual): profiles/round4_mx_lab_epilogue_overlap.txt
#endif
#ifndef MX_DMA_IN_C
#define MX_DMA_IN_C 1     // the operand DMA of stage s + 2 is issued from the C slot, among the MFMAs (0: from the L slot, among the ds_reads: 11-17 % slower, profiles/round4_mx_lab_dma_in_c.txt)
#endif

// Diagnostic build (tools/mx_lab.py, -DWFL_GEMM_STAMPS): per workgroup and wave group, cycles (s_memtime) spent in the prologue, the
// epilogues, the DMA waits, the barriers and the MFMA issue, written to GemmArgs::stamps[block][group][8] by one lane of each group.
#if defined(WFL_GEMM_STAMPS) && !defined(MX_STAMPS_LIGHT)
#define MXT() ((long long)__builtin_amdgcn_s_memtime())
#define MXACC(var, t0) do { var += MXT() - (t0); } while (0)
#else
#define MXT() 0LL
#define MXACC(var, t0) do { } while (0)
#endif

static __device__ __forceinline__ void mglds(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((mgptr_t)g, (mlptr_t)l, 16, 0, 0);
}
static __device__ __forceinline__ int mxswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }   // SINGLE, 128-byte rows
static __device__ __forceinline__ int mpswz(int row) { return row & 6; }                                       // PAIR frame rows
static __device__ __forceinline__ int mwswz(int row) { return (-(row >> 2)) & 3; }                             // PAIR weight rows (64 B)
template <int N>
static __device__ __forceinline__ void mx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// OUT: 0 bf16 (+ low half with RES), 1 one e4m3 plane at the fixed scale c8_inv_scale, 2 an e4m3 hi + lo pair at that scale
template <typename T>
__device__ __forceinline__ void mx_store_nt(T* ptr, const T& val) {     // streaming store (global_store ... nt)
  if constexpr (sizeof(T) == 16) {
    typedef int v4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(v4, val), (v4*)ptr);
  } else {
    typedef int v2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(__builtin_bit_cast(v2, val), (v2*)ptr);
  }
}

template <int ACT, int MT, bool RES, bool PAIR, int OUT>
__global__ __launch_bounds__(512) void gemm_mx_kernel(GemmArgs p) {
  static_assert(OUT == 0 || !RES, "e4m3 output: no residual");
  constexpr int BMV = MT * 32;                      // frame rows per tile
  constexpr int KS = PAIR ? 64 : 128;               // k's (bytes of one plane) per stage
  constexpr int ABYTES = BMV * 128;
  constexpr int WROW = PAIR ? 64 : 128;
  constexpr int STB = ABYTES + 256 * WROW;
  constexpr int R = 3;                              // ring depth: two stages in flight ahead of the one being computed
  constexpr int NPA = BMV / 8;                      // frame-tile DMA pieces (8 rows x 128 B each), dealt to the 8 waves:
  constexpr int NA = (NPA + 7) / 8;                 //   waves < NAX take NA, the others NA - 1
  constexpr int NAX = NPA % 8 == 0 ? 8 : NPA % 8;
  constexpr int NW = PAIR ? 2 : 4;                  // weight-tile pieces per wave
  constexpr int NSTORE = (RES ? 4 : (OUT == 2 ? 4 : 2)) * MT;
  static_assert(R * STB <= 160 * 1024, "LDS");
  static_assert(2 * (NA + NW) + NSTORE < 64, "vmcnt is six bits wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3;
  const int g = lane >> 4, c = lane & 15;
  const int G = gridDim.x;

  const int tiles_n = p.N / 256;
  const int tiles_m = (p.M + BMV - 1) / BMV;
  const int ntiles = tiles_m * tiles_n;
  const int nk = p.K / KS;
  auto tile_of = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {   // XCD-aware order: v and v + 8 share an XCD; contiguous run per XCD
    const int q = ntiles >> 3, r = ntiles & 7, x = v & 7, i = v >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const int tm = bid / tiles_n;
    m0 = tm * BMV;
    n0 = (bid - tm * tiles_n) * 256;
  };
  const int wm = grp * (MT * 16), wn = wq * 64;

  // ---- operand stream
  const bool a_full = wid < NAX;                     // this wave issues NA frame pieces (else NA - 1)
  const int ap0 = a_full ? wid * NA : NAX * NA + (wid - NAX) * (NA - 1);      // its first piece
  const char* a_src[NA];
  const char* w_src[NW];
  int pv = blockIdx.x, pkt = 0, issued = 0, islot = 0;
  auto set_src = [&](int v) __attribute__((always_inline)) {
    int m0, n0;
    tile_of(v, m0, n0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (ap0 + i) * 8 + (lane >> 3);
      int am = m0 + row;
      am = am < p.M ? am : p.M - 1;
      if (PAIR) {
        const int lc = (lane & 7) ^ mpswz(row);      // chunks 0..3: the hi plane's 64 k's, 4..7: the lo plane's
        a_src[i] = (lc < 4 ? (const char*)p.A : (const char*)p.a8_lo) + (long)am * p.lda + (lc & 3) * 16;
      } else {
        a_src[i] = (const char*)p.A + (long)am * p.lda + (((lane & 7) ^ mxswz(row)) << 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      if (PAIR) {
        const int row = (wid * NW + i) * 16 + (lane >> 2);
        w_src[i] = (const char*)p.W + (long)(n0 + row) * p.K + (((lane & 3) ^ mwswz(row)) << 4);
      } else {
        const int row = (wid * NW + i) * 8 + (lane >> 3);
        w_src[i] = (const char*)p.W + (long)(n0 + row) * p.K + (((lane & 7) ^ mxswz(row)) << 4);
      }
    }
  };
  set_src(pv);
  auto prefetch_one = [&]() __attribute__((always_inline)) {   // the DMA of the stream's next stage, if there is one
    if (pv >= ntiles) return;
    char* base = smem + islot * STB;
    islot = islot + 1 == R ? 0 : islot + 1;
    const long koff = (long)pkt * KS;
#ifndef MX_ABL_NODMA            // (diagnostic builds, tools/mx_lab.py: no operand DMA / no fragment reads / no MFMAs -- wrong results, same launch)
#pragma unroll
    for (int i = 0; i < NA - 1; ++i) mglds(a_src[i] + koff, base + (ap0 + i) * 1024);
    if (a_full) mglds(a_src[NA - 1] + koff, base + (ap0 + NA - 1) * 1024);
#pragma unroll
    for (int i = 0; i < NW; ++i) mglds(w_src[i] + koff, base + ABYTES + (wid * NW + i) * 1024);
#else
    (void)base; (void)koff;
#endif
    ++issued;
    if (++pkt == nk) {
      pkt = 0;
      pv += G;
      if (pv < ntiles) set_src(pv);
    }
  };
  // wait until this wave's pieces of global stage `need` have landed; `stores`: an epilogue's NSTORE stores were issued after them
  auto wait_stage = [&](int need, bool stores) __attribute__((always_inline)) {
    const int younger = issued - need - 1;             // stages issued after it: 0 .. R - 2
    if (younger < 0) return;
#define MX_WAITY(L)                                                                             \
    do {                                                                                        \
      if (younger >= 1) { if (stores) mx_wait_vm<(L) + NSTORE>(); else mx_wait_vm<(L)>(); }     \
      else { if (stores) mx_wait_vm<NSTORE>(); else mx_wait_vm<0>(); }                          \
    } while (0)
#ifndef MX_ABL_NODMA
    if (a_full) MX_WAITY(NA + NW); else MX_WAITY(NA - 1 + NW);
#endif
#undef MX_WAITY
  };

  // ---- fragment addressing (weight rows permuted as in gemm_stream.hip: lane (g, c) ends up with channels 8g .. 8g+7 and 32+8g .. of frame c)
  int w_off[4][PAIR ? 1 : 2], x_off[2];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = wn + 32 * (v >> 1) + 8 * (c >> 2) + 4 * (v & 1) + (c & 3);
    if (PAIR) w_off[v][0] = ABYTES + r * 64 + ((g ^ mwswz(r)) << 4);
    else {
#pragma unroll
      for (int j = 0; j < 2; ++j) w_off[v][PAIR ? 0 : j] = ABYTES + r * 128 + (((2 * g + j) ^ mxswz(r)) << 4);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)                           // + u * 2048 (the swizzles do not see multiples of 16 rows)
    x_off[j] = (wm + c) * 128 + (PAIR ? (((g + 4 * j) ^ mpswz(wm + c)) << 4) : (((2 * g + j) ^ mxswz(wm + c)) << 4));
  const int scale_w = 0x7f7f7f7f;                                   // 2^0
  const int scale_x = (PAIR && g >= 2) ? 0x7b7b7b7b : 0x7f7f7f7f;   // lanes g = 2, 3 scale the second 64 k's: the lo plane's, 2^-4

  f32x4 acc[MT][4];
#pragma unroll
  for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- epilogue of tile (m0, n0): registers -> HBM (the register epilogue of gemm_stream.hip: 16-byte stores, never waited for,
  //      masked stores go to a scratch line so that every wave issues exactly NSTORE of them)
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    const int nb = n0 + wn + 8 * g;
    const bool has_bias = p.bias != nullptr;
    int mrow0 = m0 + wm + c;                         // opaque: computed here, not hoisted out of the K loop as MT loop-invariant registers
    asm volatile("" : "+v"(mrow0));             //   (those spill, and a spill's reload waits on vmcnt behind the operand DMA)
    f32x4 bj[4], cj[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bj[2 * h + q] = has_bias ? *(const f32x4*)(p.bias + nb + 32 * h + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        cj[2 * h + q] = *(const f32x4*)(p.w8_scale + nb + 32 * h + 4 * q);
      }
    float sa[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      int m = mrow0 + 16 * u;
      m = m < p.M ? m : p.M - 1;
      sa[u] = p.a8_scale ? p.a8_scale[p.a8_lead + m] : p.a8_static;
    }
    const float invP = 1.0f / (float)p.P;
    int orow[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      const int m = mrow0 + 16 * u;
      int b = (int)((float)m * invP);
      int t = m - b * p.P;
      if (t < 0) { t += p.P; --b; }
      if (t >= p.P) { t -= p.P; ++b; }
      orow[u] = (m < p.M && t < p.T) ? (int)p.c_lead + b * p.c_pitch + t : -1;
    }
#ifdef MX_ABL_FULL_LINES             // diagnostic build (wrong results, same bytes): a store / load instruction covers 8 rows x 128 B instead of 16 rows x 64 B
#define OROW(u, h) (orow[u] + 8 * ((h) - (c >> 3)))
#define NBH(h) (n0 + wn + 8 * g + 32 * (c >> 3))
#else
#define OROW(u, h) orow[u]
#define NBH(h) (nb + 32 * (h))
#endif
    constexpr int RING = 2;                          // (three tiles of residual halves in flight spill: a reload waits on vmcnt behind the DMA)
    bf16x8 rr[RES ? RING : 1][2], rl[RES ? RING : 1][2];
    const bf16_t* res_lo = p.res_lo ? p.res_lo : p.res;
    const float lo_scale = p.res_lo ? 1.f : 0.f;
    auto load_res = [&](int u, int slot) __attribute__((always_inline)) {
#ifdef MX_ABL_EPI_NORES              // diagnostic build: the residual is not read (wrong results)
      rr[slot][0] = rr[slot][1] = rl[slot][0] = rl[slot][1] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      return;
#endif
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long r = OROW(u, h) >= 0 ? OROW(u, h) : p.c_lead;
        rr[slot][h] = *(const bf16x8*)(p.res + r * p.ldres + NBH(h));
        rl[slot][h] = *(const bf16x8*)(res_lo + r * p.ldres + NBH(h));
      }
    };
    if (RES) {
#pragma unroll
      for (int u = 0; u < RING && u < MT; ++u) load_res(u, u);
    }
    char* trash = (char*)p.trash + lane * 16;
    // A plain bf16 output (q | k | v: 370 MB at large-v3 size, read once by the attention launch) is stored with the streaming hint: 3-9 % per
    // launch in tools/mx_lab.py.  Not the e4m3 outputs (8-byte stores: 3-9 % slower with it) and not the residual stream (re-read next).
#ifndef MX_ST_NT
#define MX_ST_NT (OUT == 0 && !RES)
#endif
#define MX_ST(ptr, val) do { if constexpr (MX_ST_NT) mx_store_nt((ptr), (val)); else *(ptr) = (val); } while (0)
#ifdef MX_ABL_EPI_NOSTORE            // diagnostic build: every store goes to the scratch line
#define MX_KEEP(k) false
#else
#define MX_KEEP(k) (k)
#endif
    float amax8 = 0.f;                               // e4m3 output: largest stored |x| * scale (above 448 it did not fit)
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 oo[2], ool[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (OUT == 2) __builtin_amdgcn_sched_barrier(0);      // (the pair output's two runs one after the other: together they spill)
        float x[8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[u][2 * h + q][e] * cj[2 * h + q][e] * sa[u];
            v = apply_act<ACT>(v + bj[2 * h + q][e]);
            if (RES) v = (bf2f(rr[u % RING][h][4 * q + e]) + lo_scale * bf2f(rl[u % RING][h][4 * q + e])) + p.alpha * v;
            x[4 * q + e] = v;
          }
        const bool keep = orow[u] >= 0 && nb + 32 * h < p.n_valid;
        if (OUT != 0) {
          const float keepf = keep ? 1.f : 0.f;
          int wv[2] = {0, 0}, lv[2] = {0, 0};        // e4m3 bytes of the run of 8 channels: hi plane, lo plane
#pragma unroll
          for (int e = 0; e < 8; e += 2) {           // two values at a time: a packed hi word half, then what its rounding left behind
            typedef __attribute__((ext_vector_type(2))) float f32x2_;
            const float t0 = x[e] * p.c8_inv_scale, t1 = x[e + 1] * p.c8_inv_scale;
            amax8 = fmaxf(amax8, fmaxf(fabsf(t0), fabsf(t1)) * keepf);
            const float y0 = fminf(fmaxf(t0, -448.f), 448.f), y1 = fminf(fmaxf(t1, -448.f), 448.f);
            const bool up = (e & 2) != 0;
            wv[e >> 2] = up ? __builtin_amdgcn_cvt_pk_fp8_f32(y0, y1, wv[e >> 2], true) : __builtin_amdgcn_cvt_pk_fp8_f32(y0, y1, wv[e >> 2], false);
            if (OUT == 2) {
              const f32x2_ hv = up ? __builtin_amdgcn_cvt_pk_f32_fp8(wv[e >> 2], true) : __builtin_amdgcn_cvt_pk_f32_fp8(wv[e >> 2], false);
              const float z0 = 16.f * (y0 - hv[0]), z1 = 16.f * (y1 - hv[1]);
              lv[e >> 2] = up ? __builtin_amdgcn_cvt_pk_fp8_f32(z0, z1, lv[e >> 2], true) : __builtin_amdgcn_cvt_pk_fp8_f32(z0, z1, lv[e >> 2], false);
            }
          }
          char* d8 = (char*)(p.c8 + (long)OROW(u, h) * p.ldc8 + NBH(h));
          d8 = MX_KEEP(keep) ? d8 : trash;
          MX_ST((uint2*)d8, make_uint2((unsigned)wv[0], (unsigned)wv[1]));
          if (OUT == 2) {
            char* dl = (char*)(p.c8_lo + (long)OROW(u, h) * p.ldc8 + NBH(h));
            dl = MX_KEEP(keep) ? dl : trash;
            MX_ST((uint2*)dl, make_uint2((unsigned)lv[0], (unsigned)lv[1]));
          }
          continue;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(x[e]);
        if (MX_FULL_LINE_ST) {                       // both halves first, stored below
          oo[h] = o;
          if (RES) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ool[h][e] = f2bf(x[e] - bf2f(o[e]));
          }
          continue;
        }
        char* dst = (char*)((bf16_t*)p.C + (long)OROW(u, h) * p.ldc + NBH(h));
        dst = MX_KEEP(keep) ? dst : trash;
        MX_ST((bf16x8*)dst, o);
        if (RES) {
          bf16x8 ol;
#pragma unroll
          for (int e = 0; e < 8; ++e) ol[e] = f2bf(x[e] - bf2f(o[e]));
          char* dl = (char*)(p.c_lo + (long)OROW(u, h) * p.ldc + NBH(h));
          dl = MX_KEEP(keep && p.c_lo) ? dl : trash;
          MX_ST((bf16x8*)dl, ol);
        }
      }
      if (OUT == 0 && MX_FULL_LINE_ST) {
        // A lane holds channels 8 g .. 8 g + 7 and 32 + 8 g .. of frame row c: stored as they are, an instruction writes 16 rows x 64 bytes --
        // half lines.  Rows c and c + 8 trade one half each (DPP row_ror:8 inside the 16 lanes of a g), after which the lanes c < 8 hold the
        // first 64 bytes of rows c and c + 8 and the lanes c >= 8 the second: every store instruction writes 8 rows x 128 bytes, whole
        // lines (3-10 % per launch with the addresses permuted alone: profiles/round4_mx_lab_epilogue_memory.txt).
        const bool lowhalf = c < 8;
        auto trade = [&](const bf16x8& h0, const bf16x8& h1, bf16x8& a, bf16x8& b) __attribute__((always_inline)) {
          const i32x4 s0 = __builtin_bit_cast(i32x4, h0), s1 = __builtin_bit_cast(i32x4, h1);
          i32x4 ra, rb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int send = lowhalf ? s1[j] : s0[j];
            const int recv = __builtin_amdgcn_update_dpp(0, send, 0x128, 0xf, 0xf, false);   // row_ror:8: lane c <-> lane c ^ 8
            ra[j] = lowhalf ? s0[j] : recv;
            rb[j] = lowhalf ? recv : s1[j];
          }
          a = __builtin_bit_cast(bf16x8, ra);
          b = __builtin_bit_cast(bf16x8, rb);
        };
        const int po = __builtin_amdgcn_update_dpp(0, orow[u], 0x128, 0xf, 0xf, false);
        const int rowa = lowhalf ? orow[u] : po, rowb = lowhalf ? po : orow[u];      // rows (c & 7) and (c & 7) + 8 of the block
        const int nbx = nb + (lowhalf ? 0 : 32);
        const bool colok = nbx < p.n_valid;
        bf16x8 a, b;
        trade(oo[0], oo[1], a, b);
        char* da = (char*)((bf16_t*)p.C + (long)rowa * p.ldc + nbx);
        char* db = (char*)((bf16_t*)p.C + (long)rowb * p.ldc + nbx);
        da = MX_KEEP(rowa >= 0 && colok) ? da : trash;
        db = MX_KEEP(rowb >= 0 && colok) ? db : trash;
        MX_ST((bf16x8*)da, a);
        MX_ST((bf16x8*)db, b);
        if (RES) {
          trade(ool[0], ool[1], a, b);
          char* la = (char*)(p.c_lo + (long)rowa * p.ldc + nbx);
          char* lb = (char*)(p.c_lo + (long)rowb * p.ldc + nbx);
          la = MX_KEEP(rowa >= 0 && colok && p.c_lo) ? la : trash;
          lb = MX_KEEP(rowb >= 0 && colok && p.c_lo) ? lb : trash;
          MX_ST((bf16x8*)la, a);
          MX_ST((bf16x8*)lb, b);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (RES && u + RING < MT) load_res(u + RING, u % RING);
    }
    if (OUT != 0 && p.err) {
      if (__builtin_amdgcn_ballot_w64(amax8 > 448.f) && lane == 0) atomicOr(p.err, 2u);
    }
  };

  // ---- prologue: two stages in flight, stage 0 landed, group 1 one barrier behind
  long long st_ep = 0, st_wait = 0, st_bar = 0, st_mma = 0, st_l = 0;
#ifdef WFL_GEMM_STAMPS
  const long long st_t0 = (long long)__builtin_amdgcn_s_memtime();
#else
  const long long st_t0 = 0;
#endif
#pragma unroll
  for (int t = 0; t < R - 1; ++t) prefetch_one();
#if MX_STAGGER_CYC > 0
  if (RES && ntiles > 2 * G) {
    // Residual launches of several rounds: every workgroup alternates a K loop (no HBM traffic to speak of: the operands come from L2 / MALL)
    // and an epilogue that moves 8 bytes per output element through HBM.  Started together, all 256 run their epilogues together -- HBM
    // idles through the K loops and is the bound through the epilogues.  A start phase per workgroup (0 .. 3 quarters of MX_STAGGER_CYC,
    // spread over every XCD) takes them out of step.
    const long long until = (long long)__builtin_amdgcn_s_memtime() + (long long)((blockIdx.x >> 3) & 3) * (MX_STAGGER_CYC / 4);
    while ((long long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(16);
  }
#endif
  wait_stage(0, false);
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();
  const long long st_t1 = MXT();

#define MSB() __builtin_amdgcn_sched_barrier(0)
  constexpr bool EARLY1 = MX_EPI_G1_EARLY && !RES;
  i32x8 fw[4], fx[MT];
  int s = 0, rslot = 0;
  int pm0 = 0, pn0 = 0;
  bool have_prev = false;
  bool pre_issued = false;                           // the coming C slot's stage went out in front of an epilogue already
  for (int tv = blockIdx.x; tv < ntiles; tv += G) {
    int m0, n0;
    tile_of(tv, m0, n0);
    const bool last_tile = tv + G >= ntiles;
    for (int kt = 0; kt < nk; ++kt) {
      // ---- L slot (the other group issues its MFMAs meanwhile): the previous tile's epilogue, this stage's fragments, the DMA of stage s + 2
      long long tq = MXT();
      if (kt == 0 && have_prev && !(EARLY1 && grp)) {
#if MX_PRE_ISSUE
        // The stage this step's C slot would issue goes out HERE, in front of the epilogue: a stage issued behind the stores is, in the wave's
        // in-order vmcnt queue, a wait for the stores' acknowledgement as soon as it is awaited (one step later); this way the first such
        // stage is the next step's, awaited two steps later.  (Its ring slot held stage s - 1, read by both groups before this interval.)
        prefetch_one();
        pre_issued = true;
#endif
        epilogue(pm0, pn0); MXACC(st_ep, tq); tq = MXT();
      }
      const char* sb = smem + rslot * STB;
#ifdef MX_ABL_NOLDS
      if (s == 0)
#endif
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const i32x4 lo = *(const i32x4*)(sb + w_off[v][0]);
        const i32x4 hi = PAIR ? lo : *(const i32x4*)(sb + w_off[v][PAIR ? 0 : 1]);
        fw[v] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#ifdef MX_ABL_NOLDS
      if (s == 0)
#endif
#pragma unroll
      for (int u = 0; u < MT; ++u) {
        const i32x4 lo = *(const i32x4*)(sb + x_off[0] + u * 2048), hi = *(const i32x4*)(sb + x_off[1] + u * 2048);
        fx[u] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#if !MX_DMA_IN_C
      prefetch_one();
#endif
      const bool stores = have_prev && kt < R - 2 + (MX_PRE_ISSUE ? 1 : 0);   // the epilogue's stores are younger than the stage awaited next
      MXACC(st_l, tq); tq = MXT();
      if (grp) wait_stage(s + 1, stores);             // group 1 waits before the barrier ...
      __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the fragments are in registers
      MXACC(st_wait, tq); tq = MXT();
      __builtin_amdgcn_s_barrier();
      MXACC(st_bar, tq); tq = MXT();
      MSB();
      // ---- C slot
#ifndef MX_ABL_NOMMA
#pragma unroll
      for (int u = 0; u < MT; ++u) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          acc[u][v] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[v], fx[u], acc[u][v], 0, 0, 0, scale_w, 0, scale_x);
#if MX_DMA_IN_C
        if (u == 0) { if (pre_issued) pre_issued = false; else prefetch_one(); }   // the DMA of stage s + 2 among the MFMAs: an LDS-DMA instruction issued here costs a third
#endif                                               //   of one issued among the L slot's ds_reads; in one clump - spread over the slot it loses 2-7 %
      }
#else
      acc[0][0][0] += (float)fw[0][0] + (float)fx[0][0];
#endif
      MSB();
      MXACC(st_mma, tq); tq = MXT();
      // group 1 stores its tile right behind the tile's last MFMAs, i.e. in the same barrier interval in which group 0 (one barrier ahead) runs
      // ITS epilogue at the head of the next tile's first L slot: the two epilogues overlap instead of following each other.  (Same order in
      // the wave's vmcnt queue as before: the DMA issued in this step, the stores, the DMA of the next step.)
      // (Not for the residual epilogues: those are bound by their HBM bytes, two at once gain nothing: measured -2 ... +3 %; the others +3 ... +8 %.)
      if (EARLY1 && grp && kt == nk - 1) {
#if MX_PRE_ISSUE
        prefetch_one();                               // (the next step's stage, in front of the stores: as above)
        pre_issued = true;
#endif
        epilogue(m0, n0); MXACC(st_ep, tq); tq = MXT();
      }
      if (!grp) wait_stage(s + 1, stores);            // ... group 0 after its MFMAs
      MXACC(st_wait, tq); tq = MXT();
      if (!(grp && last_tile && kt == nk - 1)) __builtin_amdgcn_s_barrier();
      MXACC(st_bar, tq);
      MSB();
      ++s;
      rslot = rslot + 1 == R ? 0 : rslot + 1;
    }
    pm0 = m0; pn0 = n0;
    have_prev = true;
  }
  {
    const long long tq = MXT();
    if (have_prev && !(EARLY1 && grp)) epilogue(pm0, pn0);
    MXACC(st_ep, tq);
  }
#ifdef WFL_GEMM_STAMPS
  if (p.stamps && wq == 0 && lane == 0) {
    unsigned long long* o = p.stamps + ((long)blockIdx.x * 2 + grp) * 8;
    o[0] = st_t1 - st_t0; o[1] = (long long)__builtin_amdgcn_s_memtime() - st_t0; o[2] = st_ep; o[3] = st_wait; o[4] = st_bar; o[5] = st_mma; o[6] = st_l; o[7] = s;
  }
#endif
#undef MSB
}

template <int ACT, int MT, bool RES, bool PAIR, int OUT>
static int launch_mx(const GemmArgs& a, hipStream_t s) {
  constexpr int BMV = MT * 32;
  constexpr int lds = 3 * (BMV * 128 + 256 * (PAIR ? 64 : 128));
  const int tiles = ((a.M + BMV - 1) / BMV) * (a.N / 256);
  auto k = gemm_mx_kernel<ACT, MT, RES, PAIR, OUT>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  g_wfl_gemm_kernel_id = 7;
  hipLaunchKernelGGL(k, dim3(tiles < MXNCU ? tiles : MXNCU), dim3(512), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

bool wfl_gemm_mx_takes(const GemmArgs& a) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("WFL_GEMM_NO_MX"); off = e && atoi(e) ? 1 : 0; }
  if (off) return false;
  if (a.a8 != 2 && a.a8 != 3) return false;
  if (!a.w8_scale || a.glu || a.out_f32 || a.pos || a.clip_bias || a.ln_s || a.stats_out || a.stats_in || a.tap_wrap || a.clip_T) return false;
  if (a.cin < a.K || a.N % 256 || a.K % 128 || a.K < 512 || a.lda % 16 || a.n_valid % 8) return false;
  if (a.a8 == 3 && !a.a8_lo) return false;
  if (a.c8 && (a.res || a.ldc8 % 8)) return false;
  if (a.c8_lo && !a.c8) return false;
  if (a.res && a.act != WFL_ACT_NONE) return false;
  if (a.act != WFL_ACT_NONE && a.act != WFL_ACT_GELU) return false;
  return true;
}


// Returns 1 when this kernel does not take the launch.
int wfl_launch_gemm_mx(const GemmArgs& a, hipStream_t s) {
  if (!wfl_gemm_mx_takes(a)) return 1;
  static void* trash[32] = {nullptr};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 31;
  if (!trash[dev]) {
    if (hipMalloc(&trash[dev], 4096) != hipSuccess) return -2;
  }
  GemmArgs g = a;
  g.trash = trash[dev];
  const bool pair = a.a8 == 3;
  const int out = a.c8 ? (a.c8_lo ? 2 : 1) : 0;
#define MX_PICK(ACT_, RES_, OUT_) (pair ? launch_mx<ACT_, 6, RES_, true, OUT_>(g, s) : launch_mx<ACT_, 5, RES_, false, OUT_>(g, s))
  if (g.res) return out ? -1 : MX_PICK(WFL_ACT_NONE, true, 0);
  if (g.act == WFL_ACT_GELU) {
    if (out == 2) return MX_PICK(WFL_ACT_GELU, false, 2);
    if (out == 1) return MX_PICK(WFL_ACT_GELU, false, 1);
    return MX_PICK(WFL_ACT_GELU, false, 0);
  }
  if (out) return -1;
  return MX_PICK(WFL_ACT_NONE, false, 0);
#undef MX_PICK
}
