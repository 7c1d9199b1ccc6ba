// Flash-style self-attention for gfx950: softmax(q k^T) v per (clip, head), scores never materialised.
// Replaces HF modeling_whisper.py:215-238 / SDPA (Whisper, head_dim 64) and nn.MultiheadAttention inside the
// Conformer block (/root/reference/model.py:26, 42; head_dim = d / conformer_heads = 256 for Whisper-base).
//
// One workgroup = 4 waves = one (clip, head, block of query frames); each wave owns QT tiles of 16 query frames.
// K tiles and V tiles (64 keys x HD, both row-major exactly as the packed q|k|v projection wrote them) are staged in
// LDS.  Scores are computed TRANSPOSED, S^T = K . Q^T with MFMA 16x16x32 (A = K rows, B = Q rows), so every
// lane holds scores of ONE query frame (column lane&15) and the row max/sum need only two cross-lane steps; the
// S^T accumulators are then already laid out as the B operand of O^T = V^T . P^T, whose A operand (V^T fragments:
// 8 keys of one channel per lane) comes out of the row-major V tile by ds_read_b64_tr_b16, the hardware transposing
// read (key order inside a 32-key k-step is permuted identically for P and for the V reads), so neither P nor a
// transposed copy of V ever exists.
// q is pre-scaled by head_dim^-1/2 * log2(e) at weight-pack time, so the softmax is exp2(s - max).
#include "common.h"
#include <cstdlib>

#define KT 64          // keys per tile
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// V tile rows are HD*2 + 32 bytes apart: the 8 key rows x 32 bytes one 32-lane half of a ds_read_b64_tr_b16 touches then
// fall into 8 different 32-byte windows of the 256-byte bank row (conflict-free for HD = 32 .. 640).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_t;
static __device__ __forceinline__ bf16x4 ds_read_tr(const char* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p));
}

#if defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 2      // (diagnostic build of tools/micro/conv0_probe.hip: no matrix instructions; results are wrong)
static __device__ __forceinline__ f32x4 attn_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c[e] = fmaf(bf2f(a[e]), bf2f(b[e + 4]), c[e]);
  return c;
}
#else
#define attn_mfma(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

#ifndef WFL_ATTN_DMA
#define WFL_ATTN_DMA 1     // the prefetching kernels (head_dim 32 / 64 / 128) stage K and V tiles with LDS-DMA: no staging registers (158 -> 139
#endif                     // VGPRs at head_dim 64), no ds_write, no wait between a tile's loads and its stores.  Round 3, tools/attn_bench.py:
                           // 97.5 -> 89.6 us at cfg2 size, 550 -> 493 us at cfg5 size, bit-identical output.  0: the register-staged form of rounds 1-2
typedef __attribute__((address_space(1))) const void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;
// DMA layout of the V tile: rows HD * 2 bytes apart (an LDS-DMA instruction writes 1 KiB contiguously: no padding), the 32-byte windows
// (one output-channel tile each) of a row XORed so that the 8 key rows x 32 bytes one half of a ds_read_b64_tr_b16 touches fall into 8
// different windows of the 256-byte bank row
template <int HD>
static __host__ __device__ constexpr bool attn_dma(bool prefetch) {
  return WFL_ATTN_DMA && ((prefetch && (HD == 32 || HD == 64 || HD == 128)) || (!prefetch && HD == 256));
}
template <int HD>
static __device__ __forceinline__ int v_swz(int row) {
  return HD == 32 ? (row >> 2) & 1 : HD == 64 ? (row >> 1) & 3 : row & 7;      // (head_dim 256: a row is two bank rows; the low three window bits)
}

template <int HD>
static __device__ __forceinline__ int k_swz(int row) {
  constexpr int CPR = HD / 8;                 // 16-byte chunks per K row
  if (CPR >= 16) return row & 15;
  if (CPR == 8) return (row >> 1) & 7;
  return (row >> 2) & 3;                      // CPR == 4
}

// SPLIT ("model.precision: high", AttnArgs::QK_lo / V_lo): q, k, v arrive as bf16 pairs hi + lo and P is split in registers; the scores
// and the context are three MFMA passes each (the products of two low halves are dropped: 2^-18 relative).  A tile holds four images.
#ifndef WFL_ATTN_SPLIT_P
#define WFL_ATTN_SPLIT_P 1   // 0: P stays ONE bf16 value (five MFMA passes instead of six).  Measured (round 4): 1 % of the precision-high step, and the
#endif                       //   goldens' logits error grows from 0.0004 to 0.003-0.006, held-out raw mismatches from 3-4 to 11 of 96 000, 62 -> 58 identical clips: kept split
template <int HD, int QT, bool PREFETCH, bool BIAS, bool OUT8 = false, bool SPLIT = false>
// (Four waves per SIMD instead of three -- __launch_bounds__(256, 4) -- measured slower twice: 99 us against 90 at 139 registers / 7
//  spilled, 93 against 91 at 129 registers / 5 spilled: a fourth workgroup per CU adds more LDS and L2 contention than latency hiding.
//  A software-pipelined loop -- the scores of tile kt + 1 issued before the softmax of tile kt, K staged one tile ahead of V -- was built
//  and removed again: hipcc wants 218 registers for it (two waves per SIMD: 106.5 us against 94.0 on the same box), and capped at
//  three waves per SIMD it spills 66-86 of them.
//  Round 4: an eight-wave workgroup whose two wave groups alternate between the matrix pipe and the softmax / fragment reads, barriers
//  between the phases (attention_pp.hip in the history), was built, equal in output, and 17-30 % SLOWER: one wave per SIMD issuing
//  v_mfma_f32_16x16x32 leaves the other wave 8 of every 16 issue cycles, its softmax needs more -- profiles/round4_attn_pp_dead_end.txt.)
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
  static_assert(!SPLIT || !OUT8, "split precision: bf16 output");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DMA = attn_dma<HD>(PREFETCH);
  constexpr int VPITCH = DMA ? HD * 2 : HD * 2 + 32;
  constexpr int TILE1 = KT * HD * 2 + KT * VPITCH;          // K tile [KT][HD] (16-byte chunks XOR-swizzled) + V tile [KT][VPITCH]
  constexpr int TILE_BYTES = SPLIT ? 2 * TILE1 : TILE1;     // SPLIT: the low halves' images behind the high ones
  // PREFETCH variants keep TWO tiles in LDS: tile kt+1 is written (from the registers its global loads landed in) right
  // after tile kt's MFMAs, so a key tile costs one workgroup barrier instead of two
  char* Ks = smem;
  char* Vs = smem + KT * HD * 2;
  constexpr int CPR = HD / 8;
  constexpr int KCH = KT * CPR / 256;    // K (and V) chunks per thread per tile
  constexpr int KS = HD / 32;            // k-steps over head_dim
  constexpr int DT = HD / 16;            // output channel tiles

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  // XCD-aware order: blocks i and i+8 share an XCD (and its L2).  All query blocks of one (clip, head) read the
  // same K / V^T, so give each XCD a contiguous run of logical blocks ordered (clip, head, query block): K/V are
  // then fetched from HBM once per (clip, head) instead of once per XCD (PMC: 419 MB -> algorithmic 74 MB per launch).
  const int nqb = (p.T + 4 * QT * 16 - 1) / (4 * QT * 16);     // (grid geometry: the batch-wide frame count)
  const int nblk = nqb * p.heads * p.B;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int qb = bid % nqb;
  const int bh = bid / nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  // a batch of clips of different lengths (AttnArgs::clip_T): this block's clip has its own frame count; keys and queries beyond it
  // do not exist -- exactly the clip labelled alone
  const int T = p.clip_T ? p.clip_T[b] : p.T;
  if (qb * (4 * QT * 16) >= T) return;                          // (block-uniform: before any barrier)
  const int q0 = qb * (4 * QT * 16) + wid * (QT * 16);
  const long row0 = p.lead + (long)b * p.P;
  const bf16_t* Kg = p.QK + p.d + h * HD;                 // + row * ldqk
  const bf16_t* Vg = p.V + h * HD;                        // + row * ldv
  const bf16_t* Kgl = SPLIT ? p.QK_lo + p.d + h * HD : nullptr;
  const bf16_t* Vgl = SPLIT ? p.V_lo + h * HD : nullptr;

  // ---- Q fragments (B operand): lane -> query frame q0 + 16*qt + c, channels 32*ks + 8*g .. +8
  bf16x8 qf[QT][KS], qfl[SPLIT ? QT : 1][SPLIT ? KS : 1];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + c;
    q = q < T ? q : T - 1;
    const bf16_t* qp = p.QK + (row0 + q) * p.ldqk + h * HD + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = *(const bf16x8*)(qp + ks * 32);
    if (SPLIT) {
      const bf16_t* ql = p.QK_lo + (row0 + q) * p.ldqk + h * HD + g * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qfl[qt][ks] = *(const bf16x8*)(ql + ks * 32);
    }
  }

  // Online softmax with a DEFERRED reference: scores are produced as s - mref (the S^T accumulators start at -mref, so
  // the subtraction rides on the MFMA's C operand), P = exp2(s - mref), and mref moves only when a tile holds a score
  // more than RESCALE_THR above it (tile 0 always sets it) -- then, and only then, O and the row sums are rescaled.
  // Softmax is shift invariant, so any mref is exact as long as exp2 stays in range: P <= 2^THR by construction, and
  // every row keeps the term exp2(0) = 1 of the key that set its mref, so the row sum cannot underflow.
  // Row sums come from the matrix pipe: one extra V^T "channel" that is all ones (a constant A fragment, lane c == 0),
  // accumulated like an output tile (osum[qt][0] of lanes g == 0), over the same bf16-rounded P as the numerator.
  constexpr float RESCALE_THR = 8.0f;
  f32x4 o[QT][DT], osum[QT];
  float negm[QT];                        // minus the reference of query tile qt (one value per lane's query: the accumulators start from it)
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

  const int ntiles = (T + KT - 1) / KT;
  bf16x8 kreg[KCH], vreg[KCH];
  bf16x8 kregl[(SPLIT && PREFETCH) ? KCH : 1], vregl[(SPLIT && PREFETCH) ? KCH : 1];    // (non-prefetching kernels reuse kreg / vreg for the low tile)

  // per-lane element offsets inside a tile are fixed; the tile base is block-uniform (scalar base + 32-bit lane offset)
  int koff[KCH], voff[KCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int ch = tid + 256 * i;
    const int r = ch / CPR, cc = ch % CPR;
    koff[i] = r * (int)p.ldqk + cc * 8;
    voff[i] = r * (int)p.ldv + cc * 8;
  }
  auto load_tile = [&](int kt) {
    const bf16_t* kb = Kg + (row0 + (long)kt * KT) * p.ldqk;
    const bf16_t* vb = Vg + (row0 + (long)kt * KT) * p.ldv;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      kreg[i] = *(const bf16x8*)(kb + koff[i]);
      vreg[i] = *(const bf16x8*)(vb + voff[i]);
    }
    if (SPLIT && PREFETCH) {
      const bf16_t* kl = Kgl + (row0 + (long)kt * KT) * p.ldqk;
      const bf16_t* vl = Vgl + (row0 + (long)kt * KT) * p.ldv;
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        kregl[i] = *(const bf16x8*)(kl + koff[i]);
        vregl[i] = *(const bf16x8*)(vl + voff[i]);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int ch = tid + 256 * i;
      const int r = ch / CPR, cc = ch % CPR;
      *(bf16x8*)(Ks + r * (HD * 2) + ((cc ^ k_swz<HD>(r)) << 4)) = kreg[i];
      *(bf16x8*)(Vs + r * VPITCH + cc * 16) = vreg[i];
      if (SPLIT && PREFETCH) {
        *(bf16x8*)(Ks + TILE1 + r * (HD * 2) + ((cc ^ k_swz<HD>(r)) << 4)) = kregl[i];
        *(bf16x8*)(Vs + TILE1 + r * VPITCH + cc * 16) = vregl[i];
      }
    }
  };
  auto load_store_lo = [&](int kt) {           // SPLIT without prefetch: the low tile through the same registers, after the high one
    const bf16_t* kl = Kgl + (row0 + (long)kt * KT) * p.ldqk;
    const bf16_t* vl = Vgl + (row0 + (long)kt * KT) * p.ldv;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      kreg[i] = *(const bf16x8*)(kl + koff[i]);
      vreg[i] = *(const bf16x8*)(vl + voff[i]);
    }
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int ch = tid + 256 * i;
      const int r = ch / CPR, cc = ch % CPR;
      *(bf16x8*)(Ks + TILE1 + r * (HD * 2) + ((cc ^ k_swz<HD>(r)) << 4)) = kreg[i];
      *(bf16x8*)(Vs + TILE1 + r * VPITCH + cc * 16) = vreg[i];
    }
  };

  // DMA staging: a wave's instruction j covers 1 KiB of the tile image (rows of HD * 2 bytes); lane l fetches the 16-byte chunk that
  // belongs at its LDS position under the K / V swizzles
  constexpr int NDI = DMA ? KT * HD * 2 / 1024 / 4 : 1;   // instructions per wave and image
  int kdoff[NDI], vdoff[NDI];
  if (DMA) {
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
      const int byte = (wid * NDI + j) * 1024 + lane * 16;
      const int r = byte / (HD * 2), pos = (byte % (HD * 2)) >> 4;
      kdoff[j] = r * (int)p.ldqk + ((pos ^ k_swz<HD>(r)) << 3);
      vdoff[j] = r * (int)p.ldv + (((((pos >> 1) ^ v_swz<HD>(r)) << 1) | (pos & 1)) << 3);
    }
  }
  auto dma_tile = [&](int kt, int buf) {
    const bf16_t* kb = Kg + (row0 + (long)kt * KT) * p.ldqk;
    const bf16_t* vb = Vg + (row0 + (long)kt * KT) * p.ldv;
    char* dst = smem + buf * TILE_BYTES + wid * NDI * 1024;
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
      __builtin_amdgcn_global_load_lds((attn_gptr_t)(kb + kdoff[j]), (attn_lptr_t)(dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((attn_gptr_t)(vb + vdoff[j]), (attn_lptr_t)(dst + KT * HD * 2 + j * 1024), 16, 0, 0);
    }
    if (SPLIT) {
      const bf16_t* kl = Kgl + (row0 + (long)kt * KT) * p.ldqk;
      const bf16_t* vl = Vgl + (row0 + (long)kt * KT) * p.ldv;
#pragma unroll
      for (int j = 0; j < NDI; ++j) {
        __builtin_amdgcn_global_load_lds((attn_gptr_t)(kl + kdoff[j]), (attn_lptr_t)(dst + TILE1 + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((attn_gptr_t)(vl + vdoff[j]), (attn_lptr_t)(dst + TILE1 + KT * HD * 2 + j * 1024), 16, 0, 0);
      }
    }
  };

  // BIAS: the slice of the relative-position table a key tile needs -- offsets key - query for this workgroup's 4 * QT * 16 queries
  // and the tile's 64 keys, 64 + 4 * QT * 16 - 1 entries -- is staged in LDS one tile ahead (it rides on the tile's barrier); round 1
  // gathered every score's entry from global memory inside the loop (the BIAS kernel ran at half the plain kernel's rate)
  constexpr int NQW = 4 * QT * 16;                       // queries per workgroup
  constexpr int BT = KT + NQW;                           // table entries per tile (one spare)
  float* btab = (float*)(smem + (PREFETCH ? 2 : 1) * TILE_BYTES);   // [2][BT]
  const int qwg0 = qb * NQW;
  const float* tabc = BIAS ? p.bias + (long)h * (2 * p.T - 1) + (p.T - 1) : nullptr;    // (table and gate are laid out for the batch-wide T)
  auto stage_bias = [&](int kt) {
    if (BIAS && tid < BT - 1) {
      int off = kt * KT - qwg0 - (NQW - 1) + tid;        // key - query for entry tid
      off = off < -(p.T - 1) ? -(p.T - 1) : (off > p.T - 1 ? p.T - 1 : off);
      btab[(kt & 1) * BT + tid] = tabc[off];
    }
  };
  float gq[QT];
  if (BIAS) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      int q = q0 + qt * 16 + c;
      q = q < T ? q : T - 1;
      gq[qt] = p.gate[((long)b * p.heads + h) * p.T + q];
    }
    stage_bias(0);
  }
  if (DMA && PREFETCH) {
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (PREFETCH) {
    load_tile(0);
    store_tile();                        // tile 0 -> buffer 0
    if (ntiles > 1) load_tile(1);        // in flight under tile 0's MFMAs
    __syncthreads();
  }
  for (int kt = 0; kt < ntiles; ++kt) {
    if (DMA && PREFETCH && kt + 1 < ntiles) dma_tile(kt + 1, (kt + 1) & 1);     // into the buffer every wave left before the last barrier
    if (!PREFETCH) {
      __syncthreads();                   // every wave is done reading the previous tile
      if (DMA) {                         // (one buffer: the co-resident workgroup covers the wait)
        dma_tile(kt, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        load_tile(kt);
        store_tile();
        if (SPLIT) load_store_lo(kt);
      }
      __syncthreads();
    } else {
      Ks = smem + (kt & 1) * TILE_BYTES;
      Vs = Ks + KT * HD * 2;
    }

    // ---- S^T = K . Q^T : st[qt][kk][e] = score(query q0+16qt+c, key 64kt + 16kk + 4g + e)
    f32x4 st[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) st[qt][kk] = (f32x4){negm[qt], negm[qt], negm[qt], negm[qt]};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int r = kk * 16 + c;
#if defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 3       // (diagnostic: no fragment reads from LDS; results are wrong)
        const bf16x8 kf = qf[0][ks];
#else
        const bf16x8 kf = *(const bf16x8*)(Ks + r * (HD * 2) + (((ks * 4 + g) ^ k_swz<HD>(r)) << 4));
#endif
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          st[qt][kk] = attn_mfma(kf, qf[qt][ks], st[qt][kk]);
        if (SPLIT) {
          const bf16x8 kfl = *(const bf16x8*)(Ks + TILE1 + r * (HD * 2) + (((ks * 4 + g) ^ k_swz<HD>(r)) << 4));
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            st[qt][kk] = attn_mfma(kf, qfl[qt][ks], st[qt][kk]);
            st[qt][kk] = attn_mfma(kfl, qf[qt][ks], st[qt][kk]);
          }
        }
      }
    }
    if (BIAS) {
      // WavLM gated relative position bias (HF modeling_wavlm.py:167-180, 243-271): score += gate[b,h,q] * table[h][k-q],
      // table = rel_attn_embed[bucket(k - q)][h] * log2(e), one row of 2T-1 entries per head, built at load time
      const float* bt = btab + (kt & 1) * BT + (NQW - 1) - (wid * (QT * 16) + c) + g * 4;   // entry of (key 0 of the tile, this lane's query 0)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e) st[qt][kk][e] = fmaf(gq[qt], bt[kk * 16 + e - qt * 16], st[qt][kk][e]);
    }
    if (kt * KT + KT > T) {            // last tile: keys >= T do not exist
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kt * KT + kk * 16 + g * 4 + e >= T) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) st[qt][kk][e] = -INFINITY;
          }
    }

    // ---- softmax numerators (per query frame = per lane column c; the 4 lane groups g share a frame)
    bf16x8 pf[QT][2], pfl[SPLIT ? QT : 1][2];
    {
      float mx[QT];
      bool over = kt == 0;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float m = st[qt][0][0];
#if !(defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 5)  // (diagnostic: no running maximum; results are wrong)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e) m = fmaxf(m, st[qt][kk][e]);
#endif
        mx[qt] = m;
        over = over || m > RESCALE_THR;
      }
      if (__builtin_amdgcn_ballot_w64(over) != 0) {      // wave-uniform and rare after the first tiles
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float m = mx[qt];
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          // raise-only (tile 0: take the tile's max, whatever its sign); a fully masked tile cannot occur (kt*KT < T)
          const float d = kt == 0 ? m : fmaxf(m, 0.f);    // new mref - old mref
          const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);   // (O and the sums are still zero at tile 0)
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
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#if defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 1     // (diagnostic build of tools/micro/conv0_probe.hip: no transcendental instructions)
          t[e] = f2bf(st[qt][2 * s2][e] * 0.001f);
          t[4 + e] = f2bf(st[qt][2 * s2 + 1][e] * 0.001f);
#elif defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 8   // (diagnostic: no softmax arithmetic at all -- no exp2, no conversion)
          t[e] = __builtin_bit_cast(bf16x2_t, st[qt][2 * s2][e])[1];
          t[4 + e] = __builtin_bit_cast(bf16x2_t, st[qt][2 * s2 + 1][e])[1];
#else
          t[e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2][e]));
          t[4 + e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2 + 1][e]));
#endif
        }
        pf[qt][s2] = t;
        osum[qt] = attn_mfma(ones, t, osum[qt]);
        if (SPLIT && WFL_ATTN_SPLIT_P) {          // what the rounding of P left behind, as a second operand
          bf16x8 tl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            tl[e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2][e]) - bf2f(t[e]));
            tl[4 + e] = f2bf(__builtin_amdgcn_exp2f(st[qt][2 * s2 + 1][e]) - bf2f(t[4 + e]));
          }
          pfl[qt][s2] = tl;
          osum[qt] = attn_mfma(ones, tl, osum[qt]);
        }
      }
    }

    // ---- O^T += V^T . P^T : k-slot 8g+j of k-step s2 is key 32*s2 + 4g + j (j<4) / 32*s2 + 16 + 4g + j-4.
    // Transposing read: lane 4q+pp of a 16-lane group addresses key row (block base + q), channels 4pp..4pp+3 and
    // receives channel (lane & 15) of the block's 4 key rows.
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int vr = 32 * s2 + 4 * g + (c >> 2);        // (row vr + 16 has the same window swizzle)
        const char* vp = Vs + vr * VPITCH + ((DMA ? dt ^ v_swz<HD>(vr) : dt) * 16 + 4 * (c & 3)) * 2;
#if defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 3
        const bf16x8 vf = qf[0][s2];
        (void)vp;
#else
        const bf16x4 lo = ds_read_tr(vp);
        const bf16x4 hi = ds_read_tr(vp + 16 * VPITCH);
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#endif
#if defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 7       // (diagnostic: the V fragments are read, the context MFMAs are not issued)
        asm volatile("" :: "v"(vf));
#else
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          o[qt][dt] = attn_mfma(vf, pf[qt][s2], o[qt][dt]);
#endif
        if (SPLIT) {
          const bf16x4 llo = ds_read_tr(vp + TILE1);
          const bf16x4 lhi = ds_read_tr(vp + TILE1 + 16 * VPITCH);
          const bf16x8 vfl = {llo[0], llo[1], llo[2], llo[3], lhi[0], lhi[1], lhi[2], lhi[3]};
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            if (WFL_ATTN_SPLIT_P) o[qt][dt] = attn_mfma(vf, pfl[qt][s2], o[qt][dt]);
            o[qt][dt] = attn_mfma(vfl, pf[qt][s2], o[qt][dt]);
          }
        }
      }
    }
    if (DMA && PREFETCH && kt + 1 < ntiles) {
      stage_bias(kt + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt + 1 has landed (this wave's pieces; the barrier covers the others')
      __syncthreads();
    } else if (PREFETCH && kt + 1 < ntiles) {
      // tile kt+1 (in registers since the previous iteration) -> the other buffer, which every wave left before the last
      // barrier; then start tile kt+2's loads
      Ks = smem + ((kt + 1) & 1) * TILE_BYTES;
      Vs = Ks + KT * HD * 2;
#if !(defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 4)  // (diagnostic: the first tile over and over; results are wrong)
      store_tile();
      if (kt + 2 < ntiles) load_tile(kt + 2);
#endif
      stage_bias(kt + 1);
#if !(defined(WFL_ABL_ATTN) && WFL_ABL_ATTN == 6)  // (diagnostic: no barrier -- racy, results are wrong)
      __syncthreads();
#endif
    }
  }

  // ---- normalise and store: lane holds channels h*HD + 16dt + 4g + e of query frame q0 + 16qt + c
  bool sat8 = false;                                 // OUT8: a stored element did not fit e4m3 at the fixed scale -> bit 1 of the error word
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = __shfl(osum[qt][0], c);          // row sum of query c lives in lane (g = 0, c), register 0
    const float inv = 1.0f / l;
    const int q = q0 + qt * 16 + c;
    if (OUT8) {
      if (q < T) {                                   // e4m3 with a fixed scale, saturating: four channels = four bytes per lane and tile
        unsigned char* op8 = p.O8 + (row0 + q) * p.ldo8 + h * HD + g * 4;
        const float sc = inv * p.o8_scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = o[qt][dt][e] * sc;
            sat8 |= !(fabsf(y) <= 448.f);            // (NaN included)
            x[e] = fminf(fmaxf(y, -448.f), 448.f);
          }
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
          *(int*)(op8 + dt * 16) = w;
          if (p.O8_lo) {                             // (wave-uniform) the pair's low plane: what the rounding left behind, 16 x
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
        if (p.O_lo) {                                 // (precision: high) what the rounding left behind
          bf16x4 ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) ol[e] = f2bf(o[qt][dt][e] * inv - bf2f(ov[e]));
          *(bf16x4*)(p.O_lo + (op - p.O) + dt * 16) = ol;
        }
      }
    }
  }
  if (OUT8 && p.err) {
    if (__builtin_amdgcn_ballot_w64(sat8) && lane == 0) atomicOr(p.err, 2u);
  }
}

template <int HD, int QT, bool PREFETCH, bool BIAS, bool OUT8 = false, bool SPLIT = false>
static int launch_attn(const AttnArgs& a, hipStream_t s) {
  static_assert(!BIAS || PREFETCH, "the bias table slice is staged on the prefetch barrier");
  constexpr int lds = (PREFETCH ? 2 : 1) * (SPLIT ? 2 : 1) * (KT * HD * 2 + KT * (attn_dma<HD>(PREFETCH) ? HD * 2 : HD * 2 + 32)) +
                      (BIAS ? 2 * (KT + 4 * QT * 16) * 4 : 0);
  auto k = attn_kernel<HD, QT, PREFETCH, BIAS, OUT8, SPLIT>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  const int qb = 4 * QT * 16;
  dim3 grid(((a.T + qb - 1) / qb) * a.heads * a.B);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wfl_launch_attention_big(const AttnArgs& a, hipStream_t s);   // attention_big.hip: head_dim 384 / 512 / 640

// WFL_ATTN_VARIANT=1 selects round 1's kernel (attention_big.hip) for head_dim 384 (A/B runs)
static int attn_variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("WFL_ATTN_VARIANT"); v = e ? atoi(e) : 0; }
  return v;
}

int wfl_launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.heads <= 0 || a.d % a.heads || a.P % 8 || a.ldqk % 8 || a.ldv % 8 || a.ldo % 4 || a.T <= 0 || !a.V) return -1;
  const int hd = a.d / a.heads;
  if (a.O8) {
    if (a.bias || hd != 64 || a.ldo8 % 4) return -4;
    // (64 queries per wave here too: cfg5 with e4m3 activations 56.0 against 57.1 ms per step, whole-model A/B; WFL_ATTN_VARIANT=5: 32)
    return attn_variant() == 5 ? launch_attn<64, 2, true, false, true>(a, s) : launch_attn<64, 4, true, false, true>(a, s);
  }
  if (a.QK_lo && a.V_lo && !a.O8) {                     // "model.precision: high": three passes over split operands
    if (a.bias && a.gate && hd == 64) return launch_attn<64, 2, true, true, false, true>(a, s);     // (WavLM-base / -large)
    if (a.bias && a.gate && hd == 32) return launch_attn<32, 2, true, true, false, true>(a, s);
    if (!a.bias && hd == 32) return launch_attn<32, 2, true, false, false, true>(a, s);
    if (!a.bias && hd == 64) return launch_attn<64, 2, true, false, false, true>(a, s);
    if (!a.bias && hd == 128) return launch_attn<128, 2, true, false, false, true>(a, s);
    if (!a.bias && hd == 256) return launch_attn<256, 1, false, false, false, true>(a, s);
  }
  if (a.bias) {
    if (!a.gate) return -1;
    switch (hd) {
      case 32: return launch_attn<32, 2, true, true>(a, s);
      case 64: return launch_attn<64, 2, true, true>(a, s);
    }
    return -4;
  }
  switch (hd) {
    case 32: return launch_attn<32, 2, true, false>(a, s);
    // head_dim 64 at cfg2 size (tools/attn_bench.py 512 8 16; WFL_ATTN_VARIANT=2 / 3: 16 / 64 queries per wave): see DESIGN.md section 4
    // Round 4: 64 queries per wave is the default.  Per launch it equals 32 (89.1 against 89.6 us), in the whole model -- beside the other stream's
    // launches -- it is worth 1.1 % of cfg2's step (3.62-3.64 against 3.66-3.68 ms, two runs each on one box; WFL_ATTN_VARIANT=5 selects 32 again).
    case 64: return attn_variant() == 2 ? launch_attn<64, 1, true, false>(a, s)
                  : attn_variant() == 5 ? launch_attn<64, 2, true, false>(a, s)
                  : attn_variant() == 4 ? launch_attn<64, 3, true, false>(a, s) : launch_attn<64, 4, true, false>(a, s);
    case 128: return launch_attn<128, 2, true, false>(a, s);
    // head_dim 256, measured at cfg2 size (tools/attn_bench.py): <256, 1, no prefetch> 125 us (two workgroups per CU hide each other's
    // tile loads); <256, 2, prefetch> 166; <256, 2, no prefetch> 168; <256, 1, prefetch> 212 (one workgroup per CU each)
    case 256: return launch_attn<256, 1, false, false>(a, s);
    // head_dim 384 (Whisper-small's Conformer heads): 32 queries per wave with Q in registers (506 VGPRs, one wave per SIMD) halves the
    // K / V tile traffic per query against attention_big's 16: 845 vs 2 635 us at 64 x 1500 frames (tools/attn_bench.py)
    case 384: return attn_variant() == 1 ? wfl_launch_attention_big(a, s) : launch_attn<384, 2, false, false>(a, s);
    case 512: case 640: return wfl_launch_attention_big(a, s);
  }
  return -4;   // unsupported head_dim
}
