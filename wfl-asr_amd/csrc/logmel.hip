// Whisper log-mel front-end on gfx950, fp32 end to end.  Replaces the CPU round trip at
// /root/reference/model.py:153-154 (HF feature_extraction_whisper.py:135-168, 300-307): zero-pad/truncate to
// n_samples, centred STFT (n_fft 400, hop 160, periodic Hann, reflect pad 200), power, Slaney mel, log10
// (clamp 1e-10), per-clip floor (max - 8), (x + 4) / 4.
//
// Kernel 1 (one workgroup = FT = 32 frames of one clip): the 400-point real DFT is a GEMM
//   [FT frames x 400 samples] . [400 x 2*201 (Hann-folded cos | -sin)]
// run on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, so fp32 like torch.stft).
// The frames operand is Toeplitz (frame f, sample n = signal[160 f + n]), so the workgroup stages only the
// 5 360-sample signal segment in LDS, skewed by one word per 160 samples so that the 32 frames a fragment
// read touches fall in 32 different banks.  Power spectra go through LDS to the sparse mel projection
// (each bin feeds <= 2 triangles), log10, and an ordered-int atomic max per clip.
// Kernel 2 applies the per-clip floor + affine and writes bf16 channels-last frame rows (the Whisper stem's
// GEMM operand) and, for parity tests, an optional fp32 copy in the reference's [B, n_mels, frames] layout.
//
// The same STFT kernel, instantiated with POWER = true, is the `encoder_type: none` front-end (/root/reference/model.py:82-91,
// 149-150: torchaudio.transforms.MelSpectrogram(sample_rate, n_fft=400, hop_length=frame_duration * sample_rate, n_mels), i.e.
// periodic Hann, centred / reflect-padded frames, power 2, HTK mel triangles without normalisation, NO log): frames
// 1 + L / hop, the plain mel power goes out as fp32 [B][frames][n_mels] (the hidden states themselves) and kernel 2' rounds it
// into the head's bf16 frame rows.  Hop 160 and 320 are instantiated (frame_duration 0.01 / 0.02 s at 16 kHz).
#include "common.h"

#define NFFT 400
#define NBIN_PAD 224     // 201 bins padded to 7 MFMA column tiles
#define FT 32            // frames per workgroup: 50 KB of LDS; at 200 registers TWO workgroups share a CU and one's staging /
                         // mel phases run under the other's MFMA loop (64 frames = 100 KB = one per CU ran 2x slower).  Round 3 tried
                         // three per CU (__launch_bounds__(256, 3), twiddle ring 10 deep, 14 registers spilled, the wave with a single
                         // column tile rotated over the SIMDs): the median workgroup took 55 us instead of 39 and the launch 145 us
                         // instead of 147 -- no gain, reverted (tools/micro/logmel_bench.hip prints the phases)
#define RT (FT / 32)     // 32-frame MFMA row tiles per workgroup
#define PPITCH 225
template <int HOP> struct Seg {
  static constexpr int SEG = (FT - 1) * HOP + NFFT;         // hop 160: 5360 samples
  static constexpr int SEG_LDS = SEG + SEG / HOP + 2;       // skewed
};
#ifdef WFL_LOGMEL_STAMPS
#define LSTAMP(k) do { if (threadIdx.x == 0 && p.stamps) p.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif

// (struct LogmelArgs: common.h -- one definition for the kernel and for model.hip)

static __device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static __device__ __forceinline__ float ord2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <int HOP, bool POWER>
__global__ __launch_bounds__(256) void logmel_power_kernel(LogmelArgs p) {
  constexpr int SEG = Seg<HOP>::SEG, SEG_LDS = Seg<HOP>::SEG_LDS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* seg = (float*)smem;
  float* pw = seg + ((SEG_LDS + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, f0 = blockIdx.x * FT;
  const int len = p.lens ? min(p.lens[b], min(p.L, p.n_samples)) : min(p.L, p.n_samples);
  // POWER with per-clip lengths: the clip is reflected about ITS last sample and has 1 + len / HOP frames (a clip labelled alone);
  // the frames up to the batch-wide count are written as zeros
  const int nsamp = (POWER && p.lens) ? len : p.n_samples;
  const int nfr = (POWER && p.lens) ? (len > NFFT / 2 ? 1 + len / HOP : 0) : p.n_frames;
  const float* w = p.wav + (long)b * p.ldw;
  LSTAMP(0);

  // ---- stage the signal segment: sample index i = 160*f0 - 200 + j, reflect about 0 and n_samples-1
  // All 41 loads of a thread are issued before the first LDS write (unconditional, index clamped, value selected afterwards):
  // a load-then-store loop serialises 41 HBM round trips per workgroup (it was 2/3 of this kernel's time).
  const int s0 = f0 * HOP - NFFT / 2;
  constexpr int NIT = (SEG + 255) / 256;
  float sv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int i = s0 + tid + 256 * it;
    if (i < 0) i = -i;
    if (i >= nsamp) i = 2 * (nsamp - 1) - i;
    const bool ok = i >= 0 && i < len;
    const int ic = min(max(i, 0), p.L - 1);
    const float v = w[ic];
    sv[it] = ok ? v : 0.f;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int j = tid + 256 * it;
    if (j < SEG) seg[j + j / HOP] = sv[it];
  }
  __syncthreads();
  LSTAMP(1);

  // ---- DFT on the fp32 MFMA.  Lane l: A[row = l&31][k = l>>5], B[k = l>>5][col = l&31].
  // The periodic Hann window and the twiddles are symmetric about n = 200 (w[400-n] = w[n], cos(2 pi (400-n) k/400) =
  // cos(2 pi n k/400), sin(...) = -sin(...)) and w[0] = 0, so
  //     Re[k] = sum_{j=1..200} (x[j] + x[400-j]) * Wc[j][k],   Im[k] = sum_{j=1..200} (x[j] - x[400-j]) * Ws[j][k]
  // with Wc[200] carrying a factor 1/2 (x[200] is counted twice by the fold) and Ws[200] = 0: 100 K-steps of 2
  // instead of 200.  The twiddle rows come from L2 through a register ring PD steps deep (one wave per SIMD has
  // nobody else to hide a ~1 us L2 round trip).
  const int r = lane & 31, kh = lane >> 5;
  constexpr int PD = 20;                       // prefetch depth in K-steps (2 MFMAs = 128 cycles each); 100 steps = 5 rounds of PD
  for (int ct = wid; ct < NBIN_PAD / 32; ct += 4) {
    f32x16 re[RT], im[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { re[rt][e] = 0.f; im[rt][e] = 0.f; }
    const float* fa = seg + (HOP + 1) * r;                   // frame r of row tile 0 (skewed: + n + n/HOP); tile rt at + (HOP+1)*32*rt
    const float* bc = p.Wc + (long)kh * NBIN_PAD + ct * 32 + r;   // row j-1 of the folded tables, j = 1 + 2*step + kh
    const float* bs = p.Ws + (long)kh * NBIN_PAD + ct * 32 + r;
    float wcv[PD], wsv[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) { wcv[q] = bc[(long)(2 * q) * NBIN_PAD]; wsv[q] = bs[(long)(2 * q) * NBIN_PAD]; }
    // the signal samples of a K-step are read from LDS one step ahead of the MFMAs that consume them
    auto ldx = [&](int step, float (&xa)[RT], float (&xb)[RT]) __attribute__((always_inline)) {
      const int j = 1 + 2 * step + kh;                       // 1..200
      const int n2 = NFFT - j;                               // 200..399
      const int a1 = j + (j >= HOP ? 1 : 0);                 // (j <= 200 < 2 HOP)
      const int a2 = n2 + (n2 >= 2 * HOP ? 2 : (n2 >= HOP ? 1 : 0));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { xa[rt] = fa[(HOP + 1) * 32 * rt + a1]; xb[rt] = fa[(HOP + 1) * 32 * rt + a2]; }
    };
    float na[RT], nb[RT];
    ldx(0, na, nb);
    for (int sb = 0; sb < 100; sb += PD) {
#pragma unroll
      for (int q = 0; q < PD; ++q) {
        const int step = sb + q;
        float xa[RT], xb[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { xa[rt] = na[rt]; xb[rt] = nb[rt]; }
        const float c = wcv[q], sn = wsv[q];
        __builtin_amdgcn_sched_barrier(0);            // pin the prefetches here: hipcc otherwise sinks the twiddle loads to
        if (step + PD < 100) {                         // their use PD steps later and every step eats an L2 round trip
          wcv[q] = bc[(long)(2 * (step + PD)) * NBIN_PAD];
          wsv[q] = bs[(long)(2 * (step + PD)) * NBIN_PAD];
        }
        if (step + 1 < 100) ldx(step + 1, na, nb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          re[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[rt] + xb[rt], c, re[rt], 0, 0, 0);
          im[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[rt] - xb[rt], sn, im[rt], 0, 0, 0);
        }
      }
    }
    // D[row][col]: col = lane&31 (bin), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (frame in the row tile)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int fr = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        pw[fr * PPITCH + ct * 32 + r] = re[rt][e] * re[rt][e] + im[rt][e] * im[rt][e];
      }
  }
  LSTAMP(2);
  __syncthreads();
  LSTAMP(3);

  // ---- sparse mel projection + log10; thread -> frame (tid % FT), mel bands (tid / FT) + (256 / FT) * i.
  // The filter table (weights, first bin, width per band: ~10 KB) moves into the now idle signal segment first: read from
  // global memory inside the band loop, every tap was a dependent L2 round trip (~1/3 of this kernel's time).
  const int f = tid % FT;
  const bool fvalid = f0 + f < p.n_frames;
  float mx = -INFINITY;
  const int nw = p.n_mels * p.mel_maxw;
  const bool tab_lds = nw + 2 * p.n_mels <= SEG_LDS;
  float* lw = seg;
  int* llo = (int*)(seg + nw);
  int* lcnt = llo + p.n_mels;
  if (tab_lds) {
    for (int i = tid; i < nw; i += 256) lw[i] = p.mel_w[i];
    for (int i = tid; i < p.n_mels; i += 256) { llo[i] = p.mel_lo[i]; lcnt[i] = p.mel_cnt[i]; }
    __syncthreads();
  }
  for (int m = tid / FT; m < p.n_mels; m += 256 / FT) {
    const int lo = tab_lds ? llo[m] : p.mel_lo[m], cnt = tab_lds ? lcnt[m] : p.mel_cnt[m];
    const float* mw = tab_lds ? lw + m * p.mel_maxw : p.mel_w + (long)m * p.mel_maxw;
    float acc = 0.f;
    for (int i = 0; i < cnt; ++i) acc += mw[i] * pw[f * PPITCH + lo + i];
    const float lv = POWER ? (f0 + f < nfr ? acc : 0.f) : log10f(fmaxf(acc, 1e-10f));
    if (fvalid) {
      p.raw[((long)b * p.n_frames + f0 + f) * p.n_mels + m] = lv;
      mx = fmaxf(mx, lv);
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
  if (!POWER && lane == 0 && mx > -INFINITY) atomicMax(p.clipmax + b, f2ord(mx));
  LSTAMP(4);
}

__global__ __launch_bounds__(256) void logmel_finish_kernel(const float* __restrict__ raw, const unsigned* __restrict__ clipmax,
                                                            int B, int n_frames, int n_mels, bf16_t* __restrict__ out,
                                                            long ldo, long lead, int P, float* __restrict__ ref_out, bf16_t* __restrict__ out_lo) {
  const long total = (long)B * n_frames * n_mels;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % n_mels);
    const long bf = i / n_mels;
    const int t = (int)(bf % n_frames), b = (int)(bf / n_frames);
    const float floorv = ord2f(clipmax[b]) - 8.0f;
    const float v = (fmaxf(raw[i], floorv) + 4.0f) / 4.0f;
    if (out) {
      const bf16_t hv = f2bf(v);
      out[(lead + (long)b * P + t) * ldo + m] = hv;
      if (out_lo) out_lo[(lead + (long)b * P + t) * ldo + m] = f2bf(v - bf2f(hv));     // (precision: high: the stem's split-precision operand)
    }
    if (ref_out) ref_out[((long)b * n_mels + m) * n_frames + t] = v;
  }
}

// `none` front-end: raw [B][frames][n_mels] fp32 -> bf16 frame rows; channel m lands in column m + (m >= split ? shift : 0)
// (the head's padded channel layout: model.hip, pad_head_state), the other columns of a row are left as they are (zero).
__global__ __launch_bounds__(256) void melpower_rows_kernel(const float* __restrict__ raw, int B, int n_frames, int n_mels,
                                                            bf16_t* __restrict__ out, long ldo, long lead, int P, int split, int shift) {
  const long total = (long)B * n_frames * n_mels;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % n_mels);
    const long bf = i / n_mels;
    const int t = (int)(bf % n_frames), b = (int)(bf / n_frames);
    out[(lead + (long)b * P + t) * ldo + m + (m >= split ? shift : 0)] = f2bf(raw[i]);
  }
}

template <int HOP, bool POWER>
static int launch_power(const LogmelArgs& a, hipStream_t s) {
  constexpr int lds = (((Seg<HOP>::SEG_LDS + 3) & ~3) + FT * PPITCH) * 4;
  auto k = logmel_power_kernel<HOP, POWER>;
  static WflOncePerDevice attr_once;
  if (attr_once.need()) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
  }
  dim3 grid((a.n_frames + FT - 1) / FT, a.B);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return 0;
}

// encoder_type none: a.raw receives the mel power (= the hidden states, fp32 [B][frames][n_mels]); a.n_samples = a.L > 200; with
// a.lens every clip is reflected about its own last sample and has its own 1 + len / hop frames (zeros behind them).
int wfl_launch_melpower(const LogmelArgs& a, int hop, bf16_t* out, long ldo, long lead, int P, int split, int shift, hipStream_t s) {
  if (a.n_samples != a.L || a.L <= NFFT / 2 || a.n_frames != 1 + a.L / hop || a.n_mels <= 0 || a.B <= 0) return -1;
  int r;
  if (hop == 160) r = launch_power<160, true>(a, s);
  else if (hop == 320) r = launch_power<320, true>(a, s);
  else return -1;
  if (r) return r;
  const long total = (long)a.B * a.n_frames * a.n_mels;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(melpower_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.raw, a.B, a.n_frames, a.n_mels, out, ldo, lead,
                     P, split, shift);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wfl_launch_logmel(const LogmelArgs& a, bf16_t* out, long ldo, long lead, int P, float* ref_out, hipStream_t s, bf16_t* out_lo) {
  if (a.n_frames * 160 != a.n_samples || a.n_mels <= 0 || a.B <= 0) return -1;
  if (wfl_launch_fill_i32((int*)a.clipmax, a.B, 0, s)) return -3;          // (a kernel, not a memset node: common.h)
  if (int r = launch_power<160, false>(a, s)) return r;
  const long total = (long)a.B * a.n_frames * a.n_mels;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(logmel_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.raw, a.clipmax, a.B, a.n_frames,
                     a.n_mels, out, ldo, lead, P, ref_out, out_lo);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
