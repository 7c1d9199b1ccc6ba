// Shared device/host definitions for the gfx950 kernels of the WFL-ASR labeling hot path.
//
// HBM layout of every activation ("frame rows"): a flat run of rows, C channels each, bf16:
//
//     row(b, t) = lead + b * pitch + t          0 <= t < T,   pitch = T + halo  (pitch % 8 == 0)
//
// Rows outside [0, T) of a clip (the `halo` rows between clips, the `lead` rows in front of clip 0 and the
// slack behind the last clip) are ZERO and are never written by any kernel, so a k-tap Conv1d over time is a
// plain GEMM whose A row for output frame t is the contiguous run of k*C channels starting at frame t - pad
// (channels-last makes the taps adjacent), a stride-s conv is the same GEMM with lda = s*C reading a buffer
// whose pitch is s times the output pitch, and tiles may straddle clips: a kernel walks the flat row index m,
// and only decides `m % pitch < T` when it stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WFL_ACT_NONE 0
#define WFL_ACT_GELU 1
#define WFL_ACT_RELU 2
#define WFL_ACT_SIGMOID 3

// One bf16 GEMM launch:  C[m][n] = epi( sum_k A(m, k) * W[n][k] )
struct GemmArgs {
  const bf16_t* A;      // row m starts at A + m * lda
  long lda;             // elements
  int cin;              // channels per tap; k -> tap = k / cin, c = k % cin, at +tap * tap_stride + c
  long tap_stride;      // elements between taps (= dilation * C of the input buffer); cin >= K => single tap
  int tap_wrap;         // > 0 ("model.precision: high", one-launch form): K holds THREE segments of tap_wrap taps each -- tap index t reads at
  long seg_off;         //   (t % tap_wrap) * tap_stride, plus seg_off for the middle segment (the operand's low half, lo_delta away):
                        //   [A_hi | A_lo | A_hi] against weights packed [W_hi | W_hi | W_lo]
  const bf16_t* W;      // [N][K] row-major, N % 128 == 0, K % 64 == 0 (zero padded)
  int M, N, K;
  int n_valid;          // columns >= n_valid are not stored
  int P, T;             // flat row m -> clip b = m / P, frame t = m % P, stored iff t < T
  void* C;              // output rows: C + (c_lead + b * c_pitch + t) * ldc
  long ldc;
  long c_lead;
  int c_pitch;
  const float* bias;        // [N] or null
  const float* clip_bias;   // [n_clip_rows][clip_ld] per-clip additive bias table or null
  const int* clip_idx;      // [B] row of clip_bias used by clip b
  int clip_ld;
  const bf16_t* res;        // residual, same row mapping as C, or null
  long ldres;
  // The residual stream is carried as a bf16 PAIR hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits), so that the
  // dozens of residual adds of a forward do not each round the running sum to 8 bits.  GEMM operands read `hi` only -- exactly
  // the bf16 rounding of x an MFMA operand is anyway.  Both optional:
  const bf16_t* res_lo;     //   low half of the residual (ld = ldres), null: the residual is hi alone
  bf16_t* c_lo;             //   low half of the output (bf16 outputs only; ld = ldc, same row mapping as C), null: not kept
  int acc_f32;              // out_f32 only: C += result (the split-precision classifier adds its passes up in fp32)
  // fp8 weights (BASELINE configs[4]): W is [N][K] OCP e4m3 bytes, one fp32 scale per output channel; the tile is converted to
  // bf16 in registers (exact) in front of the bf16 MFMA, the scale multiplies the accumulator in the epilogue (gemm_stream.hip)
  const float* w8_scale;    //   [N] per-output-channel scale, non-null <=> W holds e4m3 bytes
  // fp8 ACTIVATIONS as well (round 3; gemm_stream.hip, template A8): A holds OCP e4m3 bytes (lda in bytes, single tap), the value of
  // element (m, k) is a8(m, k) * sa(m) with sa(m) = a8_scale[a8_lead + m] (one fp32 scale per A row, written by the kernel that
  // quantised the row) or a8_static when a8_scale is null.  Needs w8_scale; the MFMA is v_mfma_f32_16x16x32_fp8_fp8, both scales
  // multiply the accumulator in the epilogue.
  int a8;                   // 1: gemm_stream's A8 form (non-scaled fp8 MFMA); 2 / 3 (round 4, gemm_mx.hip): the block-scaled K = 128 MFMA with
                            //   A as one e4m3 plane / as an e4m3 PAIR hi + lo (a8_lo: lo = e4m3(16 (x / s - hi)), same lda and scale)
  const unsigned char* a8_lo;
  const float* a8_scale;
  long a8_lead;
  float a8_static;
  // fp8 OUTPUT (with a8): instead of C, e4m3(out * c8_inv_scale) goes to c8 (ldc8 bytes per row, same row mapping as C)
  unsigned char* c8;
  unsigned char* c8_lo;     // gemm_mx.hip: the output as an e4m3 pair -- e4m3(16 (out * c8_inv_scale - hi)) goes here (ld = ldc8)
  long ldc8;
  float c8_inv_scale;
  unsigned* err;            // the forward's error word or null: an e4m3 output that saturates (|x| * c8_inv_scale > 448, or NaN) ORs bit 1
                            //   into it -- a fixed output scale never clips silently (round 4)
  float alpha;              // out = res + alpha * act(v)
  const bf16_t* pos;        // [T][ldpos] added after the activation (positional table) or null
  long ldpos;
  const bf16_t* pos_lo;     //   the table's low half ("model.precision: high", gemm256.hip's three-segment walk only), or null
  int act;
  int glu;                  // weights row-interleaved in groups of 16 (a | gate); N_out = N / 2
  int out_f32;
  const float* ln_s;        // LayerNorm folded in front of the GEMM (gemm_stream.hip): s[n] = sum_k W'[n][k], W' = gamma o W,
  float ln_eps;             //   bias = b + W beta;  out = rstd_m * (acc - mean_m * s[n]) + bias[n];  statistics over the K columns
  float* stats_out;         // residual launches (gemm_stream.hip): per output row, per 256-column tile, (sum, sum of squares) of
                            //   the bf16-rounded outputs: stats_out[(row * 4 + tile) * 2 + {0, 1}], row = the C row index, N <= 1024
  const float* stats_in;    // with ln_s: take the row statistics from a producer's stats_out (row = A row m + stats_lead)
  int stats_nsl;            //   instead of summing the fragments in the kernel; stats_nsl = tiles per row (producer's N / 256)
  long stats_lead;
  void* trash;              // set by the launcher: >= 1 KiB scratch line for stores that must not land (gemm_stream.hip)
  const int* clip_T;        // [clips] valid frames of clip b (<= T), or null: every clip has T.  Rows t >= clip_T[b] are not stored --
                            //   batches of clips of different lengths behind the WavLM / mel front-ends (model.hip, "ragged")
#ifdef WFL_GEMM_STAMPS
  unsigned long long* stamps;   // diagnostic build: [blocks][8]
#endif
};

struct AttnArgs {
  const bf16_t* QK;   // frame rows, ld = ldqk: [q (d) | k (d)] per row, q pre-scaled by hd^-1/2 * log2(e)
  long ldqk;
  long lead;          // row(b,t) = lead + b*P + t
  const bf16_t* V;    // frame rows, ld = ldv, head h at column h*hd (normally the v columns of the packed q|k|v rows)
  long ldv;
  bf16_t* O;          // frame rows, ld = ldo, head h at column h*hd
  long ldo;
  int B, T, P, heads, d;
  const float* bias;      // optional additive score bias (WavLM gated rel-pos), see attention.hip
  const float* gate;      // [B][heads][T] per-query gate multiplying bias[h][q][k]
  const int* clip_T;      // [B] valid frames per clip (keys and queries >= clip_T[b] do not exist), or null: T for all
  bf16_t* O_lo;           // optional low half of the context (ld = ldo): o - bf16(o), for the split-precision GEMM that consumes it
  const bf16_t* QK_lo;    // "model.precision: high": the low halves of the packed q | k rows and of the v columns (same layout and leading
  const bf16_t* V_lo;     //   dimensions as QK / V).  Both set: scores and context run as three MFMA passes over split operands
                          //   (q_hi k_hi + q_lo k_hi + q_hi k_lo; p_hi v_hi + p_lo v_hi + p_hi v_lo, P split in registers); head_dim 64 / 256
  // fp8 context (BASELINE configs[4], round 3): instead of O, e4m3(context * o8_scale) goes to O8 (ldo8 bytes per row, same row mapping)
  // -- the out-projection's fp8 operand, with the fixed scale 1 / o8_scale (GemmArgs::a8_static).  head_dim 64 only.
  unsigned char* O8;
  unsigned char* O8_lo;   // round 4: the context as an e4m3 PAIR -- e4m3(16 (context * o8_scale - hi)) goes here (ld = ldo8), or null
  long ldo8;
  float o8_scale;
  unsigned* err;          // the forward's error word or null: a context element that saturates e4m3 (or is NaN) ORs bit 1 into it
};

// One BiLSTM layer's recurrence (lstm.hip)
struct LstmArgs {
  const float* gx; long ldgx;      // frame rows, fp32: col = dir*4H + 4*unit + gate
  const bf16_t* whh;               // [2][G][4U][H] bf16, slice rows = 4*u_local + gate
  const bf16_t* whh_lo;            // "model.precision: high": W_hh - bf16(W_hh), same layout; with out_lo it selects the split-precision
  bf16_t* out_lo;                  //   recurrence (h carried and exchanged as a bf16 pair, three MFMA passes; H <= 256) -- else null
  bf16_t* out; long ldo;           // frame rows: col = dir*H + unit
  long lead;
  int B, T, P, H, U, G;
  unsigned long long* hx;          // exchange granules: [2 dir][groups][2 halves: hi, lo][2 parity][H/8 blocks of 8 units][16 clips][4]
  unsigned* error;                 // the forward's error word (ORed into)
  int grp0;                        // first clip group of this launch
  int ngroups;                     // clip groups of the whole batch (exchange indexing)
  int ngroups_launch;              // clip groups of this launch
  unsigned long long* roll;        // roll-call granules: [2 dir][groups][64]
  const int* clip_T;               // [B] valid frames per clip or null: clip b runs steps 0 .. clip_T[b] - 1, its backward direction
                                   //   starts at its own last frame; the other steps of a shorter clip compute on and store nothing
#ifdef WFL_LSTM_STAMPS
  unsigned long long* stamps;      // diagnostic build (tools/micro/lstm_bench.hip): [steps 64..95][8] phase stamps of WG (0,0,0) wave 0
  int team_shift;                  // diagnostic build: the first team_shift team slots of the grid stay empty (moves the teams to other XCDs)
#endif
};

// ---- launch-argument structs shared by a kernel's translation unit and model.hip (round 4: each used to be declared twice)
// logmel.hip
struct LogmelArgs {
  const float* wav; long ldw;       // [B][ldw]
  const int* lens;                  // [B] valid samples per clip, or null (=L)
  int L;                            // samples present per row
  int B, n_samples, n_frames, n_mels;
  const float* Wc; const float* Ws; // [200][224] folded tables: row j-1 <-> n = j (1..200)
  const int* mel_lo; const int* mel_cnt; const float* mel_w; int mel_maxw;
  float* raw;                       // [B][n_frames][n_mels] log10 mel
  unsigned* clipmax;                // [B] ordered-uint max of raw (zeroed by the caller)
#ifdef WFL_LOGMEL_STAMPS
  unsigned long long* stamps;       // diagnostic build (tools/micro/logmel_bench.hip): [blocks][8] 100 MHz phase stamps
#endif
};

// precise.hip
struct PreciseFinishArgs {
  const float* acc; long ld_acc;          // [B * P rows][ld_acc]: the three passes' sum (rows b * P + t)
  int B, P, T, n_out;                     // output columns (GLU: half the accumulator's)
  int glu;                                // accumulator columns interleaved in groups of 16: (a | gate) -> a * sigmoid(gate)
  const float* bias;                      // [N of the accumulator] or null
  const float* clip_bias; const int* clip_idx; int clip_ld;
  int act; float alpha;
  const bf16_t* pos; long ldpos;          // [T][ldpos] added after the activation, or null
  const bf16_t* pos_lo;                   //   and the table's low half (same layout), or null
  const bf16_t* res; const bf16_t* res_lo; long ldres;     // residual rows (same row mapping as the output) or null
  bf16_t* out; bf16_t* out_lo; long ldc; long c_lead; int c_pitch;
  const int* clip_T;                      // ragged batches: rows t >= clip_T[b] are not stored
};

// head.hip
struct TagArgs {
  const float* logits; long ldl;     // [rows][C] fp32, rows = B*T compact
  int rows, C;
  float threshold; int o_id;
  int* ids;                          // argmax, or o_id when max prob < threshold
  int* argmax;                       // raw argmax (may be null)
  float* maxprob;
  // offsets (optional)
  const bf16_t* hid; long ldh; long lead; int P, T, d;   // frame rows of the offset head's hidden activation
  const bf16_t* hid_lo;              //   precision high: their low halves (same layout) or null
  const float* w2;                   // [2][d]
  const float* b2;                   // [2]
  float* offsets;                    // [rows][2]
  const unsigned* status_src;        // the forward's device-side error word -> *status_dst (both optional)
  int* status_dst;
  const int* clip_T; int Tmax;       // [clips] valid frames per clip (rows = clips x Tmax) or null: frames beyond a clip's own count are
                                     //   tagged "O" with probability 0 and zero offsets, whatever the logits buffer holds there
};

// wavlm.hip
struct Conv0Args {
  const float* wav; long ldw; int L;      // [B][ldw]
  const double* wstats;                   // [B][2] sum, sumsq of the waveform, or null (do_normalize off)
  const float* w;                         // [C][10]
  const float* bias;                      // [C] or null
  const float* gamma; const float* beta;  // [C]
  int B, T0, C;
  const int* lens;                        // [B] samples per clip or null (= L): statistics, the conv's input range and the frame count
                                          //   (len - 10) / 5 + 1 follow the clip's own length
  double* cstats;                         // [B][C][2] (group mode)
  float* cpart;                           // [B][time blocks][C][2] per-workgroup partial sums (group mode, pass 1)
  bf16_t* out; long lead; int P;          // frame rows [.., C]
  bf16_t* out_lo;                         // precision high: the rows' low halves (same layout) or null
};

// posconv.hip
struct PosConvArgs {
  const bf16_t* xg;               // [groups][R][64] regrouped rows (zero outside valid frames / channels)
  long R;                         // rows per group
  long lead;                      // row of (clip 0, frame 0)
  int B, P, T;                    // flat frame m = b * P + t, stored iff t < T
  int groups, cpg, taps;          // channels per group (<= 64, % 8 == 0), taps (even, <= 128)
  const bf16_t* w[16];            // per group [>= cpg rows][ldw]: k = tap * 64 + channel
  const float* bias[16];          // per group [>= cpg]
  long ldw;
  const bf16_t* res;              // x rows (ld), its low half (or null)
  const bf16_t* res_lo;
  bf16_t* out;                    // y rows (ld), low half (or null)
  bf16_t* out_lo;
  long ld;
  const int* clip_T;              // [B] valid frames per clip or null
};

// norm.hip
struct ZeroMulti {
  int n;
  char* buf[10];
  long ld_bytes[10];
  long lead[10];
  int P[10], T[10];
  long tail_rows[10];
  int B;
  unsigned* err_word;   // the forward's device-side error word, cleared here (first kernel of every forward); may be null
};

// erf GELU (nn.GELU() default / HF "gelu"):  gelu(x) = max(x, 0) - |x| * Phi(-|x|),  Phi(-t) = 0.5 * erfc(t / sqrt 2).
// log2 Phi(-t) is smooth and nearly quadratic, so Phi(-t) = exp2(q(t)) with q a degree-5 minimax fit on [0, 6] weighted
// by the error it causes in gelu (tools/fit_gelu.py): |gelu error| <= 6.4e-7 over all x in fp32 (the bf16 rounding of the
// stored value is ~4e-3 relative), q -> -inf beyond the fit range so large |x| need no clamp.  5 FMA + 1 v_exp + 3 VALU,
// against 2 transcendentals + 13 VALU for the Abramowitz-Stegun 7.1.26 form used before (9 us of a 27 us fc1 tile).
static __device__ __forceinline__ float gelu_erf(float x) {
  const float t = fabsf(x);
  float q = fmaf(-4.732937668e-04f, t, 7.084457669e-03f);
  q = fmaf(q, t, -5.182715505e-02f);
  q = fmaf(q, t, -4.599926770e-01f);
  q = fmaf(q, t, -1.150787711e+00f);
  q = fmaf(q, t, -1.000037670e+00f);
  return fmaxf(x, 0.f) - t * __builtin_amdgcn_exp2f(q);
}
static __device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int ACT>
static __device__ __forceinline__ float apply_act(float v) {
  if (ACT == WFL_ACT_GELU) return gelu_erf(v);
  if (ACT == WFL_ACT_RELU) return fmaxf(v, 0.0f);
  if (ACT == WFL_ACT_SIGMOID) return sigmoidf_(v);
  return v;
}

static __device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
static __device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// hipFuncSetAttribute and small scratch allocations are per DEVICE: a launcher remembers, per kernel instantiation, on which
// devices it has run (one process may hold models on several GPUs; normally one process drives one GPU).
struct WflOncePerDevice {
  unsigned mask = 0;
  bool need() {
    int d = 0;
    (void)hipGetDevice(&d);
    if ((mask >> (d & 31)) & 1u) return false;
    mask |= 1u << (d & 31);
    return true;
  }
};

// host-side launchers (one per .hip translation unit)
int wfl_launch_gemm(const GemmArgs& a, hipStream_t s);
// which kernel the last wfl_launch_gemm used (profiling labels): 1 gemm_stream<.,6>, 2 gemm256<.,6>, 3 gemm256<.,8>, 4 gemm_bf16 (128 tile), 5 gemm_stream<.,8>, 6 gemm_stream conv mode
extern int g_wfl_gemm_kernel_id;
bool wfl_gemm_mx_takes(const GemmArgs& a);       // gemm_mx.hip: fp8 x fp8 on the block-scaled MFMA (GemmArgs::a8 == 2, 3)
int wfl_launch_gemm_mx(const GemmArgs& a, hipStream_t s);
bool wfl_gemm256_tri_takes(const GemmArgs& a);  // gemm256.hip: a three-segment launch its slice-by-slice walk takes (any number of rows)
bool wfl_gemm_stream_takes(const GemmArgs& a);   // gemm_stream.hip: would the streaming (LayerNorm-folding) kernel take it
int wfl_launch_attention(const AttnArgs& a, hipStream_t s);
int wfl_launch_layernorm(const bf16_t* x, long ldx, bf16_t* y, long ldy, const float* g, const float* b, float eps,
                         long lead, int B, int P, int T, int C, hipStream_t s);
// Plain fill / copy kernels.  The forward never uses hipMemsetAsync / hipMemcpyAsync: a memset NODE of 64 bytes or more captured
// into a HIP graph writes garbage from its second replay on (ROCm 7.2, gfx950; tools/micro/graph_memset.py) -- the cause of the
// graph-replay divergence seen in round 1 (the per-clip log-mel maximum was "cleared" to garbage).
int wfl_launch_fill_i32(int* dst, long n, int value, hipStream_t s);
int wfl_launch_copy16(void* dst, const void* src, long bytes, hipStream_t s);   // bytes % 16 == 0, both 16-byte aligned
int wfl_launch_zero_halo(bf16_t* buf, long ld_bytes, long lead, int B, int P, int T, long tail_rows, hipStream_t s);
