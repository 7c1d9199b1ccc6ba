/* wfl_asr.h — C ABI of libwfl_asr_hip.so, the MI355X (gfx950) implementation of the WFL-ASR labeling hot path.
 *
 * The reference (usamireko/WFL-ASR) has no plugin / operator / FFI layer: its hot path sits behind plain Python
 * callables.  This header is therefore the build's own boundary; every entry point names the
 * reference interface it replaces so a maintainer can bind it from the reference's Python (ctypes stub in
 * INTEGRATION.md).  Conventions:
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers unless the name ends in `_host`;
 *   - the caller owns every buffer (PyTorch-ROCm allocations are fine: pass tensor.data_ptr());
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it and nothing here synchronises the device;
 *   - return value 0 = ok, negative = error (wfl_last_error() gives the text); no exceptions cross the ABI;
 *   - one model handle per process per GPU; a handle is not re-entrant.  A model lives on the HIP device that was current
 *     when wfl_finalize ran (wfl_device()); every later call must be made with that device current and with buffers and
 *     stream of that device (checked: a mismatch is an error, not a cross-device access).
 */
#ifndef WFL_ASR_H
#define WFL_ASR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfl_model wfl_model;

#define WFL_ENC_WHISPER 0
#define WFL_ENC_WAVLM 1
#define WFL_ENC_NONE 2     /* no encoder: the hidden states are a power mel spectrogram (model.py:82-91, 149-150) */

#define WFL_ABI_VERSION 2

/* Architecture = what /root/reference/model.py:54-146 reads from config.yaml plus the HF encoder config it
 * fetches by name (model.py:69-70, 74-80).  All int32, 64 slots, zero-initialise then fill. */
typedef struct wfl_arch {
  int32_t abi_version;       /* WFL_ABI_VERSION */
  int32_t encoder_type;      /* WFL_ENC_* */
  int32_t d_model;
  int32_t enc_layers;
  int32_t enc_heads;
  int32_t enc_ffn;
  int32_t n_mels;            /* whisper: 80 | 128 */
  int32_t max_positions;     /* whisper: encoder frames (1500); mel frames = 2x */
  /* head (model.py:96-142) */
  int32_t num_classes;       /* len(phonemes.txt) */
  int32_t o_id;              /* class index of "O" */
  int32_t num_languages;
  int32_t lang_emb_dim;
  int32_t enable_bilstm;
  int32_t bilstm_layers;
  int32_t n_conformer;
  int32_t conformer_heads;
  int32_t conformer_ff_expansion;
  int32_t conformer_kernel;
  int32_t enable_dilated;
  int32_t dilated_depth;
  int32_t dilated_kernel;
  /* wavlm (HF WavLMConfig) */
  int32_t wavlm_n_conv;              /* 7 */
  int32_t wavlm_conv_dim[8];
  int32_t wavlm_conv_kernel[8];
  int32_t wavlm_conv_stride[8];
  int32_t wavlm_group_norm;          /* 1: feat_extract_norm == "group" */
  int32_t wavlm_conv_bias;
  int32_t wavlm_stable_layer_norm;
  int32_t wavlm_pos_conv_kernel;
  int32_t wavlm_pos_conv_groups;
  int32_t wavlm_num_buckets;
  int32_t wavlm_max_distance;
  int32_t wavlm_do_normalize;
  int32_t fp8_weights;               /* 1: the Whisper encoder layers' q|k|v, out_proj, fc1, fc2 weights are kept as OCP e4m3 with one
                                        fp32 scale per output channel (BASELINE configs[4]); 0: bf16 */
  int32_t mel_hop;                   /* WFL_ENC_NONE: hop of the mel front-end = int(frame_duration * sample_rate) (model.py:88);
                                        d_model = n_mels = the hidden width (model.py:91); 160 and 320 are built */
  int32_t precision;                 /* 0: bf16 operands (default).  1 ("model.precision: high", round 3): every GEMM, the attention's two
                                        products and the BiLSTM recurrence (hidden size <= 256 per direction) run as three bf16 MFMA passes
                                        over split operands -- A_hi W_hi + A_hi W_lo + A_lo W_hi, summed in fp32 -- with every activation
                                        carried as a bf16 pair hi + lo: the reference's tag indices at ~2.7x the forward time; the
                                        workspace doubles (wfl_workspace_bytes) */
  int32_t fp8_activations;           /* fp8_weights models only ("model.activation_dtype", round 4): how the four GEMM inputs of every encoder
                                        layer are carried.
                                        0: bf16 -- the e4m3 weights are converted in registers, bf16 MFMA;
                                        3: e4m3 PAIRS hi + lo (eight significant bits, what a bf16 operand carries) on the block-scaled
                                           fp8 MFMA v_mfma_scale_f32_16x16x128_f8f6f4: lo rides in the same instruction with a 2^-4 block
                                           scale.  Both hold the reference's arithmetic on the fp8-rounded checkpoint (logits within
                                           0.13 / 0.022);
                                        2: ONE e4m3 value per activation on the same MFMA: the fastest form, and 5-9 % of the raw tag
                                           decisions then differ from that reference (three mantissa bits) -- an explicit opt-in;
                                        1: round 3's form of 2 on the non-scaled fp8 MFMA (kept for A/B runs). */
  int32_t reserved[6];
} wfl_arch;

const char* wfl_last_error(void);
int32_t wfl_abi_version(void);

/* Replaces BIOPhonemeTagger.__init__ (model.py:55-146): builds an empty model for `arch`. */
int32_t wfl_create(const wfl_arch* arch, wfl_model** out);
void wfl_destroy(wfl_model* m);

/* Replaces nn.Module.load_state_dict(strict=True) as called at /root/reference/infer.py:206-207: call once per
 * state-dict tensor with the reference's own key names (fp32 host data; `num_batches_tracked` is ignored). */
int32_t wfl_load_tensor(wfl_model* m, const char* name, const float* data_host, const int64_t* shape, int32_t ndim);

/* Strict key check + weight packing (q/k/v packing and scaling, BatchNorm fold into the k=31 conv, conv weights to
 * tap-major GEMM form, GLU row interleave, language-bias table, bf16 conversion, upload).  After this the model
 * is immutable.  Missing or unexpected keys are an error, as with strict loading. */
int32_t wfl_finalize(wfl_model* m);

/* HIP device ordinal the weights were uploaded to by wfl_finalize (-1 before). */
int32_t wfl_device(const wfl_model* m);

/* The language ids WFL_LANG_AVERAGE averages over: the reference loops over the ids listed in langs.txt
 * (/root/reference/infer.py:147-156, 266-276), which may be a subset of the embedding rows.  Default: every id
 * 0 .. num_languages-1.  ids_host: n host int32, each in [0, num_languages). */
int32_t wfl_set_average_languages(wfl_model* m, const int32_t* ids_host, int32_t n);

/* Output frames for L input samples per clip (Whisper: always max_positions; WavLM: conv arithmetic; WFL_ENC_NONE:
 * 1 + L / mel_hop, the centred STFT's frame count, 0 when L <= 200). */
int32_t wfl_num_frames(const wfl_model* m, int32_t L);
int64_t wfl_workspace_bytes(const wfl_model* m, int32_t B, int32_t L);

/* lang_mode for wfl_forward */
#define WFL_LANG_NONE 0     /* forward(lang_id=None): skip lang_proj (model.py:176) */
#define WFL_LANG_IDS 1      /* lang_id[B] given */
#define WFL_LANG_AVERAGE 2  /* infer.py:146-156 / 266-276: every language id, mean of logits and of offsets
                               (the encoder runs once; only the head depends on the language) */

/* Replaces BIOPhonemeTagger.forward (model.py:148-194) + decode_predictions (196-198) + the softmax/threshold
 * half of suppress_low_confidence (infer.py:86-96) for a batch of clips.
 *   wav        [B][ldw] fp32 16 kHz samples, L valid columns; lens (optional, [B] int32) marks shorter clips
 *   ids        [B][T] int32  argmax class, or o_id where max prob < threshold
 *   argmax     [B][T] int32  raw argmax                      (optional)
 *   maxprob    [B][T] fp32   max softmax probability
 *   offsets    [B][T][2] fp32 sigmoid sub-frame offsets
 *   logits     [B][T][C] fp32                                (optional)
 *   hidden     [B][T][d] fp32 encoder output                 (optional, parity tests; WFL_ENC_NONE: the mel power, d = n_mels)
 *   lens with WavLM / WFL_ENC_NONE (whose input the reference never pads): clip b is treated as lens[b] samples long through the whole
 *   forward -- its own waveform / GroupNorm statistics, conv frame counts, attention keys, positional-conv padding and backward-LSTM
 *   start -- and comes out bit for bit as if it were labelled alone; frames behind its own count are tagged o_id with probability 0.
 *   With Whisper, lens only says where a clip's samples end inside the 30 s window (the encoder always sees 1500 frames).
 *   status     [1] int32 (device, optional): 0, or a bit mask of device-side errors of THIS forward (bit 0: an
 *              inter-workgroup wait of the BiLSTM recurrence timed out; bit 1, fp8_activations only: an e4m3 activation saturated
 *              at its scale or was NaN -- the tags are invalid either way).  Written by the last kernel
 *              of the forward, so it can ride in the same D2H copy as the tags.
 */
int32_t wfl_forward(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L,
                    const int32_t* lang_id, int32_t lang_mode, float threshold, void* workspace,
                    int64_t workspace_bytes, int32_t* ids, int32_t* argmax, float* maxprob, float* offsets,
                    float* logits, float* hidden, int32_t* status, void* stream);

/* The two halves of wfl_forward, for callers that work on the encoder output in between -- the reference's training-time
 * validation forward pads / truncates hidden_states to max_label_len frames before the head (model.py:166-174).
 *   wfl_encode: model.py:149-161 (feature extractor + encoder) -> hidden [B][T][d] fp32, T = wfl_num_frames(L).
 *   wfl_head:   model.py:176-194 + the tag decision on hidden [B][T][d] fp32 with ANY T >= 1 (outputs as wfl_forward);
 *               workspace size from wfl_head_workspace_bytes(B, T). */
int32_t wfl_encode(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L, void* workspace,
                   int64_t workspace_bytes, float* hidden, void* stream);
int64_t wfl_head_workspace_bytes(const wfl_model* m, int32_t B, int32_t T);
int32_t wfl_head(wfl_model* m, const float* hidden, int32_t B, int32_t T, const int32_t* lang_id, int32_t lang_mode,
                 float threshold, void* workspace, int64_t workspace_bytes, int32_t* ids, int32_t* argmax, float* maxprob,
                 float* offsets, float* logits, int32_t* status, void* stream);

/* Synchronise `stream` and report the device-side error word of the last forward run on this workspace (same bits as
 * `status`: every kernel of a forward ORs into one word that the forward clears when it starts).  Optional; 0 = ok.
 * B, L: those of that forward (the word's place in the workspace follows the plan).  A workspace no forward has run on holds
 * whatever its allocator left there: zero it once, or call this only after a forward. */
int32_t wfl_check(wfl_model* m, void* workspace, int64_t workspace_bytes, int32_t B, int32_t L, void* stream);

/* ---- single stages, exported for unit parity tests and profiling ---- */

/* Replaces WhisperFeatureExtractor.__call__ at model.py:153-154 (HF feature_extraction_whisper.py:135-168).
 * out [B][n_mels][2*max_positions] fp32, the reference's layout. */
int32_t wfl_logmel(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L,
                   float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* bf16 MFMA GEMM with fused epilogue on frame rows (see csrc/common.h for the row layout):
 *   C[row(b,t)][n] = res + alpha * act( sum_k A[m][k] W[n][k] + bias[n] )      m = b*P + t, stored iff t < T
 * A, W, C(bf16 unless out_f32), res are bf16; K % 64 == 0, N % 128 == 0; k is split into taps of `cin` channels
 * `tap_stride` elements apart (cin >= K: one contiguous run).  glu: W rows interleaved (16 a | 16 gate). */
int32_t wfl_op_gemm(const void* A, int64_t lda, int32_t cin, int64_t tap_stride, const void* W, int32_t M, int32_t N,
                    int32_t K, int32_t n_valid, int32_t P, int32_t T, void* C, int64_t ldc, int64_t c_lead,
                    int32_t c_pitch, const float* bias, const void* res, int64_t ldres, float alpha, int32_t act,
                    int32_t glu, int32_t out_f32, void* stream);

/* The same Linear / Conv1d in the exact-label mode (`model.precision: high`; /root/reference/model.py:18-19, 26, 31-37, 131, 140 and
 * HF modeling_whisper.py:309-354, 391-407 computed to fp32-like accuracy so that infer.py:86-96 yields the reference's tag indices):
 * operands and results are bf16 PAIRS hi + lo (hi = bf16(x), lo = bf16(x - hi)), and
 *   out = res_hi + res_lo + alpha * act( A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T + bias ),   C = bf16(out), C_lo = bf16(out - C).
 * W3 is [N][3 K] bf16 with rows [W_hi | W_hi | W_lo]; A_lo has A_hi's leading dimension and tap layout (both inside one allocation);
 * K is the layer's K (a whole number of `cin`-channel taps), K % 32 == 0, N % 256 == 0 for the MFMA kernels' fast forms. */
int32_t wfl_op_gemm_split(const void* A_hi, const void* A_lo, int64_t lda, int32_t cin, int64_t tap_stride, const void* W3, int32_t M,
                          int32_t N, int32_t K, int32_t n_valid, int32_t P, int32_t T, void* C, void* C_lo, int64_t ldc,
                          int64_t c_lead, int32_t c_pitch, const float* bias, const void* res, const void* res_lo, int64_t ldres,
                          float alpha, int32_t act, int32_t glu, void* stream);

/* LayerNorm folded into the GEMM that consumes it (HF modeling_whisper.py:384-385, 401-402; model.py:18, 45):
 *   C = act( rstd_m * (A W'^T - mean_m * ln_s) + bias ),  W' = gamma o W (bf16), ln_s[n] = sum_k W'[n][k], bias = b + W beta;
 * mean / rstd are the statistics of row m of A over its K columns, computed inside the kernel.  M >= 2048, N % 256 == 0. */
int32_t wfl_op_gemm_ln(const void* A, int64_t lda, const void* W, int32_t M, int32_t N, int32_t K, int32_t n_valid, int32_t P,
                       int32_t T, void* C, int64_t ldc, int64_t c_lead, int32_t c_pitch, const float* bias, const float* ln_s,
                       float ln_eps, int32_t act, void* stream);

/* softmax(q k^T) v per (clip, head); QK rows hold [q | k] (q pre-scaled by hd^-1/2 * log2 e), V rows (ld = ldv, same row
 * indexing) hold v; normally all three are columns of one packed q|k|v projection output. */
int32_t wfl_op_attention(const void* QK, int64_t ldqk, int64_t lead, const void* V, int64_t ldv, void* O, int64_t ldo, int32_t B,
                         int32_t T, int32_t P, int32_t heads, int32_t d, void* stream);

/* fp8 x fp8 GEMM on the block-scaled MFMA (csrc/gemm_mx.hip, round 4; the encoder GEMMs of an fp8-weight model, HF modeling_whisper.py:
 * 309-354, 391-407 via /root/reference/model.py:155-156):
 *   C[row(b,t)][n] = res + alpha * act( w_scale[n] * sa(m) * sum_k a(m, k) W8[n][k] + bias[n] )
 * W8 [N][K] OCP e4m3 bytes; a(m, k) = A8[m][k] (e4m3 bytes, lda bytes per row), or A8[m][k] + A8_lo[m][k] / 16 when A8_lo is given
 * (an e4m3 PAIR: lo = e4m3(16 (x / sa - hi))); sa(m) = a_scale[m] or a_static when a_scale is null.  N % 256 == 0, K % 128 == 0, K >= 512.
 * res / res_lo / c_lo are bf16 rows with C's leading dimension and row mapping.
 * Output: bf16 rows C (+ the low half c_lo with a residual), or -- c8 given -- e4m3(out * c8_inv_scale) into c8 and, c8_lo given, the
 * remainder e4m3(16 (out * c8_inv_scale - hi)) into c8_lo (ldc8 bytes per row); an e4m3 store that saturates ORs 2 into *status. */
int32_t wfl_op_gemm_mx(const void* A8, const void* A8_lo, int64_t lda, const void* W8, const float* w_scale, const float* a_scale,
                       float a_static, int32_t M, int32_t N, int32_t K, int32_t P, int32_t T, void* C, int64_t ldc, int64_t c_lead,
                       int32_t c_pitch, const float* bias, const void* res, const void* res_lo, void* c_lo, float alpha, int32_t act,
                       void* c8, void* c8_lo, int64_t ldc8, float c8_inv_scale, int32_t* status, void* stream);

/* Frame rows -> e4m3 with one fp32 scale per row (row maximum -> 448), optionally LayerNorm(gamma, beta, eps) first (gamma null: plain
 * quantisation) and optionally as a PAIR (y8_lo given: e4m3(16 (x / scale - hi))): the producers of gemm_mx's frame operand
 * (csrc/norm.hip rows_fp8_kernel; HF modeling_whisper.py:384, 399).  x (+ x_lo when given) are bf16 rows, C % 8 == 0, C <= 2048. */
int32_t wfl_op_rows_fp8(const void* x, int64_t ldx, const void* x_lo, const float* gamma, const float* beta, float eps, int64_t lead,
                        int32_t B, int32_t P, int32_t T, int32_t C, void* y8, void* y8_lo, int64_t ldy8, float* scale, void* stream);

int32_t wfl_op_layernorm(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                         float eps, int64_t lead, int32_t B, int32_t P, int32_t T, int32_t C, void* stream);

/* softmax max-prob / argmax / threshold over fp32 logits rows (infer.py:86-96). */
int32_t wfl_op_tag_decide(const float* logits, int64_t ldl, int32_t rows, int32_t C, float threshold, int32_t o_id,
                          int32_t* ids, int32_t* argmax, float* maxprob, void* stream);

/* Per-kernel timing hook for bench.py's roofline: when enabled, wfl_forward brackets every GEMM launch with
 * hipEvents on `stream`.  wfl_gemm_profile_read synchronises on them and returns, per GEMM kernel variant
 * (key = act | glu<<2 | out_f32<<3 | res<<4 | kernel<<5 | ln_fold<<8 | stats<<10, one template instantiation = one
 * rocprof kernel name), the launch
 * count, summed milliseconds and summed algorithmic FLOPs (2 * valid_rows * n_valid * K) since the last reset. */
int32_t wfl_gemm_profile_enable(wfl_model* m, int32_t on);
int32_t wfl_gemm_profile_read(wfl_model* m, int32_t max_variants, int32_t* keys, int64_t* launches, double* total_ms,
                              double* total_flops, int32_t* n_variants, int32_t reset);

/* ---- host-side label logic (no device access): thresholded ids + offsets of one clip -> segments -> .lab text.
 * Replaces the per-frame Python of /root/reference/utils.py:10-81, 148-186 and the scipy median filter call at
 * /root/reference/infer.py:170-171, 298-299; bit-exact with them (times are doubles computed in the same order).
 *   kind[c] : 0 the "O" tag, 1 "B-x", 2 "I-x", 3 anything else (ignored);  phon[c] : phoneme index of tag c (-1 for "O")
 *   merge mode : 0 none, 1 right, 2 left, 3 previous (config postprocess.merge_segments)
 * wfl_host_decode_bio returns the number of segments (-3 when max_segments was too small, -4 when a run closes at a frame
 * beyond the n_off offsets rows, where the reference raises IndexError), wfl_host_merge_segments the
 * new count (in place), wfl_host_format_lab the number of bytes the text needs (written only if it fits in cap). */
int32_t wfl_host_median_filter(const int32_t* ids, int32_t n, int32_t size, int32_t* out);
int32_t wfl_host_decode_bio(const int32_t* ids, int32_t T, const float* offsets, int32_t n_off, const int32_t* kind,
                            const int32_t* phon, int32_t n_labels, double frame_duration, double* seg_start,
                            double* seg_end, int32_t* seg_ph, int32_t max_segments);
int32_t wfl_host_merge_segments(double* start, double* end, int32_t* ph, int32_t n, int32_t mode);
int64_t wfl_host_format_lab(const double* start, const double* end, const int32_t* ph, int32_t n,
                            const char* const* names, int32_t n_names, char* out, int64_t cap);

/* ---- audio ingest on the host (replaces soundfile.read + the float64 peak normalisation of /root/reference/infer.py:217-218,
 * 234-235 for 16-bit/24-bit/32-bit/float WAV files with 1 or 2 channels): decode, mono mix, audio / (max|audio| + 1e-8) in
 * float64, stored as float32 -- bit-identical to wfl-asr_amd/audio.py.  status: 0 ok, 1 unsupported encoding, 2 more than two
 * channels, 3 longer than cap samples (n_samples = the length), 4 cannot open.  The sample rate is reported, not converted.
 * wfl_host_load_wavs fills rows out + i * ld of a batch buffer with `threads` worker threads. */
int32_t wfl_host_load_wav(const char* path, float* out, int64_t cap, int32_t* n_samples, int32_t* sample_rate);
int32_t wfl_host_load_wavs(const char* const* paths, int32_t n, float* out, int64_t ld, int64_t cap, int32_t* n_samples,
                           int32_t* sample_rates, int32_t* status, int32_t threads);

/* The general ingest path of ONE file (replaces /root/reference/infer.py:217-220 soundfile.read + torchaudio resample, 234-235
 * whole-clip peak normalisation, 237-244 + 19-28 split_audio for clips longer than chunk_samples, 114-115 per-chunk
 * re-normalisation): rows out + r * ld receive the float32 work items, lens[r] their lengths.  Resampling = torchaudio's sinc /
 * Hann algorithm restated (parity with torchaudio UNPINNED: library absent).  Status as wfl_host_load_wav, 5 = more than max_rows
 * chunks (n_rows = the count needed, nothing written). */
int32_t wfl_host_load_wav_chunks(const char* path, int32_t target_sr, int64_t chunk_samples, float* out, int64_t ld, int32_t max_rows,
                                 int32_t* n_rows, int32_t* lens, int32_t* sample_rate);

/* ---- audio ingest on the GPU (round 3): a file that is not at the model's rate no longer costs host cores.
 * wfl_host_read_pcm16 copies the 16-bit PCM samples of `n` WAV files as they are (interleaved, 1 or 2 channels) into rows
 * out + i * ld of a (pinned) int16 buffer -- status 0 ok, 1 not 16-bit PCM, 2 more than two channels, 3 more than cap_samples values,
 * 4 cannot open.  wfl_resample_pcm16 (device pointers, one HIP stream) turns B such rows of ONE sample rate into float32 rows ready
 * for wfl_forward: decode ((l + r) / 2 for two channels), band-limited sinc resampling to new_sr -- torchaudio.functional.resample's
 * published algorithm in float64 (/root/reference/infer.py:217-220; parity with torchaudio UNPINNED: library absent), bit-identical to
 * wfl_host_load_wav_chunks -- and whole-clip peak normalisation x / (max|x| + 1e-8) in float64 (infer.py:234-235).  Row b of `out`
 * receives min(ceil(len_b * new_sr / orig_sr), out_cap) samples followed by zeros up to out_cap; clips that come out longer than
 * out_cap (= 30 s: they would be cut into chunks) belong to wfl_host_load_wav_chunks.  orig_sr == new_sr is taken (round 4): no
 * resampling, as torchaudio returns such a waveform as it is -- decode and normalisation only. */
int32_t wfl_host_read_pcm16(const char* const* paths, int32_t n, int16_t* out, int64_t ld, int64_t cap_samples, int32_t* n_frames,
                            int32_t* channels, int32_t* sample_rates, int32_t* status, int32_t threads);
int64_t wfl_resample_workspace_bytes(int32_t B, int32_t out_cap);
int32_t wfl_resample_pcm16(const int16_t* pcm, int64_t ld_in, const int32_t* n_in, const int32_t* channels, int32_t B, int32_t orig_sr,
                           int32_t new_sr, float* out, int64_t ld_out, int32_t out_cap, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* ---- boundary-snapping features on the GPU (round 3; /root/reference/correct_label.py:15-24, the step the
 * reference's notebook runs after inference): for B clips of up to L samples at 16 kHz (device pointers, one HIP stream),
 *   flux[b][t]    spectral flux of |librosa.stft(y, n_fft=512, hop_length=160)| (flux[b][0] = 0),
 *   mfcc[b][c][t] librosa.feature.mfcc(y, sr=16000, n_mfcc=13, hop_length=160): n_fft 2048, 128 Slaney mel bands, dB with top_db 80,
 *                 orthonormal DCT-II,
 * t < F = 1 + L / 160.  mel_w [128][1025] and dct [13][128] are the caller's (wfl-asr_amd/correct_label.py builds them; its numpy
 * restatement of the same features is what tests hold this to).  Restated from librosa's documented definitions: parity with librosa
 * itself is UNPINNED (library absent, the reference holds no fixtures). */
int64_t wfl_boundary_workspace_bytes(int32_t B, int32_t L);
int32_t wfl_boundary_features(const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L, const float* mel_w,
                              const float* dct, float* flux, float* mfcc, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
