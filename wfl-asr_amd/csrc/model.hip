// Model handle, weight packing and the forward schedule behind the C ABI of include/wfl_asr.h.
//
// This file is the native replacement for BIOPhonemeTagger (/root/reference/model.py:54-201): it owns the packed
// bf16 weights in HBM, lays the caller's workspace out as frame-row buffers (common.h) and enqueues the kernel
// sequence of one batched forward on the caller's HIP stream.  No torch types, no allocation and no
// synchronisation inside wfl_forward (graph-capturable).
#include "common.h"
#include "../../include/wfl_asr.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

// ---- launchers defined in the kernel translation units
int wfl_launch_logmel(const LogmelArgs& a, bf16_t* out, long ldo, long lead, int P, float* ref_out, hipStream_t s, bf16_t* out_lo = nullptr);
int wfl_launch_precise_finish(const PreciseFinishArgs& a, hipStream_t s);
int wfl_launch_melpower(const LogmelArgs& a, int hop, bf16_t* out, long ldo, long lead, int P, int split, int shift, hipStream_t s);

int wfl_launch_tag_decide(const TagArgs& a, hipStream_t s);
int wfl_launch_f32_to_rows(const float* in, bf16_t* x, long ldx, long lead, int B, int P, int T, int C, hipStream_t s, int split,
                           int shift, bf16_t* x_lo = nullptr);

int wfl_launch_lstm(LstmArgs a, void* exchange, hipStream_t s);

int wfl_launch_wav_stats(const float* wav, long ldw, int B, int L, double* stats, hipStream_t s, const int* lens = nullptr);
int wfl_launch_conv0(const Conv0Args& a, int group_norm, hipStream_t s);
int wfl_launch_clip_frames(const int* lens, int B, int L, int n, const int* kernel, const int* stride, int min_len, int* out, hipStream_t s);
int wfl_launch_posconv(const PosConvArgs& a, hipStream_t s);
int wfl_launch_regroup(const bf16_t* x, int d, int groups, int cpg, long R, long lead, int B, int P, int T, bf16_t* xg, hipStream_t s,
                       const int* clip_T = nullptr);
int wfl_launch_relpos_gate(const bf16_t* x, long ldx, long lead, int B, int P, int T, int heads, int hd, const float* w8,
                           const float* b8, const float* cst, float* gate, hipStream_t s, const bf16_t* x_lo = nullptr);
int wfl_launch_relpos_table(const float* rel_emb, const int* bucket_of_delta, int max_t, int heads, int T, float* table, hipStream_t s);
int wfl_launch_layernorm_act(const bf16_t* x, long ldx, bf16_t* y, long ldy, const float* g, const float* b, float eps, long lead,
                             int B, int P, int T, int C, int gelu, hipStream_t s, const bf16_t* x_lo = nullptr, bf16_t* y_lo = nullptr,
                             int n_div = 0, const int* clip_T = nullptr);
int wfl_lstm_units_per_wg(int H);
long wfl_lstm_exchange_bytes(int H, int B);
bool wfl_lstm_split_precision_supported(int H);
int wfl_launch_rows_fp8(const bf16_t* x, long ldx, const bf16_t* x_lo, const float* g, const float* b, float eps, long lead, int B, int P,
                        int T, int C, unsigned char* y8, long ldy8, float* scale, hipStream_t s, unsigned char* y8_lo = nullptr);          // norm.hip
int wfl_launch_axpy(float* dst, const float* src, long n, float alpha, int init, hipStream_t s);
int wfl_launch_zero_halo_multi(const ZeroMulti& z, hipStream_t s);
int wfl_launch_rows_to_f32(const bf16_t* x, long ldx, long lead, int B, int P, int T, int C, float* out, hipStream_t s, int split,
                           int shift, const bf16_t* x_lo = nullptr);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                            \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) return fail(-10, std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

static inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }
#define WAVLM_MAX_T 4096   // frames per clip the relative-position bucket table covers (81 s of audio)

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  long numel() const { long n = 1; for (auto s : shape) n *= s; return n; }
};

// A packed Linear / Conv-as-GEMM operand
struct Lin {
  bf16_t* W = nullptr;   // [N][K] padded
  float* bias = nullptr; // [N] padded (may be null)
  int N = 0, K = 0, n_valid = 0;
  float* ln_s = nullptr; // set on a LayerNorm-folded operand: s[n] = sum_k W'[n][k] (gemm_stream.hip), bias = b + W beta
  float* w8 = nullptr;   // set on an fp8 operand: W holds OCP e4m3 bytes [N][K], w8[n] = the row's scale (GemmArgs::w8_scale)
  bf16_t* W_lo = nullptr; // "model.precision: high": bf16(w - bf16(w)), same layout as W (precise.hip)
  bf16_t* W3 = nullptr;   //   and the one-launch form's operand [N][3 K]: rows [W_hi | W_hi | W_lo] (GemmArgs::tap_wrap)
};
struct LNp { float* g = nullptr; float* b = nullptr; };

struct EncLayer { LNp ln1, ln2; Lin qkv, out, fc1, fc2, qkv_ln, fc1_ln; };   // *_ln: the LayerNorm in front folded in
struct WavlmLayer { LNp ln1, ln2; Lin qkv, out, fc1, fc2, fc1_ln; float *w8 = nullptr, *b8 = nullptr, *cst = nullptr; };
struct ConfLayer {
  LNp ff1_ln, ff2_ln, ln1, ln2;
  Lin ff1_a, ff1_b, ff2_a, ff2_b, qkv, out, pw1, conv, pw2, ff1_a_ln, ff2_a_ln;
};

// GEMM launches are timed per kernel variant (template instantiation), keyed act | glu<<2 | out_f32<<3 | res<<4 | kernel id<<5
// | LayerNorm-fold mode<<8 | statistics-emitting epilogue<<10
struct GemmProf {
  long launches[2048] = {0};
  double flops[2048] = {0};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<int> key;
};

struct wfl_model {
  wfl_arch a{};
  bool finalized = false;
  int device = -1;                   // HIP device the weights live on (current device at wfl_finalize)
  std::vector<int> avg_langs;        // WFL_LANG_AVERAGE: the ids averaged over (default: all)
  std::map<std::string, HostTensor> host;
  std::vector<void*> dev_allocs;
  // geometry
  int halo = 16;
  // `encoder_type: none` (WFL_ENC_NONE): the hidden width dv = n_mels (80) is no multiple of the 64 columns / supported head sizes
  // the kernels are built for, so every d-wide tensor of the head lives in d >= dv columns (a.d_model is rewritten to d at
  // wfl_create), valid channel c in column c + (c >= dv/2 ? d/2 - dv/2 : 0) -- the two halves a BiLSTM writes -- and zeros elsewhere;
  // pad_head_state() moves the checkpoint's tensors into that layout, LayerNorm divides by dv.  dv == d for every other model.
  int dv = 0;
  // Conformer attention: head size the attention kernels are built for (32 .. 640) at or above d / conformer_heads, and the width
  // heads * that of its q | k | v and context rows.  Whisper-tiny's 192 (384 / 2 heads) runs as 256: the q | k | v projection writes
  // zero columns behind every head (QKp / ATTp buffers of the plan), out_proj skips them.  Equal to d for every BASELINE config.
  int conf_hd = 0, conf_da = 0;
  int pad_split() const { return dv == a.d_model ? a.d_model : dv / 2; }
  int pad_shift() const { return dv == a.d_model ? 0 : a.d_model / 2 - dv / 2; }
  // whisper front-end tables
  float *Wc = nullptr, *Ws = nullptr, *mel_w = nullptr;
  int *mel_lo = nullptr, *mel_cnt = nullptr;
  int mel_maxw = 0;
  Lin conv1, conv2;
  bf16_t* pos = nullptr;
  bf16_t* pos_lo = nullptr;     // precision high: what the table's bf16 rounding left behind
  std::vector<EncLayer> enc;
  LNp enc_ln;
  // wavlm
  float *conv0_w = nullptr, *conv0_b = nullptr;
  LNp conv0_norm;
  std::vector<Lin> fconv;            // feature-encoder layers 1..n-1 as GEMMs
  std::vector<LNp> fconv_ln;         // "layer" checkpoints: LayerNorm after every conv
  LNp fp_ln;
  Lin fp_proj;
  std::vector<Lin> posconv;          // one GEMM per group over [rows][64]-regrouped channels
  LNp wenc_ln;
  std::vector<WavlmLayer> wl;
  float* rel_emb = nullptr;          // [num_buckets][heads]
  int* bucket_of_delta = nullptr;    // [2*WAVLM_MAX_T - 1]
  // head
  Lin lang;
  float* lang_table = nullptr;    // [num_languages][d]
  std::vector<Lin> lstm_in;       // per layer: both directions' input projection, rows [dir][unit][gate]
  std::vector<bf16_t*> lstm_whh;  // per layer: [2][G][4U][H]
  std::vector<bf16_t*> lstm_whh_lo;   // precision high: the same slices' low halves (null entries where the recurrence stays bf16)
  int lstm_U = 0;
  std::vector<ConfLayer> conf;
  std::vector<Lin> dil;
  Lin cls, off1;
  Lin cls_lo, cls_hi2;             // split-precision classifier: W - bf16(W), and [bf16(W) | bf16(W)] for the [hi | lo] input taps
  Lin cls3;                        //   and the one-launch form [W_hi | W_hi | W_lo] against the tap segments [h_hi | h_lo | h_hi] (GemmArgs::tap_wrap)
  float *off_w2 = nullptr, *off_b2 = nullptr;
  // profiling
  bool prof_on = false;
  GemmProf prof;
  size_t prof_used = 0;
};

const char* wfl_last_error(void) { return g_err.c_str(); }
int32_t wfl_abi_version(void) { return WFL_ABI_VERSION; }

int32_t wfl_create(const wfl_arch* arch, wfl_model** out) {
  if (!arch || !out) return fail(-1, "wfl_create: null argument");
  if (arch->abi_version != WFL_ABI_VERSION) return fail(-1, "wfl_create: ABI version mismatch");
  const wfl_arch& a = *arch;
  if (a.encoder_type != WFL_ENC_WHISPER && a.encoder_type != WFL_ENC_WAVLM && a.encoder_type != WFL_ENC_NONE)
    return fail(-1, "unknown encoder_type");
  const bool none = a.encoder_type == WFL_ENC_NONE;
  int d_pad = a.d_model;
  if (none) {
    // model.py:82-91: hidden_size = n_mels; the head runs at that width
    if (a.n_mels <= 0 || a.d_model != a.n_mels) return fail(-1, "encoder_type none: d_model must equal n_mels");
    if (a.mel_hop != 160 && a.mel_hop != 320) return fail(-1, "encoder_type none: the mel front-end is built for hop 160 and 320 (frame_duration 0.01 / 0.02 s at 16 kHz)");
    if (a.d_model % 2 && a.enable_bilstm) return fail(-1, "encoder_type none: odd n_mels with a BiLSTM");
    if (a.n_conformer > 0) {
      if (a.conformer_heads <= 0 || a.d_model % a.conformer_heads) return fail(-1, "bad conformer_heads");
      d_pad = 0;
      for (int hd : {32, 64, 128, 256}) {
        const int dp = hd * a.conformer_heads;
        if (hd >= a.d_model / a.conformer_heads && dp % 64 == 0) { d_pad = dp; break; }
      }
      if (!d_pad) return fail(-1, "encoder_type none: n_mels / conformer_heads above 256 is not supported");
    } else {
      d_pad = (int)round_up(a.d_model, 64);
    }
  } else {
    if (a.d_model <= 0 || a.d_model % 64) return fail(-1, "d_model must be a positive multiple of 64");
    if (a.enc_heads <= 0 || a.d_model % a.enc_heads) return fail(-1, "enc_heads must divide d_model");
  }
  if (!none && a.enc_layers <= 0) return fail(-1, "enc_layers must be positive");
  if (a.num_classes <= 0 || a.o_id < 0 || a.o_id >= a.num_classes) return fail(-1, "bad num_classes / o_id");
  if (a.n_conformer > 0 && (a.conformer_heads <= 0 || a.d_model % a.conformer_heads)) return fail(-1, "bad conformer_heads");
  if (a.n_conformer > 0 && a.conformer_kernel % 2 == 0) return fail(-1, "even conformer_kernel_size is not supported");
  int conf_hd = 0;
  if (a.n_conformer > 0) {
    const int hd = d_pad / a.conformer_heads;
    for (int v : {32, 64, 128, 256, 384, 512, 640})
      if (!conf_hd && v >= hd) conf_hd = v;
    if (!conf_hd) return fail(-1, "conformer head size " + std::to_string(hd) + " (d_model / conformer_heads) is above 640");
  }
  if (a.enable_dilated && a.dilated_kernel % 2 == 0) return fail(-1, "even dilated_conv_kernel is not supported");
  if (a.fp8_weights && (a.encoder_type != WFL_ENC_WHISPER || a.d_model % 256 || a.enc_ffn % 256))
    return fail(-1, "fp8_weights: Whisper encoders with d_model and ffn multiples of 256 only");
  if (a.fp8_activations && !a.fp8_weights) return fail(-1, "fp8_activations needs fp8_weights");
  if (a.fp8_activations < 0 || a.fp8_activations > 3) return fail(-1, "fp8_activations: 0 (bf16), 1 / 2 (e4m3) or 3 (e4m3 pairs)");
  if (a.precision && a.fp8_weights) return fail(-1, "precision = 1 (three-pass bf16 pairs) and fp8_weights contradict each other");
  wfl_model* m = new wfl_model();
  m->a = a;
  m->dv = a.d_model;
  m->a.d_model = d_pad;
  m->conf_hd = conf_hd;
  m->conf_da = conf_hd * (a.n_conformer > 0 ? a.conformer_heads : 0);
  int pad = 1;
  if (a.n_conformer > 0) pad = std::max(pad, a.conformer_kernel / 2);
  if (a.enable_dilated)
    for (int i = 0; i < a.dilated_depth; ++i) pad = std::max(pad, (1 << i) * (a.dilated_kernel - 1) / 2);
  if (a.encoder_type == WFL_ENC_WAVLM) {
    if (a.wavlm_n_conv < 2 || a.wavlm_n_conv > 8 || a.wavlm_conv_kernel[0] != 10 || a.wavlm_conv_stride[0] != 5) {
      delete m;
      return fail(-1, "WavLM feature encoder: layer 0 must be k=10 s=5 (all released checkpoints)");
    }
    for (int i = 1; i < a.wavlm_n_conv; ++i)
      if (a.wavlm_conv_stride[i] != 2 || a.wavlm_conv_dim[i] != a.wavlm_conv_dim[0] || a.wavlm_conv_dim[i] % 8 || a.wavlm_conv_dim[i] > 512) {
        delete m;
        return fail(-1, "WavLM feature encoder: layers 1.. must be stride 2 with one common width <= 512");
      }
    if (a.wavlm_pos_conv_kernel % 2 || a.d_model % a.wavlm_pos_conv_groups || a.d_model / a.wavlm_pos_conv_groups > 64 ||
        (a.d_model / a.wavlm_pos_conv_groups) % 8) {
      delete m;
      return fail(-1, "WavLM positional conv: even kernel and d_model/groups a multiple of 8, <= 64");
    }
    pad = std::max(pad, a.wavlm_pos_conv_kernel / 2);
  }
  m->halo = (int)round_up(pad + 1, 8);
  *out = m;
  return 0;
}

void wfl_destroy(wfl_model* m) {
  if (!m) return;
  for (void* p : m->dev_allocs) (void)hipFree(p);
  for (auto& e : m->prof.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete m;
}

int32_t wfl_load_tensor(wfl_model* m, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!m || !name || (!data && ndim > 0)) return fail(-1, "wfl_load_tensor: null argument");
  if (m->finalized) return fail(-1, "wfl_load_tensor: model already finalized");
  HostTensor t;
  long n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
  if (data) t.data.assign(data, data + n);
  m->host[name] = std::move(t);
  return 0;
}

// ------------------------------------------------------------------------------------------------ packing helpers
namespace {

struct Packer {
  wfl_model* m;
  std::set<std::string> used;
  std::string err;

  const HostTensor* get(const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = m->host.find(name);
    if (it == m->host.end()) { if (err.empty()) err = "missing key in state_dict: " + name; return nullptr; }
    std::vector<int64_t> want(shape);
    if (it->second.shape != want) {
      if (err.empty()) {
        err = "size mismatch for " + name + ": got [";
        for (auto s : it->second.shape) err += std::to_string(s) + ",";
        err += "] expected [";
        for (auto s : want) err += std::to_string(s) + ",";
        err += "]";
      }
      return nullptr;
    }
    used.insert(name);
    return &it->second;
  }

  template <class T>
  T* upload(const std::vector<T>& v) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(v.size() * sizeof(T), 16)) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    m->dev_allocs.push_back(d);
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
      if (err.empty()) err = "hipMemcpy failed";
      return nullptr;
    }
    return (T*)d;
  }

  // rows[n][k] fp32 (n_valid x k_valid) -> padded bf16 [N][K] + fp32 bias [N]
  Lin pack(const std::vector<float>& rows, int n_valid, int k_valid, const std::vector<float>* bias) {
    Lin L;
    L.n_valid = n_valid;
    L.N = (int)round_up(n_valid, 128);
    L.K = (int)round_up(k_valid, 64);
    std::vector<uint16_t> w((size_t)L.N * L.K, 0);
    for (int n = 0; n < n_valid; ++n)
      for (int k = 0; k < k_valid; ++k) w[(size_t)n * L.K + k] = f32_to_bf16_bits(rows[(size_t)n * k_valid + k]);
    L.W = (bf16_t*)upload(w);
    if (m->a.precision) {                      // the weights' low halves for the split-precision passes (precise.hip)
      std::vector<uint16_t> wl((size_t)L.N * L.K, 0);
      for (int n = 0; n < n_valid; ++n)
        for (int k = 0; k < k_valid; ++k) {
          const float x = rows[(size_t)n * k_valid + k];
          const uint32_t u = (uint32_t)w[(size_t)n * L.K + k] << 16;
          float xh;
          memcpy(&xh, &u, 4);
          wl[(size_t)n * L.K + k] = f32_to_bf16_bits(x - xh);
        }
      L.W_lo = (bf16_t*)upload(wl);
      std::vector<uint16_t> w3((size_t)L.N * 3 * L.K);
      for (int n = 0; n < L.N; ++n) {
        memcpy(&w3[((size_t)n * 3 + 0) * L.K], &w[(size_t)n * L.K], (size_t)L.K * 2);
        memcpy(&w3[((size_t)n * 3 + 1) * L.K], &w[(size_t)n * L.K], (size_t)L.K * 2);
        memcpy(&w3[((size_t)n * 3 + 2) * L.K], &wl[(size_t)n * L.K], (size_t)L.K * 2);
      }
      L.W3 = (bf16_t*)upload(w3);
    }
    std::vector<float> b((size_t)L.N, 0.f);
    if (bias) for (int n = 0; n < n_valid; ++n) b[n] = (*bias)[n];
    L.bias = upload(b);
    return L;
  }

  // OCP e4m3 (fn) encode, round to nearest even, saturating at 448: nearest entry of the 127 non-negative code points
  static uint8_t e4m3_encode(float x) {
    static float tab[127];
    static bool init = false;
    if (!init) {
      for (int b = 0; b < 127; ++b) {
        const int e = b >> 3, mnt = b & 7;
        tab[b] = e == 0 ? (float)mnt / 8.0f * 0.015625f : (1.0f + (float)mnt / 8.0f) * std::ldexp(1.0f, e - 7);
      }
      init = true;
    }
    const uint8_t sign = std::signbit(x) ? 0x80 : 0;
    const float a = std::fabs(x);
    if (!(a == a)) return 0x7F;
    if (a >= 448.0f) return sign | 126;
    int lo = 0, hi = 126;                       // tab[lo] <= a < tab[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tab[mid] <= a) lo = mid; else hi = mid; }
    const float dl = a - tab[lo], dh = tab[hi] - a;
    const int pick = dl < dh ? lo : (dh < dl ? hi : ((lo & 1) ? hi : lo));
    return sign | (uint8_t)pick;
  }
  static float e4m3_decode(uint8_t b) {
    const int e = (b >> 3) & 15, mnt = b & 7;
    const float v = e == 0 ? (float)mnt / 8.0f * 0.015625f : (1.0f + (float)mnt / 8.0f) * std::ldexp(1.0f, e - 7);
    return (b & 0x80) ? -v : v;
  }

  // rows[n][k] fp32 -> e4m3 bytes [N][K] (padded) + per-row fp32 scale (row max -> 448) + fp32 bias [N]
  Lin pack8(const std::vector<float>& rows, int n_valid, int k_valid, const std::vector<float>* bias) {
    Lin L;
    L.n_valid = n_valid;
    L.N = (int)round_up(n_valid, 256);
    L.K = (int)round_up(k_valid, 64);
    std::vector<uint8_t> w((size_t)L.N * L.K, 0);
    std::vector<float> sc((size_t)L.N, 1.0f);
    for (int n = 0; n < n_valid; ++n) {
      float mx = 0.f;
      for (int k = 0; k < k_valid; ++k) mx = std::max(mx, std::fabs(rows[(size_t)n * k_valid + k]));
      const float scale = mx > 0.f ? mx / 448.0f : 1.0f;
      sc[n] = scale;
      for (int k = 0; k < k_valid; ++k) w[(size_t)n * L.K + k] = e4m3_encode(rows[(size_t)n * k_valid + k] / scale);
    }
    L.W = (bf16_t*)upload(w);
    L.w8 = upload(sc);
    std::vector<float> b((size_t)L.N, 0.f);
    if (bias) for (int n = 0; n < n_valid; ++n) b[n] = (*bias)[n];
    L.bias = upload(b);
    return L;
  }
  Lin linear8(const std::string& p, int out_f, int in_f) {
    const HostTensor* w = get(p + ".weight", {out_f, in_f});
    const HostTensor* b = get(p + ".bias", {out_f});
    if (!w || !b) return Lin();
    return pack8(w->data, out_f, in_f, &b->data);
  }

  // LayerNorm(gamma, beta) folded into the Linear that consumes it (gemm_stream.hip): W' = gamma o W rounded to bf16,
  // s[n] = sum_k W'[n][k] over the ROUNDED weights (what the MFMA multiplies), bias' = bias + W beta.
  Lin pack_ln(const std::vector<float>& rows, int n_valid, int k_valid, const std::vector<float>* bias, const std::string& ln_name) {
    const HostTensor* g = get(ln_name + ".weight", {k_valid});
    const HostTensor* b = get(ln_name + ".bias", {k_valid});
    if (!g || !b) return Lin();
    std::vector<float> wf((size_t)n_valid * k_valid), bf(n_valid), sv;
    for (int n = 0; n < n_valid; ++n) {
      double acc = bias ? (*bias)[n] : 0.0;
      for (int k = 0; k < k_valid; ++k) {
        const float w = rows[(size_t)n * k_valid + k];
        wf[(size_t)n * k_valid + k] = w * g->data[k];
        acc += (double)w * b->data[k];
      }
      bf[n] = (float)acc;
    }
    Lin L = pack(wf, n_valid, k_valid, &bf);
    sv.assign((size_t)L.N, 0.f);
    for (int n = 0; n < n_valid; ++n) {
      double acc = 0.0;
      for (int k = 0; k < k_valid; ++k) {
        const uint32_t u = (uint32_t)f32_to_bf16_bits(wf[(size_t)n * k_valid + k]) << 16;
        float r;
        memcpy(&r, &u, 4);
        acc += r;
      }
      sv[n] = (float)acc;
    }
    L.ln_s = upload(sv);
    return L;
  }
  Lin linear_ln(const std::string& p, int out_f, int in_f, const std::string& ln_name) {
    const HostTensor* w = get(p + ".weight", {out_f, in_f});
    const HostTensor* b = get(p + ".bias", {out_f});
    if (!w || !b) return Lin();
    return pack_ln(w->data, out_f, in_f, &b->data, ln_name);
  }

  Lin linear(const std::string& p, int out_f, int in_f, bool has_bias = true) {
    const HostTensor* w = get(p + ".weight", {out_f, in_f});
    const HostTensor* b = has_bias ? get(p + ".bias", {out_f}) : nullptr;
    if (!w || (has_bias && !b)) return Lin();
    return pack(w->data, out_f, in_f, has_bias ? &b->data : nullptr);
  }

  // Conv1d weight [cout][cin][k] -> tap-major rows [cout][k*cin]; optional per-output-channel scale/shift (BN fold)
  Lin conv(const std::string& p, int cout, int cin, int k, const std::vector<float>* scale = nullptr,
           const std::vector<float>* shift = nullptr, bool has_bias = true) {
    const HostTensor* w = get(p + ".weight", {cout, cin, k});
    const HostTensor* b = has_bias ? get(p + ".bias", {cout}) : nullptr;
    if (!w || (has_bias && !b)) return Lin();
    std::vector<float> rows((size_t)cout * k * cin), bias(cout, 0.f);
    for (int co = 0; co < cout; ++co) {
      const float sc = scale ? (*scale)[co] : 1.f;
      for (int ci = 0; ci < cin; ++ci)
        for (int j = 0; j < k; ++j) rows[((size_t)co * k + j) * cin + ci] = w->data[((size_t)co * cin + ci) * k + j] * sc;
      float bv = has_bias ? b->data[co] : 0.f;
      bias[co] = bv * sc + (shift ? (*shift)[co] : 0.f);
    }
    return pack(rows, cout, k * cin, &bias);
  }

  LNp ln(const std::string& p, int d) {
    LNp r;
    const HostTensor* g = get(p + ".weight", {d});
    const HostTensor* b = get(p + ".bias", {d});
    if (!g || !b) return r;
    r.g = upload(g->data);
    r.b = upload(b->data);
    return r;
  }
};

// Slaney mel filter bank, as HF audio_utils.py:638-729 with feature_extraction_whisper.py:97-105's arguments
// (0..8000 Hz, 201 bins, norm="slaney", mel_scale="slaney"); float64 then rounded to fp32 like the reference.
// htk = true: torchaudio.functional.melscale_fbanks(n_freqs = 201, 0, sr / 2, n_mels, sr, norm = None, mel_scale = "htk"), the bank
// behind torchaudio.transforms.MelSpectrogram's defaults (/root/reference/model.py:85-90): m = 2595 log10(1 + f / 700), plain
// triangles of height 1.
static void build_mel(int n_mels, std::vector<int>& lo, std::vector<int>& cnt, std::vector<float>& w, int& maxw, bool htk = false,
                      double fmax = 8000.0) {
  const int nf = 201;
  auto hz2mel = [htk](double f) {
    if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
    return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4)) : 3.0 * f / 200.0;
  };
  auto mel2hz = [htk](double mm) {
    if (htk) return 700.0 * (std::pow(10.0, mm / 2595.0) - 1.0);
    return mm >= 15.0 ? 1000.0 * std::exp((std::log(6.4) / 27.0) * (mm - 15.0)) : 200.0 * mm / 3.0;
  };
  const double m0 = hz2mel(0.0), m1 = hz2mel(fmax);
  std::vector<double> ff(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel2hz(m0 + (m1 - m0) * i / (n_mels + 1));
  std::vector<std::vector<float>> fb(n_mels, std::vector<float>(nf, 0.f));
  for (int k = 0; k < nf; ++k) {
    const double f = fmax * k / (nf - 1);
    for (int i = 0; i < n_mels; ++i) {
      const double down = (f - ff[i]) / (ff[i + 1] - ff[i]);
      const double up = (ff[i + 2] - f) / (ff[i + 2] - ff[i + 1]);
      const double v = std::max(0.0, std::min(down, up)) * (htk ? 1.0 : 2.0 / (ff[i + 2] - ff[i]));
      fb[i][k] = (float)v;
    }
  }
  lo.assign(n_mels, 0);
  cnt.assign(n_mels, 0);
  maxw = 1;
  for (int i = 0; i < n_mels; ++i) {
    int a = nf, b = -1;
    for (int k = 0; k < nf; ++k)
      if (fb[i][k] != 0.f) { a = std::min(a, k); b = std::max(b, k); }
    if (b >= a) { lo[i] = a; cnt[i] = b - a + 1; maxw = std::max(maxw, cnt[i]); }
  }
  w.assign((size_t)n_mels * maxw, 0.f);
  for (int i = 0; i < n_mels; ++i)
    for (int j = 0; j < cnt[i]; ++j) w[(size_t)i * maxw + j] = fb[i][lo[i] + j];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ finalize
// STFT front-end tables: Hann-folded DFT matrix (periodic Hann as torch.hann_window(400)) + the sparse mel bank
static void build_frontend_tables(wfl_model* m, Packer& P, bool htk, double fmax) {
  // folded about n = 200 (see logmel.hip): row j-1 <-> sample j = 1..200; Wc[200] carries the factor 1/2
  std::vector<float> wc((size_t)200 * 224, 0.f), ws((size_t)200 * 224, 0.f);
  for (int j = 1; j <= 200; ++j) {
    const double hann = 0.5 - 0.5 * std::cos(2.0 * M_PI * j / 400.0);
    const double fold = j == 200 ? 0.5 : 1.0;
    for (int k = 0; k <= 200; ++k) {
      const int nk = (int)(((long)j * k) % 400);
      const double ang = 2.0 * M_PI * nk / 400.0;
      wc[(size_t)(j - 1) * 224 + k] = (float)(fold * hann * std::cos(ang));
      ws[(size_t)(j - 1) * 224 + k] = (float)(-hann * std::sin(ang));
    }
  }
  m->Wc = P.upload(wc);
  m->Ws = P.upload(ws);
  std::vector<int> lo, cnt;
  std::vector<float> w;
  build_mel(m->a.n_mels, lo, cnt, w, m->mel_maxw, htk, fmax);
  m->mel_lo = P.upload(lo);
  m->mel_cnt = P.upload(cnt);
  m->mel_w = P.upload(w);
}

// `encoder_type: none`: move the head's checkpoint tensors (width dv) into the padded channel layout (width d, see wfl_model::dv).
// Rows / columns that carry no channel are zero, so every padded activation column stays exactly zero through Linear, Conv1d,
// GELU / ReLU / GLU, LayerNorm (gamma = beta = 0), eval-mode BatchNorm (scale 0), attention (zero q, k, v columns) and the LSTM
// (zero gates: i = f = o = 1/2, g = 0, so c = h = 0).
static void pad_head_state(wfl_model* m, Packer& P) {
  const wfl_arch& a = m->a;
  const int dv = m->dv, d = a.d_model, e = a.lang_emb_dim;
  if (dv == d) return;
  const int split = m->pad_split(), shift = m->pad_shift();
  auto pi = [&](int c) { return c + (c >= split ? shift : 0); };
  std::vector<int> PI(dv), ID_C(a.num_classes), ID2(2);
  for (int c = 0; c < dv; ++c) PI[c] = pi(c);
  for (int c = 0; c < a.num_classes; ++c) ID_C[c] = c;
  ID2[0] = 0; ID2[1] = 1;
  auto ident = [](int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = i; return v; };
  // tensor [rows][cols][taps] (taps = 1: a matrix; cols = 0: a vector) -> [n_out][k_out][taps]
  auto move = [&](const std::string& name, const std::vector<int>& rmap, int n_out, const std::vector<int>& cmap, int k_out, int taps,
                  float fill = 0.f) {
    auto it = m->host.find(name);
    if (it == m->host.end()) return;                  // finalize_head reports the missing key
    HostTensor& t = it->second;
    std::vector<int64_t> want{(int64_t)rmap.size()};
    if (!cmap.empty()) want.push_back((int64_t)cmap.size());
    if (taps > 1 || t.shape.size() == 3) want.push_back(taps);
    if (t.shape != want) {
      if (P.err.empty()) {
        P.err = "size mismatch for " + name + ": got [";
        for (auto v : t.shape) P.err += std::to_string(v) + ",";
        P.err += "] expected [";
        for (auto v : want) P.err += std::to_string(v) + ",";
        P.err += "]";
      }
      return;
    }
    HostTensor o;
    const int kin = cmap.empty() ? 1 : (int)cmap.size(), kout = cmap.empty() ? 1 : k_out;
    o.data.assign((size_t)n_out * kout * taps, fill);
    for (size_t r = 0; r < rmap.size(); ++r)
      for (int c = 0; c < kin; ++c)
        for (int j = 0; j < taps; ++j)
          o.data[((size_t)rmap[r] * kout + (cmap.empty() ? 0 : cmap[c])) * taps + j] = t.data[((size_t)r * kin + c) * taps + j];
    o.shape.push_back(n_out);
    if (!cmap.empty()) o.shape.push_back(k_out);
    if (t.shape.size() == 3) o.shape.push_back(taps);
    t = std::move(o);
  };
  const std::vector<int> none;
  {                                                   // lang_proj: [dv][dv + e]: hidden columns through pi, embedding columns behind them
    std::vector<int> cm(dv + e);
    for (int c = 0; c < dv; ++c) cm[c] = pi(c);
    for (int c = 0; c < e; ++c) cm[dv + c] = d + c;
    move("lang_proj.weight", PI, d, cm, d + e, 1);
    move("lang_proj.bias", PI, d, none, 0, 1);
  }
  if (a.enable_bilstm) {
    const int Hv = dv / 2, Hp = d / 2;
    std::vector<int> gm(4 * Hv);
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < Hv; ++u) gm[g * Hv + u] = g * Hp + u;
    for (int layer = 0; layer < a.bilstm_layers; ++layer)
      for (int dir = 0; dir < 2; ++dir) {
        const std::string suf = "_l" + std::to_string(layer) + (dir ? "_reverse" : "");
        move("bilstm.weight_ih" + suf, gm, 4 * Hp, PI, d, 1);          // layer 0 reads the hidden states, deeper ones [fwd | bwd]: both pi
        move("bilstm.weight_hh" + suf, gm, 4 * Hp, ident(Hv), Hp, 1);
        move("bilstm.bias_ih" + suf, gm, 4 * Hp, none, 0, 1);
        move("bilstm.bias_hh" + suf, gm, 4 * Hp, none, 0, 1);
      }
  }
  const int x = a.conformer_ff_expansion, kk = a.conformer_kernel;
  for (int i = 0; i < a.n_conformer; ++i) {
    const std::string p = "conformer_layers." + std::to_string(i) + ".";
    const int hv = dv / a.conformer_heads, hp = d / a.conformer_heads;
    std::vector<int> hm(dv), qm(3 * dv), gl(2 * dv);
    for (int c = 0; c < dv; ++c) hm[c] = (c / hv) * hp + c % hv;             // per-head padding of the attention's channel axis
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < dv; ++c) qm[b * dv + c] = b * d + hm[c];
    for (int c = 0; c < dv; ++c) { gl[c] = pi(c); gl[dv + c] = d + pi(c); }   // GLU: value half | gate half
    for (const char* ff : {"ff1", "ff2"}) {
      const std::string q = p + ff + ".net.";
      move(q + "0.weight", PI, d, none, 0, 1);
      move(q + "0.bias", PI, d, none, 0, 1);
      move(q + "1.weight", ident(dv * x), d * x, PI, d, 1);
      move(q + "1.bias", ident(dv * x), d * x, none, 0, 1);
      move(q + "4.weight", PI, d, ident(dv * x), d * x, 1);
      move(q + "4.bias", PI, d, none, 0, 1);
    }
    move(p + "self_attn.in_proj_weight", qm, 3 * d, PI, d, 1);
    move(p + "self_attn.in_proj_bias", qm, 3 * d, none, 0, 1);
    move(p + "self_attn.out_proj.weight", PI, d, hm, d, 1);
    move(p + "self_attn.out_proj.bias", PI, d, none, 0, 1);
    for (const char* ln : {"ln1", "ln2"}) {
      move(p + ln + ".weight", PI, d, none, 0, 1);
      move(p + ln + ".bias", PI, d, none, 0, 1);
    }
    move(p + "conv.0.weight", gl, 2 * d, PI, d, 1);
    move(p + "conv.0.bias", gl, 2 * d, none, 0, 1);
    move(p + "conv.2.weight", PI, d, PI, d, kk);
    move(p + "conv.2.bias", PI, d, none, 0, 1);
    move(p + "conv.3.weight", PI, d, none, 0, 1);
    move(p + "conv.3.bias", PI, d, none, 0, 1);
    move(p + "conv.3.running_mean", PI, d, none, 0, 1);
    move(p + "conv.3.running_var", PI, d, none, 0, 1, 1.0f);
    move(p + "conv.5.weight", PI, d, PI, d, 1);
    move(p + "conv.5.bias", PI, d, none, 0, 1);
  }
  if (a.enable_dilated)
    for (int i = 0; i < a.dilated_depth; ++i) {
      const std::string p = "dilated_conv_stack." + std::to_string(2 * i);
      move(p + ".weight", PI, d, PI, d, a.dilated_kernel);
      move(p + ".bias", PI, d, none, 0, 1);
    }
  move("classifier.weight", ID_C, a.num_classes, PI, d, 1);
  move("boundary_offset_head.0.weight", PI, d, PI, d, 3);
  move("boundary_offset_head.0.bias", PI, d, none, 0, 1);
  move("boundary_offset_head.2.weight", ID2, 2, PI, d, 1);
}

static int finalize_whisper(wfl_model* m, Packer& P) {
  const wfl_arch& a = m->a;
  const int d = a.d_model, hd = d / a.enc_heads;
  build_frontend_tables(m, P, false, 8000.0);
  m->conv1 = P.conv("encoder.conv1", d, a.n_mels, 3);
  m->conv2 = P.conv("encoder.conv2", d, d, 3);
  if (const HostTensor* pe = P.get("encoder.embed_positions.weight", {a.max_positions, d})) {
    std::vector<uint16_t> pb(pe->data.size());
    for (size_t i = 0; i < pb.size(); ++i) pb[i] = f32_to_bf16_bits(pe->data[i]);
    m->pos = (bf16_t*)P.upload(pb);
    if (a.precision) {
      for (size_t i = 0; i < pb.size(); ++i) {
        const uint32_t u = (uint32_t)pb[i] << 16;
        float xh;
        memcpy(&xh, &u, 4);
        pb[i] = f32_to_bf16_bits(pe->data[i] - xh);
      }
      m->pos_lo = (bf16_t*)P.upload(pb);
    }
  }
  const float qs = (float)(std::pow((double)hd, -0.5) * 1.4426950408889634);   // hd^-1/2 * log2(e)
  m->enc.resize(a.enc_layers);
  for (int i = 0; i < a.enc_layers; ++i) {
    const std::string p = "encoder.layers." + std::to_string(i) + ".";
    EncLayer& L = m->enc[i];
    L.ln1 = P.ln(p + "self_attn_layer_norm", d);
    L.ln2 = P.ln(p + "final_layer_norm", d);
    const HostTensor* wq = P.get(p + "self_attn.q_proj.weight", {d, d});
    const HostTensor* bq = P.get(p + "self_attn.q_proj.bias", {d});
    const HostTensor* wk = P.get(p + "self_attn.k_proj.weight", {d, d});
    const HostTensor* wv = P.get(p + "self_attn.v_proj.weight", {d, d});
    const HostTensor* bv = P.get(p + "self_attn.v_proj.bias", {d});
    if (wq && bq && wk && wv && bv) {
      std::vector<float> rows((size_t)3 * d * d), bias((size_t)3 * d, 0.f);
      for (size_t j = 0; j < (size_t)d * d; ++j) {
        rows[j] = wq->data[j] * qs;
        rows[(size_t)d * d + j] = wk->data[j];
        rows[(size_t)2 * d * d + j] = wv->data[j];
      }
      for (int j = 0; j < d; ++j) { bias[j] = bq->data[j] * qs; bias[2 * d + j] = bv->data[j]; }
      if (a.fp8_weights) {
        L.qkv = P.pack8(rows, 3 * d, d, &bias);
      } else {
        L.qkv = P.pack(rows, 3 * d, d, &bias);
        L.qkv_ln = P.pack_ln(rows, 3 * d, d, &bias, p + "self_attn_layer_norm");
      }
    }
    if (a.fp8_weights) {
      // fp8 weights (BASELINE configs[4]): the four big operands of every layer as e4m3 + per-channel scale; no LayerNorm folding
      // (its gamma would have to go into the quantised tensor), the LayerNorm kernel feeds the plain fp8 GEMM
      L.out = P.linear8(p + "self_attn.out_proj", d, d);
      L.fc1 = P.linear8(p + "fc1", a.enc_ffn, d);
      L.fc2 = P.linear8(p + "fc2", d, a.enc_ffn);
    } else {
      L.out = P.linear(p + "self_attn.out_proj", d, d);
      L.fc1 = P.linear(p + "fc1", a.enc_ffn, d);
      L.fc1_ln = P.linear_ln(p + "fc1", a.enc_ffn, d, p + "final_layer_norm");
      L.fc2 = P.linear(p + "fc2", d, a.enc_ffn);
    }
  }
  m->enc_ln = P.ln("encoder.layer_norm", d);
  return 0;
}

static int finalize_wavlm(wfl_model* m, Packer& P) {
  const wfl_arch& a = m->a;
  const int d = a.d_model, hd = d / a.enc_heads, C = a.wavlm_conv_dim[0], nconv = a.wavlm_n_conv;
  const bool group = a.wavlm_group_norm != 0, cbias = a.wavlm_conv_bias != 0;
  const std::string fe = "encoder.feature_extractor.conv_layers.";
  if (const HostTensor* w0 = P.get(fe + "0.conv.weight", {C, 1, 10})) m->conv0_w = P.upload(w0->data);
  if (cbias)
    if (const HostTensor* b0 = P.get(fe + "0.conv.bias", {C})) m->conv0_b = P.upload(b0->data);
  m->conv0_norm = P.ln(fe + "0.layer_norm", C);
  for (int i = 1; i < nconv; ++i) {
    m->fconv.push_back(P.conv(fe + std::to_string(i) + ".conv", C, C, a.wavlm_conv_kernel[i], nullptr, nullptr, cbias));
    if (!group) m->fconv_ln.push_back(P.ln(fe + std::to_string(i) + ".layer_norm", C));
  }
  m->fp_ln = P.ln("encoder.feature_projection.layer_norm", C);
  m->fp_proj = P.linear("encoder.feature_projection.projection", d, C);
  // positional conv: weight = g * v / ||v||, norm over dims (0, 1) per tap (weight_norm dim=2); HF modeling_wavlm.py:48-90
  {
    const int G = a.wavlm_pos_conv_groups, cpg = d / G, K = a.wavlm_pos_conv_kernel;
    const std::string pc = "encoder.encoder.pos_conv_embed.conv.";
    const HostTensor* g0 = P.get(pc + "parametrizations.weight.original0", {1, 1, K});
    const HostTensor* v = P.get(pc + "parametrizations.weight.original1", {d, cpg, K});
    const HostTensor* pb = P.get(pc + "bias", {d});
    if (g0 && v && pb) {
      std::vector<double> nrm(K, 0.0);
      for (int o = 0; o < d; ++o)
        for (int c = 0; c < cpg; ++c)
          for (int k = 0; k < K; ++k) { const double x = v->data[((size_t)o * cpg + c) * K + k]; nrm[k] += x * x; }
      for (int k = 0; k < K; ++k) nrm[k] = std::sqrt(nrm[k]);
      for (int gi = 0; gi < G; ++gi) {
        // rows = the group's cpg output channels, k index = tap * 64 + channel (channels padded to 64)
        std::vector<float> rows((size_t)cpg * K * 64, 0.f), bias(cpg);
        for (int o = 0; o < cpg; ++o) {
          const int oc = gi * cpg + o;
          bias[o] = pb->data[oc];
          for (int c = 0; c < cpg; ++c)
            for (int k = 0; k < K; ++k)
              rows[((size_t)o * K + k) * 64 + c] = (float)(g0->data[k] * v->data[((size_t)oc * cpg + c) * K + k] / nrm[k]);
        }
        m->posconv.push_back(P.pack(rows, cpg, K * 64, &bias));
      }
    }
  }
  m->wenc_ln = P.ln("encoder.encoder.layer_norm", d);
  const float qs = (float)(std::pow((double)hd, -0.5) * 1.4426950408889634);
  m->wl.resize(a.enc_layers);
  for (int i = 0; i < a.enc_layers; ++i) {
    const std::string p = "encoder.encoder.layers." + std::to_string(i) + ".";
    WavlmLayer& L = m->wl[i];
    L.ln1 = P.ln(p + "layer_norm", d);
    L.ln2 = P.ln(p + "final_layer_norm", d);
    const HostTensor* wq = P.get(p + "attention.q_proj.weight", {d, d});
    const HostTensor* bq = P.get(p + "attention.q_proj.bias", {d});
    const HostTensor* wk = P.get(p + "attention.k_proj.weight", {d, d});
    const HostTensor* bk = P.get(p + "attention.k_proj.bias", {d});
    const HostTensor* wv = P.get(p + "attention.v_proj.weight", {d, d});
    const HostTensor* bv = P.get(p + "attention.v_proj.bias", {d});
    if (wq && bq && wk && bk && wv && bv) {
      std::vector<float> rows((size_t)3 * d * d), bias((size_t)3 * d);
      for (size_t j = 0; j < (size_t)d * d; ++j) {
        rows[j] = wq->data[j] * qs;
        rows[(size_t)d * d + j] = wk->data[j];
        rows[(size_t)2 * d * d + j] = wv->data[j];
      }
      for (int j = 0; j < d; ++j) { bias[j] = bq->data[j] * qs; bias[d + j] = bk->data[j]; bias[2 * d + j] = bv->data[j]; }
      L.qkv = P.pack(rows, 3 * d, d, &bias);
    }
    L.out = P.linear(p + "attention.out_proj", d, d);
    L.fc1 = P.linear(p + "feed_forward.intermediate_dense", a.enc_ffn, d);
    // (stable layers: final_layer_norm could fold into fc1 like Whisper's; measured on cfg3 it costs more in the residual GEMM's
    //  statistics epilogue -- four 256-column tiles at d = 1024 -- and the folded fc1 than the LayerNorm launch it saves:
    //  GEMM family +1.2 ms, LayerNorm -0.84 ms per 64 x 10 s step.  Set WFL_WAVLM_LN_FOLD=1 to try it.)
    if (a.wavlm_stable_layer_norm && getenv("WFL_WAVLM_LN_FOLD") && atoi(getenv("WFL_WAVLM_LN_FOLD")))
      L.fc1_ln = P.linear_ln(p + "feed_forward.intermediate_dense", a.enc_ffn, d, p + "final_layer_norm");
    L.fc2 = P.linear(p + "feed_forward.output_dense", d, a.enc_ffn);
    if (const HostTensor* w8 = P.get(p + "attention.gru_rel_pos_linear.weight", {8, hd})) L.w8 = P.upload(w8->data);
    if (const HostTensor* b8 = P.get(p + "attention.gru_rel_pos_linear.bias", {8})) L.b8 = P.upload(b8->data);
    if (const HostTensor* cst = P.get(p + "attention.gru_rel_pos_const", {1, a.enc_heads, 1, 1})) L.cst = P.upload(cst->data);
    if (i == 0) {
      if (const HostTensor* re = P.get(p + "attention.rel_attn_embed.weight", {a.wavlm_num_buckets, a.enc_heads})) {
        std::vector<float> scaled(re->data);
        for (auto& x : scaled) x *= 1.4426950408889634f;           // scores live in the log2 domain
        m->rel_emb = P.upload(scaled);
      }
    }
  }
  // relative-position buckets (HF modeling_wavlm.py:243-271), float32 arithmetic like torch
  {
    std::vector<int> bod(2 * WAVLM_MAX_T - 1);
    const int nb = a.wavlm_num_buckets / 2, max_exact = nb / 2;
    for (int delta = -(WAVLM_MAX_T - 1); delta <= WAVLM_MAX_T - 1; ++delta) {
      int bkt = delta > 0 ? nb : 0;
      const int ad = delta < 0 ? -delta : delta;
      if (ad < max_exact) bkt += ad;
      else {
        const float v = std::log((float)ad / (float)max_exact) / (float)std::log((double)a.wavlm_max_distance / max_exact) * (float)(nb - max_exact);
        int large = max_exact + (int)v;
        if (large > nb - 1) large = nb - 1;
        bkt += large;
      }
      bod[delta + WAVLM_MAX_T - 1] = bkt;
    }
    m->bucket_of_delta = P.upload(bod);
  }
  return 0;
}

static int finalize_head(wfl_model* m, Packer& P) {
  const wfl_arch& a = m->a;
  const int d = a.d_model, e = a.lang_emb_dim;
  // language conditioning: cat(h, emb) @ W^T + b  ==  h @ W[:, :d]^T + (W[:, d:] @ emb + b)   (model.py:176-180)
  const HostTensor* emb = P.get("lang_emb.weight", {a.num_languages, e});
  const HostTensor* lw = P.get("lang_proj.weight", {d, d + e});
  const HostTensor* lb = P.get("lang_proj.bias", {d});
  if (emb && lw && lb) {
    std::vector<float> rows((size_t)d * d), table((size_t)std::max(a.num_languages, 1) * d, 0.f);
    for (int n = 0; n < d; ++n) {
      for (int k = 0; k < d; ++k) rows[(size_t)n * d + k] = lw->data[(size_t)n * (d + e) + k];
      for (int l = 0; l < a.num_languages; ++l) {
        double acc = lb->data[n];
        for (int k = 0; k < e; ++k) acc += (double)lw->data[(size_t)n * (d + e) + d + k] * emb->data[(size_t)l * e + k];
        table[(size_t)l * d + n] = (float)acc;
      }
    }
    m->lang = P.pack(rows, d, d, nullptr);
    m->lang_table = P.upload(table);
  }
  if (a.enable_bilstm) {
    // nn.LSTM(d, d/2, num_layers, bidirectional): model.py:104-111.  Gate rows i|f|g|o (blocks of H) are reordered
    // unit-major / gate-minor for the recurrence kernel (lstm.hip); b_ih + b_hh fold into the projection bias.
    const int H = d / 2;
    m->lstm_U = wfl_lstm_units_per_wg(H);
    if (m->lstm_U <= 0 && P.err.empty()) P.err = "BiLSTM hidden size " + std::to_string(H) + " is not supported";
    const int U = std::max(m->lstm_U, 1), G = H / U;
    for (int layer = 0; layer < a.bilstm_layers && P.err.empty(); ++layer) {
      const int din = d;      // layer 0: encoder width d; deeper layers: 2H = d
      std::vector<float> rows((size_t)8 * H * din), bias((size_t)8 * H);
      std::vector<uint16_t> whh((size_t)2 * 4 * H * H), whh_lo((size_t)2 * 4 * H * H);
      for (int dir = 0; dir < 2; ++dir) {
        const std::string suf = "_l" + std::to_string(layer) + (dir ? "_reverse" : "");
        const HostTensor* wih = P.get("bilstm.weight_ih" + suf, {4 * H, din});
        const HostTensor* whh_t = P.get("bilstm.weight_hh" + suf, {4 * H, H});
        const HostTensor* bih = P.get("bilstm.bias_ih" + suf, {4 * H});
        const HostTensor* bhh = P.get("bilstm.bias_hh" + suf, {4 * H});
        if (!wih || !whh_t || !bih || !bhh) break;
        for (int u = 0; u < H; ++u)
          for (int gate = 0; gate < 4; ++gate) {
            const int src = gate * H + u;
            const size_t dst = (size_t)dir * 4 * H + 4 * u + gate;
            memcpy(&rows[dst * din], &wih->data[(size_t)src * din], sizeof(float) * din);
            bias[dst] = bih->data[src] + bhh->data[src];
            // slice-major packing: [dir][slice][4*u_local + gate][H]
            const int sl = u / U, ul = u % U;
            const size_t wrow = ((size_t)(dir * G + sl) * 4 * U) + 4 * ul + gate;
            for (int k = 0; k < H; ++k) {
              const float x = whh_t->data[(size_t)src * H + k];
              const uint16_t hb = f32_to_bf16_bits(x);
              const uint32_t u32 = (uint32_t)hb << 16;
              float xh;
              memcpy(&xh, &u32, 4);
              whh[wrow * H + k] = hb;
              whh_lo[wrow * H + k] = f32_to_bf16_bits(x - xh);
            }
          }
      }
      if (!P.err.empty()) break;
      m->lstm_in.push_back(P.pack(rows, 8 * H, din, &bias));
      m->lstm_whh.push_back((bf16_t*)P.upload(whh));
      m->lstm_whh_lo.push_back((a.precision && wfl_lstm_split_precision_supported(H)) ? (bf16_t*)P.upload(whh_lo) : nullptr);
    }
  }
  m->conf.resize(a.n_conformer);
  const int x = a.conformer_ff_expansion, kk = a.conformer_kernel;
  for (int i = 0; i < a.n_conformer; ++i) {
    const std::string p = "conformer_layers." + std::to_string(i) + ".";
    ConfLayer& C = m->conf[i];
    C.ff1_ln = P.ln(p + "ff1.net.0", d);
    C.ff1_a = P.linear(p + "ff1.net.1", d * x, d);
    C.ff1_a_ln = P.linear_ln(p + "ff1.net.1", d * x, d, p + "ff1.net.0");
    C.ff1_b = P.linear(p + "ff1.net.4", d, d * x);
    C.ff2_ln = P.ln(p + "ff2.net.0", d);
    C.ff2_a = P.linear(p + "ff2.net.1", d * x, d);
    C.ff2_a_ln = P.linear_ln(p + "ff2.net.1", d * x, d, p + "ff2.net.0");
    C.ff2_b = P.linear(p + "ff2.net.4", d, d * x);
    const HostTensor* iw = P.get(p + "self_attn.in_proj_weight", {3 * d, d});
    const HostTensor* ib = P.get(p + "self_attn.in_proj_bias", {3 * d});
    if (iw && ib) {
      const int hd = m->dv / a.conformer_heads;              // (the true head size; padded columns are zero)
      const float qs = (float)(std::pow((double)hd, -0.5) * 1.4426950408889634);
      std::vector<float> rows(iw->data), bias(ib->data);
      for (size_t j = 0; j < (size_t)d * d; ++j) rows[j] *= qs;
      for (int j = 0; j < d; ++j) bias[j] *= qs;
      const int da = m->conf_da, hp = m->conf_hd, hv = d / a.conformer_heads;
      if (da == d) {
        C.qkv = P.pack(rows, 3 * d, d, &bias);
      } else {                                               // head h's rows move to [h * hp, h * hp + hv) of each da-wide block
        std::vector<float> prow((size_t)3 * da * d, 0.f), pbias((size_t)3 * da, 0.f);
        for (int b3 = 0; b3 < 3; ++b3)
          for (int c = 0; c < d; ++c) {
            const size_t dst = (size_t)b3 * da + (c / hv) * hp + c % hv;
            memcpy(&prow[dst * d], &rows[((size_t)b3 * d + c) * d], sizeof(float) * d);
            pbias[dst] = bias[(size_t)b3 * d + c];
          }
        C.qkv = P.pack(prow, 3 * da, d, &pbias);
      }
    }
    if (m->conf_da == d) {
      C.out = P.linear(p + "self_attn.out_proj", d, d);
    } else {
      const HostTensor* ow = P.get(p + "self_attn.out_proj.weight", {d, d});
      const HostTensor* ob = P.get(p + "self_attn.out_proj.bias", {d});
      if (ow && ob) {
        const int da = m->conf_da, hp = m->conf_hd, hv = d / a.conformer_heads;
        std::vector<float> prow((size_t)d * da, 0.f);
        for (int n = 0; n < d; ++n)
          for (int c = 0; c < d; ++c) prow[(size_t)n * da + (c / hv) * hp + c % hv] = ow->data[(size_t)n * d + c];
        C.out = P.pack(prow, d, da, &ob->data);
      }
    }
    C.ln1 = P.ln(p + "ln1", d);
    C.ln2 = P.ln(p + "ln2", d);
    // pointwise conv d -> 2d followed by GLU: interleave rows in groups of 16 (a | gate)
    const HostTensor* pw = P.get(p + "conv.0.weight", {2 * d, d, 1});
    const HostTensor* pb = P.get(p + "conv.0.bias", {2 * d});
    if (pw && pb) {
      std::vector<float> rows((size_t)2 * d * d), bias((size_t)2 * d);
      for (int gidx = 0; gidx < d / 16; ++gidx)
        for (int r = 0; r < 16; ++r) {
          const int ca = gidx * 16 + r, cg = d + gidx * 16 + r;
          const int ra = gidx * 32 + r, rg = gidx * 32 + 16 + r;
          memcpy(&rows[(size_t)ra * d], &pw->data[(size_t)ca * d], sizeof(float) * d);
          memcpy(&rows[(size_t)rg * d], &pw->data[(size_t)cg * d], sizeof(float) * d);
          bias[ra] = pb->data[ca];
          bias[rg] = pb->data[cg];
        }
      C.pw1 = P.pack(rows, 2 * d, d, &bias);
    }
    // dense k-tap conv with eval-mode BatchNorm folded in: y = (conv(x) - mean) * g / sqrt(var + eps) + beta
    const HostTensor* bw = P.get(p + "conv.3.weight", {d});
    const HostTensor* bb = P.get(p + "conv.3.bias", {d});
    const HostTensor* bm = P.get(p + "conv.3.running_mean", {d});
    const HostTensor* bvar = P.get(p + "conv.3.running_var", {d});
    if (m->host.count(p + "conv.3.num_batches_tracked")) P.used.insert(p + "conv.3.num_batches_tracked");
    if (bw && bb && bm && bvar) {
      std::vector<float> sc(d), sh(d);
      for (int c = 0; c < d; ++c) {
        sc[c] = bw->data[c] / std::sqrt(bvar->data[c] + 1e-5f);
        sh[c] = bb->data[c] - bm->data[c] * sc[c];
      }
      C.conv = P.conv(p + "conv.2", d, d, kk, &sc, &sh);
    }
    C.pw2 = P.conv(p + "conv.5", d, d, 1);
  }
  if (a.enable_dilated) {
    for (int i = 0; i < a.dilated_depth; ++i)
      m->dil.push_back(P.conv("dilated_conv_stack." + std::to_string(2 * i), d, d, a.dilated_kernel));
  }
  m->cls = P.linear("classifier", a.num_classes, d);
  // The classifier runs in split precision (its inputs are the residual stream's hi + lo halves, its weights hi + lo):
  // logits = [h_hi | h_lo] . [W_hi | W_hi]^T + h_hi . W_lo^T + b, three bf16 MFMA passes summed in fp32 -- 0.4 % of the forward's FLOPs
  // for a logit error a plain bf16 pass would double (tests/study_quant.py).
  if (const HostTensor* cw = P.get("classifier.weight", {a.num_classes, d})) {
    std::vector<float> lo((size_t)a.num_classes * d), hi2((size_t)a.num_classes * 2 * d), w3((size_t)a.num_classes * 3 * d);
    for (int n = 0; n < a.num_classes; ++n)
      for (int k = 0; k < d; ++k) {
        const float w = cw->data[(size_t)n * d + k];
        const uint32_t u = (uint32_t)f32_to_bf16_bits(w) << 16;
        float wh;
        memcpy(&wh, &u, 4);
        lo[(size_t)n * d + k] = w - wh;
        hi2[(size_t)n * 2 * d + k] = wh;
        hi2[(size_t)n * 2 * d + d + k] = wh;
        w3[(size_t)n * 3 * d + k] = wh;
        w3[(size_t)n * 3 * d + d + k] = wh;
        w3[(size_t)n * 3 * d + 2 * d + k] = w - wh;
      }
    const HostTensor* cb = P.get("classifier.bias", {a.num_classes});
    m->cls_lo = P.pack(lo, a.num_classes, d, nullptr);
    if (cb) m->cls_hi2 = P.pack(hi2, a.num_classes, 2 * d, &cb->data);
    if (cb && d % 64 == 0) m->cls3 = P.pack(w3, a.num_classes, 3 * d, &cb->data);
  }
  m->off1 = P.conv("boundary_offset_head.0", d, d, 3);
  const HostTensor* w2 = P.get("boundary_offset_head.2.weight", {2, d, 1});
  const HostTensor* b2 = P.get("boundary_offset_head.2.bias", {2});
  if (w2 && b2) {
    m->off_w2 = P.upload(w2->data);
    m->off_b2 = P.upload(b2->data);
  }
  return 0;
}

int32_t wfl_finalize(wfl_model* m) {
  if (!m) return fail(-1, "wfl_finalize: null model");
  if (m->finalized) return fail(-1, "wfl_finalize: already finalized");
  Packer P{m};
  if (hipGetDevice(&m->device) != hipSuccess) return fail(-10, "wfl_finalize: hipGetDevice failed");
  if (m->a.encoder_type == WFL_ENC_WHISPER) finalize_whisper(m, P);
  else if (m->a.encoder_type == WFL_ENC_WAVLM) finalize_wavlm(m, P);
  else {
    build_frontend_tables(m, P, true, 8000.0);      // MelSpectrogram(f_min = 0, f_max = sample_rate / 2): 16 kHz input, like the rest of the path
    pad_head_state(m, P);
  }
  if (P.err.empty()) finalize_head(m, P);
  if (!P.err.empty()) return fail(-2, "wfl_finalize: " + P.err);
  for (auto& kv : m->host)
    if (!P.used.count(kv.first)) return fail(-2, "wfl_finalize: unexpected key in state_dict: " + kv.first);
  m->host.clear();
  m->avg_langs.clear();
  for (int i = 0; i < m->a.num_languages; ++i) m->avg_langs.push_back(i);
  m->finalized = true;
  return 0;
}

int32_t wfl_device(const wfl_model* m) { return m ? m->device : -1; }

int32_t wfl_set_average_languages(wfl_model* m, const int32_t* ids_host, int32_t n) {
  if (!m || !m->finalized) return fail(-1, "wfl_set_average_languages: model not finalized");
  if (!ids_host || n <= 0) return fail(-1, "wfl_set_average_languages: empty id list");
  for (int i = 0; i < n; ++i)
    if (ids_host[i] < 0 || ids_host[i] >= m->a.num_languages) return fail(-1, "wfl_set_average_languages: id out of range");
  m->avg_langs.assign(ids_host, ids_host + n);
  return 0;
}

// Every entry point that touches the device: the model's device must be the current one (a handle on GPU 1 driven with GPU 0
// current would dereference GPU-1 weight pointers from GPU-0 kernels).
static int check_device(const wfl_model* m, const char* who) {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) return fail(-10, std::string(who) + ": hipGetDevice failed");
  if (d != m->device)
    return fail(-11, std::string(who) + ": the model lives on HIP device " + std::to_string(m->device) + " but device " +
                         std::to_string(d) + " is current (call hipSetDevice / torch.cuda.set_device first)");
  return 0;
}

// ------------------------------------------------------------------------------------------------ workspace plan
namespace {

struct Plan {
  int B, L, T, P, lead, tail;
  int T2, P2, lead2;            // whisper stem input rows (mel frames)
  long R, R2;
  int d, ffw;
  // wavlm feature-encoder levels: level i = output of conv layer i (frames T_i, pitch P_i = P * 2^(n-1-i), lead 8)
  int nlev, Tl[8], Pl[8], leadl[8];
  long Rl[8];
  // byte offsets
  long mel, c1, X, Y, ATT, QK, FF, stats, raw, clipmax, logits, logits2, offs2, gx, lstm_x, enc2;
  long FA, FB, XG, gate, rtab, wstats, cstats, cpart, err, Xlo, Ylo, QKp, ATTp, clipT, total;
  long X8, FF8, rs8;            // fp8 activations (fp8_weights models): e4m3 rows [R][d], [R][ffw], fp32 row scales [R]
  long X8lo, FF8lo;             //   and their lo planes (e4m3 pairs, gemm_mx.hip)
  long lo_delta, hp32;          // "model.precision: high": every activation buffer has its low half lo_delta bytes further on (a twin of
  long hp32_floats;             //   the whole activation area); hp32 = the fp32 sums of the three passes, [rows][columns]
  int da;                       // Conformer attention width (wfl_model::conf_da); QKp / ATTp exist when it differs from d
};

static int wavlm_frames(const wfl_arch& a, int L) {
  long t = L;
  for (int i = 0; i < a.wavlm_n_conv; ++i) {
    if (t < a.wavlm_conv_kernel[i]) return 0;
    t = (t - a.wavlm_conv_kernel[i]) / a.wavlm_conv_stride[i] + 1;
  }
  return (int)t;
}

// T_frames > 0: a head-only plan over that many frames per clip (wfl_head): no encoder-side buffers.
static Plan make_plan(const wfl_model* m, int B, int L, int T_frames = 0) {
  const wfl_arch& a = m->a;
  const bool whisper = a.encoder_type == WFL_ENC_WHISPER;
  Plan p{};
  p.B = B; p.L = L;
  const bool none = a.encoder_type == WFL_ENC_NONE;
  p.T = T_frames > 0 ? T_frames : (whisper ? a.max_positions : (none ? 1 + L / a.mel_hop : wavlm_frames(a, L)));
  p.P = (int)round_up(p.T + m->halo, 8);
  p.lead = m->halo;
  p.tail = 256;
  p.T2 = 2 * p.T; p.P2 = 2 * p.P; p.lead2 = 8;
  p.R = p.lead + (long)B * p.P + p.tail;
  p.R2 = p.lead2 + (long)B * p.P2 + 2 * p.tail;
  p.d = a.d_model;
  p.ffw = std::max(a.enc_ffn, a.d_model * std::max(a.conformer_ff_expansion, 1));
  long off = 0;
  auto take = [&](long bytes) { long o = off; off = round_up(off + bytes, 256); return o; };
  if (T_frames > 0) {
  } else if (whisper) {
    p.mel = take(p.R2 * a.n_mels * 2 + 1024);
    p.c1 = take(p.R2 * p.d * 2);
    p.raw = take((long)B * p.T2 * a.n_mels * 4);
  } else if (none) {
    p.raw = take((long)B * p.T * a.n_mels * 4);        // the mel power = the hidden states, fp32 [B][T][n_mels]
  } else {
    const int n = a.wavlm_n_conv, C = a.wavlm_conv_dim[0];
    p.nlev = n;
    long t = L;
    for (int i = 0; i < n; ++i) {
      t = t >= a.wavlm_conv_kernel[i] ? (t - a.wavlm_conv_kernel[i]) / a.wavlm_conv_stride[i] + 1 : 0;
      p.Tl[i] = (int)t;
    }
    for (int i = n - 1; i >= 0; --i) {
      p.Pl[i] = i == n - 1 ? p.P : p.Pl[i + 1] * 2;
      p.leadl[i] = i == n - 1 ? p.lead : 8;
      p.Rl[i] = p.leadl[i] + (long)B * p.Pl[i] + p.tail;
    }
    p.FA = take(p.Rl[0] * C * 2);                     // levels 0, 2, 4, 6
    p.FB = take(p.Rl[1] * C * 2);                     // levels 1, 3, 5
    p.XG = take((long)a.wavlm_pos_conv_groups * p.R * 64 * 2);
    p.gate = take((long)B * a.enc_heads * p.T * 4);
    p.rtab = take((long)a.enc_heads * (2L * p.T + 1) * 4);
    p.wstats = take((long)B * 2 * 8);
    p.cstats = take((long)B * C * 2 * 8);
    p.cpart = take(a.wavlm_group_norm ? (long)B * ((p.Tl[0] + 511) / 512) * C * 2 * 4 : 16);   // GroupNorm partials per 512-step block
  }
  p.X = take(p.R * p.d * 2);
  p.Y = take(p.R * p.d * 2);
  p.Xlo = take(p.R * p.d * 2);                          // low halves of the residual stream (GemmArgs::res_lo)
  p.Ylo = take(p.R * p.d * 2);
  p.ATT = take(p.R * p.d * 2);
  p.QK = take(p.R * 3 * p.d * 2);                   // packed q | k | v rows
  p.FF = take(p.R * p.ffw * 2);
  p.da = m->conf_da > 0 ? m->conf_da : p.d;
  if (p.da != p.d) {
    p.QKp = take(p.R * 3 * p.da * 2);
    p.ATTp = take(p.R * p.da * 2);
  }
  if (a.fp8_weights && T_frames <= 0) {
    p.X8 = take(p.R * (long)p.d);
    p.FF8 = take(p.R * (long)p.ffw);
    p.rs8 = take(p.R * 4L);
    p.X8lo = take(p.R * (long)p.d);
    p.FF8lo = take(p.R * (long)p.ffw);
  }
  p.stats = take(p.R * 4L * 2 * 4);                     // per row, per 256-column tile (<= 4): (sum, sum of squares)
  p.enc2 = take(p.R * p.d * 2);
  p.clipmax = take((long)B * 4);
  p.logits = take((long)B * p.T * round_up(a.num_classes, 4) * 4);      // (rows of a multiple of four floats when the caller does not ask for them)
  p.logits2 = take((long)B * p.T * a.num_classes * 4);
  p.offs2 = take((long)B * p.T * 2 * 4);
  p.gx = p.lstm_x = 0;
  if (a.enable_bilstm) {
    p.gx = take(p.R * 4L * p.d * 4);                       // fp32 [rows][8H]
    p.lstm_x = take(wfl_lstm_exchange_bytes(p.d / 2, B));
  }
  p.clipT = take(8L * B * 4);                            // per-clip frame counts per front-end level (ragged batches), [level][B]
  p.err = take(256);                                     // the forward's device-side error word
  if (a.precision) {
    p.lo_delta = round_up(off, 256);
    off = 2 * p.lo_delta;                                // the twin
    long rows = std::max(p.R2, p.R);
    for (int i = 1; i < p.nlev; ++i) rows = std::max(rows, p.Rl[i]);                       // (WavLM: the feature encoder's levels are GEMM outputs too)
    const long cols = std::max<long>(std::max<long>(3L * std::max(p.d, p.da), 2L * p.ffw), 1024);
    p.hp32_floats = rows * cols;
    p.hp32 = take(p.hp32_floats * 4);
  }
  p.total = off;
  return p;
}

}  // namespace

int32_t wfl_num_frames(const wfl_model* m, int32_t L) {
  if (!m) return -1;
  if (m->a.encoder_type == WFL_ENC_NONE) return L > 200 ? 1 + L / m->a.mel_hop : 0;    // (reflect padding needs L > n_fft / 2)
  return m->a.encoder_type == WFL_ENC_WHISPER ? m->a.max_positions : wavlm_frames(m->a, L);
}

int64_t wfl_workspace_bytes(const wfl_model* m, int32_t B, int32_t L) {
  if (!m || B <= 0) return -1;
  return make_plan(m, B, L).total;
}

// ------------------------------------------------------------------------------------------------ forward
namespace {

struct Runner {
  wfl_model* m;
  Plan p;
  char* ws;
  hipStream_t s;
  int rc = 0;

  bf16_t* buf(long off) const { return (bf16_t*)(ws + off); }
  // Ragged batches (WavLM / mel front-ends with per-clip lengths): clipT = [B] frames of every clip at the encoder's rate,
  // levelT[i] = the same at level i of the WavLM conv stack (both in the workspace, written by clip_frames_kernel); null otherwise.
  // Every kernel that stores frame rows skips rows t >= clipT[b]; the shared buffers are zeroed whole at the start of the forward,
  // so those rows stay zero like the halo rows do, and what a clip sees is exactly what it sees when it is labelled alone.
  const int* clipT = nullptr;
  const int* levelT[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int* clip_T_for(int P) const {
    if (!clipT) return nullptr;
    if (P == p.P) return clipT;
    for (int i = 0; i < p.nlev; ++i) if (p.Pl[i] == P) return levelT[i];
    return nullptr;
  }
  // Residual stream hi + lo (common.h, GemmArgs::res_lo): X and Y have low halves; lo_ok says whether the low half of the
  // tensor currently held in X / Y is valid (a kernel that writes only the high half invalidates it).
  // "model.precision: high" (precise.hip): EVERY activation buffer has a low half, lo_delta bytes further on, and lo_ok tracks the
  // buffers whose producer wrote it (X, Y, ATT, QK, FF, mel, c1, enc2, QKp, ATTp, and WavLM's FA, FB, XG: indices 0 .. 12; any pointer
  // inside the buffer counts)
  bool precise() const { return m->a.precision != 0 && p.lo_delta > 0; }
  bool lo_ok[13] = {false, false, false, false, false, false, false, false, false, false, false, false, false};
  int lo_idx(const void* ptr) const {               // default mode -- 0: inside X's first row (a column offset is allowed), 1: Y, else -1
    if (precise()) {
      const long o = (const char*)ptr - ws;
      const bool wavlm = m->a.encoder_type == WFL_ENC_WAVLM;
      const long C0 = wavlm ? m->a.wavlm_conv_dim[0] : 0;
      const long st[13] = {p.X, p.Y, p.ATT, p.QK, p.FF, p.mel, p.c1, p.enc2, p.QKp, p.ATTp, p.FA, p.FB, p.XG};
      const long sz[13] = {p.R * p.d * 2, p.R * p.d * 2, p.R * p.d * 2, p.R * 3L * p.d * 2, p.R * (long)p.ffw * 2, p.mel > 0 || m->a.encoder_type == WFL_ENC_WHISPER ? p.R2 * m->a.n_mels * 2 + 1024 : 0,
                           m->a.encoder_type == WFL_ENC_WHISPER ? p.R2 * (long)p.d * 2 : 0, p.R * p.d * 2, p.da != p.d ? p.R * 3L * p.da * 2 : 0, p.da != p.d ? p.R * (long)p.da * 2 : 0,
                           wavlm ? p.Rl[0] * C0 * 2 : 0, wavlm ? p.Rl[1] * C0 * 2 : 0, wavlm ? (long)m->a.wavlm_pos_conv_groups * p.R * 64 * 2 : 0};
      for (int i = 0; i < 13; ++i)
        if (sz[i] > 0 && o >= st[i] && o < st[i] + sz[i]) return i;
      return -1;
    }
    const long dx = (const char*)ptr - (ws + p.X), dy = (const char*)ptr - (ws + p.Y);
    if (dx >= 0 && dx < (long)p.d * 2) return 0;
    if (dy >= 0 && dy < (long)p.d * 2) return 1;
    return -1;
  }
  bf16_t* lo_of(const void* ptr) const {
    const int i = lo_idx(ptr);
    if (i < 0) return nullptr;
    if (precise()) return (bf16_t*)((char*)ptr + p.lo_delta);
    return (bf16_t*)(ws + (i == 0 ? p.Xlo : p.Ylo) + ((const char*)ptr - (ws + (i == 0 ? p.X : p.Y))));
  }
  const bf16_t* lo_in(const void* ptr) const { const int i = lo_idx(ptr); return (i >= 0 && lo_ok[i]) ? lo_of(ptr) : nullptr; }
  bool next_stats = false;      // the next gemm() (a residual launch) feeds a LayerNorm-folded GEMM: have it emit the row statistics
  bool next_lo_out = false;     // the next gemm() produces a residual-stream tensor: keep its low half
  bool next_acc_f32 = false;    // the next gemm() (fp32 output) adds to what is there
  double next_flops = -1.0;     // >= 0: algorithmic FLOPs to book for the next gemm() instead of 2 M N K
  int next_tap_wrap = 0;        // the next gemm()'s K holds three tap segments (GemmArgs::tap_wrap / seg_off)
  long next_seg_off = 0;
  // fp8 activations (GemmArgs::a8): the next gemm()'s A operand is e4m3 bytes with these scales / its output goes out as e4m3
  bool next_a8 = false;
  const float* next_a8_scale = nullptr;
  float next_a8_static = 1.f;
  unsigned char* next_c8 = nullptr;
  unsigned char* next_c8_lo = nullptr;            // gemm_mx.hip: e4m3 pair output
  const unsigned char* next_a8_lo = nullptr;      //   and pair input (next_a8_mode == 3)
  int next_a8_mode = 1;                           // GemmArgs::a8 of the next launch: 1 gemm_stream A8, 2 / 3 gemm_mx single / pair
  long next_ldc8 = 0;
  float next_c8_inv = 1.f;
  // LayerNorm statistics left behind by the last residual GEMM (gemm_stream.hip, STATS): valid for the rows of `stats_for`
  const void* stats_for = nullptr;
  int stats_nsl = 0;
  bool stats_in_next = false;   // the next gemm() call (a folded operand) reads them

  void gemm(const bf16_t* A, long lda, const Lin& W, int M, int P, int T, void* C, long ldc, long c_lead, int c_pitch,
            int act = WFL_ACT_NONE, const bf16_t* res = nullptr, long ldres = 0, float alpha = 1.f, int cin = 0,
            long tap_stride = 0, bool glu = false, bool out_f32 = false,
            const bf16_t* pos = nullptr, long ldpos = 0, const float* clip_bias = nullptr, const int* clip_idx = nullptr,
            int clip_ld = 0) {
    if (rc) return;
    if (precise() && !out_f32 && !W.w8 && !next_a8 && W.W_lo) {
      gemm_precise(A, lda, W, M, P, T, C, ldc, c_lead, c_pitch, act, res, ldres, alpha, cin, tap_stride, glu, pos, ldpos, clip_bias, clip_idx,
                   clip_ld);
      return;
    }
    if (precise() && out_f32 && !next_acc_f32 && act == WFL_ACT_NONE && W.W_lo && !W.w8 && !glu && !res && !pos && !clip_bias && !W.ln_s &&
        next_tap_wrap == 0) {
      // fp32 output (the BiLSTM's input projection): the two correction passes first -- A_hi W_lo^T, then A_lo W_hi^T added to it, no
      // bias -- and the plain launch below adds A_hi W_hi^T + b to them
      const bf16_t* A_lo = lo_in(A);
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !A_lo) break;
        GemmArgs g{};
        g.A = pass == 1 ? A_lo : A; g.lda = lda;
        g.cin = cin > 0 ? cin : W.K; g.tap_stride = tap_stride;
        g.W = pass == 0 ? W.W_lo : W.W; g.M = M; g.N = W.N; g.K = W.K; g.n_valid = W.n_valid;
        g.P = P; g.T = T; g.clip_T = clip_T_for(P);
        g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch;
        g.alpha = 1.f; g.act = WFL_ACT_NONE; g.out_f32 = 1; g.acc_f32 = pass;
        const int r = wfl_launch_gemm(g, s);
        if (r) { rc = fail(r, "gemm launch failed (precision high, fp32 output, pass " + std::to_string(pass) + ")"); return; }
      }
      next_acc_f32 = true;
    }
    GemmArgs g{};
    g.ln_s = W.ln_s; g.ln_eps = 1e-5f;
    g.w8_scale = W.w8;
    if (W.ln_s && stats_in_next) { g.stats_in = (const float*)(ws + p.stats); g.stats_nsl = stats_nsl; g.stats_lead = p.lead; }
    stats_in_next = false;
    g.A = A; g.lda = lda;
    g.cin = cin > 0 ? cin : W.K; g.tap_stride = tap_stride;
    g.tap_wrap = next_tap_wrap; g.seg_off = next_seg_off;
    next_tap_wrap = 0; next_seg_off = 0;
    g.W = W.W; g.M = M; g.N = W.N; g.K = W.K; g.n_valid = W.n_valid;
    g.P = P; g.T = T;
    g.clip_T = clip_T_for(P);
    g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch;
    g.bias = W.bias; g.clip_bias = clip_bias; g.clip_idx = clip_idx; g.clip_ld = clip_ld;
    g.res = res; g.ldres = ldres; g.alpha = alpha;
    g.pos = pos; g.ldpos = ldpos;
    g.act = act; g.glu = glu ? 1 : 0; g.out_f32 = out_f32 ? 1 : 0;
    g.acc_f32 = (out_f32 && next_acc_f32) ? 1 : 0;
    if (next_a8) {
      g.a8 = next_a8_mode; g.a8_lo = next_a8_lo; g.a8_scale = next_a8_scale; g.a8_lead = p.lead; g.a8_static = next_a8_static;
      g.c8 = next_c8; g.c8_lo = next_c8_lo; g.ldc8 = next_ldc8; g.c8_inv_scale = next_c8_inv;
      g.err = (unsigned*)(ws + p.err);
      next_a8 = false; next_a8_scale = nullptr; next_c8 = nullptr; next_a8_lo = nullptr; next_c8_lo = nullptr;
    }
    if (!out_f32 && !glu && (res || next_lo_out)) g.c_lo = lo_of(C);
    if (res) g.res_lo = lo_in(res);
    { const int ci = out_f32 ? -1 : lo_idx(C); if (ci >= 0) lo_ok[ci] = g.c_lo != nullptr; }
    const double flops_booked = next_flops;
    next_lo_out = next_acc_f32 = false;
    next_flops = -1.0;
    if (C == stats_for) stats_for = nullptr;                       // the rows they describe are being overwritten
    const bool want_stats = next_stats;
    next_stats = false;
    if (want_stats && res && !out_f32 && !glu && act == WFL_ACT_NONE && ldc == p.d && W.n_valid == p.d && c_lead == p.lead &&
        c_pitch == p.P && P == p.P && ln_fold_mode() == 1 && !W.w8) {
      g.stats_out = (float*)(ws + p.stats);
      if (wfl_gemm_stream_takes(g)) { stats_for = C; stats_nsl = W.N / 256; }
      else g.stats_out = nullptr;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->prof_on) {
      if (m->prof_used >= m->prof.ev.size()) {
        hipEvent_t a_, b_;
        if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { rc = fail(-10, "hipEventCreate"); return; }
        m->prof.ev.push_back({a_, b_});
      }
      e0 = m->prof.ev[m->prof_used].first; e1 = m->prof.ev[m->prof_used].second;
      ++m->prof_used;
      (void)hipEventRecord(e0, s);
    }
    const int r = wfl_launch_gemm(g, s);
    if (m->prof_on) {
      (void)hipEventRecord(e1, s);
      const int key = (act & 3) | (glu ? 4 : 0) | (out_f32 ? 8 : 0) | (res ? 16 : 0) | ((g_wfl_gemm_kernel_id & 7) << 5) |
                      ((g.ln_s ? (g.stats_in ? 2 : 1) : 0) << 8) | (g.stats_out ? 1024 : 0);
      m->prof.key.push_back(key);
      m->prof.launches[key] += 1;
      m->prof.flops[key] += flops_booked >= 0.0 ? flops_booked : 2.0 * (double)(M / P) * T * (double)W.n_valid * (double)W.K;
    }
    if (r) rc = fail(r, "gemm launch failed (" + std::to_string(r) + ")");
  }

  // "model.precision: high": the same Linear / Conv1d as three bf16 passes over split operands, summed in fp32, then the layer's
  // epilogue as its own kernel (precise.hip).  A's low half is used when its producer wrote it (lo_ok).
  void gemm_precise(const bf16_t* A, long lda, const Lin& W, int M, int P, int T, void* C, long ldc, long c_lead, int c_pitch, int act,
                    const bf16_t* res, long ldres, float alpha, int cin, long tap_stride, bool glu, const bf16_t* pos, long ldpos,
                    const float* clip_bias, const int* clip_idx, int clip_ld) {
    next_lo_out = next_acc_f32 = next_stats = false;
    next_flops = -1.0;
    stats_in_next = false;
    if (C == stats_for) stats_for = nullptr;
    const int B = M / P;
    const bf16_t* A_lo = lo_in(A);
    // One launch instead of three + the finish kernel, whenever the operand has its low half and the fused epilogues can do what the
    // layer needs: K' = 3 K with the weights packed [W_hi | W_hi | W_lo] against the taps [A_hi | A_lo | A_hi] (GemmArgs::tap_wrap; the
    // low half of A is lo_delta away), the accumulator never leaves the registers, the epilogue writes hi AND lo.  WFL_PRECISE_FUSED=0
    // keeps the three-launch form (A/B runs); the positional-table launch (the Whisper stem's conv2) joined it in round 4 (below).
    static int fused_on = -1;
    if (fused_on < 0) { const char* e = getenv("WFL_PRECISE_FUSED"); fused_on = e ? atoi(e) : 1; }
    GemmArgs g{};
    if (fused_on && W.W3 && A_lo && lo_of(C)) {
      g.A = A; g.lda = lda;
      g.cin = cin > 0 ? cin : W.K; g.tap_stride = tap_stride;
      g.tap_wrap = W.K / g.cin; g.seg_off = (long)(A_lo - A);
      g.W = W.W3; g.M = M; g.N = W.N; g.K = 3 * W.K; g.n_valid = W.n_valid;
      g.P = P; g.T = T; g.clip_T = clip_T_for(P);
      g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch;
      g.bias = W.bias; g.clip_bias = clip_bias; g.clip_idx = clip_idx; g.clip_ld = clip_ld;
      g.res = res; g.ldres = ldres; g.alpha = alpha; g.res_lo = res ? lo_in(res) : nullptr;
      g.act = act; g.glu = glu ? 1 : 0; g.ln_eps = 1e-5f;
      g.c_lo = lo_of(C);
      // the positional-table launch (the Whisper stem's conv2): one launch too when the slice-by-slice walk takes it -- its epilogue adds
      // the table as a pair; any other kernel would drop the low half, so those shapes keep the three-launch form below
      if (pos) { g.pos = pos; g.ldpos = ldpos; g.pos_lo = (pos == m->pos) ? m->pos_lo : nullptr; }
    }
    static int pos_fused = -1;                          // WFL_PRECISE_POS_FUSED=0: the table launch in the three-launch form (A/B runs)
    if (pos_fused < 0) { const char* e = getenv("WFL_PRECISE_POS_FUSED"); pos_fused = e ? atoi(e) : 1; }
    if (g.W && (!pos || (pos_fused && g.pos_lo && !glu && wfl_gemm256_tri_takes(g) && !wfl_gemm_stream_takes(g)))) {
      { const int ci = lo_idx(C); if (ci >= 0) lo_ok[ci] = true; }
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (m->prof_on) {
        if (m->prof_used >= m->prof.ev.size()) {
          hipEvent_t a_, b_;
          if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { rc = fail(-10, "hipEventCreate"); return; }
          m->prof.ev.push_back({a_, b_});
        }
        e0 = m->prof.ev[m->prof_used].first; e1 = m->prof.ev[m->prof_used].second;
        ++m->prof_used;
        (void)hipEventRecord(e0, s);
      }
      const int r = wfl_launch_gemm(g, s);
      if (m->prof_on) {
        (void)hipEventRecord(e1, s);
        const int key = (act & 3) | (glu ? 4 : 0) | (res ? 16 : 0) | ((g_wfl_gemm_kernel_id & 7) << 5);
        m->prof.key.push_back(key);
        m->prof.launches[key] += 1;
        m->prof.flops[key] += 3.0 * 2.0 * (double)B * T * (double)W.n_valid * (double)W.K;
      }
      if (r) rc = fail(r, "gemm launch failed (precision high, one-launch form: " + std::to_string(r) + ")");
      return;
    }
    if ((long)M * W.N > p.hp32_floats) { rc = fail(-1, "precision high: the fp32 accumulator is too small for this launch"); return; }
    float* acc = (float*)(ws + p.hp32);
    for (int pass = 0; pass < 3; ++pass) {
      if (pass == 2 && !A_lo) break;
      GemmArgs g{};
      g.A = pass == 2 ? A_lo : A; g.lda = lda;
      g.cin = cin > 0 ? cin : W.K; g.tap_stride = tap_stride;
      g.W = pass == 1 ? W.W_lo : W.W; g.M = M; g.N = W.N; g.K = W.K; g.n_valid = W.N;
      g.P = P; g.T = T; g.clip_T = clip_T_for(P);
      g.C = acc; g.ldc = W.N; g.c_lead = 0; g.c_pitch = P;
      g.alpha = 1.f; g.act = WFL_ACT_NONE; g.out_f32 = 1; g.acc_f32 = pass > 0 ? 1 : 0;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (m->prof_on) {
        if (m->prof_used >= m->prof.ev.size()) {
          hipEvent_t a_, b_;
          if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { rc = fail(-10, "hipEventCreate"); return; }
          m->prof.ev.push_back({a_, b_});
        }
        e0 = m->prof.ev[m->prof_used].first; e1 = m->prof.ev[m->prof_used].second;
        ++m->prof_used;
        (void)hipEventRecord(e0, s);
      }
      const int r = wfl_launch_gemm(g, s);
      if (m->prof_on) {
        (void)hipEventRecord(e1, s);
        const int key = 8 | ((g_wfl_gemm_kernel_id & 7) << 5);
        m->prof.key.push_back(key);
        m->prof.launches[key] += 1;
        m->prof.flops[key] += 2.0 * (double)B * T * (double)W.n_valid * (double)W.K;
      }
      if (r) { rc = fail(r, "gemm launch failed (precision high, pass " + std::to_string(pass) + ": " + std::to_string(r) + ")"); return; }
    }
    PreciseFinishArgs f{};
    f.acc = acc; f.ld_acc = W.N; f.B = B; f.P = P; f.T = T;
    f.glu = glu ? 1 : 0;
    f.n_out = glu ? W.n_valid / 2 : W.n_valid;
    f.bias = W.bias; f.clip_bias = clip_bias; f.clip_idx = clip_idx; f.clip_ld = clip_ld;
    f.act = act; f.alpha = alpha; f.pos = pos; f.ldpos = ldpos;
    f.pos_lo = (pos && pos == m->pos) ? m->pos_lo : nullptr;
    f.res = res; f.res_lo = res ? lo_in(res) : nullptr; f.ldres = ldres;
    f.out = (bf16_t*)C; f.out_lo = lo_of(C); f.ldc = ldc; f.c_lead = c_lead; f.c_pitch = c_pitch;
    f.clip_T = clip_T_for(P);
    { const int ci = lo_idx(C); if (ci >= 0) lo_ok[ci] = f.out_lo != nullptr; }
    if (f.n_out % 8) { rc = fail(-1, "precision high: output width must be a multiple of 8"); return; }
    const int r = wfl_launch_precise_finish(f, s);
    if (r) rc = fail(r, "precise_finish launch failed");
  }

  // WFL_LN_FOLD: 1 (default) fold a LayerNorm into the GEMM that consumes it whenever the GEMM that produced its input left
  // the row statistics behind; 2 fold always, summing the statistics inside the consumer (slower: tools/gemm_lab.py qkvLN);
  // 0 never (LayerNorm kernel + plain GEMM).
  static int ln_fold_mode() {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("WFL_LN_FOLD"); mode = e ? atoi(e) : 1; }
    return mode;
  }
  // y = act(LayerNorm(x) W^T + b)
  void ln_gemm(const bf16_t* x, bf16_t* scratch, const LNp& w, const Lin& plain, const Lin& folded, int M, void* C, long ldc,
               int act) {
    if (rc) return;
    const int mode = (m->dv != p.d || precise()) ? 0 : ln_fold_mode();   // (the folded form divides by K: not for a zero-padded width;
                                                                          //  precision high: the LayerNorm kernel writes hi + lo)
    if (folded.ln_s && (mode == 2 || (mode == 1 && stats_for == x))) {
      GemmArgs g{};
      g.M = M; g.N = folded.N; g.K = folded.K; g.cin = folded.K; g.n_valid = folded.n_valid; g.act = act; g.ln_s = folded.ln_s;
      if (mode == 1) { g.stats_in = (const float*)(ws + p.stats); g.stats_nsl = stats_nsl; }
      if (wfl_gemm_stream_takes(g)) {
        stats_in_next = mode == 1;
        gemm(x + (long)p.lead * p.d, p.d, folded, M, p.P, p.T, C, ldc, p.lead, p.P, act);
        return;
      }
    }
    ln(x, scratch, w);
    gemm(scratch + (long)p.lead * p.d, p.d, plain, M, p.P, p.T, C, ldc, p.lead, p.P, act);
  }

  // lo_out: the output is (or may become) a residual-stream tensor -- keep its low half; the input's low half is read when valid
  void ln(const bf16_t* x, bf16_t* y, const LNp& w, bool lo_out = false) {
    if (rc) return;
    if (y == stats_for) stats_for = nullptr;
    const bf16_t* x_lo = lo_in(x);
    bf16_t* y_lo = (lo_out || precise()) ? lo_of(y) : nullptr;
    { const int yi = lo_idx(y); if (yi >= 0) lo_ok[yi] = y_lo != nullptr; }
    prof_begin();
    const int r = wfl_launch_layernorm_act(x, p.d, y, p.d, w.g, w.b, 1e-5f, p.lead, p.B, p.P, p.T, p.d, 0, s, x_lo, y_lo, m->dv, clipT);
    prof_end(2043, 0.0);
    if (r) rc = fail(r, "layernorm launch failed");
  }

  // Timing hook for the non-GEMM kernel families (keys 2040 attention, 2041 BiLSTM recurrence, 2042 log-mel, 2043 LayerNorm)
  hipEvent_t prof_e1 = nullptr;
  void prof_begin() {
    prof_e1 = nullptr;
    if (!m->prof_on || rc) return;
    if (m->prof_used >= m->prof.ev.size()) {
      hipEvent_t a_, b_;
      if (hipEventCreate(&a_) != hipSuccess || hipEventCreate(&b_) != hipSuccess) { rc = fail(-10, "hipEventCreate"); return; }
      m->prof.ev.push_back({a_, b_});
    }
    (void)hipEventRecord(m->prof.ev[m->prof_used].first, s);
    prof_e1 = m->prof.ev[m->prof_used].second;
    ++m->prof_used;
  }
  void prof_end(int key, double flops) {
    if (!prof_e1) return;
    (void)hipEventRecord(prof_e1, s);
    m->prof.key.push_back(key);
    m->prof.launches[key] += 1;
    m->prof.flops[key] += flops;
    prof_e1 = nullptr;
  }

  // padded: the Conformer attention at width p.da != p.d (head size rounded up to a built one): q | k | v rows in QKp, context in ATTp
  void attn(int heads, const float* bias = nullptr, const float* gate = nullptr, bool padded = false, unsigned char* o8 = nullptr,
            long ldo8 = 0, float o8_scale = 1.f, unsigned char* o8_lo = nullptr) {
    if (rc) return;
    AttnArgs a{};
    a.bias = bias; a.gate = gate;
    a.O8 = o8; a.O8_lo = o8_lo; a.ldo8 = ldo8; a.o8_scale = o8_scale;
    a.err = (unsigned*)(ws + p.err);
    if (precise() && !o8) {                               // the context's low half for the out-projection's third pass
      bf16_t* o_hi = buf(padded ? p.ATTp : p.ATT);
      a.O_lo = lo_of(o_hi);
      const int oi = lo_idx(o_hi);
      if (oi >= 0) lo_ok[oi] = a.O_lo != nullptr;
    }
    const int w = padded ? p.da : p.d;
    bf16_t* qk = buf(padded ? p.QKp : p.QK);
    static const bool split_attn = std::getenv("WFL_SPLIT_ATTN") == nullptr;       // (set to anything: bf16 q, k, v, P in precision high -- A/B runs)
    if (precise() && !o8 && split_attn)
      if (const bf16_t* ql = lo_in(qk)) { a.QK_lo = ql; a.V_lo = ql + 2 * w; }     // q | k | v as hi + lo (their projection wrote both halves)
    a.QK = qk; a.ldqk = 3 * w; a.lead = p.lead; a.V = qk + 2 * w; a.ldv = 3 * w; a.O = buf(padded ? p.ATTp : p.ATT); a.ldo = w;
    a.B = p.B; a.T = p.T; a.P = p.P; a.heads = heads; a.d = w;
    a.clip_T = clipT;
    prof_begin();
    const int r = wfl_launch_attention(a, s);
    prof_end(2040, 4.0 * (double)p.B * p.T * (double)p.T * (double)p.d);
    if (r) rc = fail(r, "attention launch failed (" + std::to_string(r) + "; head_dim " + std::to_string(w / heads) + ")");
  }

  ZeroMulti zm{};
  void zero_add(long off, long ld_elems, long lead, int P, int T, long tail) {     // queued; zero_flush() launches once
    const int k = zm.n++;
    zm.buf[k] = ws + off; zm.ld_bytes[k] = ld_elems * 2; zm.lead[k] = lead; zm.P[k] = P; zm.T[k] = T;
    zm.tail_rows[k] = (long)(P - T) + tail;
  }
  void zero_flush() {
    if (rc || zm.n == 0) return;
    zm.B = p.B;
    const int r = wfl_launch_zero_halo_multi(zm, s);
    zm.n = 0;
    if (r) rc = fail(r, "zero_halo launch failed");
  }

  void zero(long off, long ld_elems, long lead, int P, int T, long tail) {
    if (rc) return;
    // `tail` = rows behind the last clip's pitch; the kernel counts from the last clip's last valid frame
    const int r = wfl_launch_zero_halo(buf(off), ld_elems * 2, lead, p.B, P, T, (long)(P - T) + tail, s);
    if (r) rc = fail(r, "zero_halo launch failed");
  }
};

}  // namespace

static int run_logmel(wfl_model* m, const Plan& p, char* ws, const float* wav, long ldw, const int* lens, float* ref_out,
                      hipStream_t s) {
  LogmelArgs a{};
  a.wav = wav; a.ldw = ldw; a.lens = lens; a.L = p.L; a.B = p.B;
  a.n_frames = p.T2; a.n_samples = p.T2 * 160; a.n_mels = m->a.n_mels;
  a.Wc = m->Wc; a.Ws = m->Ws; a.mel_lo = m->mel_lo; a.mel_cnt = m->mel_cnt; a.mel_w = m->mel_w; a.mel_maxw = m->mel_maxw;
  a.raw = (float*)(ws + p.raw); a.clipmax = (unsigned*)(ws + p.clipmax);
  bf16_t* mel_lo = (m->a.precision && p.lo_delta > 0) ? (bf16_t*)(ws + p.mel + p.lo_delta) : nullptr;
  return wfl_launch_logmel(a, (bf16_t*)(ws + p.mel), m->a.n_mels, p.lead2, p.P2, ref_out, s, mel_lo);
}

int32_t wfl_logmel(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L, float* out,
                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (!m || !m->finalized) return fail(-1, "wfl_logmel: model not finalized");
  if (m->a.encoder_type != WFL_ENC_WHISPER) return fail(-1, "wfl_logmel: not a Whisper model");
  if (B <= 0 || L < 0 || ldw < L) return fail(-1, "wfl_logmel: bad shape");
  if (int r = check_device(m, "wfl_logmel")) return r;
  const Plan p = make_plan(m, B, L);
  if (workspace_bytes < p.total) return fail(-1, "wfl_logmel: workspace too small");
  const int r = run_logmel(m, p, (char*)workspace, wav, ldw, lens, out, (hipStream_t)stream);
  return r ? fail(r, "logmel launch failed") : 0;
}

// Halo rows of the buffers every stage shares (cheap; keeps the layout invariant independent of the workspace's history) and
// the forward's error word.
static int begin_forward(Runner& R, bool with_encoder) {
  const wfl_arch& a = R.m->a;
  const Plan& p = R.p;
  const int d = p.d;
  R.zm.err_word = (unsigned*)(R.ws + p.err);      // cleared by the halo kernel (a kernel, not a memset node: graph replay)
  R.zero_add(p.X, d, p.lead, p.P, p.T, p.tail);
  R.zero_add(p.Y, d, p.lead, p.P, p.T, p.tail);
  R.zero_add(p.ATT, d, p.lead, p.P, p.T, p.tail);
  R.zero_add(p.QK, 3 * d, p.lead, p.P, p.T, p.tail);
  R.zero_add(p.FF, p.ffw, p.lead, p.P, p.T, p.tail);
  if (p.da != d) {
    R.zero_add(p.QKp, 3 * p.da, p.lead, p.P, p.T, p.tail);
    R.zero_add(p.ATTp, p.da, p.lead, p.P, p.T, p.tail);
  }
  if (with_encoder && a.encoder_type == WFL_ENC_WHISPER) {
    R.zero_add(p.mel, a.n_mels, p.lead2, p.P2, p.T2, 2 * p.tail);
    R.zero_add(p.c1, d, p.lead2, p.P2, p.T2, 2 * p.tail);
  }
  R.zero_flush();
  if (R.precise()) {                               // the low halves' halo rows are taps of the split-precision convolutions too
    const long D = p.lo_delta;
    R.zero_add(p.X + D, d, p.lead, p.P, p.T, p.tail);
    R.zero_add(p.Y + D, d, p.lead, p.P, p.T, p.tail);
    R.zero_add(p.ATT + D, d, p.lead, p.P, p.T, p.tail);
    R.zero_add(p.QK + D, 3 * d, p.lead, p.P, p.T, p.tail);
    R.zero_add(p.FF + D, p.ffw, p.lead, p.P, p.T, p.tail);
    if (p.da != d) {
      R.zero_add(p.QKp + D, 3 * p.da, p.lead, p.P, p.T, p.tail);
      R.zero_add(p.ATTp + D, p.da, p.lead, p.P, p.T, p.tail);
    }
    if (with_encoder && a.encoder_type == WFL_ENC_WHISPER) {
      R.zero_add(p.mel + D, a.n_mels, p.lead2, p.P2, p.T2, 2 * p.tail);
      R.zero_add(p.c1 + D, d, p.lead2, p.P2, p.T2, 2 * p.tail);
    }
    R.zm.err_word = nullptr;                       // (cleared by the first flush)
    R.zero_flush();
  }
  // a zero-padded width (encoder_type none): the front-end / wfl_head write only the valid columns of the Y rows
  if (!R.rc && R.m->dv != d && wfl_launch_fill_i32((int*)(R.ws + p.Y), p.R * d / 2, 0, R.s)) return fail(-3, "fill launch failed");
  return R.rc;
}

// Ragged batches: every row of the buffers the stages share starts at zero (begin_forward zeroed their halo rows only); producers skip
// the rows behind a clip's own frame count, so those stay zero for the whole forward -- for the convolutions' taps and the positional
// conv's padding they are the same zeros the halo rows are.
static int zero_shared_buffers(Runner& R) {
  const Plan& p = R.p;
  const long d = p.d;
  int r = wfl_launch_fill_i32((int*)(R.ws + p.X), p.R * d / 2, 0, R.s);
  if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.Y), p.R * d / 2, 0, R.s);
  if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.ATT), p.R * d / 2, 0, R.s);
  if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.QK), p.R * 3 * d / 2, 0, R.s);
  if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.FF), p.R * (long)p.ffw / 2, 0, R.s);
  if (!r && p.da != p.d) {
    r = wfl_launch_fill_i32((int*)(R.ws + p.QKp), p.R * 3L * p.da / 2, 0, R.s);
    if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.ATTp), p.R * (long)p.da / 2, 0, R.s);
  }
  if (!r && R.precise()) {                         // ... and their low halves
    const long D = p.lo_delta;
    r = wfl_launch_fill_i32((int*)(R.ws + p.X + D), p.R * d / 2, 0, R.s);
    if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.Y + D), p.R * d / 2, 0, R.s);
    if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.ATT + D), p.R * d / 2, 0, R.s);
    if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.QK + D), p.R * 3 * d / 2, 0, R.s);
    if (!r) r = wfl_launch_fill_i32((int*)(R.ws + p.FF + D), p.R * (long)p.ffw / 2, 0, R.s);
  }
  return r ? fail(r, "fill launch failed") : 0;
}

// Feature extractor + encoder (model.py:149-161): leaves the encoder output in the Y rows.
static int run_encoder(Runner& R, const float* wav, int64_t ldw, const int32_t* lens) {
  wfl_model* m = R.m;
  const wfl_arch& a = m->a;
  const Plan& p = R.p;
  const int B = p.B, L = p.L, d = p.d;
  const long Mrows = (long)B * p.P;
  bf16_t *X = R.buf(p.X), *Y = R.buf(p.Y), *ATT = R.buf(p.ATT), *QK = R.buf(p.QK), *FF = R.buf(p.FF);
  if (a.encoder_type == WFL_ENC_WHISPER) {
    // ---- Whisper encoder (HF modeling_whisper.py:618-642)
    R.prof_begin();
    const int r = run_logmel(m, p, R.ws, wav, ldw, lens, nullptr, R.s);
    R.prof_end(2042, 0.96e9 * (double)B * (a.n_mels / 80.0));
    if (r) return fail(r, "logmel launch failed");
    if (R.precise()) R.lo_ok[5] = true;                // (the log-mel kernel wrote the features' low half)
    bf16_t* mel = R.buf(p.mel);
    bf16_t* c1 = R.buf(p.c1);
    // conv1 k3 p1: frame t reads mel rows t-1..t+1 = 3*n_mels contiguous channels
    R.gemm(mel + (long)(p.lead2 - 1) * a.n_mels, a.n_mels, m->conv1, B * p.P2, p.P2, p.T2, c1, d, p.lead2, p.P2, WFL_ACT_GELU);
    // conv2 k3 s2 p1: frame t reads c1 rows 2t-1..2t+1; pitch(c1) = 2 * pitch(X) makes it one flat GEMM with lda = 2d
    R.next_lo_out = true;                            // the residual stream starts here
    R.gemm(c1 + (long)(p.lead2 - 1) * d, 2 * d, m->conv2, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_GELU, nullptr, 0,
           1.f, 0, 0, false, false, m->pos, d);
    // fp8 x fp8 (BASELINE configs[4]; WFL_FP8_ACT=0: fp8 weights with bf16 activations, round 2's form): every GEMM operand of a layer
    // is e4m3 -- LayerNorm outputs and the attention context with one scale per row (norm.hip), fc1's GELU output straight from its
    // epilogue with a fixed scale -- and the four GEMMs run on v_mfma_f32_16x16x32_fp8_fp8 (gemm_stream.hip, A8)
    // Round 4: e4m3 activations are an explicit opt-in (wfl_arch::fp8_activations, "model.activation_dtype: fp8"); the default for an
    // fp8-weight model is bf16 activations, which holds the reference's arithmetic on the fp8-rounded checkpoint.  WFL_FP8_ACT=0 / 1
    // overrides either way (A/B runs).
    static int fp8_act_env = -2;
    if (fp8_act_env == -2) { const char* e = getenv("WFL_FP8_ACT"); fp8_act_env = e ? atoi(e) : -1; }
    const int fp8_act = fp8_act_env >= 0 ? fp8_act_env : a.fp8_activations;
    const bool act8 = a.fp8_weights && fp8_act && p.X8 > 0 && p.ffw == a.enc_ffn;
    // fp8_act: 1 single e4m3 activations on gemm_stream's non-scaled fp8 MFMA (round 3); 2 the same on the block-scaled K = 128 MFMA
    // (gemm_mx.hip, twice the rate); 3 e4m3 PAIRS hi + lo on that MFMA -- eight significant bits per activation, the parity of bf16
    // activations at the fp8 MFMA's price per pass
    const bool pair8 = fp8_act == 3;
    const int a8_mode = fp8_act == 1 ? 1 : (pair8 ? 3 : 2);
    for (int i = 0; act8 && i < a.enc_layers; ++i) {
      const EncLayer& L_ = m->enc[i];
      unsigned char* X8 = (unsigned char*)(R.ws + p.X8);
      unsigned char* FF8 = (unsigned char*)(R.ws + p.FF8);
      unsigned char* X8lo = pair8 ? (unsigned char*)(R.ws + p.X8lo) : nullptr;
      unsigned char* FF8lo = pair8 ? (unsigned char*)(R.ws + p.FF8lo) : nullptr;
      float* rs8 = (float*)(R.ws + p.rs8);
      // fixed scales of the two operands whose producer cannot see a whole row: single e4m3 stores e4m3(8 x) (|x| <= 56, normal numbers
      // down to 2^-9); a pair carries eight significant bits, so e4m3(4 x) + lo covers |x| <= 112 down to 2^-13.  A value beyond the
      // range sets bit 1 of the status word (GemmArgs::err / AttnArgs::err): never a silent clip.
      const float ff_scale = pair8 ? 4.0f : 8.0f;
      const float att_scale = pair8 ? 4.0f : 8.0f;
      auto rows8 = [&](const bf16_t* x, const bf16_t* x_lo, const LNp* w) {
        if (R.rc) return;
        R.prof_begin();
        const int r = wfl_launch_rows_fp8(x, d, x_lo, w ? w->g : nullptr, w ? w->b : nullptr, 1e-5f, p.lead, B, p.P, p.T, d, X8, d, rs8, R.s, X8lo);
        R.prof_end(2043, 0.0);
        if (r) R.rc = fail(r, "rows_fp8 launch failed");
      };
      auto next_in = [&](const float* row_scale, float stat, const unsigned char* lo_plane) {
        R.next_a8 = true; R.next_a8_mode = a8_mode; R.next_a8_scale = row_scale; R.next_a8_static = stat; R.next_a8_lo = lo_plane;
      };
      rows8(X, R.lo_in(X), &L_.ln1);
      next_in(rs8, 1.f, X8lo ? X8lo + (long)p.lead * d : nullptr);
      R.gemm((const bf16_t*)(X8 + (long)p.lead * d), d, L_.qkv, (int)Mrows, p.P, p.T, QK, 3 * d, p.lead, p.P, WFL_ACT_NONE);
      // the attention context leaves the attention kernel as e4m3 (a pair in mode 3) with a fixed scale (head_dim 64; other head sizes:
      // bf16, then quantised per row): a context row is a convex combination of value rows, |x| <= max |v|
      if (d / a.enc_heads == 64) {
        R.attn(a.enc_heads, nullptr, nullptr, false, X8, d, att_scale, X8lo);
        next_in(nullptr, 1.0f / att_scale, X8lo ? X8lo + (long)p.lead * d : nullptr);
      } else {
        R.attn(a.enc_heads);
        rows8(ATT, nullptr, nullptr);
        next_in(rs8, 1.f, X8lo ? X8lo + (long)p.lead * d : nullptr);
      }
      R.gemm((const bf16_t*)(X8 + (long)p.lead * d), d, L_.out, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_NONE, X, d, 1.f);
      rows8(X, R.lo_in(X), &L_.ln2);
      next_in(rs8, 1.f, X8lo ? X8lo + (long)p.lead * d : nullptr);
      R.next_c8 = FF8; R.next_c8_lo = FF8lo; R.next_ldc8 = p.ffw; R.next_c8_inv = ff_scale;
      R.gemm((const bf16_t*)(X8 + (long)p.lead * d), d, L_.fc1, (int)Mrows, p.P, p.T, FF, p.ffw, p.lead, p.P, WFL_ACT_GELU);
      next_in(nullptr, 1.0f / ff_scale, FF8lo ? FF8lo + (long)p.lead * p.ffw : nullptr);
      R.gemm((const bf16_t*)(FF8 + (long)p.lead * p.ffw), p.ffw, L_.fc2, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_NONE, X, d, 1.f);
    }
    for (int i = 0; !act8 && i < a.enc_layers; ++i) {
      const EncLayer& L_ = m->enc[i];
      R.ln_gemm(X, Y, L_.ln1, L_.qkv, L_.qkv_ln, (int)Mrows, QK, 3 * d, WFL_ACT_NONE);
      R.attn(a.enc_heads);
      R.next_stats = true;                                            // (consumed by the folded fc1)
      R.gemm(ATT + (long)p.lead * d, d, L_.out, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_NONE, X, d, 1.f);
      R.ln_gemm(X, Y, L_.ln2, L_.fc1, L_.fc1_ln, (int)Mrows, FF, p.ffw, WFL_ACT_GELU);
      R.next_stats = i + 1 < a.enc_layers;                            // (the next layer's folded q|k|v; the final LayerNorm is a kernel)
      R.gemm(FF + (long)p.lead * p.ffw, p.ffw, L_.fc2, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_NONE, X, d, 1.f);
    }
    R.ln(X, Y, m->enc_ln, true);   // encoder output in Y (with lang_id None it is the head's residual stream)
  } else if (a.encoder_type == WFL_ENC_NONE) {
    // ---- no encoder: hidden = MelSpectrogram(wav).transpose(1, 2) (model.py:149-150).  Like WavLM input it is never padded by the
    // reference: one clip length per call.
    if (lens) {                                        // (every clip keeps its own frame count: Runner::clipT)
      int* ct = (int*)(R.ws + p.clipT);
      const int k0 = 0, s0 = a.mel_hop;
      const int cr = wfl_launch_clip_frames(lens, B, L, 1, &k0, &s0, 201, ct, R.s);
      if (cr) return fail(cr, "clip_frames launch failed");
      R.clipT = ct;
      if (int zr = zero_shared_buffers(R)) return zr;
    }
    if (L <= 200) return fail(-1, "wfl_forward: encoder_type none needs more than 200 samples (reflect padding of the STFT)");
    LogmelArgs la{};
    la.wav = wav; la.ldw = ldw; la.lens = lens; la.L = L; la.B = B;
    la.n_frames = p.T; la.n_samples = L; la.n_mels = a.n_mels;
    la.Wc = m->Wc; la.Ws = m->Ws; la.mel_lo = m->mel_lo; la.mel_cnt = m->mel_cnt; la.mel_w = m->mel_w; la.mel_maxw = m->mel_maxw;
    la.raw = (float*)(R.ws + p.raw); la.clipmax = nullptr;
    R.prof_begin();
    const int r = wfl_launch_melpower(la, a.mel_hop, Y, d, p.lead, p.P, m->pad_split(), m->pad_shift(), R.s);
    R.prof_end(2042, 0.0);
    if (r) return fail(r, "mel launch failed");
    R.lo_ok[1] = false;
    if (R.precise()) {                                 // the fp32 mel power once more, as hi + lo rows
      const int r2 = wfl_launch_f32_to_rows(la.raw, Y, d, p.lead, B, p.P, p.T, a.n_mels, R.s, m->pad_split(), m->pad_shift(), R.lo_of(Y));
      if (r2) return fail(r2, "f32_to_rows launch failed");
      R.lo_ok[1] = true;
    }
  } else {
    // ---- WavLM (HF modeling_wavlm.py:1032-1088).  The reference never pads WavLM input: with `lens` every clip keeps its own sample
    // and frame counts through the whole forward (Runner::clipT), i.e. what it gets when it is labelled alone.
    if (lens) {
      int* ct = (int*)(R.ws + p.clipT);
      const int cr = wfl_launch_clip_frames(lens, B, L, p.nlev, a.wavlm_conv_kernel, a.wavlm_conv_stride, 0, ct, R.s);
      if (cr) return fail(cr, "clip_frames launch failed");
      for (int i = 0; i < p.nlev; ++i) R.levelT[i] = ct + (long)i * B;
      R.clipT = R.levelT[p.nlev - 1];
      if (int zr = zero_shared_buffers(R)) return zr;
    }
    if (p.T > WAVLM_MAX_T) return fail(-1, "wfl_forward: clip too long for the WavLM relative-position table");
    const int C = a.wavlm_conv_dim[0], n = p.nlev;
    bf16_t* F[2] = {R.buf(p.FA), R.buf(p.FB)};
    double* wstats = nullptr;
    if (a.wavlm_do_normalize) {
      wstats = (double*)(R.ws + p.wstats);
      const int r = wfl_launch_wav_stats(wav, ldw, B, L, wstats, R.s, lens);
      if (r) return fail(r, "wav_stats launch failed");
    }
    // Rows that are not valid frames of a level are zeroed before the level is produced (the two buffers alternate
    // between levels; a K-padded conv GEMM may read a few channels past the last valid frame).
    const long D = R.precise() ? p.lo_delta : 0;       // precision high: the feature extractor's rows as bf16 pairs too
    R.zero(p.FA, C, p.leadl[0], p.Pl[0], p.Tl[0], p.tail);
    if (D) R.zero(p.FA + D, C, p.leadl[0], p.Pl[0], p.Tl[0], p.tail);
    if (R.rc) return R.rc;
    {
      Conv0Args c{};
      if (D) { c.out_lo = R.lo_of(F[0]); R.lo_ok[10] = c.out_lo != nullptr; }
      c.wav = wav; c.ldw = ldw; c.L = L; c.wstats = wstats; c.w = m->conv0_w; c.bias = m->conv0_b;
      c.gamma = m->conv0_norm.g; c.beta = m->conv0_norm.b; c.B = B; c.T0 = p.Tl[0]; c.C = C; c.lens = lens;
      c.cstats = (double*)(R.ws + p.cstats); c.cpart = (float*)(R.ws + p.cpart); c.out = F[0]; c.lead = p.leadl[0]; c.P = p.Pl[0];
      const int r = wfl_launch_conv0(c, a.wavlm_group_norm, R.s);
      if (r) return fail(r, "conv0 launch failed");
    }
    for (int i = 1; i < n; ++i) {
      // conv k, stride 2, no padding: output frame t reads input rows 2t .. 2t+k-1 = k*C contiguous channels, lda = 2C
      const bf16_t* in = F[(i - 1) & 1] + (long)p.leadl[i - 1] * C;
      bf16_t* out = F[i & 1];
      const bool group = a.wavlm_group_norm != 0;
      R.zero((i & 1) ? p.FB : p.FA, C, p.leadl[i], p.Pl[i], p.Tl[i], p.tail);
      if (D) R.zero(((i & 1) ? p.FB : p.FA) + D, C, p.leadl[i], p.Pl[i], p.Tl[i], p.tail);
      R.gemm(in, 2 * C, m->fconv[i - 1], B * p.Pl[i], p.Pl[i], p.Tl[i], out, C, p.leadl[i], p.Pl[i], group ? WFL_ACT_GELU : WFL_ACT_NONE);
      if (!group && !R.rc) {
        const int r = wfl_launch_layernorm_act(out, C, out, C, m->fconv_ln[i - 1].g, m->fconv_ln[i - 1].b, 1e-5f, p.leadl[i], B,
                                               p.Pl[i], p.Tl[i], C, 1, R.s, R.lo_in(out), D ? R.lo_of(out) : nullptr, 0, R.levelT[i]);
        if (r) return fail(r, "layernorm launch failed");
      }
    }
    if (R.rc) return R.rc;
    bf16_t* feats = F[(n - 1) & 1];                    // level n-1 has the main geometry (lead, P, T)
    {
      const int r = wfl_launch_layernorm_act(feats, C, feats, C, m->fp_ln.g, m->fp_ln.b, 1e-5f, p.lead, B, p.P, p.T, C, 0, R.s, R.lo_in(feats),
                                             D ? R.lo_of(feats) : nullptr, 0, R.clipT);
      if (r) return fail(r, "layernorm launch failed");
    }
    R.next_lo_out = true;
    R.gemm(feats + (long)p.lead * C, C, m->fp_proj, (int)Mrows, p.P, p.T, X, d, p.lead, p.P);
    // positional conv: x + GELU(grouped conv k, pad k/2, last step dropped), one contiguous-tap GEMM per group
    {
      const int G = a.wavlm_pos_conv_groups, cpg = d / G, K = a.wavlm_pos_conv_kernel;
      bf16_t* XG = R.buf(p.XG);
      if (!R.rc) {
        int r = wfl_launch_regroup(X, d, G, cpg, p.R, p.lead, B, p.P, p.T, XG, R.s, R.clipT);
        R.lo_ok[12] = false;
        if (!r && D && R.lo_in(X)) {                 // the low halves in the same layout: the per-group GEMMs' third pass
          r = wfl_launch_regroup(R.lo_in(X), d, G, cpg, p.R, p.lead, B, p.P, p.T, R.lo_of(XG), R.s, R.clipT);
          R.lo_ok[12] = true;
        }
        if (r) return fail(r, "regroup launch failed");
      }
      // all groups in one tap-stationary launch (posconv.hip); WFL_POSCONV_GEMM=1 keeps rounds 1-2's GEMM per group (A/B runs)
      static int as_gemm = -1;
      if (as_gemm < 0) { const char* e = getenv("WFL_POSCONV_GEMM"); as_gemm = e && atoi(e) ? 1 : 0; }
      int taken = 1;
      if (!as_gemm && !D && !R.rc && G <= 16) {      // (precision high: the GEMM per group, three passes each)
        PosConvArgs pc{};
        pc.xg = XG; pc.R = p.R; pc.lead = p.lead; pc.B = B; pc.P = p.P; pc.T = p.T; pc.groups = G; pc.cpg = cpg; pc.taps = K;
        for (int gi = 0; gi < G; ++gi) { pc.w[gi] = m->posconv[gi].W; pc.bias[gi] = m->posconv[gi].bias; }
        pc.ldw = m->posconv[0].K;
        pc.res = X; pc.res_lo = R.lo_in(X); pc.out = Y; pc.out_lo = R.lo_of(Y); pc.ld = d;
        pc.clip_T = R.clipT;
        R.prof_begin();
        taken = wfl_launch_posconv(pc, R.s);
        R.prof_end(2044, 2.0 * (double)B * p.T * (double)d * (double)cpg * (double)K);
        if (taken < 0) return fail(taken, "posconv launch failed");
        if (taken == 0) { R.lo_ok[1] = true; R.stats_for = nullptr; }
      }
      for (int gi = 0; gi < G && taken == 1; ++gi)
        R.gemm(XG + ((long)gi * p.R + p.lead - K / 2) * 64, 64, m->posconv[gi], (int)Mrows, p.P, p.T, Y + gi * cpg, d, p.lead, p.P,
               WFL_ACT_GELU, X + gi * cpg, d, 1.f);
    }
    bf16_t *H = Y, *S = X;                             // current hidden states / scratch
    const bool stable = a.wavlm_stable_layer_norm != 0;
    if (!stable) { R.ln(H, S, m->wenc_ln, true); std::swap(H, S); }
    float* gate = (float*)(R.ws + p.gate);
    float* rtab = (float*)(R.ws + p.rtab);
    if (!R.rc) {
      const int r = wfl_launch_relpos_table(m->rel_emb, m->bucket_of_delta, WAVLM_MAX_T, a.enc_heads, p.T, rtab, R.s);
      if (r) return fail(r, "relpos_table launch failed");
    }
    const int hd = d / a.enc_heads;
    for (int i = 0; i < a.enc_layers && !R.rc; ++i) {
      const WavlmLayer& L_ = m->wl[i];
      bf16_t* A_in = H;                                // what the attention block sees
      if (stable) { R.ln(H, S, L_.ln1); A_in = S; }
      if (R.rc) break;
      int r = wfl_launch_relpos_gate(A_in, d, p.lead, B, p.P, p.T, a.enc_heads, hd, L_.w8, L_.b8, L_.cst, gate, R.s, R.lo_in(A_in));
      if (r) return fail(r, "relpos_gate launch failed");
      R.gemm(A_in + (long)p.lead * d, d, L_.qkv, (int)Mrows, p.P, p.T, QK, 3 * d, p.lead, p.P);
      R.attn(a.enc_heads, rtab, gate);
      if (stable) {
        // x = x + attn; x = x + FFN(LN(x))
        R.next_stats = L_.fc1_ln.ln_s != nullptr;
        R.gemm(ATT + (long)p.lead * d, d, L_.out, (int)Mrows, p.P, p.T, H, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
        R.ln_gemm(H, S, L_.ln2, L_.fc1, L_.fc1_ln, (int)Mrows, FF, p.ffw, WFL_ACT_GELU);
        R.gemm(FF + (long)p.lead * p.ffw, p.ffw, L_.fc2, (int)Mrows, p.P, p.T, H, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
      } else {
        // x = LN(x + attn); x = LN_final(x + FFN(x))
        R.gemm(ATT + (long)p.lead * d, d, L_.out, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
        R.ln(S, H, L_.ln1, true);
        R.gemm(H + (long)p.lead * d, d, L_.fc1, (int)Mrows, p.P, p.T, FF, p.ffw, p.lead, p.P, WFL_ACT_GELU);
        R.gemm(FF + (long)p.lead * p.ffw, p.ffw, L_.fc2, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
        R.ln(S, H, L_.ln2, true);
      }
    }
    if (stable) { R.ln(H, S, m->wenc_ln, true); std::swap(H, S); }
    if (H != Y && !R.rc) {                             // the head expects the encoder output in Y
      R.stats_for = nullptr;
      if (wfl_launch_copy16(Y, H, p.R * d * 2, R.s)) return fail(-3, "copy launch failed");
      if (R.lo_in(H) && wfl_launch_copy16(R.lo_of(Y), R.lo_of(H), p.R * d * 2, R.s)) return fail(-3, "copy launch failed");
      R.lo_ok[1] = R.lo_in(H) != nullptr;
    }
  }
  return R.rc;
}

// The encoder output as the caller sees it: compact fp32 [B][T][hidden_size]
static int emit_hidden(Runner& R, float* hidden) {
  const wfl_model* m = R.m;
  const Plan& p = R.p;
  if (m->a.encoder_type == WFL_ENC_NONE)           // the mel power itself, before its rounding into the bf16 rows
    return wfl_launch_axpy(hidden, (const float*)(R.ws + p.raw), (long)p.B * p.T * m->a.n_mels, 1.f, 1, R.s);
  // (the default build hands out the high halves only: wfl_head rounds its input to bf16 again, and bf16(hi + lo) is not always hi --
  //  lo can round up to exactly half a unit of hi -- which would cost wfl_encode + wfl_head == wfl_forward its bit-exactness)
  return wfl_launch_rows_to_f32(R.buf(p.Y), p.d, p.lead, p.B, p.P, p.T, m->dv, hidden, R.s, m->pad_split(), m->pad_shift(),
                                R.precise() ? R.lo_in(R.buf(p.Y)) : nullptr);
}

// Head (model.py:176-194) + tag decision on the encoder output in the Y rows.
static int run_head(Runner& R, const int32_t* lang_id, int32_t lang_mode, float threshold, int32_t* ids, int32_t* argmax,
                    float* maxprob, float* offsets, float* logits, int32_t* status) {
  wfl_model* m = R.m;
  const wfl_arch& a = m->a;
  const Plan& p = R.p;
  const int B = p.B, d = p.d;
  const long Mrows = (long)B * p.P;
  bf16_t *X = R.buf(p.X), *Y = R.buf(p.Y), *ATT = R.buf(p.ATT), *QK = R.buf(p.QK), *FF = R.buf(p.FF);
  // ---- head (model.py:176-194); with WFL_LANG_AVERAGE it runs once per language on the same encoder output
  const int n_pass = lang_mode == WFL_LANG_AVERAGE ? (int)m->avg_langs.size() : 1;
  float* lg = logits ? logits : (float*)(R.ws + p.logits);
  bf16_t* ENC = Y;
  // the head needs Y as scratch: keep the encoder output in ATT when more than one pass reads it
  if (n_pass > 1) {
    if (wfl_launch_copy16(R.buf(p.enc2), Y, p.R * d * 2, R.s)) return fail(-3, "copy launch failed");
    ENC = R.buf(p.enc2);
    if (R.precise()) {                                   // ... and its low half
      const bool have_lo = R.lo_in(Y) != nullptr && R.lo_of(ENC) != nullptr;
      if (have_lo && wfl_launch_copy16(R.lo_of(ENC), R.lo_of(Y), p.R * d * 2, R.s)) return fail(-3, "copy launch failed");
      const int ei = R.lo_idx(ENC);
      if (ei >= 0) R.lo_ok[ei] = have_lo;
    }
  }
  int* lang_dev = nullptr;
  for (int pass = 0; pass < n_pass; ++pass) {
    bf16_t* H;      // current activation
    bf16_t* S;      // scratch of the same shape
    if (lang_mode == WFL_LANG_NONE) {
      H = Y; S = X;
    } else {
      const int* idx = lang_id;
      if (lang_mode == WFL_LANG_AVERAGE) {
        // clip_idx = this pass's language id for every clip: reuse the clipmax slot region (B ints) as a constant index vector
        lang_dev = (int*)(R.ws + p.clipmax);
        const int fr = wfl_launch_fill_i32(lang_dev, B, m->avg_langs[pass], R.s);
        if (fr) return fail(fr, "fill launch failed");
        idx = lang_dev;
      }
      R.next_lo_out = true;
      R.gemm(ENC + (long)p.lead * d, d, m->lang, (int)Mrows, p.P, p.T, X, d, p.lead, p.P, WFL_ACT_NONE, nullptr, 0, 1.f, 0, 0,
             false, false, nullptr, 0, m->lang_table, idx, d);
      H = X; S = Y;
    }
    if (a.enable_bilstm) {
      const int Hh = d / 2;
      float* GX = (float*)(R.ws + p.gx);
      for (int layer = 0; layer < a.bilstm_layers; ++layer) {
        R.gemm(H + (long)p.lead * d, d, m->lstm_in[layer], (int)Mrows, p.P, p.T, GX, 8 * Hh, p.lead, p.P, WFL_ACT_NONE, nullptr, 0,
               1.f, 0, 0, false, true);
        if (R.rc) return R.rc;
        LstmArgs la{};
        la.gx = GX; la.ldgx = 8 * Hh; la.whh = m->lstm_whh[layer]; la.out = S; la.ldo = d; la.lead = p.lead;
        la.B = B; la.T = p.T; la.P = p.P; la.H = Hh; la.U = m->lstm_U;
        la.error = (unsigned*)(R.ws + p.err);
        la.clip_T = R.clipT;
        R.stats_for = nullptr;
        // the recurrence writes plain bf16 rows -- or, precision high with H <= 256, h as a bf16 pair like every other activation
        const bool split = R.precise() && m->lstm_whh_lo[layer] != nullptr && R.lo_of(S) != nullptr;
        if (split) { la.whh_lo = m->lstm_whh_lo[layer]; la.out_lo = R.lo_of(S); }
        { const int si = R.lo_idx(S); if (si >= 0) R.lo_ok[si] = split; }
        R.prof_begin();
        const int lr = wfl_launch_lstm(la, R.ws + p.lstm_x, R.s);
        R.prof_end(2041, 2.0 * (double)B * p.T * 2.0 * 4.0 * (double)Hh * (double)Hh);
        if (lr) return fail(lr, lr == -5 ? "BiLSTM: hidden size too large (more than 64 slice workgroups per direction)"
                                          : "lstm launch failed (" + std::to_string(lr) + ")");
        std::swap(H, S);
      }
    }
    for (int i = 0; i < a.n_conformer; ++i) {
      const ConfLayer& C = m->conf[i];
      // x = x + 0.5 * FF1(x)
      R.ln_gemm(H, S, C.ff1_ln, C.ff1_a, C.ff1_a_ln, (int)Mrows, FF, p.ffw, WFL_ACT_GELU);
      R.gemm(FF + (long)p.lead * p.ffw, p.ffw, C.ff1_b, (int)Mrows, p.P, p.T, H, d, p.lead, p.P, WFL_ACT_NONE, H, d, 0.5f);
      // x = LN1(x + MHA(x))
      if (p.da == d) {
        R.gemm(H + (long)p.lead * d, d, C.qkv, (int)Mrows, p.P, p.T, QK, 3 * d, p.lead, p.P);
        R.attn(a.conformer_heads);
        R.gemm(ATT + (long)p.lead * d, d, C.out, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
      } else {
        R.gemm(H + (long)p.lead * d, d, C.qkv, (int)Mrows, p.P, p.T, R.buf(p.QKp), 3 * p.da, p.lead, p.P);
        R.attn(a.conformer_heads, nullptr, nullptr, true);
        R.gemm(R.buf(p.ATTp) + (long)p.lead * p.da, p.da, C.out, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
      }
      R.ln(S, H, C.ln1, true);
      // x = x + pw2(GELU(BN(conv_k(GLU(pw1(LN2(x)))))))
      R.ln(H, S, C.ln2);
      R.gemm(S + (long)p.lead * d, d, C.pw1, (int)Mrows, p.P, p.T, ATT, d, p.lead, p.P, WFL_ACT_NONE, nullptr, 0, 1.f, 0, 0, true);
      // dense k-tap conv: taps are adjacent rows (cin = d, tap stride = one row) -> the streaming GEMM's tap-stationary mode
      R.gemm(ATT + (long)(p.lead - a.conformer_kernel / 2) * d, d, C.conv, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_GELU,
             nullptr, 0, 1.f, d, d);
      R.next_stats = true;                                            // (consumed by the folded ff2)
      R.gemm(S + (long)p.lead * d, d, C.pw2, (int)Mrows, p.P, p.T, H, d, p.lead, p.P, WFL_ACT_NONE, H, d, 1.f);
      // x = x + 0.5 * FF2(x)
      R.ln_gemm(H, S, C.ff2_ln, C.ff2_a, C.ff2_a_ln, (int)Mrows, FF, p.ffw, WFL_ACT_GELU);
      R.next_stats = i + 1 < a.n_conformer;                           // (the next block's folded ff1)
      R.gemm(FF + (long)p.lead * p.ffw, p.ffw, C.ff2_b, (int)Mrows, p.P, p.T, H, d, p.lead, p.P, WFL_ACT_NONE, H, d, 0.5f);
    }
    if (a.enable_dilated) {
      for (int i = 0; i < a.dilated_depth; ++i) {
        const int dil = 1 << i, pad = dil * (a.dilated_kernel - 1) / 2;
        R.gemm(H + (long)(p.lead - pad) * d, d, m->dil[i], (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_RELU, nullptr, 0, 1.f,
               d, (long)dil * d);
        std::swap(H, S);
      }
    }
    float* lg_pass = (n_pass > 1 && pass > 0) ? (float*)(R.ws + p.logits2) : lg;
    // (the internal logits buffer has rows of a multiple of four floats: the fp32 epilogue then stores 16 bytes at a time instead of
    //  141 single floats per row; the caller's own buffer and the language-averaging passes keep the compact layout)
    const long ldlg = (!logits && n_pass == 1) ? round_up(a.num_classes, 4) : a.num_classes;
    // classifier in split precision: h_hi . W_lo^T first, then [h_hi | h_lo] . [W_hi | W_hi]^T + b added to it in fp32 (the two
    // input halves are taps one buffer apart; without a valid low half the second pass is the plain K = d one)
    static const bool cls_one = std::getenv("WFL_CLS_TWO_LAUNCHES") == nullptr;       // (A/B hook: rounds 2-3's two-launch form)
    const bf16_t* hlo = R.lo_in(H);
    if (hlo && m->cls3.W && cls_one) {
      // one launch (round 3): K' = 3 d over the tap segments [h_hi | h_lo | h_hi] against [W_hi | W_hi | W_lo], bias once, the sum never
      // leaves the accumulators
      R.next_flops = 2.0 * (double)B * p.T * (double)a.num_classes * (double)d;
      R.next_tap_wrap = 1;
      R.next_seg_off = (long)(hlo - H);
      R.gemm(H + (long)p.lead * d, d, m->cls3, (int)Mrows, p.P, p.T, lg_pass, ldlg, 0, p.T, WFL_ACT_NONE, nullptr, 0, 1.f, d, 0, false, true);
    } else {
      R.next_flops = 0.0;
      R.gemm(H + (long)p.lead * d, d, m->cls_lo, (int)Mrows, p.P, p.T, lg_pass, ldlg, 0, p.T, WFL_ACT_NONE, nullptr, 0, 1.f,
             0, 0, false, true);
      R.next_acc_f32 = true;
      R.next_flops = 2.0 * (double)B * p.T * (double)a.num_classes * (double)d;
      if (hlo)
        R.gemm(H + (long)p.lead * d, d, m->cls_hi2, (int)Mrows, p.P, p.T, lg_pass, ldlg, 0, p.T, WFL_ACT_NONE, nullptr, 0,
               1.f, d, (long)(hlo - H), false, true);
      else
        R.gemm(H + (long)p.lead * d, d, m->cls, (int)Mrows, p.P, p.T, lg_pass, ldlg, 0, p.T, WFL_ACT_NONE, nullptr, 0, 1.f,
               0, 0, false, true);
    }
    R.gemm(H + (long)(p.lead - 1) * d, d, m->off1, (int)Mrows, p.P, p.T, S, d, p.lead, p.P, WFL_ACT_GELU, nullptr, 0, 1.f, d, d);
    if (R.rc) return R.rc;
    TagArgs t{};
    t.rows = B * p.T; t.C = a.num_classes; t.threshold = threshold; t.o_id = a.o_id;
    t.ids = ids; t.argmax = argmax; t.maxprob = maxprob;
    t.hid = S; t.hid_lo = R.lo_in(S); t.ldh = d; t.lead = p.lead; t.P = p.P; t.T = p.T; t.d = d; t.w2 = m->off_w2; t.b2 = m->off_b2;
    t.clip_T = R.clipT; t.Tmax = p.T;
    if (n_pass == 1) {
      t.logits = lg; t.ldl = ldlg; t.offsets = offsets;
      t.status_src = (const unsigned*)(R.ws + p.err); t.status_dst = status;
      const int r = wfl_launch_tag_decide(t, R.s);
      if (r) return fail(r, "tag_decide launch failed");
    } else {
      // offsets of this pass only; logits are decided after the mean
      t.logits = nullptr;
      t.offsets = pass == 0 ? offsets : (float*)(R.ws + p.offs2);
      int r = wfl_launch_tag_decide(t, R.s);
      if (r) return fail(r, "tag_decide launch failed");
      const float w = 1.0f / (float)n_pass;
      const long nl = (long)B * p.T * a.num_classes, no = (long)B * p.T * 2;
      if (pass == 0) {
        r = wfl_launch_axpy(lg, lg, nl, w, 1, R.s);
        if (!r) r = wfl_launch_axpy(offsets, offsets, no, w, 1, R.s);
      } else {
        r = wfl_launch_axpy(lg, (float*)(R.ws + p.logits2), nl, w, 0, R.s);
        if (!r) r = wfl_launch_axpy(offsets, (float*)(R.ws + p.offs2), no, w, 0, R.s);
      }
      if (r) return fail(r, "axpy launch failed");
      if (pass == n_pass - 1) {
        TagArgs f{};
        f.logits = lg; f.ldl = a.num_classes; f.rows = B * p.T; f.C = a.num_classes; f.threshold = threshold;
        f.o_id = a.o_id; f.ids = ids; f.argmax = argmax; f.maxprob = maxprob;
        f.clip_T = R.clipT; f.Tmax = p.T;
        f.status_src = (const unsigned*)(R.ws + p.err); f.status_dst = status;
        r = wfl_launch_tag_decide(f, R.s);
        if (r) return fail(r, "tag_decide launch failed");
      }
    }
  }
  return R.rc;
}

static int check_forward_args(const wfl_model* m, const char* who, int32_t B, const int32_t* lang_id, int32_t lang_mode,
                              const int32_t* ids, const float* maxprob, const float* offsets) {
  if (!ids || !maxprob || !offsets) return fail(-1, std::string(who) + ": ids, maxprob and offsets are required");
  if (B <= 0) return fail(-1, std::string(who) + ": bad batch size");
  if (lang_mode == WFL_LANG_IDS && !lang_id) return fail(-1, std::string(who) + ": lang_id missing");
  if (lang_mode == WFL_LANG_AVERAGE && m->avg_langs.empty()) return fail(-1, std::string(who) + ": no languages to average");
  if (lang_mode < WFL_LANG_NONE || lang_mode > WFL_LANG_AVERAGE) return fail(-1, std::string(who) + ": bad lang_mode");
  return 0;
}

int32_t wfl_forward(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L,
                    const int32_t* lang_id, int32_t lang_mode, float threshold, void* workspace, int64_t workspace_bytes,
                    int32_t* ids, int32_t* argmax, float* maxprob, float* offsets, float* logits, float* hidden,
                    int32_t* status, void* stream) {
  if (!m || !m->finalized) return fail(-1, "wfl_forward: model not finalized");
  if (!wav || B <= 0 || L <= 0 || ldw < L) return fail(-1, "wfl_forward: bad input shape");
  if (int r = check_forward_args(m, "wfl_forward", B, lang_id, lang_mode, ids, maxprob, offsets)) return r;
  if (int r = check_device(m, "wfl_forward")) return r;
  Runner R{m, make_plan(m, B, L), (char*)workspace, (hipStream_t)stream};
  const Plan& p = R.p;
  if (!workspace || workspace_bytes < p.total) return fail(-1, "wfl_forward: workspace too small");
  if (p.T <= 0) return fail(-1, "wfl_forward: clip too short for the encoder");
  if (int r = begin_forward(R, true)) return r;
  if (int r = run_encoder(R, wav, ldw, lens)) return r;
  if (hidden) {
    const int r = emit_hidden(R, hidden);
    if (r) return fail(r, "rows_to_f32 launch failed");
  }
  return run_head(R, lang_id, lang_mode, threshold, ids, argmax, maxprob, offsets, logits, status);
}

int32_t wfl_encode(wfl_model* m, const float* wav, int64_t ldw, const int32_t* lens, int32_t B, int32_t L, void* workspace,
                   int64_t workspace_bytes, float* hidden, void* stream) {
  if (!m || !m->finalized) return fail(-1, "wfl_encode: model not finalized");
  if (!wav || !hidden || B <= 0 || L <= 0 || ldw < L) return fail(-1, "wfl_encode: bad argument");
  if (int r = check_device(m, "wfl_encode")) return r;
  Runner R{m, make_plan(m, B, L), (char*)workspace, (hipStream_t)stream};
  const Plan& p = R.p;
  if (!workspace || workspace_bytes < p.total) return fail(-1, "wfl_encode: workspace too small");
  if (p.T <= 0) return fail(-1, "wfl_encode: clip too short for the encoder");
  if (int r = begin_forward(R, true)) return r;
  if (int r = run_encoder(R, wav, ldw, lens)) return r;
  const int r = emit_hidden(R, hidden);
  return r ? fail(r, "rows_to_f32 launch failed") : 0;
}

int64_t wfl_head_workspace_bytes(const wfl_model* m, int32_t B, int32_t T) {
  if (!m || B <= 0 || T <= 0) return -1;
  return make_plan(m, B, 0, T).total;
}

int32_t wfl_head(wfl_model* m, const float* hidden, int32_t B, int32_t T, const int32_t* lang_id, int32_t lang_mode,
                 float threshold, void* workspace, int64_t workspace_bytes, int32_t* ids, int32_t* argmax, float* maxprob,
                 float* offsets, float* logits, int32_t* status, void* stream) {
  if (!m || !m->finalized) return fail(-1, "wfl_head: model not finalized");
  if (!hidden || T <= 0) return fail(-1, "wfl_head: bad argument");
  if (int r = check_forward_args(m, "wfl_head", B, lang_id, lang_mode, ids, maxprob, offsets)) return r;
  if (int r = check_device(m, "wfl_head")) return r;
  Runner R{m, make_plan(m, B, 0, T), (char*)workspace, (hipStream_t)stream};
  const Plan& p = R.p;
  if (!workspace || workspace_bytes < p.total) return fail(-1, "wfl_head: workspace too small");
  if (int r = begin_forward(R, false)) return r;
  const int r = wfl_launch_f32_to_rows(hidden, R.buf(p.Y), p.d, p.lead, B, p.P, p.T, m->dv, R.s, m->pad_split(), m->pad_shift(),
                                       R.precise() ? R.lo_of(R.buf(p.Y)) : nullptr);
  if (r) return fail(r, "f32_to_rows launch failed");
  if (R.precise()) R.lo_ok[1] = true;
  return run_head(R, lang_id, lang_mode, threshold, ids, argmax, maxprob, offsets, logits, status);
}

int32_t wfl_check(wfl_model* m, void* workspace, int64_t workspace_bytes, int32_t B, int32_t L, void* stream) {
  if (!m || !m->finalized || !workspace) return fail(-1, "wfl_check: bad argument");
  if (int r = check_device(m, "wfl_check")) return r;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  const Plan p = make_plan(m, B, L);
  if (workspace_bytes < p.total) return fail(-1, "wfl_check: workspace too small");
  unsigned err = 0;
  HIPCHK(hipMemcpy(&err, (char*)workspace + p.err, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (err & 1u) return fail(-20, "BiLSTM recurrence: an inter-workgroup wait timed out (results of the last forward are invalid)");
  if (err & 2u) return fail(-22, "fp8 activations: a value did not fit e4m3 at its scale (results of the last forward are invalid; run this "
                                 "checkpoint with bf16 activations)");
  if (err) return fail(-21, "device-side error word " + std::to_string(err));
  return 0;
}

// ------------------------------------------------------------------------------------------------ single-op exports
int32_t wfl_op_gemm_mx(const void* A8, const void* A8_lo, int64_t lda, const void* W8, const float* w_scale, const float* a_scale,
                       float a_static, int32_t M, int32_t N, int32_t K, int32_t P, int32_t T, void* C, int64_t ldc, int64_t c_lead,
                       int32_t c_pitch, const float* bias, const void* res, const void* res_lo, void* c_lo, float alpha, int32_t act,
                       void* c8, void* c8_lo, int64_t ldc8, float c8_inv_scale, int32_t* status, void* stream) {
  GemmArgs g{};
  g.A = (const bf16_t*)A8; g.a8_lo = (const unsigned char*)A8_lo; g.a8 = A8_lo ? 3 : 2; g.lda = lda; g.cin = K; g.W = (const bf16_t*)W8;
  g.w8_scale = w_scale; g.a8_scale = a_scale; g.a8_static = a_static; g.a8_lead = 0;
  g.M = M; g.N = N; g.K = K; g.n_valid = N; g.P = P; g.T = T; g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch;
  g.bias = bias; g.res = (const bf16_t*)res; g.res_lo = (const bf16_t*)res_lo; g.c_lo = (bf16_t*)c_lo; g.ldres = ldc; g.alpha = alpha; g.act = act;
  g.c8 = (unsigned char*)c8; g.c8_lo = (unsigned char*)c8_lo; g.ldc8 = ldc8; g.c8_inv_scale = c8_inv_scale; g.err = (unsigned*)status;
  if (!A8 || !W8 || !w_scale || (!C && !c8) || M <= 0 || P <= 0 || P % 8) return fail(-1, "wfl_op_gemm_mx: bad argument");
  const int r = wfl_launch_gemm_mx(g, (hipStream_t)stream);
  return r ? fail(r == 1 ? -1 : r, "wfl_op_gemm_mx: shape or argument not taken by gemm_mx.hip (" + std::to_string(r) + ")") : 0;
}

int32_t wfl_op_rows_fp8(const void* x, int64_t ldx, const void* x_lo, const float* gamma, const float* beta, float eps, int64_t lead,
                        int32_t B, int32_t P, int32_t T, int32_t C, void* y8, void* y8_lo, int64_t ldy8, float* scale, void* stream) {
  const int r = wfl_launch_rows_fp8((const bf16_t*)x, ldx, (const bf16_t*)x_lo, gamma, beta, eps, lead, B, P, T, C, (unsigned char*)y8, ldy8,
                                    scale, (hipStream_t)stream, (unsigned char*)y8_lo);
  return r ? fail(r, "wfl_op_rows_fp8: invalid arguments or launch failure") : 0;
}

int32_t wfl_op_gemm(const void* A, int64_t lda, int32_t cin, int64_t tap_stride, const void* W, int32_t M, int32_t N,
                    int32_t K, int32_t n_valid, int32_t P, int32_t T, void* C, int64_t ldc, int64_t c_lead, int32_t c_pitch,
                    const float* bias, const void* res, int64_t ldres, float alpha, int32_t act, int32_t glu,
                    int32_t out_f32, void* stream) {
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.cin = cin > 0 ? cin : K; g.tap_stride = tap_stride;
  g.W = (const bf16_t*)W; g.M = M; g.N = N; g.K = K; g.n_valid = n_valid; g.P = P; g.T = T;
  g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch; g.bias = bias;
  g.res = (const bf16_t*)res; g.ldres = ldres; g.alpha = alpha; g.act = act; g.glu = glu; g.out_f32 = out_f32;
  const int r = wfl_launch_gemm(g, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_gemm: invalid arguments or launch failure (" + std::to_string(r) + ")") : 0;
}

int32_t wfl_op_gemm_ln(const void* A, int64_t lda, const void* W, int32_t M, int32_t N, int32_t K, int32_t n_valid, int32_t P,
                       int32_t T, void* C, int64_t ldc, int64_t c_lead, int32_t c_pitch, const float* bias, const float* ln_s,
                       float ln_eps, int32_t act, void* stream) {
  if (!ln_s) return fail(-1, "wfl_op_gemm_ln: ln_s is required");
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.cin = K; g.tap_stride = 0;
  g.W = (const bf16_t*)W; g.M = M; g.N = N; g.K = K; g.n_valid = n_valid; g.P = P; g.T = T;
  g.C = C; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch; g.bias = bias; g.alpha = 1.f; g.act = act;
  g.ln_s = ln_s; g.ln_eps = ln_eps;
  const int r = wfl_launch_gemm(g, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_gemm_ln: shape not supported by the LayerNorm-folding kernel or launch failure (" + std::to_string(r) + ")") : 0;
}

int32_t wfl_op_gemm_split(const void* A_hi, const void* A_lo, int64_t lda, int32_t cin, int64_t tap_stride, const void* W3, int32_t M,
                          int32_t N, int32_t K, int32_t n_valid, int32_t P, int32_t T, void* C, void* C_lo, int64_t ldc, int64_t c_lead,
                          int32_t c_pitch, const float* bias, const void* res, const void* res_lo, int64_t ldres, float alpha,
                          int32_t act, int32_t glu, void* stream) {
  if (!A_hi || !A_lo || !W3 || !C || !C_lo) return fail(-1, "wfl_op_gemm_split: null argument");
  if (K <= 0 || (cin > 0 && K % cin)) return fail(-1, "wfl_op_gemm_split: K must be a whole number of taps");
  GemmArgs g{};
  g.A = (const bf16_t*)A_hi; g.lda = lda; g.cin = cin > 0 ? cin : K; g.tap_stride = tap_stride;
  g.tap_wrap = K / g.cin; g.seg_off = (long)((const bf16_t*)A_lo - (const bf16_t*)A_hi);
  g.W = (const bf16_t*)W3; g.M = M; g.N = N; g.K = 3 * K; g.n_valid = n_valid; g.P = P; g.T = T;
  g.C = C; g.c_lo = (bf16_t*)C_lo; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = c_pitch; g.bias = bias;
  g.res = (const bf16_t*)res; g.res_lo = res ? (const bf16_t*)res_lo : nullptr; g.ldres = ldres; g.alpha = alpha; g.act = act; g.glu = glu;
  g.ln_eps = 1e-5f;
  const int r = wfl_launch_gemm(g, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_gemm_split: invalid arguments or launch failure (" + std::to_string(r) + ")") : 0;
}

int32_t wfl_op_attention(const void* QK, int64_t ldqk, int64_t lead, const void* V, int64_t ldv, void* O, int64_t ldo, int32_t B, int32_t T,
                         int32_t P, int32_t heads, int32_t d, void* stream) {
  AttnArgs a{};
  a.QK = (const bf16_t*)QK; a.ldqk = ldqk; a.lead = lead; a.V = (const bf16_t*)V; a.ldv = ldv; a.O = (bf16_t*)O; a.ldo = ldo;
  a.B = B; a.T = T; a.P = P; a.heads = heads; a.d = d;
  const int r = wfl_launch_attention(a, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_attention: invalid arguments, unsupported head_dim or launch failure (" + std::to_string(r) + ")") : 0;
}

int32_t wfl_op_layernorm(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta, float eps,
                         int64_t lead, int32_t B, int32_t P, int32_t T, int32_t C, void* stream) {
  const int r = wfl_launch_layernorm((const bf16_t*)x, ldx, (bf16_t*)y, ldy, gamma, beta, eps, lead, B, P, T, C, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_layernorm: invalid arguments or launch failure") : 0;
}

int32_t wfl_op_tag_decide(const float* logits, int64_t ldl, int32_t rows, int32_t C, float threshold, int32_t o_id, int32_t* ids,
                          int32_t* argmax, float* maxprob, void* stream) {
  if (!logits || !ids || !maxprob) return fail(-1, "wfl_op_tag_decide: null argument");
  TagArgs t{};
  t.logits = logits; t.ldl = ldl; t.rows = rows; t.C = C; t.threshold = threshold; t.o_id = o_id;
  t.ids = ids; t.argmax = argmax; t.maxprob = maxprob;
  const int r = wfl_launch_tag_decide(t, (hipStream_t)stream);
  return r ? fail(r, "wfl_op_tag_decide: launch failure") : 0;
}

int32_t wfl_gemm_profile_enable(wfl_model* m, int32_t on) {
  if (!m) return fail(-1, "null model");
  m->prof_on = on != 0;
  return 0;
}

int32_t wfl_gemm_profile_read(wfl_model* m, int32_t max_variants, int32_t* keys, int64_t* launches, double* total_ms,
                              double* total_flops, int32_t* n_variants, int32_t reset) {
  if (!m || !keys || !launches || !total_ms || !total_flops || !n_variants) return fail(-1, "wfl_gemm_profile_read: null argument");
  std::vector<double> ms(2048, 0.0);
  for (size_t i = 0; i < m->prof_used; ++i) {
    HIPCHK(hipEventSynchronize(m->prof.ev[i].second));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, m->prof.ev[i].first, m->prof.ev[i].second));
    ms[m->prof.key[i]] += t;
  }
  int n = 0;
  for (int k = 0; k < 2048 && n < max_variants; ++k)
    if (m->prof.launches[k]) {
      keys[n] = k; launches[n] = m->prof.launches[k]; total_ms[n] = ms[k]; total_flops[n] = m->prof.flops[k];
      ++n;
    }
  *n_variants = n;
  if (reset) {
    memset(m->prof.launches, 0, sizeof(m->prof.launches));
    memset(m->prof.flops, 0, sizeof(m->prof.flops));
    m->prof.key.clear();
    m->prof_used = 0;
  }
  return 0;
}
