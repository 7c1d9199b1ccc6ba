"""Host-side mirror of the reference's model object for the inference path.

`BIOPhonemeTagger(config, label_list)` keeps the constructor, attributes and `forward` contract of
/root/reference/model.py:54-201, but owns no torch modules: weights go straight into the HIP library
(`load_state_dict` -> wfl_load_tensor/wfl_finalize) and `forward` is one `wfl_forward` call on the
current HIP stream.  PyTorch is used only for device buffers and the stream handle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .archs import MelArch, WhisperArch, head_config, resolve_encoder_arch

LANG_NONE, LANG_IDS, LANG_AVERAGE = 0, 1, 2


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _resolve_device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.WflError("this build runs on MI355X only (device must be cuda[:N]); there is no CPU path")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class TagBatch:
    """Decisions for a batch: ids/argmax [B,T] int32, maxprob [B,T], offsets [B,T,2], optional logits/hidden."""

    def __init__(self, ids, argmax, maxprob, offsets, logits=None, hidden=None):
        self.ids, self.argmax, self.maxprob, self.offsets = ids, argmax, maxprob, offsets
        self.logits, self.hidden = logits, hidden
        self.packed = None      # int32 view over ids | maxprob | offsets | status word (one contiguous D2H copy)
        self.status = None      # int32 [1]: device-side error bits of this forward (0 = ok), the last word of `packed`


def raise_on_status(word: int):
    """`word` = TagBatch.status after the forward finished (include/wfl_asr.h: bit 0 = BiLSTM hand-off time-out, bit 1 = an e4m3
    activation saturated or was NaN -- `model.activation_dtype: fp8` only)."""
    if word:
        why = []
        if int(word) & 1:
            why.append("a BiLSTM inter-workgroup wait timed out")
        if int(word) & 2:
            why.append("an fp8 activation did not fit e4m3 at its scale (model.activation_dtype: fp8; use bf16 activations for this checkpoint)")
        raise _lib.WflError(f"the forward reported a device-side error (status {int(word):#x}"
                            + (": " + "; ".join(why) if why else "") + "); its tags are invalid")


class BIOPhonemeTagger:
    def __init__(self, config: dict, label_list, device=None):
        """device: the HIP device the weights will live on (default: the current one when load_state_dict runs).  Every
        kernel launch of this object happens with that device current."""
        self.config = config
        self.device = _resolve_device(device) if device is not None else None
        self.encoder_type, self.arch = resolve_encoder_arch(config["model"], config.get("data"))
        self.head_cfg = head_config(config["model"])
        self.label_list = list(label_list)
        self.label2id = {label: i for i, label in enumerate(self.label_list)}
        self.id2label = {i: label for label, i in self.label2id.items()}
        if "O" not in self.label2id:
            raise ValueError('label_list has no "O" tag')
        self.hidden_size = self.arch.d_model
        self.num_languages = self.head_cfg["num_languages"]
        self._lib = _lib.load()
        self._handle = C.c_void_p(0)
        self._ready = False
        self._ws = None
        self._ws_extra = {}
        self._graphs = {}
        a = _lib.WflArch()
        a.abi_version = _lib.ABI_VERSION
        a.encoder_type = {"whisper": 0, "wavlm": 1, "none": 2}[self.encoder_type]
        a.d_model, a.enc_layers, a.enc_heads, a.enc_ffn = self.arch.d_model, self.arch.layers, self.arch.heads, self.arch.ffn
        if isinstance(self.arch, MelArch):
            a.n_mels, a.mel_hop = self.arch.n_mels, self.arch.hop
        elif isinstance(self.arch, WhisperArch):
            a.n_mels, a.max_positions = self.arch.n_mels, self.arch.max_positions
            a.fp8_weights = int(str(config["model"].get("weight_dtype", "bf16")).lower() in ("fp8", "e4m3", "float8_e4m3fn"))
            # model.activation_dtype (fp8-weight models; wfl_arch::fp8_activations): bf16 | fp8_pair (e4m3 hi + lo pairs on the
            # block-scaled fp8 MFMA: bf16's eight significant bits, the reference's arithmetic on the fp8 checkpoint) | fp8 (one e4m3
            # value per activation: fastest, 5-9 % of the raw tag decisions differ -- an explicit opt-in) | fp8_nonscaled (round 3's kernel)
            act = str(config["model"].get("activation_dtype", "bf16")).lower()
            modes = {"bf16": 0, "fp8_nonscaled": 1, "fp8": 2, "e4m3": 2, "float8_e4m3fn": 2, "fp8_pair": 3, "e4m3_pair": 3}
            if act not in modes:
                raise ValueError(f"model.activation_dtype: {act!r} (one of {sorted(modes)})")
            if modes[act] and not a.fp8_weights:
                raise ValueError("model.activation_dtype: fp8 needs model.weight_dtype: fp8")
            a.fp8_activations = modes[act]
            if (self.arch.n_fft, self.arch.hop) != (400, 160):
                raise ValueError("the log-mel kernel is built for n_fft=400 / hop=160 (every Whisper checkpoint)")
        else:
            w = self.arch
            a.wavlm_n_conv = len(w.conv_dim)
            for i, (c, k, s) in enumerate(zip(w.conv_dim, w.conv_kernel, w.conv_stride)):
                a.wavlm_conv_dim[i], a.wavlm_conv_kernel[i], a.wavlm_conv_stride[i] = c, k, s
            a.wavlm_group_norm = int(w.feat_extract_norm == "group")
            a.wavlm_conv_bias = int(w.conv_bias)
            a.wavlm_stable_layer_norm = int(w.stable_layer_norm)
            a.wavlm_pos_conv_kernel, a.wavlm_pos_conv_groups = w.pos_conv_kernel, w.pos_conv_groups
            a.wavlm_num_buckets, a.wavlm_max_distance = w.num_buckets, w.max_distance
            a.wavlm_do_normalize = int(w.do_normalize)
        # model.precision: high -- split-precision GEMMs (three bf16 passes, fp32 sums) for callers who need the reference's tag indices
        a.precision = int(str(config["model"].get("precision", "default")).lower() in ("high", "exact"))
        if a.precision and getattr(a, "fp8_weights", 0):
            raise ValueError("model.precision: high and model.weight_dtype: fp8 contradict each other (e4m3 weights carry 3 mantissa bits)")
        # The three-pass BiLSTM recurrence holds both halves of its W_hh slice in registers (csrc/lstm.hip): hidden sizes up to 256 per
        # direction with the loader wave, up to 640 (Whisper-small's 384, WavLM-large's 512, Whisper-large's 640) in the four-wave form
        # that has 512 registers per wave (round 4).  Beyond that the recurrence would silently stay bf16: refused.
        self.precision_note = "default (bf16 operands)"
        if a.precision:
            self.precision_note = "high (every product three bf16 passes over split operands)"
            if self.head_cfg["enable_bilstm"] and self.arch.d_model // 2 > 640:
                raise ValueError(f"model.precision: high with a BiLSTM of hidden size {self.arch.d_model // 2} per direction: the three-pass "
                                 "recurrence is built for hidden sizes up to 640 (d_model <= 1280)")
        h = self.head_cfg
        a.num_classes, a.o_id = len(self.label_list), self.label2id["O"]
        a.num_languages, a.lang_emb_dim = h["num_languages"], h["lang_emb_dim"]
        a.enable_bilstm, a.bilstm_layers = int(h["enable_bilstm"]), h["bilstm_num_layer"]
        a.n_conformer, a.conformer_heads = h["num_conformer_layers"], h["conformer_heads"]
        a.conformer_ff_expansion, a.conformer_kernel = h["conformer_ff_expansion"], h["conformer_kernel_size"]
        a.enable_dilated, a.dilated_depth, a.dilated_kernel = int(h["enable_dilated_conv"]), h["dilated_conv_depth"], h["dilated_conv_kernel"]
        _lib.check(self._lib.wfl_create(C.byref(a), C.byref(self._handle)), "wfl_create")

    # ---- nn.Module-shaped conveniences the reference's infer.py uses (infer.py:205-208)
    def load_state_dict(self, state_dict, strict: bool = True):
        if not strict:
            raise ValueError("only strict loading is supported (as at /root/reference/infer.py:207)")
        for name, t in state_dict.items():
            if isinstance(t, torch.Tensor):
                t = t.detach().cpu()
                arr = t.to(torch.float32).numpy() if t.dtype != torch.float32 else t.numpy()
            else:
                arr = np.asarray(t)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self._lib.wfl_load_tensor(self._handle, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                       f"wfl_load_tensor({name})")
        if self.device is None:
            self.device = _resolve_device("cuda")
        with torch.cuda.device(self.device):         # wfl_finalize uploads to the CURRENT device
            _lib.check(self._lib.wfl_finalize(self._handle), "load_state_dict")
        self._ready = True
        return self

    def to(self, device):
        dev = _resolve_device(device)
        if self._ready and dev != self.device:
            raise _lib.WflError(f"the weights were uploaded to {self.device}; build the tagger with device={dev} (or call "
                                ".to() before load_state_dict) instead of moving it")
        self.device = dev
        return self

    def set_average_languages(self, ids):
        """The ids `lang_id=None` averages over: the reference loops over langs.txt (infer.py:147, 268)."""
        arr = (C.c_int32 * len(ids))(*[int(i) for i in ids])
        _lib.check(self._lib.wfl_set_average_languages(self._handle, arr, len(ids)), "wfl_set_average_languages")

    def eval(self):
        return self

    def num_frames(self, L: int) -> int:
        return int(self._lib.wfl_num_frames(self._handle, int(L)))

    def effective_precision(self) -> str:
        """What `model.precision` turned into for this model (a wide BiLSTM under `precision: high` keeps a bf16 recurrence, by request)."""
        return self.precision_note

    def batches_in_flight(self) -> int:
        """How many batches a labelling loop should keep in flight (one stream + workspace slot each).  Two everywhere but one
        case: a BiLSTM behind a small Whisper encoder (tiny / base).  There the recurrence -- a few workgroups, serial in time --
        is a large share of the forward and a third batch fills the CUs it leaves idle (default config.yaml head, 16 x 30 s:
        60.4 / 78.1 / 87.0 / 91.8 k audio-s/s at 1 / 2 / 3 / 4 in flight, not monotonic beyond; with Whisper-small, WavLM-large or no BiLSTM a third
        batch costs 1-3 %: profiles/round2_inflight_sweep.json).  `WFL_INFLIGHT` overrides."""
        env = os.environ.get("WFL_INFLIGHT")
        if env:
            return max(1, int(env))
        small_whisper = self.encoder_type == "whisper" and self.arch.d_model <= 512
        return 3 if (self.head_cfg["enable_bilstm"] and small_whisper) else 2

    def _workspace(self, B: int, L: int, device, slot: int = 0, need: int = 0):
        need = need or int(self._lib.wfl_workspace_bytes(self._handle, B, L))
        if need <= 0:
            raise _lib.WflError("wfl_workspace_bytes failed")
        if device != self.device:
            raise _lib.WflError(f"input is on {device} but the model lives on {self.device}")
        if slot:                                    # extra workspaces: one per batch in flight on its own stream
            ws = self._ws_extra.get(slot)
            if ws is None or ws.numel() < need or ws.device != device:
                if any(k[-1] == slot for k in self._graphs):
                    raise _lib.WflError("workspace must not grow while captured graphs hold its address")
                self._ws_extra[slot] = ws = torch.zeros(need, dtype=torch.uint8, device=device)   # (zeros: see below)
            return ws
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            if any(k[-1] == 0 for k in self._graphs):
                raise _lib.WflError("workspace must not grow while captured graphs hold its address; "
                                    "label the largest batch first or create a new tagger")
            self._ws = None
            # zeros, not empty: the forward's error word lives in the workspace, and `check()` on a slot no forward has run on yet
            # must read 0 rather than whatever the allocator hands back (a one-time fill of about 1 GB per slot)
            self._ws = torch.zeros(need, dtype=torch.uint8, device=device)
        return self._ws

    def _check_lang(self, lang_id, B):
        """Range check on the host copy of lang_id (no device sync); returns an int32 CPU/CUDA tensor."""
        if isinstance(lang_id, torch.Tensor) and lang_id.is_cuda:
            t = lang_id.to(torch.int32)
        else:
            t = torch.as_tensor(lang_id).to(torch.int32).reshape(-1)
            if t.numel() and (int(t.max()) >= self.num_languages or int(t.min()) < 0):
                raise ValueError(f"Language ID out of range (num_languages={self.num_languages})")
        if t.numel() != B:
            raise ValueError("lang_id must have one entry per clip")
        return t

    def _launch(self, x, lens_t, lang_t, mode, threshold, out: "TagBatch", slot: int = 0):
        B, L = x.shape
        ws = self._workspace(B, L, x.device, slot)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            ldw = x.stride(0) if B > 1 else L      # (a single row's stride is arbitrary: numpy's x[None] gives 0)
            rc = self._lib.wfl_forward(self._handle, _ptr(x), ldw, _ptr(lens_t), B, L, _ptr(lang_t), mode,
                                       float(threshold), _ptr(ws), ws.numel(), _ptr(out.ids), _ptr(out.argmax),
                                       _ptr(out.maxprob), _ptr(out.offsets), _ptr(out.logits), _ptr(out.hidden),
                                       _ptr(out.status), C.c_void_p(stream))
        _lib.check(rc, "wfl_forward")

    def _alloc_out(self, B, T, dev, want_logits, want_hidden) -> "TagBatch":
        """ids | maxprob | offsets | status | argmax live in ONE allocation so the host side needs a single D2H copy
        (`TagBatch.packed` = the first 4*B*T + 1 words: ids, max-prob bits, offsets bits, the forward's status word)."""
        Cn = len(self.label_list)
        n = B * T
        blob = torch.empty(5 * n + 1, dtype=torch.int32, device=dev)
        out = TagBatch(
            blob[0:n].view(B, T), blob[4 * n + 1:5 * n + 1].view(B, T), blob[n:2 * n].view(torch.float32).view(B, T),
            blob[2 * n:4 * n].view(torch.float32).view(B, T, 2),
            torch.empty(B, T, Cn, dtype=torch.float32, device=dev) if want_logits else None,
            torch.empty(B, T, self.hidden_size, dtype=torch.float32, device=dev) if want_hidden else None)
        out.status = blob[4 * n:4 * n + 1]
        out.packed = blob[0:4 * n + 1]
        return out

    @torch.no_grad()
    def label(self, input_values: torch.Tensor, lang_id=None, threshold: float = 0.0, lens=None,
              average_languages: bool = False, want_logits: bool = False, want_hidden: bool = False,
              graph: bool = False, slot: int = 0) -> TagBatch:
        """The batched fast path: [B, L] fp32 16 kHz clips -> per-frame decisions (all on the GPU).

        graph=True (off by default) replays the whole forward (about 100 kernel launches) as one captured HIP graph per
        (B, L, mode, slot) signature; inputs are copied into static buffers and the returned tensors are the graph's static
        outputs (consume them before the next call with the same signature and slot).  The kernels are long enough that
        eager launches already keep the GPU busy (graph gain measured < 1 %).

        slot: workspace index.  Batches labelled concurrently on different streams must use different slots (each slot owns
        a workspace); the forward itself keeps no other per-call state."""
        if not self._ready:
            raise _lib.WflError("load_state_dict() has not been called")
        if not input_values.is_cuda:
            raise _lib.WflError("input_values must be a CUDA (ROCm) tensor; there is no CPU path")
        if input_values.dim() != 2:
            raise ValueError("input_values must be [B, L]")
        x = input_values.to(torch.float32)
        if x.stride(1) != 1:
            x = x.contiguous()
        B, L = x.shape
        dev = x.device
        T = self.num_frames(L)
        lang_t = None
        if average_languages:
            mode = LANG_AVERAGE
        elif lang_id is None:
            mode = LANG_NONE
        else:
            mode = LANG_IDS
            lang_t = self._check_lang(lang_id, B)
        lens_t = torch.as_tensor(lens).to(torch.int32).reshape(-1) if lens is not None else None
        if lens_t is not None and lens_t.numel() != B:
            raise ValueError("lens must have one entry per clip")
        if not graph:
            lang_d = lang_t.to(dev).contiguous() if lang_t is not None else None
            lens_d = lens_t.to(dev).contiguous() if lens_t is not None else None
            out = self._alloc_out(B, T, dev, want_logits, want_hidden)
            if lens_d is not None and out.logits is not None and self.encoder_type != "whisper":
                out.logits.zero_()             # (frames behind a shorter clip's own count are not written by the forward)
            self._launch(x, lens_d, lang_d, mode, threshold, out, slot)
            return out
        # one graph per signature AND workspace slot: a graph owns static inputs/outputs and replays on its slot's workspace,
        # so graphs of different slots may run concurrently on different streams (the slot is the key's last entry)
        key = (B, L, mode, float(threshold), want_logits, want_hidden, lens_t is not None, dev.index, slot)
        g = self._graphs.get(key)
        if g is None:
            st = dict(
                x=torch.empty(B, L, dtype=torch.float32, device=dev),
                lang=torch.zeros(B, dtype=torch.int32, device=dev) if lang_t is not None else None,
                lens=torch.zeros(B, dtype=torch.int32, device=dev) if lens_t is not None else None,
                out=self._alloc_out(B, T, dev, want_logits, want_hidden))
            st["x"].copy_(x)
            if lang_t is not None:
                st["lang"].copy_(lang_t)
            if lens_t is not None:
                st["lens"].copy_(lens_t)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):           # warm-up outside capture (function attributes, workspace)
                self._launch(st["x"], st["lens"], st["lang"], mode, threshold, st["out"], slot)
            torch.cuda.current_stream(dev).wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                self._launch(st["x"], st["lens"], st["lang"], mode, threshold, st["out"], slot)
            st["graph"] = cg
            self._graphs[key] = g = st
        g["x"].copy_(x, non_blocking=True)
        # small host-side vectors: blocking copies (an async copy from pageable memory may read the source after
        # the caller has dropped it)
        if lang_t is not None:
            g["lang"].copy_(lang_t, non_blocking=lang_t.is_cuda)
        if lens_t is not None:
            g["lens"].copy_(lens_t, non_blocking=lens_t.is_cuda)
        g["graph"].replay()
        return g["out"]

    def check(self, B: int, L: int, device=None, slot: int = 0):
        """Synchronise and raise if the last forward on workspace `slot` recorded a device-side error (BiLSTM hand-off
        time-out).  The batched loops read the same word from `TagBatch.status` instead (no extra synchronisation)."""
        dev = torch.device(device) if device is not None else self.device
        ws = self._workspace(B, L, dev, slot)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(self._lib.wfl_check(self._handle, _ptr(ws), ws.numel(), B, L, C.c_void_p(stream)), "wfl_check")

    def forward(self, input_values, lang_id=None, max_label_len=None):
        """Reference contract (model.py:148-194): returns (logits [B,T,C] f32, offsets [B,T,2] f32).  With `max_label_len`
        (the training-time validation forward, train.py:456-545) the encoder output is truncated / zero-padded to that many
        frames before the head (model.py:166-174): wfl_encode -> pad/truncate -> wfl_head."""
        if max_label_len is None:
            out = self.label(input_values, lang_id, want_logits=True)
            return out.logits, out.offsets
        hidden = self.encode(input_values)
        cur = hidden.size(1)
        if cur > max_label_len:
            hidden = hidden[:, :max_label_len, :].contiguous()
        elif cur < max_label_len:
            hidden = torch.cat([hidden, hidden.new_zeros(hidden.size(0), max_label_len - cur, hidden.size(2))], dim=1)
        out = self.head(hidden, lang_id, want_logits=True)
        return out.logits, out.offsets

    @torch.no_grad()
    def encode(self, input_values: torch.Tensor, lens=None) -> torch.Tensor:
        """[B, L] fp32 16 kHz -> encoder output [B, T, d] fp32 (model.py:149-161; wfl_encode)."""
        if not self._ready:
            raise _lib.WflError("load_state_dict() has not been called")
        x = input_values.to(torch.float32)
        if x.stride(1) != 1:
            x = x.contiguous()
        B, L = x.shape
        hidden = torch.empty(B, self.num_frames(L), self.hidden_size, dtype=torch.float32, device=x.device)
        lens_d = torch.as_tensor(lens).to(torch.int32).reshape(-1).to(x.device) if lens is not None else None
        ws = self._workspace(B, L, x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            rc = self._lib.wfl_encode(self._handle, _ptr(x), x.stride(0) if B > 1 else L, _ptr(lens_d), B, L, _ptr(ws), ws.numel(), _ptr(hidden),
                                      C.c_void_p(stream))
        _lib.check(rc, "wfl_encode")
        return hidden

    @torch.no_grad()
    def head(self, hidden: torch.Tensor, lang_id=None, threshold: float = 0.0, average_languages: bool = False,
             want_logits: bool = False) -> TagBatch:
        """Encoder output [B, T, d] fp32 (any T) -> decisions (model.py:176-194 + infer.py:86-96; wfl_head)."""
        if not self._ready:
            raise _lib.WflError("load_state_dict() has not been called")
        if hidden.dim() != 3 or hidden.size(2) != self.hidden_size or not hidden.is_cuda:
            raise ValueError("hidden must be a CUDA tensor [B, T, d_model]")
        h = hidden.to(torch.float32).contiguous()
        B, T, _ = h.shape
        dev = h.device
        lang_t = None
        if average_languages:
            mode = LANG_AVERAGE
        elif lang_id is None:
            mode = LANG_NONE
        else:
            mode = LANG_IDS
            lang_t = self._check_lang(lang_id, B).to(dev).contiguous()
        need = int(self._lib.wfl_head_workspace_bytes(self._handle, B, T))
        ws = self._workspace(B, 0, dev, slot=-1, need=need)       # its own slot: a head-only plan lays the workspace out differently
        out = self._alloc_out(B, T, dev, want_logits, False)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = self._lib.wfl_head(self._handle, _ptr(h), B, T, _ptr(lang_t), mode, float(threshold), _ptr(ws), ws.numel(),
                                    _ptr(out.ids), _ptr(out.argmax), _ptr(out.maxprob), _ptr(out.offsets), _ptr(out.logits),
                                    _ptr(out.status), C.c_void_p(stream))
        _lib.check(rc, "wfl_head")
        return out

    __call__ = forward

    def log_mel(self, input_values: torch.Tensor, lens=None) -> torch.Tensor:
        """[B, L] -> [B, n_mels, frames] fp32, the WhisperFeatureExtractor output (model.py:153-154)."""
        x = input_values.to(torch.float32).contiguous()
        B, L = x.shape
        out = torch.empty(B, self.arch.n_mels, 2 * self.arch.max_positions, dtype=torch.float32, device=x.device)
        lens_t = torch.as_tensor(lens, device=x.device).to(torch.int32).contiguous() if lens is not None else None
        ws = self._workspace(B, L, x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            rc = self._lib.wfl_logmel(self._handle, _ptr(x), x.stride(0), _ptr(lens_t), B, L, _ptr(out), _ptr(ws),
                                      ws.numel(), C.c_void_p(stream))
        _lib.check(rc, "wfl_logmel")
        return out

    def decode_predictions(self, logits):
        return torch.argmax(logits, dim=-1)

    def id_to_label(self, ids):
        return [[self.id2label[int(i)] for i in seq] for seq in ids]

    # ---- GEMM timing hook (bench.py roofline)
    def gemm_profile(self, on: bool):
        _lib.check(self._lib.wfl_gemm_profile_enable(self._handle, int(on)), "wfl_gemm_profile_enable")

    def gemm_profile_read(self, reset=True):
        n = 256
        keys = (C.c_int32 * n)()
        launches = (C.c_int64 * n)()
        ms = (C.c_double * n)()
        fl = (C.c_double * n)()
        cnt = C.c_int32(0)
        _lib.check(self._lib.wfl_gemm_profile_read(self._handle, n, keys, launches, ms, fl, C.byref(cnt), int(reset)),
                   "wfl_gemm_profile_read")
        return [dict(key=int(keys[i]), launches=int(launches[i]), ms=float(ms[i]), flops=float(fl[i])) for i in range(cnt.value)]

    def __del__(self):
        try:
            self._graphs.clear()
            if self._handle:
                self._lib.wfl_destroy(self._handle)
                self._handle = C.c_void_p(0)
        except Exception:
            pass
