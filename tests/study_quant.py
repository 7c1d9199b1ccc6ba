"""Where does the bf16 error of the cfg2 forward come from?  (diagnostic, CPU; not a test)

Re-runs the oracle's Whisper + Conformer forward with bf16 rounding injected at chosen places and reports the logit
error against the fp32 run.  Flags (any combination, comma separated on the command line):
  w     weights of every Linear / Conv rounded to bf16
  a     the input of every Linear / Conv rounded to bf16 (what a bf16 MFMA operand is)
  r     the residual stream rounded to bf16 after every residual add (and the stem output)
  attn  q, k, v and the softmax probabilities rounded to bf16; attn_qk / attn_p / attn_v: only those operands
  r_stem / r_encln / r_lang / r_ln1 / r_enc<i> / r_conf<i>   the residual stream rounded only at that place (encoder layer i, Conformer block i)
  cls   the classifier's input and weight rounded to bf16 (w / a leave the classifier alone)
  wenc / whead   weights of the encoder / of the head (classifier excepted) only
usage: python tests/study_quant.py [clips] flags [flags ...]     e.g.  python tests/study_quant.py 1 w a r attn w,a,r,attn
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wfl_oracle as O   # noqa: E402
import synthetic as synth   # noqa: E402
from wfl_asr_amd.archs import resolve_encoder_arch   # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class Q:
    def __init__(self, flags):
        self.f = set(flags)

    def w(self, t, name=""):
        cls = name.startswith("classifier")
        if cls:
            return bf(t) if "cls" in self.f else t
        if "w" in self.f or ("wenc" in self.f and name.startswith("encoder.")) or ("whead" in self.f and not name.startswith("encoder.")):
            return bf(t)
        return t

    def a(self, t, name=""):
        cls = name.startswith("classifier")
        if cls:
            return bf(t) if "cls" in self.f else t
        return bf(t) if "a" in self.f else t

    def r(self, t, kind=""):
        if "r" in self.f or (kind and ("r_" + kind) in self.f):
            return bf(t)
        return t

    def at(self, t, what=""):
        # attn: q, k, v, P all rounded; attn_qk / attn_p / attn_v: only that operand (round 4: which of precision-high's split passes pay)
        if "attn" in self.f or ("attn_" + what) in self.f or (what in ("q", "k") and "attn_qk" in self.f):
            return bf(t)
        return t


def lin(q, x, sd, p):
    return F.linear(q.a(x, p), q.w(sd[p + ".weight"], p), sd.get(p + ".bias"))


def conv(q, x, sd, p, **kw):
    return F.conv1d(q.a(x, p), q.w(sd[p + ".weight"], p), sd.get(p + ".bias"), **kw)


def attn(q_, qh, kh, vh):
    a = torch.softmax(q_.at(qh, "q") @ q_.at(kh, "k").transpose(2, 3), dim=-1)
    return q_.at(a, "p") @ q_.at(vh, "v")


def forward(q, wav, lang, sd, arch, hc):
    feats = O.whisper_log_mel(wav, arch.n_mels, arch.max_positions * 2 * arch.hop, arch.n_fft, arch.hop)
    p = "encoder."
    x = F.gelu(conv(q, feats, sd, p + "conv1", padding=1))
    x = F.gelu(conv(q, x, sd, p + "conv2", stride=2, padding=1))
    x = q.r(x.permute(0, 2, 1) + sd[p + "embed_positions.weight"], "stem")
    B, T, d = x.shape
    heads = arch.heads
    hd = d // heads
    for i in range(arch.layers):
        lp = f"{p}layers.{i}."
        h = O._ln(x, sd, lp + "self_attn_layer_norm")
        qq = (lin(q, h, sd, lp + "self_attn.q_proj") * hd ** -0.5).view(B, T, heads, hd).transpose(1, 2)
        k = lin(q, h, sd, lp + "self_attn.k_proj").view(B, T, heads, hd).transpose(1, 2)
        v = lin(q, h, sd, lp + "self_attn.v_proj").view(B, T, heads, hd).transpose(1, 2)
        a = attn(q, qq, k, v).transpose(1, 2).reshape(B, T, d)
        x = q.r(x + lin(q, a, sd, lp + "self_attn.out_proj"), f"enc{i}")
        h = O._ln(x, sd, lp + "final_layer_norm")
        h = F.gelu(lin(q, h, sd, lp + "fc1"))
        x = q.r(x + lin(q, h, sd, lp + "fc2"), f"enc{i}")
    x = q.r(O._ln(x, sd, p + "layer_norm"), "encln")
    hidden = x
    e = sd["lang_emb.weight"][lang][:, None, :].expand(-1, T, -1)
    x = q.r(lin(q, torch.cat([x, e], dim=-1), sd, "lang_proj"), "lang")
    ch = hc["conformer_heads"]
    chd = d // ch
    for i in range(hc["num_conformer_layers"]):
        cp = f"conformer_layers.{i}."

        def ff(x, pp):
            h = O._ln(x, sd, pp + ".net.0")
            return lin(q, F.gelu(lin(q, h, sd, pp + ".net.1")), sd, pp + ".net.4")

        x = q.r(x + 0.5 * ff(x, cp + "ff1"), f"conf{i}")
        qkv = F.linear(q.a(x, cp), q.w(sd[cp + "self_attn.in_proj_weight"], cp), sd[cp + "self_attn.in_proj_bias"])
        qq, k, v = qkv.chunk(3, dim=-1)
        qq = qq.view(B, T, ch, chd).transpose(1, 2) * chd ** -0.5
        k = k.view(B, T, ch, chd).transpose(1, 2)
        v = v.view(B, T, ch, chd).transpose(1, 2)
        a = attn(q, qq, k, v).transpose(1, 2).reshape(B, T, d)
        x = q.r(O._ln(x + lin(q, a, sd, cp + "self_attn.out_proj"), sd, cp + "ln1"), "ln1")
        h = O._ln(x, sd, cp + "ln2").transpose(1, 2)
        h = F.glu(conv(q, h, sd, cp + "conv.0"), dim=1)
        h = conv(q, h, sd, cp + "conv.2", padding=hc["conformer_kernel_size"] // 2)
        h = F.batch_norm(h, sd[cp + "conv.3.running_mean"], sd[cp + "conv.3.running_var"], sd[cp + "conv.3.weight"],
                         sd[cp + "conv.3.bias"], False, 0.0, 1e-5)
        h = conv(q, F.gelu(h), sd, cp + "conv.5").transpose(1, 2)
        x = q.r(x + h, f"conf{i}")
        x = q.r(x + 0.5 * ff(x, cp + "ff2"), f"conf{i}")
    logits = lin(q, x, sd, "classifier")
    return logits, hidden


def main():
    args = sys.argv[1:]
    n = int(args[0]) if args and args[0].isdigit() else 1
    combos = [a for a in args if not a.isdigit()] or ["w", "a", "r", "attn", "cls", "w,a,r,attn"]
    cfg = synth.baseline_config(1)
    enc, arch = resolve_encoder_arch(cfg["model"])
    hc = synth.head_config(cfg["model"])
    labels = synth.make_labels(70)
    sd = O.to_torch_state_dict(synth.make_state_dict(cfg, len(labels), seed=1))
    wav = torch.from_numpy(synth.make_batch(2000, n, 480000, seed=1))
    lang = torch.arange(n) % 2
    with torch.no_grad():
        ref, hid0 = forward(Q([]), wav, lang, sd, arch, hc)
        lg0, _ = O.forward(wav, lang, sd, enc, arch, hc)
        print("restatement vs oracle:", float((ref - lg0).abs().max()))
        top2 = ref.topk(2, dim=-1).values
        margin = top2[..., 0] - top2[..., 1]
        print(f"logit std {float(ref.std()):.2f}  frames with margin<=0.2: {float((margin <= 0.2).float().mean()):.3f}  <=0.5: {float((margin <= 0.5).float().mean()):.3f}")
        for c in combos:
            lg, hid = forward(Q(c.split(",")), wav, lang, sd, arch, hc)
            e = (lg - ref).abs()
            he = (hid - hid0).abs()
            flips = int((lg.argmax(-1) != ref.argmax(-1)).sum())
            print(f"{c:16s} logits err max {float(e.max()):.4f} mean {float(e.mean()):.4f} | hidden err max {float(he.max()):.4f} "
                  f"mean {float(he.mean()):.5f} | argmax flips {flips}/{ref.shape[0] * ref.shape[1]}", flush=True)


if __name__ == "__main__":
    main()
