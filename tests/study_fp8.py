"""What does an fp8 ACTIVATION format cost inside the reference's own arithmetic?  (diagnostic, CPU; not a test)

BASELINE configs[4] (Whisper-large-v3 geometry, fp8 weights).  The parity target is the oracle on the fp8-rounded checkpoint with exact
activations ("W8").  This script re-runs that oracle with the four GEMM inputs of every encoder layer rounded to a chosen format
(oracle ACT_FORMATS: bf16 / row8 / fix8 / blk8 / pair8) and reports the logit error and the raw argmax flips against W8.
usage: python tests/study_fp8.py [large_v3_4l|base_6l] [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wfl_oracle as O   # noqa: E402
import synthetic as synth   # noqa: E402
from wfl_asr_amd.archs import resolve_encoder_arch   # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "large_v3_4l"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    cfg = synth.baseline_config(4)
    if case == "large_v3_4l":
        cfg["model"]["whisper_model"] = "local/whisper-large-v3-4l"
        cfg["model"]["encoder_arch"] = dict(d_model=1280, layers=4, heads=20, ffn=5120, n_mels=128, max_positions=1500)
    else:
        cfg["model"]["whisper_model"] = "local/whisper-base-fp8"
        cfg["model"]["encoder_arch"] = dict(d_model=512, layers=6, heads=8, ffn=2048, n_mels=80, max_positions=1500)
    labels = synth.make_labels(70)
    sd_np = synth.make_state_dict(cfg, len(labels), seed=45)
    sd8 = O.to_torch_state_dict(synth.round_weights_fp8(sd_np))
    enc, arch = resolve_encoder_arch(cfg["model"])
    hc = synth.head_config(cfg["model"])
    wav = torch.from_numpy(synth.make_batch(905, 1, int(secs * 16000), seed=45))
    lang = torch.zeros(1, dtype=torch.int64)
    torch.set_num_threads(8)

    def run(fmt):
        with torch.no_grad():
            lg, of = O.forward(wav, lang, sd8, enc, arch, hc, act_fp8=fmt)
        return lg

    t0 = time.time()
    ref = run(False)
    print("W8 oracle: %.1f s, logit std %.3f" % (time.time() - t0, float(ref.std())))
    arg_ref = ref.argmax(-1)
    top2 = ref.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    P = ("ln1", "ctx", "ln2", "gelu")
    rows = [("bf16 everywhere (the W8 build)", dict.fromkeys(P, "bf16")),
            ("round 3: row8 / fix8", O.ACT_FP8_ROUND3),
            ("blk8 everywhere", dict.fromkeys(P, "blk8")),
            ("pair8 everywhere", dict.fromkeys(P, "pair8"))]
    for k in P:
        rows.append(("blk8 at %s only (others bf16)" % k, {**dict.fromkeys(P, "bf16"), k: "blk8"}))
    for k in P:
        rows.append(("pair8 at %s, blk8 elsewhere" % k, {**dict.fromkeys(P, "blk8"), k: "pair8"}))
    rows.append(("pair8 at ln1 + ln2, blk8 at ctx + gelu", {"ln1": "pair8", "ln2": "pair8", "ctx": "blk8", "gelu": "blk8"}))
    rows.append(("pair8 at ctx + gelu, blk8 at ln1 + ln2", {"ln1": "blk8", "ln2": "blk8", "ctx": "pair8", "gelu": "pair8"}))
    rows.append(("pair8 at ln1 + ln2 + ctx, blk8 at gelu", {"ln1": "pair8", "ln2": "pair8", "ctx": "pair8", "gelu": "blk8"}))
    rows.append(("pair8 at ln1 + ln2 + gelu, blk8 at ctx", {"ln1": "pair8", "ln2": "pair8", "ctx": "blk8", "gelu": "pair8"}))
    print("%-46s %9s %9s %7s %9s" % ("activation format", "max", "mean", "flips", "flips@2tau"))
    for name, fmt in rows:
        lg = run(fmt)
        err = (lg - ref).abs()
        flips = lg.argmax(-1) != arg_ref
        tau2 = 2 * 0.4 * float(ref.std()) / 6.5
        print("%-46s %9.4f %9.4f %7d %9d   (of %d; %.0f %% graded at 2 tau)" % (
            name, float(err.max()), float(err.mean()), int(flips.sum()), int((flips & (margin > tau2)).sum()), flips.numel(),
            100 * float((margin > tau2).float().mean())))


if __name__ == "__main__":
    main()
