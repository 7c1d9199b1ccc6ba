#!/usr/bin/env python3
"""Diagnostic (not product): the forward must not depend on what its workspace held before (uninitialised reads) -- every model
family is run on a zeroed workspace and on workspaces filled with 0xFF / 0x3F bytes (NaN / finite garbage); logits and offsets must
come out bit-identical.  usage: workspace_poison.py"""
import os, sys
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synthetic as synth
from wfl_asr_amd.tagger import BIOPhonemeTagger
from cases import tiny_whisper_config, tiny_wavlm_config
def run(name, cfg, L, B, lens=None):
    labels = synth.make_labels(12)
    sd = synth.make_state_dict(cfg, len(labels), seed=5)
    m = BIOPhonemeTagger(cfg, labels); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
    wav = torch.from_numpy(synth.make_batch(100, B, L, seed=5) * (0.05 if cfg["model"]["encoder_type"] == "none" else 1.0)).cuda()
    lang = np.zeros(B, np.int64)
    outs = []
    for fill in (0, 0xFF, 0x3F, 0xFF):
        m.label(wav, lang, threshold=0.4, lens=lens)            # make sure the workspace exists
        torch.cuda.synchronize()
        m._ws.fill_(fill)
        o = m.label(wav, lang, threshold=0.4, lens=lens, want_logits=True)
        torch.cuda.synchronize()
        outs.append((o.logits.clone(), o.offsets.clone(), o.ids.clone()))
    ref = outs[0]
    msg = []
    for k, o in enumerate(outs[1:]):
        dl = (o[0] - ref[0]).abs(); do = (o[1] - ref[1]).abs()
        msg.append((bool(torch.isnan(o[0]).any()), float(torch.nan_to_num(dl, nan=1e9).max()), float(torch.nan_to_num(do, nan=1e9).max())))
    print(name, "B", B, "L", L, "poisoned-workspace runs vs zeroed: (nan, max dlogit, max doffset)", msg)
run("wavlm-base cfg1", synth.baseline_config(0), 69468, 1)
run("wavlm-base cfg1", synth.baseline_config(0), 143261, 1)
run("wavlm tiny stable full head", tiny_wavlm_config(True, enable_bilstm=True), 30000, 3)
run("whisper tiny full head", tiny_whisper_config(enable_bilstm=True), 30000, 3)
run("whisper cfg2", synth.baseline_config(1), 480000, 2)
run("mel full head", synth.base_config("none"), 50000, 3)
run("whisper base default head", synth.base_config("whisper"), 480000, 2)
