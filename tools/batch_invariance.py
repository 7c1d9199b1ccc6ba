"""Diagnostic (not product): a clip labelled alone vs inside batches of 32 and 40 clips (several tiles per persistent GEMM
workgroup, in-loop statistics epilogues) must be bit-identical.  tests/test_gpu_model.py checks the same at the benchmark batch."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import synthetic as synth
from wfl_asr_amd.tagger import BIOPhonemeTagger
cfg = synth.baseline_config(1)
if os.environ.get("LAB_PRECISION") == "high":      # the exact-label mode: its slice-by-slice GEMM walk is chosen by shape, never by the batch
    cfg["model"]["precision"] = "high"
labels = synth.make_labels(70)
sd = synth.make_state_dict(cfg, len(labels), seed=1)
m = BIOPhonemeTagger(cfg, labels); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to('cuda').eval()
for B in ((16, 32, 40) if os.environ.get("LAB_PRECISION") == "high" else (32, 40)):
    wav = torch.from_numpy(synth.make_batch(500, B, 480000, seed=3)).cuda()
    lang = (torch.arange(B) % 2).to(torch.int32).cuda()
    big = m.label(wav, lang, threshold=0.5, want_logits=True)
    torch.cuda.synchronize()
    for i in (0, min(17, B - 2), B - 1):
        one = m.label(wav[i:i + 1], lang[i:i + 1], threshold=0.5, want_logits=True)
        torch.cuda.synchronize()
        print(B, i, "bit-identical logits:", torch.equal(one.logits[0], big.logits[i]), "ids:", torch.equal(one.ids[0], big.ids[i]))
