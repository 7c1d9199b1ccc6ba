import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import synthetic as synth
from wfl_asr_amd.tagger import BIOPhonemeTagger
from cases import tiny_whisper_config

cfg = tiny_whisper_config(enable_bilstm=False)
cfg["model"]["encoder_arch"]["max_positions"] = 1500
labels = synth.make_labels(5)
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=31).items()}
m = BIOPhonemeTagger(cfg, labels); m.load_state_dict(sd); m.to("cuda")
B, L = 16, 480000
w = torch.from_numpy(synth.make_batch(800, B, L, seed=5)).cuda()
lang = torch.zeros(B, dtype=torch.int32, device="cuda")
def run(graph, **kw):
    r = m.label(w, lang, threshold=0.3, want_logits=True, graph=graph, **kw)
    torch.cuda.synchronize()
    return r.logits.clone(), r.offsets.clone(), r.maxprob.clone()
ref = run(False)
print("eager repeat", (run(False)[0] - ref[0]).abs().max().item())
for i in range(5):
    g = run(True)
    d = (g[0] - ref[0]).abs().amax(dim=(1, 2))
    print("graph replay", i, "clips differing", int((d > 0).sum()), "max", round(float(d.max()), 3), "offs diff", float((g[1] - ref[1]).abs().max()))
    if i == 2:
        print("  (eager in between)", (run(False)[0] - ref[0]).abs().max().item())
