import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import synthetic as synth
from wfl_asr_amd.tagger import BIOPhonemeTagger
from cases import tiny_whisper_config

for bilstm in (False, True):
    cfg = tiny_whisper_config(enable_bilstm=bilstm)
    cfg["model"]["encoder_arch"]["max_positions"] = 1500
    labels = synth.make_labels(5)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=31).items()}
    m = BIOPhonemeTagger(cfg, labels); m.load_state_dict(sd); m.to("cuda")
    x = synth.make_clip(702, 80000, seed=31)
    lang = [0]
    def run(B, L, lens, graph, slot=0):
        w = torch.zeros(B, L)
        w[:] = 0.123          # junk beyond lens
        w[slot, :80000] = torch.from_numpy(x)
        ln = None
        if lens:
            ln = np.zeros(B, np.int32); ln[slot] = 80000
        elif L > 80000:
            w[slot, 80000:] = 0
        r = m.label(w.cuda(), [0] * B, threshold=0.3, lens=ln, want_logits=True, graph=graph)
        torch.cuda.synchronize()
        return r.logits[slot].clone()
    ref = run(1, 80000, False, False)
    for name, args in [("B1 L80000 again", (1, 80000, False, False)), ("B1 L480000 zero", (1, 480000, False, False)),
                       ("B1 L480000 lens", (1, 480000, True, False)), ("B16 lens eager slot0", (16, 480000, True, False)),
                       ("B16 lens eager slot2", (16, 480000, True, False, 2)), ("B16 lens graph slot2", (16, 480000, True, True, 2)),
                       ("B16 lens graph slot2 again", (16, 480000, True, True, 2))]:
        got = run(*args)
        d = (got - ref).abs()
        print(f"bilstm={bilstm} {name:28s} max diff {d.max().item():.6f} frames differing {(d.amax(-1) > 0).sum().item()}")
