"""CPU: the validation pass of wfl-asr_amd/validate.py (/root/reference/train.py:456-545) -- metrics against
hand-computed cases and the loop restatement in oracle/wfl_metrics.py, `evaluate` end to end on a stand-in model."""
import numpy as np
import torch

from oracle import wfl_metrics as M
from wfl_asr_amd import validate as V


def _segs(names, t0=0.0, dur=0.1):
    return [(t0 + i * dur, t0 + (i + 1) * dur, n) for i, n in enumerate(names)]


def test_metric_known_answers():
    gt = _segs(["a", "b", "c", "d"])
    assert V.phoneme_error_rate(gt, gt) == 0.0
    assert V.phoneme_error_rate(_segs(["a", "x", "c"]), gt) == 2 / 4            # one substitution, one deletion
    assert V.phoneme_error_rate(_segs(["q", "a", "b", "c", "d", "r"]), gt) == 2 / 4
    assert V.phoneme_error_rate(_segs(["a"]), []) == 1.0                         # empty reference: / max(0, 1)
    assert V.phoneme_error_rate([], gt) == 1.0
    # timing: ground truth b at [0.1, 0.2]; the FIRST predicted b is used, the later (closer) one ignored
    pred = [(0.00, 0.05, "a"), (0.13, 0.26, "en/b"), (0.1, 0.2, "b")]
    want = ((abs(0.0 - 0.0) + abs(0.1 - 0.05)) + (abs(0.1 - 0.13) + abs(0.2 - 0.26))) / 2 / 2 / 0.1
    assert abs(V.timing_error(pred, _segs(["a", "b"])) - want) < 1e-12
    assert V.timing_error(pred, _segs(["z"])) == 0.0
    assert V.phoneme_name((0.0, 1.0, [["ja/k"]])) == "k" and V.phoneme_name("en/AA") == "AA"
    assert V.framewise_accuracy(torch.tensor([1, 2, 3, 4]), torch.tensor([1, 0, 3, 0])) == 0.5
    assert V.framewise_accuracy(torch.zeros(0), torch.zeros(0)) == 0.0


def test_metrics_match_the_loop_restatement():
    rng = np.random.default_rng(5)
    names = ["a", "b", "en/c", "d", "ja/c", "e"]
    for _ in range(200):
        def draw():
            n = int(rng.integers(0, 12))
            t = np.sort(rng.uniform(0, 3, size=2 * n)).reshape(n, 2) if n else np.zeros((0, 2))
            return [(float(a), float(b), names[int(rng.integers(len(names)))]) for a, b in t]
        p, g = draw(), draw()
        assert V.phoneme_error_rate(p, g) == M.edit_rate(p, g)
        assert abs(V.timing_error(p, g) - M.timing_rate(p, g)) < 1e-12
    lg = rng.normal(size=(3, 17, 9))
    lb = rng.integers(0, 9, size=(3, 17))
    assert V.framewise_accuracy(torch.from_numpy(lg).argmax(-1), torch.from_numpy(lb)) == M.frame_accuracy(lg, lb)


class _Writer:
    def __init__(self):
        self.scalars = {}

    def add_scalar(self, k, v, step):
        self.scalars[k] = (v, step)


def test_evaluate_end_to_end_on_a_stand_in_model(capsys):
    labels = ["B-a", "I-a", "B-b", "I-b", "O"]
    T, C = 12, len(labels)
    want_ids = torch.tensor([[0, 1, 1, 4, 2, 3, 3, 3, 4, 4, 0, 1], [2, 3, 4, 4, 0, 1, 1, 1, 4, 4, 4, 4]])
    seen = {}

    def model(x, lang, max_label_len=None):
        seen["max_label_len"] = max_label_len
        lg = torch.full((x.size(0), max_label_len, C), -5.0)
        lg.scatter_(2, want_ids[:, :max_label_len, None], 5.0)
        return lg, torch.full((x.size(0), max_label_len, 2), 0.5)

    label_ids = want_ids.clone()
    label_ids[1, 0] = 4                                               # one wrong frame of 10 in clip 1
    lengths = torch.tensor([12, 10])
    gt0 = [(0.0, 0.07, "a"), (0.09, 0.17, "en/b"), (0.21, 0.23, "a")]
    gt1 = [[(0.01, 0.05, "b"), (0.09, 0.17, "a")]]                    # (train.py's collate sometimes wraps the list once more)
    batch = (torch.zeros(2, 3840), label_ids, [None, None], [gt0, gt1], ["x", "y"], torch.tensor([0, 1]), lengths)
    cfg = {"data": {"frame_duration": 0.02}, "postprocess": {"median_filter": 1, "merge_segments": "none"}}
    w = _Writer()
    loss = V.evaluate(model, [batch], labels, cfg, writer=w, step=7, criterion=torch.nn.CrossEntropyLoss())
    assert seen["max_label_len"] == 12
    r = V.evaluate.last
    assert r["clips"] == 2 and abs(r["accuracy"] - (1.0 + 0.9) / 2) < 1e-9
    # clip 0 decodes to a, b, a; clip 1 (10 frames) to b, a -- the ground truths' names carry a language prefix once: PER counts that
    assert abs(r["per"] - (1 / 3 + 0.0) / 2) < 1e-9
    assert r["ter"] > 0 and w.scalars["val/per"] == (r["per"], 7) and w.scalars["val/loss"][0] == loss and loss > 0
    assert "[Validation]" in capsys.readouterr().out
    assert V.evaluate(model, [], labels, cfg) == 0
