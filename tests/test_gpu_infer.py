"""-m gpu: the reference's inference surface (infer_audio / infer_folder / CLI) over the batched loop: config.yaml,
phonemes.txt, langs.txt, merge map, forced-alignment .txt, checkpoint .pt and .lab files exactly as the reference
lays them out."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from wfl_asr_amd import audio as A
from wfl_asr_amd import infer as I
from wfl_asr_amd import postprocess as pp
import synthetic as synth
from cases import tiny_whisper_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("wfl")
    cfg = tiny_whisper_config(enable_bilstm=True)
    cfg["model"]["encoder_arch"]["max_positions"] = 1500         # real 30 s geometry, tiny widths
    cfg["output"]["save_dir"] = str(d / "save")
    cfg["postprocess"] = {"median_filter": 3, "merge_segments": "right", "confidence_threshold": 0.3}
    os.makedirs(cfg["output"]["save_dir"])
    labels = synth.make_labels(5)
    with open(d / "save" / "phonemes.txt", "w") as f:
        f.write("\n".join(labels) + "\n")
    with open(d / "save" / "langs.txt", "w") as f:
        f.write("en,0\nja,1\n")
    with open(d / "save" / "phoneme_merge_map.json", "w") as f:
        json.dump({"p00": {"en": "AA", "ja": "a"}}, f)
    with open(d / "config.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=31).items()}
    torch.save(sd, d / "best_model.pt")
    os.makedirs(d / "wavs")
    A.write_wav(str(d / "wavs" / "a.wav"), synth.make_clip(700, 16000 * 7, seed=31) * 0.9, 16000)
    A.write_wav(str(d / "wavs" / "B.WAV"), synth.make_clip(701, 16000 * 3, seed=31) * 0.5, 16000)
    A.write_wav(str(d / "wavs" / "long.wav"), synth.make_clip(702, 16000 * 65, seed=31) * 0.8, 16000)
    A.write_wav(str(d / "wavs" / "hi.wav"), A.resample(synth.make_clip(703, 16000 * 2, seed=31).astype(np.float64), 16000, 44100) * 0.9, 44100)
    return d, cfg, labels


def _manual(lab, path, lang_id, thr):
    """The reference's loop spelled out with single-row label() calls (infer.py:237-325)."""
    lang_name = lab._lang_name(lang_id)
    out, clock = [], 0.0
    chunks = A.load_items(path, 16000)       # the product's ingest (native; held against the Python restatement in test_audio_cpu.py)
    raw = chunks
    for c, r in zip(chunks, raw):
        x = torch.from_numpy(np.ascontiguousarray(c))[None].cuda()
        res = lab.model.label(x, None if lang_id is None else [lang_id], threshold=thr, average_languages=lang_id is None)
        # the reference's own host logic (postprocess.py, pinned to the reference by fixtures) -- the product path runs the
        # native implementation (csrc/hostpost.hip), so this also holds the two against each other end to end
        ids = pp.median_filter_ids(res.ids[0].cpu().numpy(), int(lab.config["postprocess"]["median_filter"]))
        tags = [lab.model.id2label[int(i)] for i in ids]
        segs = pp.decode_bio_tags(tags, offsets=res.offsets[0].cpu().numpy())
        if lab.merge_map and lang_name:
            segs = [(s, e, pp.canonical_to_lang(ph, lang_name, lab.merge_map)) for s, e, ph in segs]
        out.extend((s + clock, e + clock, ph) for s, e, ph in segs)
        clock += len(r) / 16000
    return pp.merge_adjacent_segments(out, "right")


def test_infer_audio_and_folder(workdir):
    d, cfg, labels = workdir
    cp, ck = str(d / "config.yaml"), str(d / "best_model.pt")
    segs = I.infer_audio(str(d / "wavs" / "a.wav"), cp, ck, output_lab_path=str(d / "out" / "a.lab"), device="cuda", lang_id=1,
                         confidence_threshold=0.3)
    lab = I._labeler(cp, ck, "cuda")
    assert segs == _manual(lab, str(d / "wavs" / "a.wav"), 1, 0.3)
    text = open(d / "out" / "a.lab").read()
    assert text == "".join(f"{int(s * 1e7)} {int(e * 1e7)} {ph}\n" for s, e, ph in segs)
    assert all(0 <= s <= e for s, e, _ in segs) and all(a[1] <= b[0] + 0.021 for a, b in zip(segs, segs[1:]))
    # long file: 3 chunks, times shifted by the chunk clock
    long_segs = I.infer_audio(str(d / "wavs" / "long.wav"), cp, ck, output_lab_path=None, device="cuda", lang_id=0, confidence_threshold=0.3)
    assert long_segs == _manual(lab, str(d / "wavs" / "long.wav"), 0, 0.3)
    assert max(e for _, e, _ in long_segs) > 60.0
    # lang_id None: average over languages (infer.py:266-276), no merge-map renaming without a language
    avg = I.infer_audio(str(d / "wavs" / "a.wav"), cp, ck, device="cuda", lang_id=None, confidence_threshold=0.3)
    assert avg == _manual(lab, str(d / "wavs" / "a.wav"), None, 0.3)
    with pytest.raises(ValueError):
        I.infer_audio(str(d / "wavs" / "a.wav"), cp, ck, device="cuda", lang_id=7)
    # folder mode: every *.wav (case-insensitive), batched across files, one .lab each; 44.1 kHz input is resampled
    out = d / "labs"
    I.infer_folder(str(d / "wavs"), cp, ck, output_dir=str(out), device="cuda", lang_id=1, confidence_threshold=0.3)
    assert sorted(os.listdir(out)) == ["B.lab", "a.lab", "hi.lab", "long.lab"]
    assert open(out / "a.lab").read() == text
    # the input WAV is never overwritten (the reference's `-o .` single-file mode would)
    before = open(d / "wavs" / "B.WAV", "rb").read()
    I.infer_audio(str(d / "wavs" / "B.WAV"), cp, ck, output_lab_path=str(d / "wavs" / "B.WAV"), device="cuda", lang_id=0)
    assert open(d / "wavs" / "B.WAV", "rb").read() == before and os.path.exists(d / "wavs" / "B.lab")


def test_forced_alignment_and_cli(workdir):
    d, cfg, labels = workdir
    cp, ck = str(d / "config.yaml"), str(d / "best_model.pt")
    free = I.infer_audio(str(d / "wavs" / "a.wav"), cp, ck, device="cuda", lang_id=0, confidence_threshold=0.3)
    forced = [ph for _, _, ph in free][:-1] + ["zz"]
    with open(d / "wavs" / "a.txt", "w") as f:
        f.write(" ".join(forced))
    try:
        got = I.infer_audio(str(d / "wavs" / "a.wav"), cp, ck, device="cuda", lang_id=0, confidence_threshold=0.3)
    finally:
        os.unlink(d / "wavs" / "a.txt")
    assert [ph for _, _, ph in got if ph not in ("SP", "AP")] == forced or [ph for _, _, ph in got] == forced
    r = subprocess.run([sys.executable, os.path.join(ROOT, "infer.py"), str(d / "wavs" / "B.WAV"), "-ckpt", ck, "-c", cp, "-o",
                        str(d / "cli" / "B.lab"), "-l", "1", "-ct", "0.3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Predicted segments:" in r.stdout and os.path.exists(d / "cli" / "B.lab")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "infer.py"), str(d / "wavs"), "-ckpt", ck, "-c", cp, "-s"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "neither --top-k nor --top-p" in r.stdout


def test_three_batches_in_flight_label_like_one(workdir, monkeypatch):
    """The pipelined loops with 3 workspace slots (the default for a BiLSTM behind a small Whisper encoder) against the same loops
    with one batch at a time: 11 short files + a 65 s one, 2 rows per forward, so both the native-loader path and the chunked path
    wrap around their slots several times.  Same segments, bit for bit."""
    d, cfg, labels = workdir
    cp, ck = str(d / "config.yaml"), str(d / "best_model.pt")
    many = d / "many"
    os.makedirs(many, exist_ok=True)
    paths = []
    for i in range(11):
        p = str(many / f"{i:02d}.wav")
        A.write_wav(p, synth.make_clip(900 + i, 16000 * (2 + i % 4), seed=31) * 0.8, 16000)
        paths.append(p)
    paths.append(str(d / "wavs" / "long.wav"))
    for i in range(7):                                   # files at other rates take the general path: with long.wav 10 items in 3 waves
        p = str(many / f"r{i}.wav")
        rate = (44100, 22050, 8000)[i % 3]
        A.write_wav(p, A.resample(synth.make_clip(930 + i, 16000 * (1 + i % 3), seed=31).astype(np.float64), 16000, rate) * 0.7, rate)
        paths.append(p)
    monkeypatch.delenv("WFL_INFLIGHT", raising=False)
    lab3 = I.Labeler(cp, ck, "cuda", batch_size=2)
    lab3._wave_items = 4
    assert lab3.n_inflight == 3
    got3 = lab3.label_files(paths, lang_id=1, confidence_threshold=0.3, verbose=False)
    chunks = [c for p in paths for c in A.load_items(p, 16000)]
    slow3 = lab3._forward_items(chunks, 1, 0.3)
    monkeypatch.setenv("WFL_INFLIGHT", "1")
    lab1 = I.Labeler(cp, ck, "cuda", batch_size=2)
    assert lab1.n_inflight == 1
    got1 = lab1.label_files(paths, lang_id=1, confidence_threshold=0.3, verbose=False)
    slow1 = lab1._forward_items(chunks, 1, 0.3)
    assert got3 == got1 and all(len(g) > 0 for g in got3)
    assert len(slow3) == len(chunks) == 21
    for (i3, o3), (i1, o1) in zip(slow3, slow1):
        assert np.array_equal(i3, i1) and np.array_equal(o3, o1)
    assert got3[0] == _manual(lab1, paths[0], 1, 0.3)
    assert got3[-1] == _manual(lab1, paths[-1], 1, 0.3) and got3[11] == _manual(lab1, paths[11], 1, 0.3)   # a resampled file, the long one


def test_bench_emits_one_json_line_with_the_contract_keys():
    """bench.py is what the driver times: a short run must print exactly one JSON line with the contract's keys (roofline and
    cpu_baseline objects included) and a status-clean forward."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-calls", "1", "--cpu-clips", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 3 and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert "workload" in d["config"]
