"""CPU: `python bench.py --gpus N` starts its own N ranks (round 4).  The launcher branch and the N-rank host path
(gloo rendezvous on 127.0.0.1, shard plan, fenced region, ONE gather of packed tags, max-reduce of the clock, rank 0's JSON line) are
walked with `--dry-launch`: no GPU, no kernels, no rate."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=e)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                  # rank 0 prints ONE line; the other ranks print nothing
    return json.loads(lines[0])


def test_bare_command_with_two_gpus_starts_two_ranks():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_launch"] is True
    assert d["config"]["parallelism"] == "clip-sharded dp2" and d["config"]["clips_per_rank"] == [16, 16]
    assert d["scaling"] == "weak" and d["metric"] == "audio_seconds_labeled_per_sec_per_node"


def test_one_gpu_does_not_launch_anything():
    r = _run("--dry-launch", "--steps", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_a_world_size_that_contradicts_gpus_is_refused():
    r = _run("--gpus", "2", "--dry-launch", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_bare_command():
    # config index 9 does not exist: every rank raises, the launcher's status comes back
    r = _run("--gpus", "2", "--dry-launch", "--config-index", "9")
    assert r.returncode != 0
