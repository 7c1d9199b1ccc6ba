#!/usr/bin/env python3
"""Headline benchmark: audio-seconds labeled per second per node (BASELINE.json metric).

One process per GPU (`--gpus N`; for N > 1 launched by torch.distributed.run, RCCL backend).  A step is one pass of
the labeling hot path over one batch of synthetic clips that is already resident in HBM:
  log-mel / conv feature encoder -> Whisper or WavLM encoder -> lang_proj -> BiLSTM / Conformer / dilated-conv head ->
  classifier + offset head -> tag decision -> (N > 1: ONE RCCL gather of the packed tags to rank 0) -> tags copied to pinned
  host memory on the owning rank.
Default workload = BASELINE.json configs[1]: Whisper-base + 2 Conformer blocks, bf16, 16 x 30 s clips per GPU (weak scaling:
the clips are independent, ranks share nothing but the final gather).  `--config-index 2|3|4` selects the other BASELINE
configs at their per-GPU sizes (64 x 10 s WavLM-large + BiLSTM + dilated; 64 x 30 s Whisper-small + full head; 32 x 30 s
Whisper-large-v3), `--full-head` the reference's default config.yaml head on Whisper-base.  Steps alternate between `--inflight`
HIP streams (default: what the product's labelling loops keep -- 2, or 3 with `--full-head`, tagger.batches_in_flight), each with
its own workspace and pinned host buffer, so that many batches are in flight per GPU (every step is still a complete pass over its
own batch, and all K steps are inside the timed, fenced region).

Besides the contract fields the JSON line carries
  roofline        the dominant kernel family (bf16 MFMA GEMMs): algorithmic FLOPs / HIP-event time per launch, measured on the
                  launch stream in a second pass over the same K steps (one batch at a time, so nothing else shares the GPU with
                  the kernel being timed), against the 2.5 PFLOP/s dense bf16 MFMA peak; `breakdown_ms_per_step` gives the same
                  event timing for attention, the BiLSTM recurrence, log-mel and LayerNorm
  value_with_h2d  the same K steps with the waveforms crossing PCIe inside the step: >= 8 distinct pinned host batches in
                  rotation, copied on a separate stream that runs ahead of the compute streams (SURVEY.md §8d metric definition:
                  "from first H2D of a batch"); `value` itself keeps the inputs resident, as the bench contract asks
  cpu_baseline    the oracle (pure-torch fp32 CPU restatement of the reference forward, kind "port") timed on this box's host
                  cores on a bounded sample of the same workload, at B = 1 per call (the reference's own loop, infer.py:261)
                  and batched (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

SR = 16000
MFMA_BF16_PEAK_TFLOPS = 2500.0        # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0         # same table: ~5 PF dense fp8 (the block-scaled K = 128 forms; the non-scaled fp8 MFMA issues at
                                      # the bf16 rate, but the roof of an fp8 config is the fp8 roof whichever instruction the build chose)
# per-GPU batch and clip length of the BASELINE configs (cfg3 64 x 10 s, cfg4 512 / 8, cfg5 256 / 8)
CONFIG_SHAPE = {1: (16, 30.0), 2: (64, 10.0), 3: (64, 30.0), 4: (32, 30.0)}
OTHER_KEYS = {2040: "attention (attn_kernel*)", 2041: "BiLSTM recurrence (lstm_kernel)", 2042: "log-mel (logmel_*_kernel)",
              2043: "LayerNorm (layernorm_kernel)", 2044: "positional conv (posconv_kernel)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (default: the BASELINE config's per-GPU batch)")
    ap.add_argument("--clip-seconds", type=float, default=0.0, help="default: the BASELINE config's clip length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips per batched CPU-baseline call (default 16 for configs[1], else 4)")
    ap.add_argument("--cpu-calls", type=int, default=5)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the HIP-event pass that times every launch")
    ap.add_argument("--no-h2d", action="store_true", help="skip the with-H2D leg")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather every step's packed tags on rank 0 with one RCCL collective (wfl_asr_amd.dist.gather_packed). "
                         "Off by default: the path shards by clip and has no exchange step -- every rank keeps its own tags, as "
                         "infer_folder under torchrun writes its own .lab files")
    ap.add_argument("--graph", action="store_true",
                    help="replay one captured HIP graph per step and workspace slot instead of launching eagerly (measured gain < 1 %%)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight per GPU: step i runs on stream i %% inflight with its own workspace and host buffer")
    ap.add_argument("--config-index", type=int, default=1, help="BASELINE.json configs[] index (1..4)")
    ap.add_argument("--precision", default="default", choices=["default", "high"],
                    help="model.precision: high = every GEMM and the attention as three bf16 passes over split operands (the reference's tag indices)")
    ap.add_argument("--full-head", action="store_true",
                    help="Whisper-base + the reference's default config.yaml head (2-layer BiLSTM, 2 Conformer, 2 dilated convs)")
    ap.add_argument("--activation-dtype", default="bf16", choices=["bf16", "fp8", "fp8_pair", "fp8_nonscaled"],
                    help="--config-index 4 (fp8 weights): model.activation_dtype -- bf16 (default: the reference's arithmetic on the fp8 "
                         "checkpoint) or fp8 (e4m3 GEMM inputs too: faster, 5-9 %% of the raw tags differ; an opt-in)")
    ap.add_argument("--no-precision-high", action="store_true",
                    help="skip the second timed leg (`model.precision: high`, the label-exact mode; default workload only)")
    ap.add_argument("--high-steps", type=int, default=0, help="steps of the precision-high leg (default: min(steps, 10))")
    ap.add_argument("--dry-launch", action="store_true",
                    help="walk the launcher and the N-rank host path on CPU (gloo, no GPU, no kernels): shard plan, barrier, one gather "
                         "of fake packed tags, max-reduce of the clock; prints the JSON line with value null")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (default: a free one)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks ourselves, one process
    per GPU, as children of this process (which has not touched the GPU and never will), relay their output -- rank 0 prints the
    one JSON line -- and return the launcher's exit status.  The driver's own form (`python -m torch.distributed.run ... bench.py
    --gpus N`) arrives with WORLD_SIZE set and never gets here."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_launch(args, rank, world):
    """The host side of an N-rank run without a GPU: gloo rendezvous, the LPT shard plan every rank computes for itself, the fenced
    timed region with a fake step (a packed tag blob of the real size per rank, ONE gather to rank 0), the max-reduce of the clock
    and the JSON line.  What it cannot show is a rate."""
    import torch.distributed as dist
    from wfl_asr_amd.dist import gather_packed, shard_items, split_packed
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, clip_seconds = CONFIG_SHAPE[args.config_index]
    T = 1500 if args.config_index != 2 else 499
    plan = shard_items([int(clip_seconds * SR)] * (B * world), world)
    assert sorted(i for p in plan for i in p) == list(range(B * world)) and all(len(p) == B for p in plan)
    words = B * T * 4 + 1
    blob = torch.full((words,), rank, dtype=torch.int32)
    blob[-1] = 0

    def fence():
        if world > 1:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = gather_packed(blob, dst=0)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        assert got.shape == (world, words)
        for r in range(world):
            ids, _, _, status = split_packed(got[r], B, T)
            assert int(ids[0, 0]) == r and int(status[0]) == 0
        print(json.dumps({"metric": "audio_seconds_labeled_per_sec_per_node", "value": None, "unit": "audio-s/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "dry_launch": True,
                          "config": {"workload": "dry launch: host path of BASELINE configs[%d] on CPU (gloo), no kernels" % args.config_index,
                                     "clips_per_gpu": B, "clip_seconds": clip_seconds, "parallelism": f"clip-sharded dp{world}",
                                     "clips_per_rank": [len(p) for p in plan]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parity_record():
    """The committed held-out parity record (tests/test_gpu_heldout.py on MI355X: 64 clips, seed 777, product `Labeler` vs the oracle
    at B = 1, checkpoint as given), quoted beside the two rates so that a reader sees what each mode's labels are worth."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_parity_heldout.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        d["source"] = os.path.relpath(files[-1], ROOT)
        return d
    except (OSError, ValueError):
        return None


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled per the gfx950 note); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return {"hbm_bytes_per_launch": d["gemm_family_hbm_mb_per_launch"] * 1e6, "source": os.path.relpath(files[-1], ROOT)}
    except (OSError, KeyError, ValueError):
        return None


def host_cores() -> int:
    """Cores this process may actually use: the cgroup CPU quota when there is one (the GPU box hands a 1-GPU job a
    share of the host, not all 256 hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("WFL_CPU_BASELINE_THREADS")
    if env:
        n = int(env)
    return n


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, labels, sd_np, clips, calls, clip_seconds):
    """The oracle forward on host cores.  Only this leg (and tests / smoke) may touch oracle/."""
    from oracle import wfl_oracle as O
    import synthetic as synth
    from wfl_asr_amd.archs import resolve_encoder_arch
    cores = host_cores()
    torch.set_num_threads(cores)
    enc, arch = resolve_encoder_arch(cfg["model"])
    sd = O.to_torch_state_dict(sd_np)
    hc = synth.head_config(cfg["model"])
    L = int(clip_seconds * SR)
    wav_all = torch.from_numpy(synth.make_batch(90000, clips, L, seed=1))
    o_id = labels.index("O")

    def timed(nb):
        wav = wav_all[:nb]
        lang = torch.arange(nb) % cfg["model"]["num_languages"]

        def call():
            lg, of = O.forward(wav, lang, sd, enc, arch, hc)
            return O.tags_from_logits(lg, o_id, 0.5)

        call()                                   # warm-up
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        return nb * clip_seconds / med, med

    v1, m1 = timed(1)
    vb, mb = timed(clips)
    return {
        "value": vb, "unit": "audio-s/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"median of {calls} calls x {clips} clips x {clip_seconds:g} s in one batch (same config, fp32, torch CPU, 1 warm-up call); "
                  f"median call {mb:.3f} s",
        "value_batch1": v1,
        "sample_batch1": f"median of {calls} calls x 1 clip x {clip_seconds:g} s, B = 1 per call like the reference's loop (infer.py:261); "
                         f"median call {m1:.3f} s",
        "torch_threads": torch.get_num_threads(),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))          # nothing in this process has touched, or will touch, the GPU
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        return dry_launch(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    if args.config_index not in CONFIG_SHAPE:
        raise SystemExit("--config-index must be one of 1, 2, 3, 4")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import synthetic as synth
    from wfl_asr_amd.tagger import BIOPhonemeTagger, raise_on_status
    from wfl_asr_amd.dist import gather_packed

    cfg = synth.base_config("whisper") if args.full_head else synth.baseline_config(args.config_index)
    if args.precision != "default":
        cfg["model"]["precision"] = args.precision
    if args.activation_dtype != "bf16":
        cfg["model"]["activation_dtype"] = args.activation_dtype
    def_b, def_s = (16, 30.0) if args.full_head else CONFIG_SHAPE[args.config_index]
    B = args.batch or def_b
    clip_seconds = args.clip_seconds or def_s
    labels = synth.make_labels(70)
    sd_np = synth.make_state_dict(cfg, len(labels), seed=1)
    model = BIOPhonemeTagger(cfg, labels, device=dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model.to(dev).eval()

    L = int(clip_seconds * SR)
    # distinct clips for the first 16 rows, the rest are rolled / rescaled copies of them (generating 64 x 30 s in numpy takes
    # longer than the whole benchmark; every row still differs from every other)
    n_gen = min(B, 16)
    base = synth.make_batch(10000 + rank * n_gen, n_gen, L, seed=1)
    rows = [np.roll(base[i % n_gen], 977 * (i // n_gen)) * (1.0 - 0.03 * (i // n_gen)) for i in range(B)]
    wav_host = np.stack(rows).astype(np.float32)
    wav = torch.from_numpy(wav_host).to(dev)                                              # resident in HBM
    lang = (torch.arange(B, device=dev) % cfg["model"]["num_languages"]).to(torch.int32)
    T = model.num_frames(L)
    words = B * T * 4 + 1                                 # one rank's packed tags: ids | max-prob | offsets | status word
    nfl = args.inflight if args.inflight > 0 else model.batches_in_flight()   # what the product's loops keep: 2, or 3 for a BiLSTM
                                                                              # behind a small encoder (tagger.batches_in_flight)
    gather_on = (world > 1 and args.gather) or bool(os.environ.get("WFL_BENCH_FAKE_WORLD"))
    host_bufs = [torch.zeros((world if (rank == 0 and gather_on) else 1), words, dtype=torch.int32).pin_memory() for _ in range(nfl)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nfl - 1)]
    step_no = [0]
    # the optional gather leg (--gather); WFL_BENCH_FAKE_WORLD walks its code path on one rank
    multi = (world > 1 and args.gather) or bool(os.environ.get("WFL_BENCH_FAKE_WORLD"))
    comm_stream = [torch.cuda.Stream(dev)] if multi else [None]
    gather_bufs = [torch.empty(world, words, dtype=torch.int32, device=dev) for _ in range(nfl)] if (multi and rank == 0) else None

    use_graph = args.graph

    def step(graph=use_graph, single=False, x=None, mdl=None):
        slot = 0 if single else step_no[0] % nfl      # single: the kernel-timing pass runs one batch at a time
        step_no[0] += 1
        host_tags = host_bufs[slot]
        with torch.cuda.stream(streams[slot]):
            out = (mdl or model).label(wav if x is None else x, lang, threshold=0.5, graph=graph, slot=slot)
            if not multi:
                if not os.environ.get("WFL_BENCH_NO_D2H"):        # (diagnostic: how much the tag copy's blit kernel costs the step)
                    host_tags[0].copy_(out.packed, non_blocking=True)
                return slot
        # N > 1: the forward ran on stream `slot`; the collective and the host copy are issued from ONE separate stream (every
        # RCCL call of this process comes from that stream, in program order), which waits for that forward only, so the next
        # step's forward (other stream) still overlaps
        comm = comm_stream[0]
        comm.wait_stream(streams[slot])
        out.packed.record_stream(comm)
        with torch.cuda.stream(comm):
            got = gather_packed(out.packed, dst=0, out=gather_bufs[slot] if rank == 0 else None)   # ONE collective, no re-packing
            if rank == 0:
                host_tags.copy_(got, non_blocking=True)
        return slot

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def check_status():
        for hb in host_bufs:                       # every rank's status word of the last step on each slot
            for r in range(hb.shape[0]):
                raise_on_status(int(hb[r, words - 1]))

    # setup, not steps: every slot's workspace, stream and pinned buffer exists before the W warm-up steps (W may be smaller than
    # the number of slots; a slot first touched inside the timed region would put its workspace allocation there)
    for _ in range(nfl):
        step()
    fence()
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    check_status()

    # ---- with-H2D leg: the same K steps, each on a batch that crosses PCIe inside the step
    h2d = None
    if not args.no_h2d:
        n_rot = 8
        pinned = [torch.from_numpy(np.roll(wav_host, j, axis=0).copy()).pin_memory() for j in range(n_rot)]
        dev_in = [torch.empty(B, L, dtype=torch.float32, device=dev) for _ in range(nfl + 1)]
        copy_stream = torch.cuda.Stream(dev)
        copied = [torch.cuda.Event() for _ in range(nfl + 1)]
        consumed = [torch.cuda.Event() for _ in range(nfl + 1)]

        def step_h2d(i):
            k = i % (nfl + 1)
            slot = step_no[0] % nfl
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[k])                  # the forward that last read this device buffer is done
                dev_in[k].copy_(pinned[i % n_rot], non_blocking=True)
                copied[k].record(copy_stream)
            streams[slot].wait_event(copied[k])
            step(x=dev_in[k])
            consumed[k].record(streams[slot])

        for k in range(nfl + 1):
            consumed[k].record(streams[0])
        # warm-up: at least one crossing of EVERY pinned batch (the first DMA out of a freshly pinned buffer is several times slower than
        # the later ones -- tools/h2d_lab.py, profiles/round4_h2d_lab.txt: with W = 3 of 8 buffers warmed the leg measured that first
        # touch, 111-115 k against 127 k once every buffer had crossed; the product's loops reuse three pinned buffers for a whole folder)
        for i in range(max(args.warmup, n_rot)):
            step_h2d(i)
        fence()
        th = time.perf_counter()
        for i in range(args.steps):
            step_h2d(i)
        fence()
        el_h = time.perf_counter() - th
        check_status()
        if world > 1:
            t = torch.tensor([el_h], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_h = float(t.item())
        h2d = {"value_with_h2d": world * B * clip_seconds * args.steps / el_h, "ms_per_step_with_h2d": 1e3 * el_h / args.steps,
               "h2d_bytes_per_step": B * L * 4, "distinct_host_batches": n_rot}
        del pinned, dev_in

    # Roofline pass: the same K steps again, launched eagerly with a HIP-event pair around every GEMM / attention / LSTM /
    # log-mel / LayerNorm launch on the launch stream (events cannot live inside a replayed graph; kernels and inputs are identical).
    use_events = not args.no_kernel_events
    prof = []
    if use_events:
        step(graph=False, single=True)
        fence()
        model.gemm_profile(True)
        model.gemm_profile_read(reset=True)
        te = time.perf_counter()
        for _ in range(args.steps):
            step(graph=False, single=True)
        fence()
        eager_ms = 1e3 * (time.perf_counter() - te) / args.steps
        prof = model.gemm_profile_read(reset=True)
        model.gemm_profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- second timed leg: the same workload with `model.precision: high` -- the mode whose tag indices are the reference's (the
    # parity record below); same batch, same fences, every rank, max over ranks.  The headline `value` stays the default build's.
    high = None
    if (args.config_index == 1 and not args.full_head and args.precision == "default" and not args.no_precision_high
            and str(cfg["model"].get("weight_dtype", "bf16")) != "fp8"):
        import copy
        cfg_h = copy.deepcopy(cfg)
        cfg_h["model"]["precision"] = "high"
        model_h = BIOPhonemeTagger(cfg_h, labels, device=dev)
        model_h.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        model_h.to(dev).eval()
        k_h = args.high_steps or min(args.steps, 10)
        step_no[0] = 0
        for _ in range(nfl + min(args.warmup, 3)):
            step(mdl=model_h)
        fence()
        step_no[0] = 0
        th0 = time.perf_counter()
        for _ in range(k_h):
            step(mdl=model_h)
        fence()
        el_hi = time.perf_counter() - th0
        check_status()
        if world > 1:
            t = torch.tensor([el_hi], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_hi = float(t.item())
        high = {"value": world * B * clip_seconds * k_h / el_hi, "unit": "audio-s/s", "ms_per_step": 1e3 * el_hi / k_h, "steps": k_h,
                "dtype": "bf16 pairs (hi + lo operands, three MFMA passes per product, fp32 sums)",
                "what": "same workload, model.precision: high -- the mode that reproduces the reference's tag indices (parity below)"}
        del model_h

    if rank == 0:
        audio_s = world * B * clip_seconds * args.steps
        m = cfg["model"]
        enc_name = m["whisper_model"] if m["encoder_type"] == "whisper" else m["wavlm_model"]
        roof = None
        if prof:
            acts = {0: 0, 1: 1, 2: 2, 3: 3}
            fp8_act = str(m.get("weight_dtype", "bf16")) == "fp8" and os.environ.get("WFL_FP8_ACT", "0" if m.get("activation_dtype", "bf16") == "bf16" else "1") != "0"

            mx_act = {"fp8": "single", "e4m3": "single", "float8_e4m3fn": "single", "fp8_pair": "pair", "e4m3_pair": "pair"}.get(
                str(m.get("activation_dtype", "bf16")))

            def kname(key):
                """rocprofv3 kernel name of the template instantiation behind a profile key (model.hip: Runner::gemm)."""
                act, glu, f32, res, kid = key & 3, bool(key & 4), bool(key & 8), bool(key & 16), (key >> 5) & 7
                lnf, stats = (key >> 8) & 3, bool(key & 1024)
                tf = lambda b: "true" if b else "false"
                if kid == 7 and fp8_act and mx_act:   # both operands e4m3 on the block-scaled MFMA (gemm_mx.hip): <ACT, MT, RES, PAIR, OUT>
                    pair = mx_act == "pair"
                    return "gemm_mx_kernel<%d, %d, %s, %s, %d>" % (acts[act], 6 if pair else 5, tf(res), tf(pair),
                                                                    (2 if pair else 1) if (act == 1 and not res) else 0)
                if kid in (1, 5, 6, 7):               # 6 = the tap-stationary conv mode, 7 = fp8: e4m3 weights (W8) or both operands e4m3 (A8)
                    a8 = kid == 7 and fp8_act
                    return "gemm_stream_kernel<%d, %d, %s, %d, %s, %s, %s, %s, %s>" % (
                        acts[act], 8 if kid == 5 else 6, tf(res), lnf, tf(stats), tf(kid == 6), tf(kid == 7 and not a8), tf(a8),
                        tf(a8 and act == 1 and not res))
                if kid in (2, 3):
                    return "gemm256_kernel<%d, %s, %s, %d>" % (acts[act], tf(glu), tf(f32), 6 if kid == 2 else 8)
                return "gemm_bf16_kernel<%d, %s, %s>" % (acts[act], tf(glu), tf(f32))

            is_fp8 = str(m.get("weight_dtype", "bf16")) == "fp8"
            peak = MFMA_FP8_PEAK_TFLOPS if is_fp8 else MFMA_BF16_PEAK_TFLOPS
            gemm = [p for p in prof if p["key"] < 2040]
            other = [p for p in prof if p["key"] >= 2040]
            tot_ms = sum(p["ms"] for p in gemm)
            tot_fl = sum(p["flops"] for p in gemm)
            tot_n = sum(p["launches"] for p in gemm)
            top = max(gemm, key=lambda p: p["ms"])
            # (the committed PMC passes are of the default command: they describe BASELINE configs[1] in bf16 only)
            traffic = pmc_traffic() if (args.config_index == 1 and not args.full_head and args.precision == "default") else None
            breakdown = {"bf16 MFMA GEMM family": tot_ms / args.steps}
            for p in other:
                breakdown[OTHER_KEYS.get(p["key"], str(p["key"]))] = p["ms"] / args.steps
            roof = {
                "bound": "mfma",
                "kernel": "MFMA GEMM family (gemm_stream_kernel / gemm_mx_kernel / gemm256_kernel / gemm_bf16_kernel, all instantiations)"
                          + ("; fp8 model: every launch priced against the 5 PF dense fp8 roof" if is_fp8 else "; bf16, 2.5 PF dense roof"),
                "timing": "HIP events (launch stream) around every launch of a second pass over the same %d steps, run right after the "
                          "timed region, one batch at a time so that no other kernel shares the GPU with the one being timed "
                          "(%.3f ms/step with the events in)" % (args.steps, eager_ms),
                "achieved": tot_fl / tot_ms / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": tot_fl / tot_ms / 1e9 / peak,
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "launches_per_step": tot_n / args.steps, "avg_launch_us": 1e3 * tot_ms / tot_n,
                "gflop_per_launch": tot_fl / tot_n / 1e9, "gemm_ms_per_step": tot_ms / args.steps,
                "breakdown_ms_per_step": breakdown,
                "other_kernels": [{"kernel": OTHER_KEYS.get(p["key"], str(p["key"])), "launches": p["launches"],
                                   "avg_us": 1e3 * p["ms"] / p["launches"],
                                   "tflops": (p["flops"] / p["ms"] / 1e9) if p["flops"] else None} for p in other],
                "variants": [
                    {"kernel": kname(p["key"]), "launches": p["launches"], "avg_us": 1e3 * p["ms"] / p["launches"],
                     "tflops": p["flops"] / p["ms"] / 1e9, "frac": p["flops"] / p["ms"] / 1e9 / peak}
                    for p in sorted(gemm, key=lambda p: -p["ms"])],
                "top_variant": {"kernel": kname(top["key"]), "avg_us": 1e3 * top["ms"] / top["launches"],
                                "tflops": top["flops"] / top["ms"] / 1e9},
            }
        result = {
            "metric": "audio_seconds_labeled_per_sec_per_node", "value": audio_s / elapsed, "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ({"fp8": "fp8 (e4m3 weights, ONE e4m3 value per activation, block-scaled fp8 MFMA; bf16 elsewhere)",
                       "fp8_pair": "fp8 (e4m3 weights, activations as e4m3 pairs hi + lo, block-scaled fp8 MFMA; bf16 elsewhere)",
                       "fp8_nonscaled": "fp8 (e4m3 weights and activations, non-scaled fp8 MFMA; bf16 elsewhere)"}.get(
                           str(m.get("activation_dtype", "bf16")), "bf16 activations x fp8 (e4m3) encoder weights"))
                     if str(m.get("weight_dtype", "bf16")) == "fp8" else
                     ("bf16 pairs (model.precision: high -- hi + lo operands, three MFMA passes, fp32 sums)" if args.precision == "high" else "bf16"),
            "data": "synthetic",
            "config": {"workload": ("default config.yaml head: " if args.full_head else "BASELINE configs[%d]: " % args.config_index)
                       + "%s + %s%d Conformer blocks%s, %d x %g s clips per GPU" % (
                enc_name, "%d-layer BiLSTM + " % m["bilstm_num_layer"] if m["enable_bilstm"] else "", m["num_conformer_layers"],
                " + %d dilated convs" % m["dilated_conv_depth"] if m["enable_dilated_conv"] else "", B, clip_seconds),
                "clips_per_gpu": B, "clip_seconds": clip_seconds, "frames_per_clip": T, "tags": len(labels),
                "parallelism": f"clip-sharded dp{world}",
                "collective": ("one RCCL gather of the packed tags per step" if multi else
                               "none on the data path (every rank keeps its own tags); barrier + max-reduce of the clock only"),
                "launch": "hip graph replay" if use_graph else "eager",
                "batches_in_flight": nfl},
            "roofline": roof,
        }
        if h2d:
            result.update(h2d)
        if high:
            result["precision_high"] = high
        par = parity_record()
        if par and args.config_index == 1 and not args.full_head:
            result["parity"] = par
        if world == 1 and not args.no_cpu_baseline:
            clips = args.cpu_clips or (16 if (args.config_index == 1 and not args.full_head) else 4)
            result["cpu_baseline"] = cpu_baseline(cfg, labels, sd_np, clips, args.cpu_calls, clip_seconds)
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
