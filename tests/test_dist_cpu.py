"""CPU, world_size 2, gloo: the multi-rank path of the labeler (sharding plan + the single tag gather)."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from wfl_asr_amd import dist as wd


def test_shard_items_balanced_and_deterministic():
    costs = [480000, 10, 480000, 300000, 300000, 5, 160000, 160000, 160000]
    plan = wd.shard_items(costs, 4)
    assert sorted(i for p in plan for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in plan]
    assert max(loads) - min(loads) <= max(costs)
    assert plan == wd.shard_items(costs, 4)
    assert wd.shard_items([], 3) == [[], [], []]
    assert wd.shard_items([7], 2) == [[0], []]


def test_pack_roundtrip_bit_exact():
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 141, (3, 50), generator=g, dtype=torch.int32)
    mp_ = torch.rand(3, 50, generator=g)
    off = torch.rand(3, 50, 2, generator=g)
    mp_[0, 0] = float("nan"); off[1, 2, 1] = -0.0
    a, b, c = wd.unpack_tags(wd.pack_tags(ids, mp_, off))
    assert torch.equal(a, ids)
    assert torch.equal(b.view(torch.int32), mp_.view(torch.int32)) and torch.equal(c.view(torch.int32), off.view(torch.int32))


def _free_port():
    """A rendezvous FILE (torch.distributed FileStore): no port to race for, nothing to resolve."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="wfl_gloo_")
    os.close(fd)
    os.unlink(path)
    return path


def _worker(rank, world, port, counts, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + str(port), rank=rank, world_size=world)
    try:
        T = 37
        n = counts[rank]
        g = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, 141, (n, T), generator=g, dtype=torch.int32)
        mp_ = torch.rand(n, T, generator=g)
        off = torch.rand(n, T, 2, generator=g)
        a, b, c = wd.gather_tags(ids, mp_, off, dst=0, counts=counts)
        if rank == 0:
            q.put((a.numpy(), b.numpy(), c.numpy()))      # (plain pickled arrays: a shared-memory tensor handle can die with its sender)
        else:
            assert a is ids and b is mp_ and c is off
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[3, 3], [4, 1], [2, 0]])
def test_gather_tags_world2_gloo(counts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    a, b, c = (torch.from_numpy(x) for x in q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    T = 37
    exp_ids, exp_mp, exp_off = [], [], []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        exp_ids.append(torch.randint(0, 141, (counts[r], T), generator=g, dtype=torch.int32))
        exp_mp.append(torch.rand(counts[r], T, generator=g))
        exp_off.append(torch.rand(counts[r], T, 2, generator=g))
    assert torch.equal(a, torch.cat(exp_ids)) and torch.equal(b, torch.cat(exp_mp)) and torch.equal(c, torch.cat(exp_off))


def _worker_packed(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + str(port), rank=rank, world_size=world)
    try:
        n, T = 3, 29
        g = torch.Generator().manual_seed(500 + rank)
        blob = torch.randint(-2 ** 31, 2 ** 31 - 1, (4 * n * T + 1,), generator=g, dtype=torch.int64).to(torch.int32)
        out = wd.gather_packed(blob, dst=0)
        if rank == 0:
            q.put(out.numpy())
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_packed_world2_gloo_is_one_bit_exact_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n, T = 3, 29
    assert out.shape == (2, 4 * n * T + 1)
    for r in range(2):
        g = torch.Generator().manual_seed(500 + r)
        blob = torch.randint(-2 ** 31, 2 ** 31 - 1, (4 * n * T + 1,), generator=g, dtype=torch.int64).to(torch.int32)
        assert torch.equal(out[r], blob)
        ids, mp_, off, st = wd.split_packed(out[r], n, T)
        assert ids.shape == (n, T) and mp_.shape == (n, T) and off.shape == (n, T, 2) and st.numel() == 1
        assert torch.equal(ids.reshape(-1), blob[:n * T]) and int(st) == int(blob[-1])
        assert torch.equal(off.reshape(-1).view(torch.int32), blob[2 * n * T:4 * n * T])


def test_gather_packed_single_rank_is_a_view():
    blob = torch.arange(4 * 2 * 5 + 1, dtype=torch.int32)
    out = wd.gather_packed(blob)
    assert out.shape == (1, blob.numel()) and out.data_ptr() == blob.data_ptr()


def test_pick_device_follows_local_rank(monkeypatch):
    """infer_folder under torchrun: a bare "cuda" must resolve to cuda:LOCAL_RANK, not to device 0 on every rank."""
    from wfl_asr_amd import infer
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert infer.pick_device("cuda") == torch.device("cuda", 3)
    assert infer.pick_device("cuda:1") == torch.device("cuda", 1)      # an explicit index wins
    with pytest.raises(RuntimeError):
        infer.pick_device("cpu")
