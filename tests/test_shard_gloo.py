"""CPU test of the multi-GPU path: two gloo ranks partition the batch, run attention on their slab
and reassemble — the same code (rocwmma_fattn/shard.py) the RCCL ranks run on MI355X, with a dense
torch attention standing in for the GPU operator (the HIP kernel itself is covered by -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rocwmma_fattn.shard import gather_batch, local_batch, scatter_batch, shard_bounds, sharded_attention


def test_shard_bounds_partition():
    for total in (1, 2, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _dense(q, k, v, mask, causal, scale):
    return torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=bool(causal), scale=scale)


def _worker(rank, world, port, B, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        shape = (B, 2, 24, 16)
        full = [torch.randn(shape) for _ in range(3)]            # identical on every rank (same seed)
        # path 1: root holds the tensors -> scatter slabs -> local attention -> gather
        slabs = [scatter_batch(t if rank == 0 else None, shape, torch.float32, "cpu", src=0) for t in full]
        lo, hi = shard_bounds(B, world, rank)
        for s, t in zip(slabs, full):
            assert torch.equal(s, t[lo:hi])
        o_local = sharded_attention(*slabs, causal=True, scale=None, attention_fn=_dense)
        o_full = gather_batch(o_local, B)
        ref = _dense(*full, None, True, None)
        assert torch.allclose(o_full, ref, atol=1e-6)
        # path 2: data generated per rank (bench.py): local_batch views need no communication at all
        o2 = sharded_attention(*(local_batch(t, world, rank) for t in full), causal=False, attention_fn=_dense)
        assert torch.allclose(o2, _dense(*full, None, False, None)[lo:hi], atol=1e-6)
        # max-over-ranks reduction used for the timing
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])   # even and uneven split
def test_two_rank_scatter_compute_gather(B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, B, results), nprocs=world, join=True)
    assert dict(results) == {0: "ok", 1: "ok"}


def test_bench_refuses_more_ranks_than_gpus_and_becomes_its_own_launcher(monkeypatch, capsys):
    """bench.py --gpus N without a launcher: (a) asks for N visible GPUs (none here -> a clear refusal, not a hang); (b) with
    --same-device it replaces itself by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` (os.execv intercepted here)."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    if torch.cuda.device_count() < 8:
        monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"])
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert "only" in str(e.value) and "--gpus 8" in str(e.value)
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--same-device"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert a[a.index("--nproc-per-node") + 1] == "2" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(a[a.index("--master-port") + 1]) <= 65535
    i = a.index(os.path.join(root, "bench.py"))
    assert a[i + 1:] == ["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--same-device"]
