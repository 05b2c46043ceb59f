"""N > 1 path on CPU: world_size-2 gloo run of the shard partition + the trajectory all-gather
(the same code bench.py / users run over RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
    import importlib.util
    import torch
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("sharding", os.path.join(ROOT, "flight.jl_amd", "flightbatch", "sharding.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sh.shard_range(n_total, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    x_local = torch.stack([idx + 1000.0 * k for k in range(27)])      # field k of aircraft i holds i + 1000 k
    full = sh.all_gather_state(x_local, n_total)
    ok = bool(torch.equal(full, torch.stack([torch.arange(n_total, dtype=torch.float64) + 1000.0 * k for k in range(27)])))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, ok))


@pytest.mark.parametrize("n_total", [64, 101])   # even and ragged shards
def test_two_rank_gloo_gather(n_total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total   # contiguous cover
    assert all(r[3] for r in res)


def test_shard_range_properties():
    sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd", "flightbatch"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("sharding", os.path.join(ROOT, "flight.jl_amd", "flightbatch", "sharding.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    for n in (1, 7, 8, 1 << 20, (1 << 22) + 3):
        for w in (1, 2, 4, 8):
            r = [sh.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_range(10, 3, 2)


def test_pack_fleet_same_mix_on_every_rank():
    """Mixed-fleet packer (config 5): types interleaved in the input order; every rank gets a contiguous slice of each
    type's sorted list, the union over ranks is a partition, order within a type is preserved, unpack inverts pack."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sharding", os.path.join(ROOT, "flight.jl_amd", "flightbatch", "sharding.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    rng = np.random.default_rng(0)
    for n, world in ((1000, 1), (1001, 2), (4096, 8), (37, 8)):
        types = rng.integers(0, 2, n) * 2          # ids 0 (C172) and 2 (Robot2D), random interleaving
        seen = np.zeros(n, int)
        for r in range(world):
            parts = sh.pack_fleet(types, r, world)
            for t, idx in parts.items():
                assert (types[idx] == t).all() and (np.diff(idx) > 0).all()
                seen[idx] += 1
                tot = int((types == t).sum())
                assert abs(idx.size - tot / world) < 1.0 + 1e-9      # same mix everywhere
            vals = {t: np.vstack([idx.astype(float), -idx.astype(float)]) for t, idx in parts.items()}
            back = sh.unpack_fleet(parts, vals, n, fill=np.nan)
            for t, idx in parts.items():
                assert np.array_equal(back[0, idx], idx) and np.array_equal(back[1, idx], -idx)
        assert (seen == 1).all()
