"""Multi-GPU layout of the batch: aircraft are independent, so the batch is cut into
contiguous index ranges, one per rank (one process per GPU), with NO communication while stepping.
The only exchange is trajectory collection: one all-gather (RCCL over xGMI on GPUs; gloo in CPU tests)
of the per-rank state panels.  The reference has no counterpart (one aircraft per Simulation,
lib/FlightCore/src/sim.jl:173-255)."""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of aircraft owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_state(x_local, n_total: int | None = None):
    """Gather [nfield, n_local] panels from every rank into [nfield, n_total] (aircraft order = rank order).
    x_local: torch tensor on the rank's device (CUDA -> RCCL, CPU -> gloo). Ragged shards are padded to the
    largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    nfield, n_local = x_local.shape
    sizes = torch.tensor([n_local], dtype=torch.int64, device=x_local.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [int(s.item()) for s in all_sizes]
    n_max = max(all_sizes)
    pad = x_local
    if n_local < n_max:
        pad = torch.zeros((nfield, n_max), dtype=x_local.dtype, device=x_local.device)
        pad[:, :n_local] = x_local
    out = torch.empty((world * nfield, n_max), dtype=x_local.dtype, device=x_local.device)  # concatenation along dim 0
    dist.all_gather_into_tensor(out, pad.contiguous())
    out = out.view(world, nfield, n_max)
    parts = [out[r, :, :all_sizes[r]] for r in range(world)]
    res = torch.cat(parts, dim=1)
    if n_total is not None and res.shape[1] != n_total:
        raise RuntimeError(f"gathered {res.shape[1]} aircraft, expected {n_total}")
    return res


# ---- mixed fleets (BASELINE.json configs[4]; SURVEY.md §8e: "assigned after the model-type sort so each GPU gets the same
# type mix") ------------------------------------------------------------------------------------------------
def pack_fleet(types, rank: int = 0, world: int = 1) -> dict:
    """Packs a fleet whose vehicle types are interleaved in the caller's order into homogeneous batches.

    `types`: integer model id per vehicle (any integers), in the caller's order. Returns, per type present,
    {type: indices} — the caller-order indices this rank owns, ascending — such that every rank gets a contiguous slice
    of EACH type's sorted index list (same mix on every rank, sizes differ by at most one per type). A stable
    counting sort: vehicles of one type keep their relative order, so results scatter back with one indexed store."""
    import numpy as np
    types = np.asarray(types)
    out = {}
    for t in np.unique(types):
        idx = np.flatnonzero(types == t)
        lo, hi = shard_range(idx.size, rank, world)
        out[int(t)] = idx[lo:hi]
    return out


def unpack_fleet(parts: dict, values: dict, n_total: int, fill=0):
    """Inverse of pack_fleet on one rank: scatter per-type arrays `values[type]` ([..., n_type]) back to caller order
    ([..., n_total], rows padded to the widest type with `fill`)."""
    import numpy as np
    rows = max(np.atleast_2d(v).shape[0] for v in values.values())
    out = np.full((rows, n_total), fill, dtype=np.float64)
    for t, idx in parts.items():
        v = np.atleast_2d(values[t])
        out[:v.shape[0], idx] = v
    return out
