"""Mixed fleets (BASELINE.json configs[4]): vehicles of different models interleaved in the caller's order, packed into
one homogeneous batch per model so that no wavefront ever mixes two right-hand sides (a Cessna 172 step costs ~500x a
Robot2D step: a mixed wave would run at the slow model's pace with half its lanes idle). Each batch is its own
libflightbatch handle with its own HIP stream, so the light model's kernels overlap the heavy one's on the GPU.
The reference has no counterpart (one model per Simulation, lib/FlightCore/src/sim.jl:173-255)."""
from __future__ import annotations

import numpy as np

from .modeling import Simulation, init, step
from .sharding import pack_fleet, unpack_fleet


class MixedFleet:
    """fleet = MixedFleet(types, {type_id: lambda n: World(n, ...)}, rank, world_size)

    `types[i]` is the model id of vehicle i in the caller's order; `factories[type_id](n)` builds the batched world for the
    n vehicles of that type this rank owns. `fleet.index[type_id]` are their caller-order indices."""

    def __init__(self, types, factories: dict, rank: int = 0, world: int = 1):
        self.types = np.asarray(types)
        self.n = int(self.types.size)
        self.index = pack_fleet(self.types, rank, world)
        self.worlds = {t: factories[t](idx.size) for t, idx in self.index.items() if idx.size}
        self.sims: dict = {}

    def simulate(self, **sim_kwargs):
        """Simulation(world; ...) for every batch, same keyword arguments (dt, Δt, ...)."""
        self.sims = {t: Simulation(w, **sim_kwargs) for t, w in self.worlds.items()}
        return self.sims

    def init(self, initializers: dict):
        """init!(sim, initializer) per type; initializers[type_id] is built by the caller for that type's vehicles
        (fleet.index[type_id] tells which ones they are)."""
        for t, sim in self.sims.items():
            if t in initializers:
                init(sim, initializers[t])
            else:
                init(sim)

    def step(self, Δt_total: float):
        """step!(sim, Δt_total) on every batch: the launches of the different models are queued on different streams."""
        for sim in self.sims.values():
            step(sim, Δt_total)

    def sync(self):
        for w in self.worlds.values():
            w.sync()

    def gather(self, what: str = "x", fill=np.nan) -> np.ndarray:
        """Per-vehicle values in the CALLER's order: rows padded to the widest model ('x', 'status', ...)."""
        vals = {t: np.asarray(getattr(w, what), dtype=np.float64) for t, w in self.worlds.items()}
        return unpack_fleet({t: self.index[t] for t in self.worlds}, vals, self.n, fill)

    def close(self):
        for w in self.worlds.values():
            w.close()
