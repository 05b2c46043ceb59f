"""flightbatch — Python host mirror of the Flight.jl operator surface over libflightbatch (HIP, gfx950).

Julia is the reference's host language and is not available in this environment; this package plays the
role the `FlightBatch.jl` ccall shim (flight.jl_amd/julia/FlightBatch.jl) plays for a
Julia host. Importing it requires the built shared library; there is no CPU fallback.
"""
from ._lib import K, EXPORTED, LIB_PATH, FlightBatchError, lib  # noqa: F401
from .modeling import (BatchedWorld, Simulation, SimulationTermination, TimeSeries, TrimParameters, TrimState,  # noqa: F401
                       f_init, f_ode, f_periodic, f_step, init, run, step, checkpoint, restore)
from . import tables  # noqa: F401
from . import sharding  # noqa: F401
from .robot2d import Robot2DWorld, InitParameters  # noqa: F401
from .c172x import Cessna172Xv2World, ModeControlLon, ModeControlLat, ModeGuidance  # noqa: F401
from . import ctl_gains  # noqa: F401
from . import scenario  # noqa: F401
from .fleet import MixedFleet  # noqa: F401
