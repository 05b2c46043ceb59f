"""Host mirror of `Robot2D.Robot()` (lib/FlightApps/src/robot2d/robot2d.jl) for the batched path.

    reference                                                 here
    Model(Robot2D.Robot())                 :526-529            Robot2DWorld(n, dtype="f64"|"f32", vehicle={...})
    f_init!(robot, InitParameters(u_m,ω,η)) :208-228,563-570    f_init(world, InitParameters(...))
    robot.controller.u.mode / m_ref / v_ref / η_ref  :359-364   world.u rows [mode, m_ref, v_ref, η_ref]
    f_ode! / f_step! / f_periodic!          :537-561            f_ode / f_step / f_periodic (flightbatch.modeling)
    Simulation(mdl; dt = 0.01, Δt = 0.02)   README.md:76        Simulation(world, dt=0.01, Δt=0.02)

The verbs and `Simulation/init/step/run` of flightbatch.modeling work on this world unchanged."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
import numpy as np

from . import hdf5_min
from ._lib import K, check, lib
from .modeling import BatchedWorld, _pd

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "robot2d.h5")
MODE_M, MODE_V, MODE_ETA = 0, 1, 2   # ControlMode, robot2d.jl:347


@dataclass
class InitParameters:
    """Robot2D.InitParameters (robot2d.jl:208-212). Scalars broadcast over the batch."""
    u_m: object = 0.0
    ω: object = 0.0
    η: object = 0.0

    def pack(self, n: int) -> np.ndarray:
        ip = np.zeros((3, n))
        ip[0] = np.asarray(self.u_m, dtype=np.float64)
        ip[1] = np.asarray(self.ω, dtype=np.float64)
        ip[2] = np.asarray(self.η, dtype=np.float64)
        return ip


def robot2d_table(vehicle: dict | None = None, gains_path: str | None = None) -> np.ndarray:
    """FB_TABLE_ROBOT2D blob: Vehicle parameters (robot2d.jl:20-30; J_b, J_r < 0 = derive like the kwdef defaults), the
    LQRDataPoint stored in robot2d.h5 (robot2d.jl:419-422) and the PID gains of Controller.f_init! (:430-436)."""
    v = dict(L=0.15, R=0.05, m_b=1.0, m_r=0.1, J_b=-1.0, J_r=-1.0, k_m=0.32, b_m=0.0189, J_m=0.0014)
    v.update(vehicle or {})
    d = hdf5_min.read_all(gains_path or _DATA)
    gains = np.concatenate([d["K_fbk"].ravel(), d["K_fwd"].ravel(), d["K_int"].ravel(), d["x_trim"].ravel(), d["u_trim"].ravel(),
                            d["z_trim"].ravel()])
    blob = np.concatenate([[v[k] for k in ("L", "R", "m_b", "m_r", "J_b", "J_r", "k_m", "b_m", "J_m")], gains,
                           [0.6, 0.0, 0.0, 0.01]]).astype(np.float64)
    assert blob.size == K["FB_R2_TABLE_SIZE"]
    return blob


class Robot2DWorld(BatchedWorld):
    """N independent `Model(Robot2D.Robot())` on one GPU. State record x [10, n] and inputs u [4, n]: include/flightbatch.h."""
    MODEL = "FB_MODEL_ROBOT2D"
    _CKPT_ARRAYS = ("x", "u")

    def __init__(self, n: int, device: int = 0, dtype: str = "f64", vehicle: dict | None = None):
        self.n = int(n)
        self._h = C.c_void_p()
        check(lib.fb_create(K["FB_MODEL_ROBOT2D"], K["FB_KIN_WA"], K["FB_F64"] if dtype == "f64" else K["FB_F32"], self.n, int(device),
                            C.byref(self._h)))
        blob = robot2d_table(vehicle)
        dims = (C.c_int64 * 1)(blob.size)
        check(lib.fb_set_table(self._h, K["FB_TABLE_ROBOT2D"], blob.ctypes.data_as(C.c_void_p), dims, 1))
        self.dtype = dtype
        self.t = 0.0
        self._Δt_root = 1.0
        self._n = 0
        self.nx, self.ns, self.nu, self.ny = K["FB_R2_NX"], 0, K["FB_R2_NU"], K["FB_R2_NY"]
        u = np.zeros((4, self.n))
        u[0] = MODE_V   # ControllerU default mode (robot2d.jl:360)
        self.u = u

    @property
    def x(self):
        x = np.empty((self.nx, self.n))
        check(lib.fb_get_state(self._h, _pd(x), None))
        return x

    @x.setter
    def x(self, v):   # `mdl.x .= v`: a plain assignment (no init! semantics: clock, periodic phase and status words stay; see set_state)
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.nx, self.n)
        check(lib.fb_assign_state(self._h, _pd(v), None))

    def set_state(self, x, s=None):   # an INITIAL condition: clears the status words, restarts the clock (fb_set_state)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.nx, self.n)
        check(lib.fb_set_state(self._h, _pd(x), None))

    @property
    def s(self):
        return np.zeros((0, self.n), dtype=np.int32)

    @property
    def u(self):
        u = np.empty((self.nu, self.n))
        check(lib.fb_get_inputs(self._h, _pd(u), None))
        return u

    @u.setter
    def u(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.nu, self.n)
        check(lib.fb_set_inputs(self._h, _pd(v), None))

    @property
    def y(self):
        y = np.empty((self.ny, self.n))
        check(lib.fb_get_outputs(self._h, _pd(y)))
        return y
