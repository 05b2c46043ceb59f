"""Python mirror of the Flight.jl operator surface for the batched path.

Same verbs, argument meaning and ordering as the reference (Julia's ``!`` cannot appear in a Python
identifier, so ``f_ode!`` is ``f_ode`` etc.):

    reference (file:line)                                        here
    ------------------------------------------------------------ ------------------------------------
    Model(SimpleWorld(Cessna172Sv0()))  FC/modeling.jl:103-153    BatchedWorld(n, device=0)
    f_init!(world, C172.TrimParameters())  FP/world.jl:49-57      f_init(world, TrimParameters(...))
    f_ode!(world)       FP/world.jl:26-32                         f_ode(world)
    f_step!(world)      FP/world.jl:34-39                         f_step(world)
    f_periodic!(Unconditional(), world)  FP/world.jl:41-47        f_periodic(world)
    Simulation(mdl; dt, Δt, t_start, t_end)  FC/sim.jl:183-255     Simulation(mdl, dt=, Δt=, ...)
    init!(sim, args...)  FC/sim.jl:390-414                         init(sim, *args)
    step!(sim[, Δt_total, stop_at_tdt])  FC/sim.jl:386              step(sim[, Δt_total, stop_at_tdt])
    run!(sim)  FC/sim.jl:611-638                                    run(sim)
    TimeSeries(sim)  FC/sim.jl:644-704                              TimeSeries(sim)

All verbs return None and mutate the model in place, like the reference (FC/modeling.jl:192-194).
All arrays are [field, aircraft] (aircraft index fastest in memory = the C ABI's SoA layout).
(FC = lib/FlightCore/src, FP = lib/FlightPhysics/src, FA = lib/FlightApps/src.)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
import numpy as np

from . import tables as _tables
from ._lib import K, FlightBatchError, check, fb_params, lib

_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)


def _pd(a):
    return a.ctypes.data_as(_D)


def _pi(a):
    return a.ctypes.data_as(_I)


# ------------------------------------------------------------------------------------------------------
@dataclass
class TrimParameters:
    """C172.TrimParameters — FA/c172/c172.jl:806-818 (same defaults). Scalars broadcast over the batch;
    arrays of length n give per-aircraft values."""
    n_e: object = (1.0, 0.0, 0.0)        # Ob location, n-vector
    h_e: object = 1050.0                 # Ob ellipsoidal altitude
    ψ_nb: object = 0.0
    EAS: object = 50.0
    γ_wb_n: object = 0.0
    ψ_wb_dot: object = 0.0
    θ_wb_dot: object = 0.0
    β_a: object = 0.0
    fuel_load: object = 0.5
    mixture: object = 0.5
    flaps: object = 0.0
    payload: object = (75.0, 75.0, 0.0, 0.0, 50.0)  # PayloadY defaults, c172.jl:529-535

    def pack(self, n: int) -> np.ndarray:
        tp = np.zeros((K["FB_NTP"], n))
        ne = np.asarray(self.n_e, dtype=np.float64)
        tp[K["FB_TP_N_E"]:K["FB_TP_N_E"] + 3] = ne.reshape(3, -1) if ne.ndim == 2 else ne[:, None]
        for key, val in (("FB_TP_H_E", self.h_e), ("FB_TP_PSI_NB", self.ψ_nb), ("FB_TP_EAS", self.EAS),
                         ("FB_TP_GAMMA_WB_N", self.γ_wb_n), ("FB_TP_PSI_WB_DOT", self.ψ_wb_dot),
                         ("FB_TP_THETA_WB_DOT", self.θ_wb_dot), ("FB_TP_BETA_A", self.β_a),
                         ("FB_TP_FUEL_LOAD", self.fuel_load), ("FB_TP_MIXTURE", self.mixture), ("FB_TP_FLAPS", self.flaps)):
            tp[K[key]] = np.asarray(val, dtype=np.float64)
        pl = np.asarray(self.payload, dtype=np.float64)
        tp[K["FB_TP_PAYLOAD"]:K["FB_TP_PAYLOAD"] + 5] = pl.reshape(5, -1) if pl.ndim == 2 else pl[:, None]
        return tp


def TrimState(n: int = 1) -> np.ndarray:
    """C172.TrimState() initial guess — FA/c172/c172.jl:796-804. Returns [7, n]."""
    return np.repeat(np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])[:, None], n, axis=1).copy()


class SimulationTermination(Exception):
    """FC/sim.jl:32. In the batch nothing is thrown from the kernels: terminated aircraft carry a sticky
    status word (BatchedWorld.status) and freeze; the rest of the batch continues."""


# ------------------------------------------------------------------------------------------------------
class BatchedWorld:
    """N independent ``Model(SimpleWorld(Cessna172Sv0()))`` instances resident on one MI355X."""

    MODEL = "FB_MODEL_C172S0"

    def __init__(self, n: int, device: int = 0, tables: dict | None = None, kinematics: str = "WA", dtype: str = "f64"):
        """kinematics: "WA" (default, FP/kinematics.jl:148), "ECEF" (:250) or "NED" (:329) — Cessna172Sv0(kinematics).
        dtype "f32" (Cessna172Sv0 + WA only) steps airborne aircraft with the fp32 kernel (positions still integrated in
        fp64, ground contact handed to the fp64 kernel); state, trim, f_ode! and the ABI stay fp64."""
        self.n = int(n)
        self._h = C.c_void_p()
        self.kinematics = kinematics
        self.dtype = dtype
        check(lib.fb_create(K[self.MODEL], K["FB_KIN_" + kinematics], K["FB_F64" if dtype == "f64" else "FB_F32"], self.n, int(device),
                            C.byref(self._h)))
        nx = C.c_int32()
        check(lib.fb_dims(self._h, C.byref(nx), None, None, None))
        self.nx = nx.value
        tb = tables or _tables.default_tables()
        self._set_table("FB_TABLE_EGM96", np.asfortranarray(tb["egm96"], dtype=np.float32), (721, 1441))
        self._set_table("FB_TABLE_PROPELLER", np.asfortranarray(tb["propeller"], dtype=np.float64), (21, 21, 6))
        self._set_table("FB_TABLE_PISTON", np.ascontiguousarray(tb["piston"], dtype=np.float64), (tb["piston"].size,))
        self._set_table("FB_TABLE_AERO", np.ascontiguousarray(tb["aero"], dtype=np.float64), (tb["aero"].size,))
        self.t = 0.0
        self._Δt_root = 1.0   # FC/modeling.jl:98
        self._n = 0           # periodic update counter, FC/modeling.jl:99
        self.trim_state = None
        self.trim_success = None
        self.trim_cost = None

    # -- plumbing --
    def _set_table(self, kind, arr, dims):
        d = (C.c_int64 * len(dims))(*dims)
        check(lib.fb_set_table(self._h, K[kind], arr.ctypes.data_as(C.c_void_p), d, len(dims)))

    def close(self):
        if self._h:
            lib.fb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- world-level parameters (atmosphere.u / terrain, FP/atmosphere.jl:75-78,165; FP/terrain.jl:34-38) --
    @property
    def params(self) -> fb_params:
        p = fb_params()
        check(lib.fb_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.params
        for k, v in kw.items():
            if k == "wind_ned":
                for j in range(3):
                    p.wind_ned[j] = float(v[j])
            else:
                setattr(p, k, v)
        check(lib.fb_set_params(self._h, C.byref(p)))

    # -- per-aircraft environment: each simulation's own world.atmosphere.sl.u / .wind.u and terrain elevation (FP/atmosphere.jl:75-84,
    #    156-165; FP/terrain.jl:34-36): rows FB_ENV_WIND_N, _E, _D, FB_ENV_T_SL, FB_ENV_P_SL, FB_ENV_H_TERRAIN of an [FB_NENV, n] array. None (the
    #    default) = the batch-wide block of `params`. --
    @property
    def env(self):
        if not self.has_env:
            return None
        e = np.empty((K["FB_NENV"], self.n))
        check(lib.fb_get_env(self._h, _pd(e)))
        return e

    @env.setter
    def env(self, v):
        if v is None:
            check(lib.fb_set_env(self._h, None))
            return
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(K["FB_NENV"], self.n)
        check(lib.fb_set_env(self._h, _pd(v)))

    @property
    def has_env(self) -> bool:
        """whether the handle carries per-aircraft environment rows — asked of the handle (fb_has_env), not remembered here"""
        rc = lib.fb_has_env(self._h)
        if rc < 0:
            check(rc)
        return rc == 1

    def set_env(self, wind_ned=None, T_sl=None, p_sl=None, h_terrain=None):
        """Per-aircraft rows from keyword arrays (scalars broadcast; what is not given comes from the batch-wide `params`,
        or from the rows already set): wind_ned [3, n] or (N, E, D), T_sl [n], p_sl [n], h_terrain [n]."""
        e = self.env
        if e is None:
            p = self.params
            e = np.empty((K["FB_NENV"], self.n))
            e[K["FB_ENV_WIND_N"]], e[K["FB_ENV_WIND_E"]], e[K["FB_ENV_WIND_D"]] = p.wind_ned[0], p.wind_ned[1], p.wind_ned[2]
            e[K["FB_ENV_T_SL"]], e[K["FB_ENV_P_SL"]], e[K["FB_ENV_H_TERRAIN"]] = p.T_sl, p.p_sl, p.h_terrain
        if wind_ned is not None:
            wv = np.asarray(wind_ned, dtype=np.float64)
            e[K["FB_ENV_WIND_N"]:K["FB_ENV_WIND_D"] + 1] = wv.reshape(3, -1) if wv.ndim == 2 else wv[:, None]
        for key, val in (("FB_ENV_T_SL", T_sl), ("FB_ENV_P_SL", p_sl), ("FB_ENV_H_TERRAIN", h_terrain)):
            if val is not None:
                e[K[key]] = np.asarray(val, dtype=np.float64)
        self.env = e

    # -- mdl.x / mdl.s / mdl.u (FC/modeling.jl:89-101) --
    @property
    def x(self) -> np.ndarray:
        x = np.empty((self.nx, self.n))
        check(lib.fb_get_state(self._h, _pd(x), None))
        return x

    @x.setter
    def x(self, v):   # `mdl.x .= v`: a plain assignment (no init! semantics: clock, periodic phase and status words stay; see set_state)
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.nx, self.n)
        check(lib.fb_assign_state(self._h, _pd(v), None))

    @property
    def s(self) -> np.ndarray:
        s = np.empty((K["FB_NS"], self.n), dtype=np.int32)
        check(lib.fb_get_state(self._h, None, _pi(s)))
        return s

    @s.setter
    def s(self, v):
        v = np.ascontiguousarray(v, dtype=np.int32).reshape(K["FB_NS"], self.n)
        check(lib.fb_assign_state(self._h, None, _pi(v)))

    def set_state(self, x, s):   # an INITIAL condition: clears the status words, restarts the clock and the periodic phase (fb_set_state)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.nx, self.n)
        s = np.ascontiguousarray(s, dtype=np.int32).reshape(K["FB_NS"], self.n)
        check(lib.fb_set_state(self._h, _pd(x), _pi(s)))

    @property
    def u(self) -> np.ndarray:
        u = np.empty((K["FB_NU"], self.n))
        check(lib.fb_get_inputs(self._h, _pd(u), None))
        return u

    @u.setter
    def u(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(K["FB_NU"], self.n)
        check(lib.fb_set_inputs(self._h, _pd(v), None))

    @property
    def ui(self) -> np.ndarray:
        ui = np.empty(self.n, dtype=np.int32)
        check(lib.fb_get_inputs(self._h, None, _pi(ui)))
        return ui

    @ui.setter
    def ui(self, v):
        v = np.ascontiguousarray(v, dtype=np.int32).reshape(self.n)
        check(lib.fb_set_inputs(self._h, None, _pi(v)))

    @property
    def y(self) -> np.ndarray:
        """mdl.y of the last f_ode!: [FB_NY, n] (layout: include/flightbatch.h FB_Y_*)."""
        y = np.empty((K["FB_NY"], self.n))
        check(lib.fb_get_outputs(self._h, _pd(y)))
        return y

    def y_fields(self, *blocks: str) -> np.ndarray:
        """mdl.y.<block>... of the last f_ode!: only the named blocks ("KIN", "AIR", "AERO", "LDG", "PWP", "FUEL", "DYN") are
        copied from the device, stacked in FB_Y_* order: [sum of widths, n]."""
        mask = 0
        for b in blocks:
            mask |= K["FB_YF_" + b.upper()]
        first = [K[k] for k in ("FB_Y_KIN", "FB_Y_AIR", "FB_Y_AERO", "FB_Y_LDG", "FB_Y_PWP", "FB_Y_FUEL", "FB_Y_DYN")] + [K["FB_NY"]]
        rows = sum(first[i + 1] - first[i] for i in range(7) if mask & (1 << i))
        y = np.empty((rows, self.n))
        check(lib.fb_get_output_fields(self._h, mask, _pd(y)))
        return y

    @property
    def status(self) -> np.ndarray:
        st = np.empty(self.n, dtype=np.int32)
        check(lib.fb_status(self._h, _pi(st)))
        return st

    @property
    def termination(self):
        """(step, where) of every aircraft: the number of RK updates completed when its simulation ended (-1: still running) and the
        FB_TERM_* code of the place — ≙ sim.t and the frame of the exception the reference stops with (FC/sim.jl:561-570)."""
        step = np.empty(self.n, dtype=np.int64)
        where = np.empty(self.n, dtype=np.int32)
        check(lib.fb_get_termination(self._h, step.ctypes.data_as(C.POINTER(C.c_int64)), _pi(where)))
        return step, where

    def sync(self):
        check(lib.fb_sync(self._h))

    # -- checkpoint / restore: everything a resumed run needs, as a dict of numpy arrays (np.savez-able) --
    _CKPT_ARRAYS = ("x", "s", "u", "ui")

    def checkpoint(self) -> dict:
        self.sync()
        cnt = C.c_int64()
        check(lib.fb_get_step_count(self._h, C.byref(cnt)))
        ck = {k: getattr(self, k) for k in self._CKPT_ARRAYS}
        if self.MODEL != "FB_MODEL_ROBOT2D":   # (the marker says whether the world had rows: restore() clears a world's rows only when told to)
            ck["has_env"] = np.bool_(self.has_env)
            if ck["has_env"]:
                ck["env"] = self.env
        tstep, twhere = self.termination
        ck.update(status=self.status, term_step=tstep, term_where=twhere, step_count=np.int64(cnt.value), t=np.float64(lib.fb_time(self._h)))
        return ck

    def restore(self, ck: dict) -> None:
        # (before the state: fb_set_env invalidates the derivative carried across launches, like any input change. A checkpoint that says
        # nothing about rows — written before they existed — leaves the world's rows as they are; one that says "no rows" clears them.)
        if self.MODEL != "FB_MODEL_ROBOT2D" and ("has_env" in ck or "env" in ck):
            has = bool(ck["has_env"]) if "has_env" in ck else True
            self.env = ck["env"] if has else None
        if "s" in self._CKPT_ARRAYS:
            self.set_state(ck["x"], ck["s"])
        else:
            self.set_state(ck["x"], None)
        for k in self._CKPT_ARRAYS:
            if k not in ("x", "s"):
                setattr(self, k, ck[k])
        st = np.ascontiguousarray(ck["status"], dtype=np.int32)
        check(lib.fb_set_step_count(self._h, int(ck["step_count"]), float(ck["t"])))
        check(lib.fb_set_status(self._h, _pi(st)))
        if "term_step" in ck:   # (checkpoints written before the record was part of them: fb_set_status has marked the words FB_TERM_OUTSIDE_STEP)
            ts = np.ascontiguousarray(ck["term_step"], dtype=np.int64)
            tw = np.ascontiguousarray(ck["term_where"], dtype=np.int32)
            check(lib.fb_set_termination(self._h, ts.ctypes.data_as(C.POINTER(C.c_int64)), _pi(tw)))
        self.t = float(ck["t"])


# ---- the verbs (all return None) ---------------------------------------------------------------------
def f_init(world: BatchedWorld, init, trim_state: np.ndarray | None = None) -> None:
    """f_init!(world, trim_params): FP/world.jl:49-57 -> FA/c172/c172.jl:883-942."""
    if init is None:   # Cessna172Xv2: the avionics half of f_init!(aircraft, C172.Init(...)) on the state already set by the host
        check(lib.fb_f_init(world._h, None, 0))
        world.t = 0.0
        return None
    if hasattr(init, "pack") and not isinstance(init, TrimParameters):   # plain per-instance initializer (Robot2D.InitParameters)
        ip = init.pack(world.n)
        check(lib.fb_f_init(world._h, _pd(ip), ip.shape[0]))
        world.t = 0.0
        return None
    if not isinstance(init, TrimParameters):
        raise TypeError(f"no f_init method for {type(init).__name__}")  # MethodError, FC/modeling.jl:205-207
    tp = init.pack(world.n)
    ts = TrimState(world.n) if trim_state is None else np.ascontiguousarray(trim_state, dtype=np.float64).reshape(7, world.n)
    ok = np.zeros(world.n, dtype=np.int32)
    cost = np.zeros(world.n)
    check(lib.fb_trim(world._h, _pd(tp), _pd(ts), _pi(ok), _pd(cost)))
    world.trim_state, world.trim_success, world.trim_cost = ts, ok.astype(bool), cost
    world.t = 0.0
    return None


def f_ode(world: BatchedWorld, xdot: np.ndarray | None = None) -> None:
    """f_ode!(world): FP/world.jl:26-32. Fills ẋ (into `xdot` if given, [FB_NX, n]) and world.y."""
    if xdot is not None:
        assert xdot.dtype == np.float64 and xdot.shape[1] == world.n and xdot.flags.c_contiguous
    check(lib.fb_f_ode(world._h, _pd(xdot) if xdot is not None else None))
    return None


def f_step(world: BatchedWorld) -> None:
    """f_step!(world): FP/world.jl:34-39."""
    check(lib.fb_f_step(world._h))
    return None


def f_periodic(world: BatchedWorld) -> None:
    """f_periodic!(Unconditional(), world): FP/world.jl:41-47."""
    check(lib.fb_f_periodic(world._h))
    return None


# ---- Simulation (FC/sim.jl) --------------------------------------------------------------------------
class Simulation:
    """Simulation(mdl; algorithm = RK4(), adaptive = false, dt = 0.02, Δt = dt, t_start = 0, t_end = 10000,
    save_on = true, saveat = []) — FC/sim.jl:183-196. Only the fixed-step RK4 path exists here.

    Saving (cb_save, FC/sim.jl:210-217) is an ON-DEVICE log: every `saveat` seconds (default: every step) the rows
    `save_rows` of the output record y (names or indices of include/flightbatch.h FB_Y_*; default: none) and, with
    `save_x`, the whole state x are appended to a device buffer sized for `t_end` (capped by `log_capacity` samples);
    nothing crosses PCIe until TimeSeries(sim) is built."""

    def __init__(self, mdl: BatchedWorld, dt: float = 0.02, Δt: float | None = None, t_start: float = 0.0,
                 t_end: float = 10000.0, save_on: bool = True, saveat: float | None = None, steps_per_launch: int | None = None,
                 save_outputs: bool = False, save_rows=None, save_x: bool = True, log_capacity: int | None = None,
                 user_callback=None):
        self.mdl = mdl
        self.user_callback = user_callback   # user_callback!(mdl), FC/sim.jl:190,331-341: runs after every step's f_step!/f_periodic!
        self.dt = float(dt)
        self.Δt = float(Δt if Δt is not None else dt)
        ratio = self.Δt / self.dt
        if abs(ratio - round(ratio)) > 1e-9 or round(ratio) < 1:
            raise ValueError("Δt must be an integer multiple of dt on the batched path")
        self.t_start, self.t_end = float(t_start), float(t_end)
        self.save_on = save_on
        self.save_every = max(1, int(round((saveat if saveat else self.dt) / self.dt)))
        mdl._Δt_root = self.Δt  # FC/sim.jl:198
        mdl.set_params(dt=self.dt, periodic_n=int(round(ratio)))
        self._nstep = 0
        nx, ny = _dims(mdl)
        rows: list = []
        if save_outputs:
            rows += list(range(ny))
        for r in (save_rows or []):
            rows.append(K[r] if isinstance(r, str) else int(r))
        self.y_rows = rows
        self.save_x = bool(save_x)
        self._rows = np.array(rows + ([K["FB_LOG_X0"] + k for k in range(nx)] if save_x else []), dtype=np.int32)
        k = steps_per_launch or (self.save_every if save_on else 50)
        check(lib.fb_set_steps_per_launch(mdl._h, int(k)))
        self._manual_save = False
        if save_on and self._rows.size:
            want = int(np.ceil((self.t_end - self.t_start) / (self.save_every * self.dt))) + 2
            budget = int(log_capacity) if log_capacity else max(2, int(8e9 // (8 * self._rows.size * mdl.n)))   # <= 8 GB by default
            if log_capacity is None and budget < want:
                # the reference's defaults (save_on, t_end = 10000, every step) need more device memory than the 8 GB default budget for a
                # batch this large. The log is capped at the budget and the run can go on until it is full (fb_step then fails with "log
                # capacity exhausted", and says so): say it now, with the ways out, instead of refusing to construct the Simulation
                import warnings
                warnings.warn(f"Simulation: saving {self._rows.size} rows of {mdl.n} vehicles every {self.save_every} steps until t_end = {self.t_end} "
                              f"needs {want} samples; the default 8 GB device log holds {budget}, i.e. the run can be stepped to "
                              f"t = {self.t_start + (budget - 1) * self.save_every * self.dt:g}. Pass t_end / saveat / save_x=False / save_rows, "
                              f"log_capacity=<samples>, or save_on=False", stacklevel=2)
            self.capacity = min(want, budget)
            # with a user callback the sample must be taken AFTER it (CallbackSet order cb_step, cb_periodic, cb_user, cb_save: FC/sim.jl:204-218):
            # the device log is then driven from step() instead of from inside fb_step
            self._manual_save = user_callback is not None
            every = (1 << 62) if self._manual_save else self.save_every
            check(lib.fb_log_configure(mdl._h, every, self.capacity, _pi(self._rows), int(self._rows.size)))
        else:
            self.save_on = False
            check(lib.fb_log_configure(mdl._h, 0, 0, None, 0))

    # property forwarding, FC/sim.jl:261-275
    @property
    def t(self):
        return self.t_start + self._nstep * self.dt

    @property
    def x(self):
        return self.mdl.x

    @property
    def y(self):
        return self.mdl.y

    @property
    def u(self):
        return self.mdl.u

    @property
    def s(self):
        return self.mdl.s


def _dims(mdl):
    nx, ny = C.c_int32(), C.c_int32()
    check(lib.fb_dims(mdl._h, C.byref(nx), None, None, C.byref(ny)))
    return nx.value, ny.value


def init(sim: Simulation, *init_args, **init_kwargs) -> None:
    """init!(sim, init_args...): FC/sim.jl:390-414 — clears the log, f_init!s the model, saves y(t0)."""
    check(lib.fb_log_clear(sim.mdl._h))
    if init_args:
        f_init(sim.mdl, *init_args, **init_kwargs)
    sim._nstep = 0
    sim.mdl._n = 0  # cb_periodic_init!, FC/sim.jl:358-362
    if sim.save_on:
        check(lib.fb_log_record(sim.mdl._h))
    return None


def step(sim: Simulation, Δt_total: float | None = None, stop_at_tdt: bool = True) -> None:
    """step!(sim) / step!(sim, Δt_total, true): FC/sim.jl:386 (OrdinaryDiffEq step!). Asynchronous: the launches (and the
    device-side saves between them) are queued on the world's stream."""
    n = 1 if Δt_total is None else int(round(Δt_total / sim.dt))
    def _count():
        cnt = C.c_int64()
        check(lib.fb_get_step_count(sim.mdl._h, C.byref(cnt)))   # host-side bookkeeping, no device synchronisation
        return int(cnt.value)

    before = _count()

    def _resync():   # a failing fb_step (log capacity exhausted, ...) may have advanced part of the steps: follow the device's clock
        nonlocal before
        now = _count()
        sim._nstep += now - before
        before = now
        sim.mdl.t = sim.t

    if sim.user_callback is None:
        try:
            check(lib.fb_step(sim.mdl._h, n))
        except Exception:
            _resync()
            raise
        sim._nstep += n
        sim.mdl.t = sim.t
        return None
    # cb_user_affect! (FC/sim.jl:331-341) runs on the host after EVERY step: launches are one step long while a callback is set;
    # cb_save comes after it (FC/sim.jl:217), so the saved sample sees the callback's changes. (Remaining difference: the reference
    # saves the y of the step's last f_ode!, taken before the callbacks; fb_log_record re-evaluates f_ode! at the saved instant.)
    for _ in range(n):
        try:
            check(lib.fb_step(sim.mdl._h, 1))
        except Exception:
            _resync()
            raise
        sim._nstep += 1
        before += 1
        sim.mdl.t = sim.t
        sim.user_callback(sim.mdl)
        if sim._manual_save and sim._nstep % sim.save_every == 0:
            check(lib.fb_log_record(sim.mdl._h))
    return None


def checkpoint(sim: Simulation) -> dict:
    """Everything needed to resume `sim` later (or elsewhere): the world's arrays, the step counter and sim.t."""
    return sim.mdl.checkpoint()


def restore(sim: Simulation, ck: dict) -> None:
    sim.mdl.restore(ck)
    sim._nstep = int(ck["step_count"])
    return None


def run(sim: Simulation) -> None:
    """run!(sim) headless (pace = Inf): FC/sim.jl:611-638 — steps to t_end."""
    step(sim, sim.t_end - sim.t, True)
    sim.mdl.sync()
    return None


class TimeSeries:
    """TimeSeries(sim): FC/sim.jl:644-704 — the logged samples, read back from the device log:
    t [m] (seconds from the last init), x [m, Nx, n] (when save_x), y [m, len(y_rows), n] (rows sim.y_rows of the output
    record; all of it with save_outputs=True). ts[i] indexes samples like the reference's getindex."""

    def __init__(self, sim: Simulation, first: int = 0, count: int | None = None):
        cnt = C.c_int64()
        check(lib.fb_log_count(sim.mdl._h, C.byref(cnt)))
        m = cnt.value - first if count is None else int(count)
        nrows = int(sim._rows.size)
        nx, _ = _dims(sim.mdl)
        self.t = np.zeros(m)
        data = np.zeros((m, nrows, sim.mdl.n))
        if m > 0 and nrows > 0:
            check(lib.fb_log_read(sim.mdl._h, int(first), m, _pd(self.t), _pd(data)))
        self.t = self.t + sim.t_start
        ny = len(sim.y_rows)
        self.y_rows = list(sim.y_rows)
        self.y = data[:, :ny] if ny else None
        self.x = data[:, ny:] if sim.save_x else np.zeros((m, nx, sim.mdl.n))

    def __len__(self):
        return len(self.t)
