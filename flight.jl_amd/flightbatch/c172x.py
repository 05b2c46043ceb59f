"""Host mirror of `Cessna172Xv2()` (lib/FlightApps/src/c172/c172x/c172x2.jl:52-58) for the batched path.

    reference                                                     here
    Model(SimpleWorld(Cessna172Xv2()))                            Cessna172Xv2World(n)
    init!(sim, C172.TrimParameters())                             init(sim, TrimParameters(...))   (trim + avionics init)
    ctl = world.aircraft.avionics.ctl
    ctl.u.lon.mode_req = ModeControlLon.EAS_clm                   world.ctl.lon.mode_req = ModeControlLon.EAS_clm
    ctl.u.lat.φ_ref = π/6                                         world.ctl.lat.φ_ref = np.pi / 6      (scalars broadcast; arrays per aircraft)
    ctl.y.lon.mode, ctl.y.lon.throttle_cmd, ...                   world.ctl.y("LON_MODE"), world.ctl.y("THROTTLE_CMD")
    act.flaps.u[] = 0.3                                           u = world.u; u[K["FB_U_FLAPS"]] = 0.3; world.u = u
    Simulation(world; dt = 0.01, Δt = 0.02)                       Simulation(world, dt=0.01, Δt=0.02)

State rows follow the reference's ComponentVector (include/flightbatch.h, FB_X2_*)."""
from __future__ import annotations

import ctypes as C
import enum
import numpy as np

from . import ctl_gains
from ._lib import K, check, lib
from .modeling import BatchedWorld, _pd


class ModeControlLon(enum.IntEnum):   # c172x_ctl.jl:29-39
    direct = 0; sas = 1; thr_q = 2; thr_θ = 3; thr_EAS = 4; EAS_q = 5; EAS_θ = 6; EAS_clm = 7; EAS_alt = 8


class ModeControlLat(enum.IntEnum):   # c172x_ctl.jl:727-733
    direct = 0; sas = 1; p_β = 2; φ_β = 3; χ_β = 4


_LON_FIELDS = {"mode_req": "LON_MODE_REQ", "throttle_axis": "THROTTLE_AXIS", "throttle_offset": "THROTTLE_OFFSET",
               "elevator_axis": "ELEVATOR_AXIS", "elevator_offset": "ELEVATOR_OFFSET", "q_ref": "Q_REF", "θ_ref": "THETA_REF",
               "EAS_ref": "EAS_REF", "clm_ref": "CLM_REF", "h_ref": "H_REF"}
_LAT_FIELDS = {"mode_req": "LAT_MODE_REQ", "aileron_axis": "AILERON_AXIS", "aileron_offset": "AILERON_OFFSET",
               "rudder_axis": "RUDDER_AXIS", "rudder_offset": "RUDDER_OFFSET", "p_ref": "P_REF", "β_ref": "BETA_REF", "φ_ref": "PHI_REF",
               "χ_ref": "CHI_REF"}


class ModeGuidance(enum.IntEnum):    # c172x_gdc.jl:19-23
    direct = 0; segment = 1; circular = 2


_GDC_FIELDS = {"mode_req": "GDC_MODE_REQ", "hor_gdc_req": "SEG_HOR_REQ", "vrt_gdc_req": "SEG_VRT_REQ"}


class _Channel:
    """ctl.u.lon / ctl.u.lat: attribute access to rows of the cu array; assignments go to the device at once."""

    def __init__(self, world, fields):
        object.__setattr__(self, "_w", world)
        object.__setattr__(self, "_f", fields)

    def __getattr__(self, name):
        return self._w.cu[K["FB_CU_" + self._f[name]]]

    def __setattr__(self, name, value):
        cu = self._w.cu
        cu[K["FB_CU_" + self._f[name]]] = np.asarray(value, dtype=np.float64)
        self._w.cu = cu


class _ControlLaws:
    def __init__(self, world):
        self.lon = _Channel(world, _LON_FIELDS)
        self.lat = _Channel(world, _LAT_FIELDS)
        self.gdc = _Channel(world, _GDC_FIELDS)     # avionics.gdc.u.mode_req, gdc.seg.u.hor_gdc_req / vrt_gdc_req
        self._w = world

    def set_target(self, p1, p2):
        """gdc.seg.u.target = Segment(p1, p2): end points as (latitude, longitude, ellipsoidal altitude), each [3] or [3, n]."""
        cu = self._w.cu
        cu[K["FB_CU_SEG_P1"]:K["FB_CU_SEG_P1"] + 3] = np.asarray(p1, dtype=np.float64).reshape(3, -1)
        cu[K["FB_CU_SEG_P2"]:K["FB_CU_SEG_P2"] + 3] = np.asarray(p2, dtype=np.float64).reshape(3, -1)
        self._w.cu = cu

    def y(self, name: str) -> np.ndarray:
        """A row of the control-law record (FB_CS_* without the prefix): modes, references, commands, compensator states."""
        return self._w.cs[K["FB_CS_" + name]]


class Cessna172Xv2World(BatchedWorld):
    """N independent `Model(SimpleWorld(Cessna172Xv2()))` on one GPU."""
    MODEL = "FB_MODEL_C172X2"
    _CKPT_ARRAYS = ("x", "s", "u", "ui", "cu", "cs")

    def __init__(self, n: int, device: int = 0, tables: dict | None = None, gains: np.ndarray | None = None, kinematics: str = "WA"):
        """kinematics: Cessna172Xv2(kinematics) (FA/c172/c172x/c172x2.jl:57-59) — "WA" (34 states), "ECEF" (33) or "NED" (31)."""
        super().__init__(n, device, tables, kinematics=kinematics)
        blob = np.ascontiguousarray(gains if gains is not None else ctl_gains.ctl_gains_blob(), dtype=np.float64)
        dims = (C.c_int64 * 1)(blob.size)
        check(lib.fb_set_table(self._h, K["FB_TABLE_CTL_GAINS"], blob.ctypes.data_as(C.c_void_p), dims, 1))
        self.ctl = _ControlLaws(self)

    @property
    def cu(self) -> np.ndarray:
        cu = np.empty((K["FB_NCU"], self.n))
        check(lib.fb_get_ctl_inputs(self._h, _pd(cu)))
        return cu

    @cu.setter
    def cu(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(K["FB_NCU"], self.n)
        check(lib.fb_set_ctl_inputs(self._h, _pd(v)))

    @property
    def cs(self) -> np.ndarray:
        cs = np.empty((K["FB_NCS"], self.n))
        check(lib.fb_get_ctl_state(self._h, _pd(cs)))
        return cs

    @cs.setter
    def cs(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(K["FB_NCS"], self.n)
        check(lib.fb_set_ctl_state(self._h, _pd(v)))

    # -- scripted scenarios: the device-side user_callback! (flightbatch/scenario.py; FC/sim.jl:185, 334-336) --
    def set_scenario(self, scn, params: np.ndarray | None = None, every: int = 1, rec_init: float = 0.0) -> None:
        """Load a scenario table (scenario.Scenario, or None to switch scenarios off), the per-aircraft parameter rows [n_par, n] and the
        evaluation period in steps (1: after every step, like the reference's user_callback!). Every aircraft starts in phase 0."""
        if scn is None:
            check(lib.fb_scenario_configure(self._h, 0))
            self._scn = None
            return
        blob = np.ascontiguousarray(scn.pack(), dtype=np.float64)
        dims = (C.c_int64 * 1)(blob.size)
        check(lib.fb_set_table(self._h, K["FB_TABLE_SCENARIO"], blob.ctypes.data_as(C.c_void_p), dims, 1))
        check(lib.fb_scenario_configure(self._h, int(every)))
        self._scn = (blob, scn.n_par, scn.n_rec, int(every), list(scn.names))
        if scn.n_par:
            p = np.ascontiguousarray(params, dtype=np.float64).reshape(scn.n_par, self.n)
            check(lib.fb_scenario_set_params(self._h, _pd(p)))
        if scn.n_rec and rec_init != 0.0:
            self.set_scenario_state(np.zeros(self.n, np.int32), np.zeros(self.n, np.int64), np.full((scn.n_rec, self.n), rec_init))

    def scenario_state(self) -> dict:
        """phase [n], since [n] (step count at the entry of the phase), rec [n_rec, n]"""
        if getattr(self, "_scn", None) is None:
            raise RuntimeError("no scenario is loaded (set_scenario)")
        n_rec = self._scn[2]
        phase = np.empty(self.n, np.int32); since = np.empty(self.n, np.int64); rec = np.empty((max(n_rec, 1), self.n))
        check(lib.fb_scenario_get_state(self._h, phase.ctypes.data_as(C.POINTER(C.c_int32)), since.ctypes.data_as(C.POINTER(C.c_int64)), _pd(rec)))
        return {"phase": phase, "since": since, "rec": rec[:n_rec]}

    def set_scenario_state(self, phase, since, rec=None) -> None:
        phase = np.ascontiguousarray(phase, dtype=np.int32).reshape(self.n); since = np.ascontiguousarray(since, dtype=np.int64).reshape(self.n)
        r = None if rec is None else np.ascontiguousarray(rec, dtype=np.float64).reshape(self._scn[2], self.n)
        check(lib.fb_scenario_set_state(self._h, phase.ctypes.data_as(C.POINTER(C.c_int32)), since.ctypes.data_as(C.POINTER(C.c_int64)),
                                        _pd(r) if r is not None and r.size else None))

    def checkpoint(self) -> dict:
        ck = super().checkpoint()
        if getattr(self, "_scn", None) is not None:   # the scenario travels with the checkpoint: table, period, parameters, per-aircraft state
            blob, n_par, n_rec, every, names = self._scn
            st = self.scenario_state()
            par = np.empty((max(n_par, 1), self.n))
            if n_par:
                check(lib.fb_scenario_get_params(self._h, _pd(par)))
            ck.update(scn_blob=blob, scn_every=np.int64(every), scn_par=par[:n_par], scn_phase=st["phase"], scn_since=st["since"], scn_rec=st["rec"],
                      scn_dims=np.array([n_par, n_rec], np.int64))
        return ck

    def restore(self, ck: dict) -> None:
        super().restore(ck)
        if "scn_blob" in ck:
            blob = np.ascontiguousarray(ck["scn_blob"], dtype=np.float64)
            dims = (C.c_int64 * 1)(blob.size)
            check(lib.fb_set_table(self._h, K["FB_TABLE_SCENARIO"], blob.ctypes.data_as(C.c_void_p), dims, 1))
            check(lib.fb_scenario_configure(self._h, int(ck["scn_every"])))
            n_par, n_rec = (int(v) for v in ck["scn_dims"])
            self._scn = (blob, n_par, n_rec, int(ck["scn_every"]), [])
            if n_par:
                check(lib.fb_scenario_set_params(self._h, _pd(np.ascontiguousarray(ck["scn_par"], dtype=np.float64))))
            self.set_scenario_state(ck["scn_phase"], ck["scn_since"], ck["scn_rec"] if n_rec else None)
