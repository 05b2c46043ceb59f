"""Scripted scenarios as a table: the device-side `user_callback!`.

The reference's scenarios are closures handed to `Simulation(...; user_callback!)` that run after every step (lib/FlightCore/src/sim.jl:185,
334-336): a `phase` symbol, and per phase "set these inputs; if <condition on the model's outputs> then set those inputs and go to the next
phase" (lib/FlightApps/demos/c172_demos.jl:423-486 crosswind landing, :525-642 traffic pattern). A batch of N simulations cannot call back into
the host after every step without crossing PCIe with the whole output record; the same logic as DATA — phases, rules, actions — is evaluated
on the device by `k_scenario` (csrc/scenario_kernels.hpp) between the stepping launches, with one phase word, one entry time, `n_par`
parameters and `n_rec` record slots per aircraft. Nothing of a run touches the host.

    scn = Scenario(n_par=..., n_rec=...)
    FINAL, FLARE = scn.phase("final"), scn.phase("flare")
    scn.when(FINAL, src.H_E - par(P_H_RWY) < 6.0, [cu("SEG_VRT_REQ", 0), cu("CLM_REF", -0.3), ...], then=FLARE)
    scn.always(GROUND, [cu("THROTTLE_AXIS", 0), u("BRAKE_LEFT", 1)])
    world.set_scenario(scn, params=[n_par, n], every=1)

Semantics of one evaluation (one aircraft): the `always` actions of its phase run; then the phase's rules are tried in order and the FIRST whose
condition holds runs its actions and sets the next phase — at most one transition per evaluation, as in the demos' if / elseif chains.
Values are read when their action runs, so an action sees what the actions before it wrote.

`evaluate_on_host` is the same interpreter in numpy over arrays the caller supplies: the host-callback form of a table (tests compare the two),
and the checker's phase machine in tests/test_gpu_scenarios.py (driving the CPU oracle)."""
from __future__ import annotations

import numpy as np

from ._lib import K

MAGIC = 5.0e6 + 1   # version 1 of the blob layout
HDR, PH_REC, RULE_REC, ACT_REC, NTERM = 8, 4, 8, 14, 3
# value sources (kind, row)
SRC = {name: K["FB_SCN_SRC_" + name] for name in ("CONST", "T", "T_IN_PHASE", "X", "CS", "CU", "U", "S", "ON_GND", "H_E", "PSI", "THETA", "PHI", "CHI",
                                                     "EAS", "CLM", "PAR", "REC")}
DST = {name: K["FB_SCN_DST_" + name] for name in ("CU", "U", "UI", "REC")}
CMP = {"<": K["FB_SCN_LT"], ">": K["FB_SCN_GT"], ">=": K["FB_SCN_GE"], "<=": K["FB_SCN_LE"], "==": K["FB_SCN_EQ"], "!=": K["FB_SCN_NE"], "always": K["FB_SCN_ALWAYS"]}


class Value:
    """c0 + sum of up to three coefficient x source terms, optionally wrapped to (-pi, pi] (Attitude.wrap_to_π, FP/attitude.jl:478)"""

    def __init__(self, c0=0.0, terms=(), wrap=False):
        self.c0, self.terms, self.wrap = float(c0), tuple(terms), bool(wrap)
        if len(self.terms) > NTERM:
            raise ValueError(f"a scenario value holds at most {NTERM} terms")

    def _lift(self, o):
        return o if isinstance(o, Value) else Value(float(o))

    def __add__(self, o):
        o = self._lift(o)
        return Value(self.c0 + o.c0, self.terms + o.terms, self.wrap or o.wrap)

    __radd__ = __add__

    def __neg__(self):
        return Value(-self.c0, tuple((k, r, -c) for k, r, c in self.terms), self.wrap)

    def __sub__(self, o):
        return self + (-self._lift(o))

    def __rsub__(self, o):
        return self._lift(o) + (-self)

    def __mul__(self, c):
        return Value(self.c0 * c, tuple((k, r, cc * c) for k, r, cc in self.terms), self.wrap)

    __rmul__ = __mul__

    # comparisons build conditions: one plain source, optionally minus one parameter row, against a constant:
    #     src.H_E - par(7) < 6.0        cs_("SEG_S_2B") > -200.0
    # (evaluated as written — lhs - par, then the comparison — like `vehicle.y.kinematics.h_e - final_leg.p2.h < 6` in the demos)
    def _cond(self, op, o):
        o = self._lift(o)
        if o.terms or o.wrap or self.wrap or self.c0 != 0.0:
            raise ValueError("a scenario condition compares `source [- par(row)]` with a constant")
        plain = [t for t in self.terms if t[0] != SRC["PAR"]]
        pars = [t for t in self.terms if t[0] == SRC["PAR"]]
        if len(plain) != 1 or plain[0][2] != 1.0 or len(pars) > 1 or (pars and pars[0][2] != -1.0):
            raise ValueError("a scenario condition compares `source [- par(row)]` with a constant")
        return Condition(plain[0][0], plain[0][1], CMP[op], o.c0, pars[0][1] if pars else -1)

    def __lt__(self, o): return self._cond("<", o)
    def __gt__(self, o): return self._cond(">", o)
    def __ge__(self, o): return self._cond(">=", o)
    def __le__(self, o): return self._cond("<=", o)
    def eq(self, o): return self._cond("==", o)
    def ne(self, o): return self._cond("!=", o)


def wrap_to_pi(v: Value) -> Value:
    return Value(v.c0, v.terms, True)


class Condition:
    def __init__(self, kind, row, cmp, thr, thr_par=-1):
        self.kind, self.row, self.cmp, self.thr, self.thr_par = int(kind), int(row), int(cmp), float(thr), int(thr_par)


ALWAYS = Condition(SRC["CONST"], 0, CMP["always"], 0.0)


def _src(kind, row=0):
    return Value(0.0, ((SRC[kind], int(row), 1.0),))


class _Sources:
    """src.T, src.H_E, src.PSI, src.ON_GND, ...: what a condition or a value may read of the model after a step"""
    T = _src("T"); T_IN_PHASE = _src("T_IN_PHASE"); ON_GND = _src("ON_GND"); H_E = _src("H_E"); PSI = _src("PSI"); THETA = _src("THETA")
    PHI = _src("PHI"); CHI = _src("CHI"); EAS = _src("EAS"); CLM = _src("CLM")


src = _Sources()


def cs_(name_or_row) -> Value:
    """a row of the control-law record (avionics.y): cs_('SEG_S_2B')"""
    return _src("CS", K["FB_CS_" + name_or_row] if isinstance(name_or_row, str) else name_or_row)


def cu_(name_or_row) -> Value:
    return _src("CU", K["FB_CU_" + name_or_row] if isinstance(name_or_row, str) else name_or_row)


def s_(name_or_row) -> Value:
    """a discrete state row (FB_S_STALL, FB_S_ENG_STATE)"""
    return _src("S", K["FB_S_" + name_or_row] if isinstance(name_or_row, str) else name_or_row)


def par(row) -> Value:
    return _src("PAR", row)


def rec_(row) -> Value:
    return _src("REC", row)


class Action:
    def __init__(self, dst, row, value):
        self.dst, self.row = int(dst), int(row)
        self.value = value if isinstance(value, Value) else Value(float(value))


def cu(name, value) -> Action:
    """avionics.{gdc, ctl}.u.<field> = value (a row of the control-law inputs, FB_CU_*)"""
    return Action(DST["CU"], K["FB_CU_" + name], value)


def u(name, value) -> Action:
    """a row of the vehicle's inputs (FB_U_*: flaps, brakes, ...)"""
    return Action(DST["U"], K["FB_U_" + name], value)


def ui(bit_name, on) -> Action:
    """a bit of the discrete inputs (FB_UI_*: engine start / stop ...): set where value != 0, cleared otherwise"""
    return Action(DST["UI"], K["FB_UI_" + bit_name], on if isinstance(on, Value) else (1.0 if on else 0.0))


def rec(row, value) -> Action:
    """record slot `row` of the aircraft = value (touchdown time, position ...: read back with world.scenario_state())"""
    return Action(DST["REC"], row, value)


def target(p1_par: int, p2_par: int):
    """gdc.seg.u.target = Segment(p1, p2) with the end points in parameter rows p1_par .. p1_par + 2, p2_par .. p2_par + 2"""
    return [Action(DST["CU"], K["FB_CU_SEG_P1"] + k, par(p1_par + k)) for k in range(3)] + [Action(DST["CU"], K["FB_CU_SEG_P2"] + k, par(p2_par + k)) for k in range(3)]


class Scenario:
    def __init__(self, n_par=0, n_rec=0):
        self.n_par, self.n_rec = int(n_par), int(n_rec)
        self.names: list[str] = []
        self._always: list[list[Action]] = []
        self._rules: list[list[tuple]] = []

    def phase(self, name: str) -> int:
        self.names.append(name); self._always.append([]); self._rules.append([])
        return len(self.names) - 1

    def always(self, phase: int, actions):
        self._always[phase] += list(actions)

    def when(self, phase: int, cond: Condition, actions=(), then: int | None = None):
        self._rules[phase].append((cond, list(actions), phase if then is None else int(then)))

    def pack(self) -> np.ndarray:
        """the FB_TABLE_SCENARIO blob (include/flightbatch.h)"""
        acts: list[Action] = []
        rules, phases = [], []
        for p in range(len(self.names)):
            a0 = len(acts); acts += self._always[p]
            r0 = len(rules)
            for cond, ra, nxt in self._rules[p]:
                f = len(acts); acts += ra
                rules.append([cond.kind, cond.row, cond.cmp, cond.thr, cond.thr_par, f, len(ra), nxt])
            phases.append([a0, len(self._always[p]), r0, len(self._rules[p])])
        blob = [MAGIC, len(phases), len(rules), len(acts), self.n_par, self.n_rec, 0.0, 0.0]
        for ph in phases:
            blob += ph
        for r in rules:
            blob += r
        for a in acts:
            if a.dst == DST["REC"] and not 0 <= a.row < self.n_rec:
                raise ValueError("record row out of range")
            rowv = [a.dst, a.row, 1.0 if a.value.wrap else 0.0, a.value.c0, len(a.value.terms)]
            for k in range(NTERM):
                kind, row, c = a.value.terms[k] if k < len(a.value.terms) else (SRC["CONST"], 0, 0.0)
                if kind == SRC["PAR"] and not 0 <= row < self.n_par:
                    raise ValueError("parameter row out of range")
                rowv += [kind, row, c]
            assert len(rowv) == ACT_REC
            blob += rowv
        return np.asarray(blob, dtype=np.float64)


# ---- the same interpreter on the host, over arrays the caller supplies -------------------------------------------------------------
def _wrap(x):
    return x + 2 * np.pi * np.floor((np.pi - x) / (2 * np.pi))


def evaluate_on_host(blob: np.ndarray, st: dict, t: float, dt: float) -> None:
    """One evaluation of the table for every aircraft, in place. st: phase [n] int, since [n] int64 (step count at the entry of the phase),
    step (int: steps taken), par [n_par, n], rec [n_rec, n], cu, cs, u, ui, s (the model's arrays, modified in place), and the outputs the
    sources name: on_gnd, h_e, psi, theta, phi, chi, EAS, clm [n]; x [rows, n] (device row order) where SRC X is used; active [n] bool
    (aircraft whose simulation has ended are not evaluated)."""
    assert blob[0] == MAGIC
    n_ph, n_rule, n_act = int(blob[1]), int(blob[2]), int(blob[3])
    PH = blob[HDR:HDR + PH_REC * n_ph].reshape(n_ph, PH_REC)
    RU = blob[HDR + PH_REC * n_ph:HDR + PH_REC * n_ph + RULE_REC * n_rule].reshape(n_rule, RULE_REC)
    AC = blob[HDR + PH_REC * n_ph + RULE_REC * n_rule:].reshape(n_act, ACT_REC)
    phase0 = st["phase"].copy()
    n = phase0.size

    def source(kind, row, m):
        kind, row = int(kind), int(row)
        if kind == SRC["CONST"]: return np.ones(m.sum())
        if kind == SRC["T"]: return np.full(m.sum(), t)
        if kind == SRC["T_IN_PHASE"]: return (st["step"] - st["since"][m]) * dt
        if kind == SRC["X"]: return st["x"][row, m]
        if kind == SRC["CS"]: return st["cs"][row, m]
        if kind == SRC["CU"]: return st["cu"][row, m]
        if kind == SRC["U"]: return st["u"][row, m]
        if kind == SRC["S"]: return st["s"][row, m].astype(np.float64)
        if kind == SRC["PAR"]: return st["par"][row, m]
        if kind == SRC["REC"]: return st["rec"][row, m]
        name = {SRC["ON_GND"]: "on_gnd", SRC["H_E"]: "h_e", SRC["PSI"]: "psi", SRC["THETA"]: "theta", SRC["PHI"]: "phi", SRC["CHI"]: "chi",
                SRC["EAS"]: "EAS", SRC["CLM"]: "clm"}[kind]
        return np.asarray(st[name], dtype=np.float64)[m]

    def run(a, m):
        if not m.any():
            return
        v = np.full(m.sum(), a[3])
        for k in range(int(a[4])):
            kind, row, c = a[5 + 3 * k:8 + 3 * k]
            v = v + c * source(kind, row, m)
        if a[2] != 0:
            v = _wrap(v)
        dst, row = int(a[0]), int(a[1])
        if dst == DST["CU"]: st["cu"][row, m] = v
        elif dst == DST["U"]: st["u"][row, m] = v
        elif dst == DST["REC"]: st["rec"][row, m] = v
        else:
            w = st["ui"][m]
            st["ui"][m] = np.where(v != 0, w | row, w & ~row)

    for p in range(n_ph):
        m = (phase0 == p) & st["active"]
        if not m.any():
            continue
        a0, na, r0, nr = (int(v) for v in PH[p])
        for a in AC[a0:a0 + na]:
            run(a, m)
        left = m.copy()
        for r in RU[r0:r0 + nr]:
            if not left.any():
                break
            lhs = np.zeros(n); lhs[left] = source(r[0], r[1], left)
            if r[4] >= 0:
                lhs = lhs - st["par"][int(r[4])]
            thr = np.full(n, r[3])
            c = int(r[2])
            hold = {CMP["<"]: lhs < thr, CMP[">"]: lhs > thr, CMP[">="]: lhs >= thr, CMP["<="]: lhs <= thr, CMP["=="]: lhs == thr, CMP["!="]: lhs != thr,
                    CMP["always"]: np.ones(n, bool)}[c] & left
            for a in AC[int(r[5]):int(r[5]) + int(r[6])]:
                run(a, hold)
            st["phase"][hold] = int(r[7])
            st["since"][hold] = st["step"]
            left &= ~hold
