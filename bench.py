#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched 6-DOF hot path (BASELINE.json: aircraft-steps/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — 1 048 576 Cessna172Sv0, randomised trim on the 32 x 32 x 1024
(EAS x altitude x heading) lattice of SURVEY.md §8(d), order permuted by a fixed LCG (seed 172), fp64, dt = 0.01.
Synthetic: no recorded data exists for this path.

One "step" of the contract = one pass of the hot path over the batch = ONE launch of the fused stepping kernel advancing
every aircraft by `--inner` RK4 steps (default 50, i.e. 0.5 s of flight), including all RK stages, the output evaluation
at the new state and f_step!. Trim, table generation and upload are outside the timed region; state is resident in HBM
when timing starts.

Multi-GPU (one process per GPU, aircraft are independent, NO data-path collective). For --gpus > 1 the default is `--scaling strong`:
BASELINE.json's metric is the whole node at N = 1 M, i.e. the SAME 1 048 576 aircraft cut into contiguous shards
(flightbatch.sharding.shard_range), 131 072 per GPU at N = 8, and `value` is the whole job's rate on THAT configuration. The weak figure —
every rank stepping its own 1 048 576 aircraft (configs[2] per GPU, rank r on lattice(r)) — is measured in the same run and attached as
"weak_scaling" (`--scaling weak` makes it the headline instead). One RCCL all-gather of the final states collects the
trajectory endpoint after the timed region (gather_ms, not part of `value`). At --gpus 1 the two coincide.

On one GPU the line also carries, under "extra", one GPU's share of configs[3] (524 288 Cessna172Xv2 with the autopilot at
Δt = 0.02) and configs[4] (mixed fp32 fleet, 50 % Cessna172Sv0 / 50 % Robot2D), each with its own timing and parity sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd"))

N_TOTAL = 1 << 20
DT = 0.01
BYTES_PER_AIRCRAFT_STEP = 440.0   # 2 * Nx * 8 B + 8 B of flags, C172Sv0 fp64
BYTES_PER_AIRCRAFT_STEP_F32 = 264.0   # the fp32 stepper: 2 * (27 * 4 B + 5 rows of q_ew / h_e kept in fp64: + 5 * 4 B) + 8 B (extra_fleet's bytes_c)
BYTES_PER_X2_STEP = 756.0         # Cessna172Xv2
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X vector fp64 peak (spec): 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PROFILE_COUNTERS = "r06_counters.json"
PROFILE_COUNTERS_X2 = "r06_x2_counters.json"   # the same for the Cessna172Xv2 airborne stepper (tools/collect_profile_x2.sh)   # rocprofv3 PMC summary of the CURRENT airborne stepper (tools/collect_profile.sh)


def lattice(seed_offset: int = 0, n: int = N_TOTAL):
    """config 3 of SURVEY.md §8(d): EAS_i = 35 + 20 i/31, h_j = 200 + 2800 j/31, ψ_k = -π + 2π k/1024,
    aircraft order permuted by a fixed LCG so neighbouring lanes hold different table cells. Returns (EAS, h, ψ, cell)
    with cell = 32 i + j."""
    idx = np.arange(n, dtype=np.uint64)
    a, c = np.uint64(1664525), np.uint64(1013904223)       # full-period LCG modulo 2^20 (a ≡ 1 mod 4, c odd)
    perm = (a * idx + c + np.uint64(172 + 7919 * seed_offset)) & np.uint64(N_TOTAL - 1)
    i = (perm >> np.uint64(15)) & np.uint64(31)
    j = (perm >> np.uint64(10)) & np.uint64(31)
    k = perm & np.uint64(1023)
    EAS = 35.0 + 20.0 * i.astype(np.float64) / 31.0
    h = 200.0 + 2800.0 * j.astype(np.float64) / 31.0
    psi = -np.pi + 2 * np.pi * k.astype(np.float64) / 1024.0
    return EAS, h, psi, (i * np.uint64(32) + j).astype(np.int64)


def stratified_sample(cell: np.ndarray, per_cell: int = 4, seed: int = 2) -> np.ndarray:
    """indices drawn over the WHOLE permuted batch, `per_cell` from every (EAS, h) cell present (1024 cells x 4 = 4096)"""
    rng = np.random.default_rng(seed)
    order = np.argsort(cell, kind="stable")
    bounds = np.flatnonzero(np.diff(cell[order])) + 1
    groups = np.split(order, bounds)
    return np.sort(np.concatenate([rng.choice(g, size=min(per_cell, g.size), replace=False) for g in groups]))


def usable_cores() -> int:
    """Cores this process can really use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose
    128 logical CPUs but grant a 16-core share per GPU)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, 16) if n > 64 else n   # no quota visible on a 128-CPU host: stay within the documented 16-core share


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    return Oracle()


def cpu_baseline(x0, u0, ui0, s0, budget_s=15.0):
    """The CPU oracle (a C++ port of the reference path — NOT the Julia reference, which cannot run here) timed on this box's
    host cores on a bounded sample of the same workload. Reference-like arithmetic: 6 RHS evaluations per step, as
    OrdinaryDiffEq's RK4 does with Flight.jl's state-modifying step callback."""
    orc = _oracle()
    threads = min(orc.max_threads(), usable_cores())   # the cores this process may actually use
    env = orc.default_env()
    m = min(16384, x0.shape[1])
    sel = slice(0, m)
    xs, us, uis, ss = (np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(u0[:, sel]), np.ascontiguousarray(ui0[sel]),
                       np.ascontiguousarray(s0[:, sel]))
    t0 = time.perf_counter()
    orc.step(xs, us, uis, ss, env, DT, 20, threads=threads, reference_like=True)
    probe = time.perf_counter() - t0
    rate = m * 20 / probe
    nsteps = int(max(20, min(2000, budget_s * rate / m)))
    t0 = time.perf_counter()
    orc.step(xs, us, uis, ss, env, DT, nsteps, threads=threads, reference_like=True)
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    m1 = 256
    orc.step(np.ascontiguousarray(x0[:, :m1]), np.ascontiguousarray(u0[:, :m1]), np.ascontiguousarray(ui0[:m1]),
             np.ascontiguousarray(s0[:, :m1]), env, DT, 100, threads=1, reference_like=True)
    el1 = time.perf_counter() - t1
    return {"value": m * nsteps / el, "unit": "aircraft-steps/s", "cores": threads, "kind": "port",
            "sample": f"{m} aircraft of the same lattice x {nsteps} RK4 steps, OpenMP over aircraft, 6 RHS evaluations/step "
                      f"(reference-like), {el:.1f} s; the port is oracle/ (C++), the Julia reference cannot run on this box",
            "single_core_value": m1 * 100 / el1}


def state_floor(xo):
    """per-state scale max(|xo|, floor) with the floors of SURVEY.md §8(d) (quaternions 1, rates 1e-3 rad/s, ...)"""
    sc = np.maximum(np.abs(xo), 1e-3)
    sc[12:20] = 1.0; sc[2:8] = 1.0; sc[10:12] = 1.0; sc[0:2] = np.maximum(np.abs(xo[0:2]), 1e-2); sc[24:27] = np.maximum(np.abs(xo[24:27]), 1.0)
    return sc


def scaled_error(x, xo):
    return np.abs(x - xo) / state_floor(xo)


def parity_sample(fb, x0, u0, ui0, s0, cell, dtype, nsteps=1000):
    """The second half of BASELINE.json's metric ("fp64 rel-err vs CPU"): a stratified sample of the benchmark batch — four
    aircraft from every one of the 1024 (EAS, h) cells, drawn over the whole permuted order — stepped nsteps times on the GPU
    and by the CPU oracle (the C++ port) from the same initial condition."""
    orc = _oracle()
    sel = stratified_sample(cell)
    xs, us, uis, ss = (np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(u0[:, sel]), np.ascontiguousarray(ui0[sel]), np.ascontiguousarray(s0[:, sel]))
    m = len(sel)
    w = fb.BatchedWorld(m, dtype=dtype)
    w.set_state(xs, ss); w.u = us; w.ui = uis
    sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
    fb.step(sim, nsteps * DT); w.sync()
    xo, so, sto = orc.step(xs, us, uis, ss, orc.default_env(), DT, nsteps, threads=min(orc.max_threads(), usable_cores()))
    # every aircraft of the sample is compared, terminated ones included: the oracle stops an aircraft where the reference stops it
    # (FC/sim.jl:561-570) and so must the GPU — same status word, same frozen state
    st = w.status
    same_status = bool(np.array_equal(st, sto)) if dtype == "f64" else bool(np.array_equal(st != 0, sto != 0))
    ok = np.ones(m, bool) if dtype == "f64" else (sto == 0) & (st == 0)
    err = float(scaled_error(w.x, xo)[:, ok].max()) if same_status else float("inf")
    w.close()
    return {"max_scaled_error": err, "against": "oracle/ (C++ port of the reference path; the Julia reference cannot run here)",
            "status_words_equal": same_status, "terminated_in_sample": int((sto != 0).sum()),
            "sample": f"{int(ok.sum())} aircraft (4 per (EAS, h) cell over the whole batch) x {nsteps} RK4 steps, fp64 oracle",
            "cells_covered": int(np.unique(cell[sel]).size), "tolerance": 1e-6 if dtype == "f64" else None}


# ---------------------------------------------------------------------------------------------------------------------
# Rehearsal of the N > 1 code path on a ONE-GPU box (FLIGHTBATCH_BENCH_REHEARSAL=1): every rank uses GPU 0, the process group is gloo
# and the collectives run on host copies. The numbers mean nothing; what it exercises is every line a multi-GPU run executes
# (tools/rehearse_multi_gpu.sh). Never set by the driver.
REHEARSAL = os.environ.get("FLIGHTBATCH_BENCH_REHEARSAL", "0") == "1"


def _cdev():
    return "cpu" if REHEARSAL else "cuda"


def _gather_state(fb, x_dev, n_total):
    return fb.sharding.all_gather_state(x_dev.cpu() if REHEARSAL else x_dev, n_total)


def per_launch_ms(fb, C, w):
    """the per-launch durations recorded by fb_timing_begin_per_launch (call after fb_timing_end)"""
    nl = C.c_int64()
    fb._lib.check(fb.lib.fb_timing_launches(w._h, None, 0, C.byref(nl)))
    buf = (C.c_float * max(nl.value, 1))()
    fb._lib.check(fb.lib.fb_timing_launches(w._h, buf, nl.value, C.byref(nl)))
    return [float(buf[k]) for k in range(nl.value)]


def launch_stats(ms):
    """median / min / max / count of a list of per-launch durations (the figure quoted as kernel_ms is the MEDIAN)"""
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {"kernel_ms": float(np.median(a)), "kernel_ms_min": float(a[0]), "kernel_ms_max": float(a[-1]), "launches_timed": int(a.size),
            "kernel_ms_per_launch": [round(float(v), 4) for v in ms]}


def warm_until_stable(fb, C, w, launch, tol=0.01, cap=30, at_least=3):
    """Warm-up for the extra legs: launch until the last three launches agree within `tol` (max / min - 1), at most `cap` launches.
    Each launch is timed by its own HIP-event pair. Returns the warm-up launches' durations (kept in the bench line: they show the
    clock ramp after an idle period, profiles/r06_x2_repro.txt)."""
    seen = []
    while len(seen) < cap:
        fb._lib.check(fb.lib.fb_timing_begin_per_launch(w._h, 1))
        launch()
        ms = C.c_float(); nl = C.c_int64()
        fb._lib.check(fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl)))
        seen += per_launch_ms(fb, C, w)[:1]
        if len(seen) >= at_least and max(seen[-3:]) / min(seen[-3:]) - 1.0 < tol:
            break
    return seen


def time_c172s0(fb, torch, dist, C, EAS, h, psi, args, local_rank, world):
    """trim + warm-up + the timed region for one shard of the Cessna172Sv0 batch. Returns a dict of measurements (and the
    initial condition on the host, for the CPU legs)."""
    n = len(EAS)
    w = fb.BatchedWorld(n, device=local_rank, dtype=args.dtype)
    # the state lives in a torch tensor so that RCCL can gather it without a host round trip
    x_dev = torch.zeros((fb.K["FB_NX"], n), dtype=torch.float64, device="cuda")
    s_dev = torch.zeros((fb.K["FB_NS"], n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fb._lib.check(fb.lib.fb_attach_state(w._h, C.c_void_p(x_dev.data_ptr()), C.c_void_p(s_dev.data_ptr())))
    t0 = time.perf_counter()
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    trim_s = time.perf_counter() - t0
    trim_ok = int(w.trim_success.sum())
    sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=args.inner)
    ic = (w.x, w.s, w.u, w.ui)
    fuel0 = ic[0][8].copy()

    def barrier():
        if world > 1:
            dist.barrier()
        w.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        fb.step(sim, args.inner * DT)
    barrier()
    fb._lib.check(fb.lib.fb_timing_begin_per_launch(w._h, args.steps))   # (one event pair per launch as well: the spread rides along)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fb.step(sim, args.inner * DT)
    barrier()
    elapsed = time.perf_counter() - t0
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    # roofline.achieved divides by the AVERAGE launch duration over the timed region (the contract's definition); median / min / max ride along
    out = {"n": n, "elapsed": elapsed, "kernel_ms": ms.value / max(nl.value, 1), "launch_stats": launch_stats(per_launch_ms(fb, C, w)),
           "trim_ok": trim_ok, "trim_s": trim_s, "ic": ic}
    # full-size invariants over ALL aircraft of the shard
    st = w.status
    xf = w.x
    out["status_bad"] = int((st != 0).sum())
    out["fuel_not_decreasing"] = int((xf[8] >= fuel0).sum())
    qtol = 1.1e-8 if args.dtype == "f64" else 5e-7    # f_step! renormalises beyond 1e-8 (kinematics.jl:114-118); the fp32 stepper holds q_wb to fp32 rounding
    out["quat_off_unit"] = int(((np.abs(np.sqrt((xf[12:16] ** 2).sum(0)) - 1) > qtol) | (np.abs(np.sqrt((xf[16:20] ** 2).sum(0)) - 1) > qtol)).sum())
    out["non_finite"] = int((~np.isfinite(xf)).any(axis=0).sum())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_cdev())
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        every = [float(e.item()) for e in every]
        out["elapsed"] = max(every); out["elapsed_per_rank"] = every
        cnt = torch.tensor([out["trim_ok"], out["status_bad"], out["fuel_not_decreasing"], out["quat_off_unit"], out["non_finite"], n],
                           dtype=torch.int64, device=_cdev())
        dist.all_reduce(cnt)
        out["trim_ok"], out["status_bad"], out["fuel_not_decreasing"], out["quat_off_unit"], out["non_finite"], out["n_total"] = [int(v) for v in cnt.tolist()]
        # trajectory collection: ONE RCCL all-gather of the final states over xGMI (north star)
        torch.cuda.synchronize(); dist.barrier(); g0 = time.perf_counter()
        gathered = _gather_state(fb, x_dev, out["n_total"])
        torch.cuda.synchronize(); out["gather_ms"] = (time.perf_counter() - g0) * 1e3
        assert gathered.shape == (fb.K["FB_NX"], out["n_total"])
        del gathered
    else:
        out["n_total"] = n
    w.close()
    del x_dev, s_dev
    return out


def check_valid(m, what):
    n = m["n_total"]
    if m["trim_ok"] != n:
        raise SystemExit(f"INVALID RUN ({what}): {n - m['trim_ok']} of {n} aircraft failed to trim; every cell of the lattice has a trim")
    for key in ("status_bad", "fuel_not_decreasing", "quat_off_unit", "non_finite"):
        if m[key]:
            raise SystemExit(f"INVALID RUN ({what}): {key} = {m[key]} of {n} aircraft — refusing to report a throughput")


def time_x2(fb, torch, dist, C, args, local_rank=0, world=1, divergent=False):
    """BASELINE.json configs[3], one GPU's share per rank: 524 288 Cessna172Xv2, autopilot every 2 steps, README example 2 scenario.
    With `world` ranks this IS configs[3] at world = 8 (4 194 304 aircraft): same barrier + max-over-ranks timing as the headline, and
    the final-state gather (one RCCL all-gather of the 34-row state panels) timed behind it.
    divergent: the same number of aircraft on bench.lattice(3)'s randomised trims (as tests/test_gpu_fullsize.py flies them), every aircraft
    in its OWN pair of control modes with its own references (as extra_x2's parity sample draws them): neighbouring lanes sit in
    different gain-schedule cells, aerodynamic table cells and control-law branches — configs[3]'s identical-aircraft scenario is the
    best case of every wave-uniform load and branch in the update, this is the other end."""
    n = N_TOTAL // 2
    if float(getattr(args, "x2_pre_sleep", 0.0)) > 0:   # (tools/bench_x2.py --pre-sleep: an idle GPU in front of the leg, as round 5's driver run had behind the CPU legs)
        time.sleep(float(args.x2_pre_sleep))
    w = fb.Cessna172Xv2World(n, device=local_rank)
    x_dev = None
    if world > 1:   # the state in a torch tensor, so that RCCL gathers it in place
        x_dev = torch.zeros((fb.K["FB_X2_NX"], n), dtype=torch.float64, device="cuda")
        s_dev = torch.zeros((fb.K["FB_NS"], n), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        fb._lib.check(fb.lib.fb_attach_state(w._h, C.c_void_p(x_dev.data_ptr()), C.c_void_p(s_dev.data_ptr())))
    w.set_params(wind_ned=(1.0, 0.5, 0.0))
    sim = fb.Simulation(w, dt=DT, Δt=getattr(args, "x2_ratio", 2) * DT, save_on=False, steps_per_launch=args.x2_inner)   # (x2_ratio: tools/bench_x2_divergence.py only)
    if divergent:   # (tools/bench_x2_divergence.py passes "trim" / "modes" to time the two sources of divergence apart)
        EAS, h, psi, _ = lattice(3, n)
        fb.init(sim, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi) if divergent != "modes" else fb.TrimParameters())
        assert w.trim_success.all()
        K = fb.K
        rng = np.random.default_rng(33)
        cu = w.cu
        cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, n); cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, n)
        cu[K["FB_CU_EAS_REF"]] += rng.uniform(-3, 3, n); cu[K["FB_CU_CLM_REF"]] += rng.uniform(-1.5, 1.5, n)
        cu[K["FB_CU_PHI_REF"]] += rng.uniform(-0.3, 0.3, n); cu[K["FB_CU_CHI_REF"]] += rng.uniform(-0.5, 0.5, n)
        if divergent == "trim":   # randomised trims, but the ONE mode pair of configs[3]
            w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
            w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
        else:
            w.cu = cu
    else:
        fb.init(sim, fb.TrimParameters())
        assert w.trim_success.all()
        w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
        w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))

    def barrier():
        if world > 1:
            dist.barrier()
        w.sync()
        if world > 1:
            torch.cuda.synchronize()

    block = args.x2_inner * DT
    warm = warm_until_stable(fb, C, w, lambda: fb.step(sim, block), cap=int(getattr(args, "x2_warm_cap", 30)))
    barrier()
    blocks = int(getattr(args, "x2_blocks", 12))
    fb._lib.check(fb.lib.fb_timing_begin_per_launch(w._h, blocks))
    t0 = time.perf_counter()
    for _ in range(blocks):
        fb.step(sim, block)
    barrier()
    el = time.perf_counter() - t0
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    stats = launch_stats(per_launch_ms(fb, C, w))
    stats["warmup_ms_per_launch"] = [round(v, 4) for v in warm]
    steps = int(round(block / DT)) * blocks
    bad = int((w.status != 0).sum())
    n_total, gather_ms = n, None
    if world > 1:
        t = torch.tensor([el, float(bad)], dtype=torch.float64, device=_cdev())
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        el = max(float(e[0].item()) for e in every); bad = int(sum(float(e[1].item()) for e in every))
        n_total = n * world
        torch.cuda.synchronize(); dist.barrier(); g0 = time.perf_counter()
        gathered = _gather_state(fb, x_dev, n_total)
        torch.cuda.synchronize(); gather_ms = (time.perf_counter() - g0) * 1e3
        assert gathered.shape == (fb.K["FB_X2_NX"], n_total)
        del gathered
    w.close()
    value = n_total * steps / el
    # roofline.achieved: algorithmic bytes of one launch / the kernel's AVERAGE launch duration over the timed region (HIP events), x ranks
    gbs = BYTES_PER_X2_STEP * n * args.x2_inner / (ms.value / max(nl.value, 1) * 1e-3) / 1e9 * world
    if divergent:
        return {"metric": "aircraft-steps/sec", "value": value, "unit": "aircraft-steps/s", "dtype": "f64",
                "config": {"workload": f"N={n} Cessna172Xv2 on lattice(3)'s randomised trims (EAS 35-55 m/s x h 200-3000 m x heading), wind as configs[3], every aircraft in its own "
                                       "(lon, lat) control-mode pair drawn from all 9 x 5 with its own references: the divergent counterpart of extra.x2",
                           "rk4_steps_per_launch": args.x2_inner, "terminated_aircraft": bad},
                **stats, "kernel_ms_mean": ms.value / max(nl.value, 1),
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}
    out = {"metric": "aircraft-steps/sec", "value": value, "unit": "aircraft-steps/s", "dtype": "f64", "n_gpus": world,
           "config": {"workload": (f"N={n_total} Cessna172Xv2 over {world} GPUs, {n} per GPU (BASELINE.json configs[3]" + (")" if world == 8 else f" at {world} of its 8 GPUs)")
                                   if world > 1 else f"N={n} Cessna172Xv2 (one GPU's share of configs[3]: 4 194 304 over 8)") +
                                  ", default trim, README example 2 scenario "
                                  "(wind, EAS + climb-rate and bank + sideslip modes), autopilot every 2 steps (Δt = 0.02), fp64, dt = 0.01",
                      "rk4_steps_per_launch": args.x2_inner, "terminated_aircraft": bad},
           "stepping_launches": int(nl.value), **stats, "kernel_ms_mean": ms.value / max(nl.value, 1), "stream_ms_per_rk4_step": ms.value / steps,
           "kernel": ("fbd::k_step_duo<0, true, false> (two waves per SIMD; the control update in two halves inside the launch)" if os.environ.get("FLIGHTBATCH_DUO", "1") != "0"
                      else "fbd::k_step_air<0, true, false, false>") + " + the ground-capable pass behind it",
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * world),
                        "note": "algorithmic bytes = 756 B per aircraft-step (SURVEY §8d)"}}
    if gather_ms is not None:
        out["gather_ms"] = gather_ms
    # fp64 VALU fraction and real HBM traffic of the Xv2 stepper, from the committed PMC summary — quoted only next to the code it was
    # measured on (source hash, like the headline's roofline_valu)
    prof = os.path.join(ROOT, "profiles", PROFILE_COUNTERS_X2)
    out["roofline_valu"] = None
    if os.path.exists(prof) and args.x2_inner == 50:
        import __graft_entry__ as _ge
        pj = json.load(open(prof))
        here = _ge.source_hash()
        if pj.get("source_hash") != here:
            out["roofline"]["traffic_note"] = f"profiles/{PROFILE_COUNTERS_X2} was collected on source hash {pj.get('source_hash')}, this tree is {here}: PMC figures withheld"
        else:
            kms = out["kernel_ms"]
            tf = pj["fp64_flops_per_launch"] * (n / pj["n"]) / (kms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                                    "flops_per_aircraft_step": pj["fp64_flops_per_aircraft_step"], "valu_insts_per_aircraft_step": pj["valu_insts_per_aircraft_step"],
                                    "valu_busy": pj.get("valu_busy"), "per_gpu": True,
                                    "source": "rocprofv3 SQ_INSTS_VALU_* counters, profiles/" + PROFILE_COUNTERS_X2 + " (source hash " + here + ")"}
            if pj.get("hbm_bytes_per_launch"):
                out["roofline"]["traffic"] = pj["hbm_bytes_per_launch"] * (n / pj["n"])
                out["roofline"]["traffic_note"] = "per launch of one GPU's share (50 steps, 25 control updates): the control-law record (94 rows read, ~53 written per update) dominates"
    return out


def extra_x2(fb, C, args, timed):
    """`extra.x2` of the bench line: the timing of time_x2 (taken BEFORE the CPU legs, on one GPU or by the multi-rank run) and a
    parity sample against the oracle."""
    out = timed
    # parity sample: 512 aircraft on randomised trims, every aircraft in its own pair of modes, 500 closed-loop steps vs the oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleX
    K = fb.K
    m = 512
    rng = np.random.default_rng(31)
    tp = fb.TrimParameters(h_e=rng.uniform(300.0, 2800.0, m), EAS=rng.uniform(38.0, 52.0, m), ψ_nb=rng.uniform(-np.pi, np.pi, m))
    gains = fb.ctl_gains.ctl_gains_blob()
    ws = fb.Cessna172Xv2World(m, gains=gains)
    sims = fb.Simulation(ws, dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=args.x2_inner)
    fb.init(sims, tp)
    orc = _oracle()
    X = OracleX(orc, gains)
    env = orc.default_env()
    o = X.trim_init(tp.pack(m), fb.TrimState(m), env, 2 * DT)
    o["status"] = np.zeros(m, np.int32); o["nstep"] = 0
    cu = ws.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, m); cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, m)
    cu[K["FB_CU_EAS_REF"]] += rng.uniform(-3, 3, m); cu[K["FB_CU_CLM_REF"]] += rng.uniform(-1.5, 1.5, m)
    cu[K["FB_CU_PHI_REF"]] += rng.uniform(-0.3, 0.3, m); cu[K["FB_CU_CHI_REF"]] += rng.uniform(-0.5, 0.5, m)
    ws.cu = cu
    perm = np.array([k if k < K["FB_X2_ACT"] else (27 + k - K["FB_X2_ACT"] if k < K["FB_X2_KIN"] else k - K["FB_NACT"]) for k in range(34)])
    o["cu"] = np.ascontiguousarray(o["cu"]); o["cu"][:] = cu
    o["x"][perm] = ws.x; o["cs"] = ws.cs; o["u"] = ws.u; o["ui"] = ws.ui; o["s"] = ws.s
    fb.step(sims, 5.0); ws.sync()
    X.step(o, env, DT, 2, 500, threads=min(orc.max_threads(), usable_cores()))
    same_status = bool(np.array_equal(ws.status, o["status"]))   # terminated aircraft are compared like the rest (frozen where the reference stops)
    xo = o["x"][perm]                                   # oracle rows (27 Sv0 rows, then the actuators) in the C ABI's reference order
    sc = np.ones_like(o["x"]); sc[:27] = state_floor(o["x"][:27])
    err = np.abs(ws.x - xo) / sc[perm]
    out["rel_err_vs_cpu"] = {"max_scaled_error": float(err.max()) if same_status else float("inf"), "status_words_equal": same_status,
                             "sample": f"{m} aircraft, randomised trims and control modes x 500 closed-loop RK4 steps, fp64 oracle (C++ port)",
                             "tolerance": 1e-6}
    ws.close()
    return out


# configs[4] computes in fp32 and BASELINE.md asks for it "reported with its own tolerance": the stated bounds on an fp32 Cessna172Sv0 after
# 10 s (1000 steps) against the fp64 oracle, in physical units (q_ew and h_e are integrated in fp64: their increments are below one fp32
# ulp), and on an fp32 Robot2D against its fp64 oracle as a scaled error. The same numbers are asserted by tests/test_gpu_f32.py and by
# tests/test_gpu_fullsize.py::test_config4_mixed_fp32_fleet_as_stated.
F32_TOLERANCE = {"rates_rad_s": 1e-5, "velocity_m_s": 2e-3, "altitude_m": 0.05, "q_wb": 2e-5, "q_ew": 1e-8, "engine_speed_rad_s": 0.05,
                 "robot2d_scaled": 2e-3}


def f32_abs_errors(x, xo, ok):
    d = np.abs(x - xo)[:, ok]
    return {"rates_rad_s": float(d[21:24].max()), "velocity_m_s": float(d[24:27].max()), "altitude_m": float(d[20].max()),
            "q_wb": float(d[12:16].max()), "q_ew": float(d[16:20].max()), "engine_speed_rad_s": float(d[9].max())}


def extra_fleet(fb, C, args):
    """BASELINE.json configs[4] on one GPU: N = 1 M vehicles, 50 % Cessna172Sv0 / 50 % Robot2D interleaved, fp32, dt = 0.01, Δt = 0.02."""
    n = N_TOTAL
    KC, KR = fb.K["FB_MODEL_C172S0"], fb.K["FB_MODEL_ROBOT2D"]
    types = np.where(np.arange(n) % 2 == 0, KC, KR)
    fleet = fb.MixedFleet(types, {KC: lambda m: fb.BatchedWorld(m, dtype="f32"), KR: lambda m: fb.Robot2DWorld(m, dtype="f32")})
    fleet.simulate(dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=50)
    nc = fleet.index[KC].size
    EAS, h, psi, _ = lattice(1)
    fleet.init({KC: fb.TrimParameters(EAS=EAS[:nc], h_e=h[:nc], ψ_nb=psi[:nc]), KR: fb.InitParameters()})
    assert fleet.worlds[KC].trim_success.all()
    rw = fleet.worlds[KR]
    u = rw.u; u[0] = 1; u[2] = 0.3; rw.u = u
    fleet.step(1.0); fleet.sync()
    T = 3.0
    cw = fleet.worlds[KC]
    fb._lib.check(fb.lib.fb_timing_begin_per_launch(cw._h, 6))
    t0 = time.perf_counter(); fleet.step(T); fleet.sync(); el = time.perf_counter() - t0
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(cw._h, C.byref(ms), C.byref(nl))
    stats = launch_stats(per_launch_ms(fb, C, cw))
    steps = int(round(T / DT))
    st = fleet.gather("status", fill=-1)
    value = n * steps / el
    # algorithmic bytes per vehicle-step (the position block of the aircraft stays fp64): see DESIGN.md §5
    bytes_c, bytes_r = 2 * (27 * 4 + 5 * 4) + 8, 44.0
    gbs = (bytes_c * nc + bytes_r * (n - nc)) * steps / el / 1e9
    out = {"metric": "vehicle-steps/sec", "value": value, "unit": "vehicle-steps/s", "dtype": "f32",
           "config": {"workload": f"mixed fleet N={n}: 50% Cessna172Sv0 (fp32 airborne stepper, randomised trims) / 50% Robot2D (fp32), interleaved input order, "
                                  "packed by model onto two HIP streams, dt=0.01, Δt=0.02 (BASELINE.json configs[4] on one GPU)", "terminated": int((st != 0).sum())},
           **stats, "kernel_ms_mean": ms.value / max(nl.value, 1), "kernel": "fbf::k_step_f32 (50 steps of 524 288 aircraft per launch; its stream also carries Robot2D's waits)",
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "note": f"algorithmic bytes: {bytes_c} B per fp32 aircraft-step, {bytes_r:.0f} B per fp32 robot-step"}}
    fleet.close()
    return out


def fleet_parity(fb, out):
    """parity sample of the fp32 aircraft kernel against the fp64 oracle (physical units: fp32 cannot meet 1e-6); after the timed legs"""
    orc = _oracle()
    nc = N_TOTAL // 2
    EAS, h, psi, _ = lattice(1)
    m = 1024
    sel = np.arange(0, nc, nc // m)[:m]
    w = fb.BatchedWorld(m, dtype="f32")
    fb.f_init(w, fb.TrimParameters(EAS=EAS[sel], h_e=h[sel], ψ_nb=psi[sel]))
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
    fb.step(sim, 10.0); w.sync()
    xo, so, sto = orc.step(x0, u0, ui0, s0, orc.default_env(), DT, 1000, threads=min(orc.max_threads(), usable_cores()))
    ok = (sto == 0) & (w.status == 0)
    absolute = f32_abs_errors(w.x, xo, ok)
    out["rel_err_vs_cpu"] = {"max_scaled_error": float(scaled_error(w.x, xo)[:, ok].max()),
                             "tolerance": {k: v for k, v in F32_TOLERANCE.items() if k in absolute},
                             "within_tolerance": bool(all(absolute[k] < F32_TOLERANCE[k] for k in absolute)),
                             "absolute": absolute,
                             "sample": f"{int(ok.sum())} fp32 aircraft of the fleet's lattice x 1000 RK4 steps vs the fp64 oracle (C++ port); "
                                       "tolerance: the stated fp32 bounds, physical units (tests/test_gpu_f32.py, test_config4_mixed_fp32_fleet_as_stated)"}
    w.close()
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this process has not touched the GPU)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inner", type=int, default=50, help="RK4 steps fused per launch (= per contract step)")
    ap.add_argument("--x2-inner", type=int, default=50, help="RK4 steps per launch of the Cessna172Xv2 stepper (control laws run inside the launch)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="N > 1: strong (default) = the same 1 048 576 aircraft sharded over the ranks (BASELINE's whole-node figure); "
                         "weak = 1 048 576 aircraft per rank. The other mode is measured too and attached as an extra key")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[3] / configs[4] legs (1-GPU runs)")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="f64: the parity path (default, BASELINE.json's metric). f32: the fp32 airborne stepper (config 5's dtype)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if REHEARSAL else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")

    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() == 0:      # (does not initialise the GPU)
        raise SystemExit("bench.py needs a GPU: libflightbatch has no CPU path")
    if world > 1:   # the process group first: nothing of this process has touched a device before RCCL binds this rank to its GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSAL:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import ctypes as C
    import flightbatch as fb

    scaling = args.scaling or "strong"   # (N > 1 only; BASELINE.json: "whole node at N = 1M")
    results = {}
    for mode in ((scaling, "weak" if scaling == "strong" else "strong") if world > 1 else ("weak",)):
        if mode == "strong":
            EAS, h, psi, cell = lattice(0)
            lo, hi = fb.sharding.shard_range(N_TOTAL, rank, world)
            EAS, h, psi, cell = EAS[lo:hi], h[lo:hi], psi[lo:hi], cell[lo:hi]
        else:
            EAS, h, psi, cell = lattice(rank)
        m = time_c172s0(fb, torch, dist, C, EAS, h, psi, args, local_rank, world)
        m["cell"] = cell
        check_valid(m, mode)
        results[mode] = m
    head = results[scaling if world > 1 else "weak"]

    line = None
    if rank == 0:
        n = head["n"]
        kernel_ms = head["kernel_ms"]
        value = float(head["n_total"]) * args.inner * args.steps / head["elapsed"]
        units_per_launch = float(n) * args.inner
        bytes_per_unit = BYTES_PER_AIRCRAFT_STEP if args.dtype == "f64" else BYTES_PER_AIRCRAFT_STEP_F32
        achieved_gbs = bytes_per_unit * units_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        counters_note = None
        prof = os.path.join(ROOT, "profiles", PROFILE_COUNTERS)
        duo_on = os.environ.get("FLIGHTBATCH_DUO", "1") != "0"
        if not os.path.exists(prof):
            counters_note = "profiles/" + PROFILE_COUNTERS + " is absent: no PMC-derived figures"
        elif args.dtype != "f64" or not duo_on:
            counters_note = "the committed PMC counters describe the fp64 k_step_duo kernel, not the one this run timed"
        else:
            pj = json.load(open(prof))
            # the PMC figures are quoted only next to the code they were measured on: tools/summarize_profile.py stores a hash of
            # flight.jl_amd/csrc/* + the compiler flags (__graft_entry__.source_hash) and a stale file is refused, not silently reused
            import __graft_entry__ as _ge
            here = _ge.source_hash()
            if pj.get("source_hash") != here:
                counters_note = (f"profiles/{PROFILE_COUNTERS} was collected on source hash {pj.get('source_hash')}, this tree is {here}: "
                                 "traffic and roofline_valu withheld (re-run tools/collect_profile.sh)")
            elif pj.get("inner") == args.inner and pj.get("n"):
                scale = n / pj["n"]                        # counters are per launch of pj["n"] aircraft; traffic and flops are proportional to n
                traffic = pj.get("hbm_bytes_per_launch") * scale if pj.get("hbm_bytes_per_launch") else None
                if pj.get("fp64_flops_per_launch"):
                    tf = pj["fp64_flops_per_launch"] * scale / (kernel_ms * 1e-3) / 1e12
                    valu = {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / FP64_VALU_PEAK_TFLOPS, "flops_per_aircraft_step": pj["fp64_flops_per_launch"] / (pj["n"] * pj["inner"]),
                            "valu_insts_per_aircraft_step": pj.get("valu_insts_per_aircraft_step"),
                            "valu_busy": pj.get("valu_busy"),
                            "source": "rocprofv3 SQ_INSTS_VALU_* counters, profiles/" + PROFILE_COUNTERS + " (source hash " + here + ")"}
        line = {
            # N > 1: the metric NAME says which of the two multi-GPU figures `value` is, so that a reader (or a driver) cannot set the weak-scaling
            # number (N x 1 048 576 aircraft) against BASELINE's whole-node figure at N = 1 M, which rides along as `strong_scaling`
            "metric": "aircraft-steps/sec" + ("" if world == 1 else (" (weak scaling: 1 048 576 aircraft PER GPU)" if scaling == "weak" else " (whole node at N = 1 M: strong scaling)")),
            "value": value, "unit": "aircraft-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["elapsed"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"N={head['n_total']} Cessna172Sv0 " + (f"over {world} GPUs (contiguous shards of {n}), " if world > 1 and scaling == "strong" else
                                                                            (f"({n} on each of {world} GPUs: configs[2] PER GPU — weak scaling, NOT BASELINE's whole-node N = 1 M figure: see strong_scaling), " if world > 1 else "")) +
                                    "randomised trim (EAS 35-55 m/s x h 200-3000 m x heading lattice, LCG-permuted), " +
                                    ("fp64" if args.dtype == "f64" else "fp32 airborne stepper (positions integrated in fp64)") + ", dt=0.01 (BASELINE.json configs[2])"),
                       "aircraft_total": head["n_total"], "aircraft_per_gpu": n, "rk4_steps_per_launch": args.inner, "rk4_steps_per_contract_step": args.inner, "dt": DT,
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "trim_success_fraction": head["trim_ok"] / head["n_total"], "trim_seconds": head["trim_s"], "terminated_aircraft": head["status_bad"],
                       "checked_on_all_aircraft": "status == 0, fuel strictly decreasing, | |q| - 1 | <= " + ("1e-8" if args.dtype == "f64" else "5e-7") + " (q_wb, q_ew), all states finite"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms_median": head["launch_stats"]["kernel_ms"], "kernel_ms_min": head["launch_stats"]["kernel_ms_min"],
                         "kernel_ms_max": head["launch_stats"]["kernel_ms_max"], "kernel": (("fbd::k_step_duo<0, false, false>" if os.environ.get("FLIGHTBATCH_DUO", "1") != "0" else "fbd::k_step_air<0, false, false, false>") if args.dtype == "f64" else "fbf::k_step_f32") + " (+ the ground-capable pass behind it)", "kernel_ms": kernel_ms,
                         "note": f"algorithmic bytes = {bytes_per_unit:.0f} B per aircraft-step (SURVEY §8d" + ("" if args.dtype == "f64" else ", fp32 rows; q_ew / h_e stay fp64") + ") x N x inner steps per launch; the fused "
                                 "stepper is " + ("fp64" if args.dtype == "f64" else "fp32") + "-VALU-bound, see roofline_valu and DESIGN.md"},
            "roofline_valu": valu,
        }
        if counters_note:
            line["roofline"]["traffic_note"] = counters_note
        if world > 1:
            per = head["elapsed_per_rank"]
            line["gather_ms"] = head["gather_ms"]
            line["ms_per_step_ranks"] = {"min": min(per) / args.steps * 1e3, "max": max(per) / args.steps * 1e3}
            other = results["weak" if scaling == "strong" else "strong"]
            line[("weak" if scaling == "strong" else "strong") + "_scaling"] = {
                "value": float(other["n_total"]) * args.inner * args.steps / other["elapsed"], "unit": "aircraft-steps/s",
                "aircraft_total": other["n_total"], "aircraft_per_gpu": other["n"], "ms_per_step": other["elapsed"] / args.steps * 1e3,
                "kernel_ms": other["kernel_ms"], "gather_ms": other["gather_ms"]}
    # every GPU-timed leg runs BEFORE the CPU legs (the ~20 s of oracle work leave the GPU idle: round 5's driver run timed configs[3]
    # behind them with two warm-up launches and got 15.1 ms where 9.85 is the kernel's time — profiles/r06_x2_repro.txt)
    x2_timed = x2_lat = fleet_timed = None
    extras = not args.no_extra and args.dtype == "f64"
    if extras:
        x2_timed = time_x2(fb, torch, dist, C, args, local_rank, world)   # configs[3]: one GPU's share, or this node's split with every rank taking part
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif extras:
        x2_lat = time_x2(fb, None, None, C, args, divergent=True)
        fleet_timed = extra_fleet(fb, C, args)
    if rank == 0:
        # the CPU baseline and the parity sample ride on every line, multi-GPU ones included: rank 0, on its own shard, after the
        # process group is gone (nothing of this is inside a timed region)
        if not args.no_cpu_baseline:
            x0, s0, u0, ui0 = head["ic"]
            line["cpu_baseline"] = cpu_baseline(x0, u0, ui0, s0)
            line["rel_err_vs_cpu"] = parity_sample(fb, x0, u0, ui0, s0, head["cell"], args.dtype)
            if world > 1:
                line["cpu_baseline"]["sample"] += f"; timed on rank 0's host share while the other {world - 1} ranks idle"
                line["rel_err_vs_cpu"]["sample"] += " (rank 0's shard)"
        if not args.no_extra and args.dtype == "f64":
            head.pop("ic", None)
            line["extra"] = {"x2": extra_x2(fb, C, args, x2_timed)}
            if world == 1:
                line["extra"]["x2_lattice"] = x2_lat
                x2_lat["vs_identical_aircraft"] = x2_lat["kernel_ms"] / x2_timed["kernel_ms"]
                line["extra"]["fleet"] = fleet_parity(fb, fleet_timed)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
