#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched 6-DOF hot path (BASELINE.json: aircraft-steps/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — 1 048 576 Cessna172Sv0 per GPU, randomised trim on the
32 x 32 x 1024 (EAS x altitude x heading) lattice of SURVEY.md §8(d), order permuted by a fixed LCG
(seed 172), fp64, dt = 0.01. Synthetic: no recorded data exists for this path.

One "step" of the contract = one pass of the hot path over the batch = ONE launch of the fused stepping
kernel advancing every aircraft by `--inner` RK4 steps (default 50, i.e. 0.5 s of flight), including all
RK stages, the output evaluation at the new state and f_step!. Trim, table generation and upload are
outside the timed region; state is resident in HBM when timing starts.

Multi-GPU: the batch shards embarrassingly (aircraft are independent): every rank owns its own
1 048 576 aircraft (weak scaling), no data-path collective; one RCCL all-gather of the final states
collects the trajectory endpoint after the timed region (reported as gather_ms, not part of `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd"))

N_PER_GPU = 1 << 20
DT = 0.01
BYTES_PER_AIRCRAFT_STEP = 440.0   # 2 * Nx * 8 B + 8 B of flags, C172Sv0 fp64
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X vector fp64 peak (spec): 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PROFILE_COUNTERS = "r01p_counters.json"   # rocprofv3 PMC summary of the CURRENT k_step (tools/collect_profile.sh)


def lattice(rank: int):
    """config 3 of SURVEY.md §8(d): EAS_i = 35 + 20 i/31, h_j = 200 + 2800 j/31, ψ_k = -π + 2π k/1024,
    aircraft order permuted by a fixed LCG so neighbouring lanes hold different table cells."""
    n = N_PER_GPU
    idx = np.arange(n, dtype=np.uint64)
    a, c = np.uint64(1664525), np.uint64(1013904223)       # full-period LCG modulo 2^20 (a ≡ 1 mod 4, c odd)
    perm = (a * idx + c + np.uint64(172 + 7919 * rank)) & np.uint64(n - 1)
    i = (perm >> np.uint64(15)) & np.uint64(31)
    j = (perm >> np.uint64(10)) & np.uint64(31)
    k = perm & np.uint64(1023)
    EAS = 35.0 + 20.0 * i.astype(np.float64) / 31.0
    h = 200.0 + 2800.0 * j.astype(np.float64) / 31.0
    psi = -np.pi + 2 * np.pi * k.astype(np.float64) / 1024.0
    return EAS, h, psi


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


def cpu_baseline(x0, u0, ui0, s0, budget_s=15.0):
    """The CPU oracle (a C++ port of the reference path) timed on this box's host cores on a bounded
    sample of the same workload. Reference-like arithmetic: 6 RHS evaluations per step, as
    OrdinaryDiffEq's RK4 does with Flight.jl's state-modifying step callback."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    orc = Oracle()
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
                      f"(reference-like), {el:.1f} s",
            "single_core_value": m1 * 100 / el1}


def parity_sample(fb, x0, u0, ui0, s0, dtype, m=256, nsteps=1000):
    """The second half of BASELINE.json's metric ("fp64 rel-err vs CPU"): the first m aircraft of the benchmark batch stepped
    nsteps times on the GPU and by the CPU oracle from the same initial condition; max over aircraft and states of
    |x_gpu - x_cpu| / max(|x_cpu|, floor) with the floors of SURVEY.md §8(d) (quaternions 1, rates 1e-3 rad/s, ...)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    orc = Oracle()
    sel = slice(0, m)
    xs, us, uis, ss = (np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(u0[:, sel]), np.ascontiguousarray(ui0[sel]), np.ascontiguousarray(s0[:, sel]))
    w = fb.BatchedWorld(m, dtype=dtype)
    w.set_state(xs, ss); w.u = us; w.ui = uis
    sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
    fb.step(sim, nsteps * DT); w.sync()
    xo, so, sto = orc.step(xs, us, uis, ss, orc.default_env(), DT, nsteps)
    ok = (sto == 0) & (w.status == 0)
    sc = np.maximum(np.abs(xo), 1e-3)
    sc[12:20] = 1.0; sc[2:8] = 1.0; sc[10:12] = 1.0; sc[0:2] = np.maximum(np.abs(xo[0:2]), 1e-2); sc[24:27] = np.maximum(np.abs(xo[24:27]), 1.0)
    err = float((np.abs(w.x - xo) / sc)[:, ok].max())
    w.close()
    return {"max_scaled_error": err, "sample": f"{int(ok.sum())} aircraft x {nsteps} RK4 steps vs the CPU oracle (fp64)", "tolerance": 1e-6 if dtype == "f64" else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inner", type=int, default=50, help="RK4 steps fused per launch (= per contract step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("c172s0", "c172x2"), default="c172s0",
                    help="c172s0: BASELINE.json configs[2] (default, the headline metric). c172x2: configs[3] — 524 288 Cessna172Xv2 per GPU "
                         "with the autopilot at Δt = 0.02 in the README-example-2 scenario (fp64 only)")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="f64: the parity path (default, BASELINE.json's metric). f32: the fp32 airborne stepper (config 5's dtype)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libflightbatch has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import ctypes as C
    import flightbatch as fb

    x2 = args.workload == "c172x2"
    if x2 and args.dtype != "f64":
        raise SystemExit("--workload c172x2 is fp64 only")
    n = N_PER_GPU // 2 if x2 else N_PER_GPU
    nrows = fb.K["FB_X2_NX"] if x2 else fb.K["FB_NX"]
    w = fb.Cessna172Xv2World(n, device=local_rank) if x2 else fb.BatchedWorld(n, device=local_rank, dtype=args.dtype)
    # the state lives in a torch tensor so that RCCL can gather it without a host round trip
    x_dev = torch.zeros((nrows, n), dtype=torch.float64, device="cuda")
    s_dev = torch.zeros((fb.K["FB_NS"], n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fb._lib.check(fb.lib.fb_attach_state(w._h, C.c_void_p(x_dev.data_ptr()), C.c_void_p(s_dev.data_ptr())))
    if x2:   # README example 2: default trim, wind N = 1, E = 0.5 m/s, EAS + climb rate (2 m/s), bank + sideslip (30 deg)
        w.set_params(wind_ned=(1.0, 0.5, 0.0))
        sim = fb.Simulation(w, dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=2)
        fb.init(sim, fb.TrimParameters())
        w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
        w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
    else:
        EAS, h, psi = lattice(rank)
        fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
        sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=args.inner)
    trim_ok = float(w.trim_success.mean())
    if not x2:
        x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui

    def barrier():
        if world > 1:
            dist.barrier()
        w.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        fb.step(sim, args.inner * DT)
    barrier()
    fb.lib.fb_timing_begin(w._h)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fb.step(sim, args.inner * DT)
    barrier()
    elapsed = time.perf_counter() - t0
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    kernel_ms = ms.value / max(nl.value, 1)   # average launch duration of k_step, HIP events on its stream

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # trajectory collection: ONE RCCL all-gather of the final states over xGMI (north star)
        torch.cuda.synchronize(); g0 = time.perf_counter()
        gathered = fb.sharding.all_gather_state(x_dev, n * world)
        torch.cuda.synchronize(); gather_ms = (time.perf_counter() - g0) * 1e3
        assert gathered.shape == (nrows, n * world)
    else:
        gather_ms = None

    status_bad = int((w.status != 0).sum())
    if status_bad > n // 1000:
        raise SystemExit(f"INVALID RUN: {status_bad} of {n} aircraft terminated (status != 0) — they do no work; refusing to report a throughput")
    total_units = float(n) * world * args.inner * args.steps
    value = total_units / elapsed

    if rank == 0:
        units_per_launch = float(n) * (2 if x2 else args.inner)   # c172x2: one launch group = one control period = 2 RK4 steps
        bytes_per_unit = 756.0 if x2 else BYTES_PER_AIRCRAFT_STEP
        achieved_gbs = bytes_per_unit * units_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        prof = os.path.join(ROOT, "profiles", PROFILE_COUNTERS)
        if os.path.exists(prof) and args.dtype == "f64" and not x2:   # the committed counters describe the fp64 kernel
            pj = json.load(open(prof))
            if pj.get("n") == n and pj.get("inner") == args.inner:
                traffic = pj.get("hbm_bytes_per_launch")
                if pj.get("fp64_flops_per_launch"):
                    tf = pj["fp64_flops_per_launch"] / (kernel_ms * 1e-3) / 1e12
                    valu = {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / FP64_VALU_PEAK_TFLOPS, "flops_per_aircraft_step": pj["fp64_flops_per_launch"] / units_per_launch,
                            "source": "rocprofv3 SQ_INSTS_VALU_* counters, profiles/" + PROFILE_COUNTERS}
        line = {
            "metric": "aircraft-steps/sec", "value": value, "unit": "aircraft-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("N=524288 Cessna172Xv2 per GPU, default trim, README example 2 scenario (wind, EAS + climb-rate and bank + sideslip "
                                    "modes), autopilot every 2 steps, fp64, dt=0.01 (BASELINE.json configs[3])") if x2 else
                                   "N=1048576 Cessna172Sv0 per GPU, randomised trim (EAS 35-55 m/s x h 200-3000 m x heading lattice, "
                                   "LCG-permuted), " + ("fp64" if args.dtype == "f64" else "fp32 airborne stepper (positions integrated in fp64)") + ", dt=0.01 (BASELINE.json configs[2])",
                       "aircraft_per_gpu": n, "rk4_steps_per_launch": (2 if x2 else args.inner), "rk4_steps_per_contract_step": args.inner, "dt": DT, "parallelism": f"batch-sharded x{world}",
                       "trim_success_fraction": trim_ok, "terminated_aircraft": status_bad},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": ("fbd::k_step_air<0, false>" if args.dtype == "f64" else "fbf::k_step_f32") + " (+ the ground-capable pass behind it)", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = 440 B per aircraft-step (SURVEY §8d) x N x inner steps per launch; the fused "
                                 "stepper is fp64-VALU-bound, see roofline_valu and DESIGN.md"},
            "roofline_valu": valu,
        }
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
        if x2:
            line["roofline"]["kernel"] = "fbd::k_step<true, 0, false> + k_x2_ctl per control period"
            line["roofline"]["note"] = "algorithmic bytes = 756 B per aircraft-step (SURVEY §8d); kernel_ms = one control period: the 2-step stepping launch, its ground pass and the control-law kernel"
        if not args.no_cpu_baseline and world == 1 and not x2:
            line["cpu_baseline"] = cpu_baseline(x0, u0, ui0, s0)
            line["rel_err_vs_cpu"] = parity_sample(fb, x0, u0, ui0, s0, args.dtype)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    w.close()


if __name__ == "__main__":
    main()
