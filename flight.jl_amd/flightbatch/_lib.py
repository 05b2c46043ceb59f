"""ctypes binding of libflightbatch.so (include/flightbatch.h). There is no fallback: if the shared
library is missing the import fails loudly, and without a HIP device ``fb_create`` raises."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLIGHTBATCH_LIB") or os.path.join(os.path.dirname(_HERE), "libflightbatch.so")  # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "flightbatch.h")


class FlightBatchError(RuntimeError):
    pass


class fb_params(C.Structure):
    _fields_ = [("dt", C.c_double), ("periodic_n", C.c_int32), ("surface", C.c_int32), ("T_sl", C.c_double),
                ("p_sl", C.c_double), ("wind_ned", C.c_double * 3), ("h_terrain", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise FlightBatchError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). flightbatch has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    H, D, I32, I64, VP = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.c_void_p
    sig = {
        "fb_create": ([C.c_int32, C.c_int32, C.c_int32, I64, C.c_int32, C.POINTER(H)], C.c_int32),
        "fb_destroy": ([H], C.c_int32),
        "fb_size": ([H], I64),
        "fb_dims": ([H, I32, I32, I32, I32], C.c_int32),
        "fb_f_init": ([H, D, C.c_int32], C.c_int32),
        "fb_set_stream": ([H, VP], C.c_int32),
        "fb_attach_state": ([H, VP, VP], C.c_int32),
        "fb_set_table": ([H, C.c_int32, VP, C.POINTER(I64), C.c_int32], C.c_int32),
        "fb_set_params": ([H, C.POINTER(fb_params)], C.c_int32),
        "fb_get_params": ([H, C.POINTER(fb_params)], C.c_int32),
        "fb_set_env": ([H, D], C.c_int32),
        "fb_get_env": ([H, D], C.c_int32),
        "fb_set_state": ([H, D, I32], C.c_int32),
        "fb_assign_state": ([H, D, I32], C.c_int32),
        "fb_get_state": ([H, D, I32], C.c_int32),
        "fb_set_inputs": ([H, D, I32], C.c_int32),
        "fb_get_inputs": ([H, D, I32], C.c_int32),
        "fb_trim": ([H, D, D, I32, D], C.c_int32),
        "fb_f_ode": ([H, D], C.c_int32),
        "fb_f_step": ([H], C.c_int32),
        "fb_f_periodic": ([H], C.c_int32),
        "fb_get_outputs": ([H, D], C.c_int32),
        "fb_get_output_fields": ([H, C.c_uint32, D], C.c_int32),
        "fb_set_ctl_inputs": ([H, D], C.c_int32),
        "fb_get_ctl_inputs": ([H, D], C.c_int32),
        "fb_set_ctl_state": ([H, D], C.c_int32),
        "fb_get_ctl_state": ([H, D], C.c_int32),
        "fb_step": ([H, I64], C.c_int32),
        "fb_set_steps_per_launch": ([H, C.c_int32], C.c_int32),
        "fb_sync": ([H], C.c_int32),
        "fb_time": ([H], C.c_double),
        "fb_status": ([H, I32], C.c_int32),
        "fb_get_termination": ([H, C.POINTER(I64), I32], C.c_int32),
        "fb_get_step_count": ([H, C.POINTER(I64)], C.c_int32),
        "fb_set_step_count": ([H, I64, C.c_double], C.c_int32),
        "fb_set_status": ([H, I32], C.c_int32),
        "fb_set_termination": ([H, C.POINTER(I64), I32], C.c_int32),
        "fb_log_configure": ([H, I64, I64, I32, C.c_int32], C.c_int32),
        "fb_log_clear": ([H], C.c_int32),
        "fb_log_record": ([H], C.c_int32),
        "fb_log_count": ([H, C.POINTER(I64)], C.c_int32),
        "fb_log_read": ([H, I64, I64, D, D], C.c_int32),
        "fb_comm_unique_id": ([C.c_char_p], C.c_int32),
        "fb_comm_init": ([H, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(VP)], C.c_int32),
        "fb_comm_shard_sizes": ([VP, C.POINTER(I64), C.POINTER(I64)], C.c_int32),
        "fb_gather_state": ([H, VP, VP], C.c_int32),
        "fb_comm_destroy": ([VP], C.c_int32),
        "fb_has_env": ([H], C.c_int32),
        "fb_scenario_configure": ([H, C.c_int32], C.c_int32),
        "fb_scenario_set_params": ([H, D], C.c_int32),
        "fb_scenario_get_params": ([H, D], C.c_int32),
        "fb_scenario_get_state": ([H, I32, C.POINTER(I64), D], C.c_int32),
        "fb_scenario_set_state": ([H, I32, C.POINTER(I64), D], C.c_int32),
        "fb_timing_begin": ([H], C.c_int32),
        "fb_timing_end": ([H, C.POINTER(C.c_float), C.POINTER(I64)], C.c_int32),
        "fb_timing_begin_per_launch": ([H, I64], C.c_int32),
        "fb_timing_launches": ([H, C.POINTER(C.c_float), I64, C.POINTER(I64)], C.c_int32),
        "fb_last_error": ([], C.c_char_p),
        "fb_version": ([], C.c_char_p),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = ABI symbol missing
        fn.argtypes = args
        fn.restype = res
    return lib, sorted(sig)


lib, EXPORTED = _load()


def check(rc: int):
    if rc != 0:
        raise FlightBatchError(lib.fb_last_error().decode() or f"libflightbatch error {rc}")


def header_constants() -> dict:
    """Parse the enum constants of include/flightbatch.h (single source of truth for layouts)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out: dict = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [t.strip() for t in item.split("=", 1)]
                val = int(eval(expr, {}, out))
            else:
                name = item
                val += 1
            out[name] = val
    return out


K = header_constants()
