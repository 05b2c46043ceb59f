"""Run-to-run determinism of the scripted crosswind landing (examples/crosswind_landing.py): the same scenario three times per horizon."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "examples")); sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import crosswind_landing as cl
for T in [float(a) for a in sys.argv[1:]] or [0.1, 1.0, 10.0, 60.0, 150.0]:
    outs = [cl.run(n=64, seed=0, t_end=T) for _ in range(3)]
    d = [float(np.nanmax(np.abs(outs[0]["x"] - outs[k]["x"]))) for k in (1, 2)]
    dc = [float(np.nanmax(np.abs(outs[0]["cs"] - outs[k]["cs"]))) for k in (1, 2)]
    rows = np.nonzero(np.nanmax(np.abs(outs[0]["x"] - outs[1]["x"]), axis=1) > 0)[0]
    print(f"t_end {T:6.1f}: max|dx| vs run 0: {d}, max|dcs| {dc}, terminated {[int((o['status'] != 0).sum()) for o in outs]}, phases {[np.bincount(o['phase'], minlength=4).tolist() for o in outs]}, differing x rows {rows.tolist()[:12]}", flush=True)
