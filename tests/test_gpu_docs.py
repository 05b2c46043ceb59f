"""The Python snippets of INTEGRATION.md ("Executed today") are executed — at a reduced batch size — so that the document cannot
drift from the package (a snippet that raises at construction was shipped once). The docstring mappings `Simulation(world, dt=0.01,
Δt=0.02)` of flightbatch/c172x.py and robot2d.py are constructed too, at a size where the reference's default log would not fit."""
import os
import re
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_python_snippet_runs(fb):
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    tail = text[text.index("## Executed today"):]
    code = re.search(r"```python\n(.*?)```", tail, flags=re.S).group(1)
    assert "1_048_576" in code and "524_288" in code
    code = code.replace("1_048_576", "4096").replace("524_288", "2048")
    ns = {"np": np}
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # the documented calls must not even warn (log sized for t_end)
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    assert ns["x"].shape == (27, 4096) and (ns["world"].status == 0).all()
    t, data = ns["ts"].t, ns["ts"]
    assert len(t) == 31 and abs(t[-1] - 30.0) < 1e-9 and (ns["xw"].status == 0).all()
    ns["world"].close(); ns["xw"].close()


def test_reference_default_simulation_constructs_with_a_warning_for_a_large_batch(fb):
    """Simulation(mdl; dt = 0.01) with the reference's defaults (save_on, t_end = 10000, every step): for a batch the 8 GB device log
    cannot hold that; the Simulation is built with the log capped, a warning says how far it reaches, and stepping within it works."""
    w = fb.BatchedWorld(65536)
    with pytest.warns(UserWarning, match="default 8 GB device log holds"):
        sim = fb.Simulation(w, dt=0.01)
    fb.init(sim, fb.TrimParameters())
    fb.step(sim, 0.05); w.sync()
    assert (w.status == 0).all() and len(fb.TimeSeries(sim).t) == 6
    w.close()
    r = fb.Robot2DWorld(1 << 20)
    with pytest.warns(UserWarning):
        fb.Simulation(r, dt=0.01, Δt=0.02)
    r.close()
