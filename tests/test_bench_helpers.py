"""CPU checks of bench.py's host-side helpers: the benchmark lattice, the stratified parity sample, the per-launch statistics the bench line
carries since round 6, and the command-line defaults the driver relies on."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_lattice_is_a_permutation_of_the_32_x_32_x_1024_grid():
    b = _bench()
    EAS, h, psi, cell = b.lattice(0)
    assert EAS.size == 1 << 20 and EAS.min() == 35.0 and EAS.max() == 55.0 and h.min() == 200.0 and h.max() == 3000.0
    assert np.unique(cell).size == 1024 and np.bincount(cell).min() == 1024 and np.bincount(cell).max() == 1024     # every (EAS, h) cell 1024 times
    key = (cell.astype(np.int64) << 10) | np.round((psi + np.pi) / (2 * np.pi) * 1024).astype(np.int64)
    assert np.unique(key).size == 1 << 20                                                                        # every lattice point exactly once
    assert (np.diff(cell[:64]) != 0).any(), "neighbouring lanes sit in different table cells"
    EAS1 = b.lattice(1)[0]
    assert not np.array_equal(EAS, EAS1) and np.array_equal(np.sort(EAS), np.sort(EAS1))                          # another rank: the same points, another order


def test_stratified_sample_covers_every_cell():
    b = _bench()
    cell = b.lattice(0)[3]
    sel = b.stratified_sample(cell)
    assert sel.size == 4096 and np.unique(sel).size == 4096 and np.all(np.diff(sel) > 0)
    assert np.unique(cell[sel]).size == 1024 and np.bincount(cell[sel]).min() == 4
    assert sel.max() > (1 << 20) * 0.99 and sel.min() < (1 << 20) * 0.01                                          # drawn over the whole permuted order


def test_launch_stats_reports_the_median_and_keeps_every_launch():
    b = _bench()
    ms = [10.9, 10.1, 9.85, 9.84, 9.86, 40.0, 9.83]          # a ramp behind the trim and one stalled launch
    s = b.launch_stats(ms)
    assert s["kernel_ms"] == 9.86 and s["kernel_ms_min"] == 9.83 and s["kernel_ms_max"] == 40.0 and s["launches_timed"] == 7
    assert s["kernel_ms_per_launch"] == ms
    assert abs(np.mean(ms) - s["kernel_ms"]) > 4.0, "what the median is for: one stall moves the mean by 4 ms and the median not at all"


def test_command_line_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True).stdout
    assert "--gpus" in out and "--steps" in out and "--warmup" in out and "strong (default)" in out
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'scaling = args.scaling or "strong"' in src and 'PROFILE_COUNTERS = "r06_counters.json"' in src
