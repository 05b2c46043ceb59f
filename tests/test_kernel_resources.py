"""Register, scratch and LDS budgets of the built kernels, read from the code object inside the library (no GPU needed): a guard against the
kind of regression round 6 found twice — row addresses hoisted out of a loop and spilled (948 B of scratch in the ground-capable Cessna172Xv2
pass, 1 168 B in k_trim), which costs nothing in correctness and a great deal in time (profiles/r06_ground_launch_anatomy.txt, r06_ab_trim.txt)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flight.jl_amd", "libflightbatch.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.fail("libflightbatch.so is not built: python -c 'import __graft_entry__ as g; g.build()'")
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not found under {LLVM}")
    d = tmp_path_factory.mktemp("co")
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(d, "discard.so")], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    recs = re.findall(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", notes, re.S)
    filt = shutil.which("c++filt")
    out = {}
    for lds, name, scratch, vgpr in recs:
        dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
        out[re.sub(r"\(.*", "", dem).replace("void ", "")] = dict(lds=int(lds), scratch=int(scratch), vgpr=int(vgpr))
    return out


def test_every_stepping_instance_is_in_the_library(kernels):
    for kin in (0, 1, 2):
        for x in ("false", "true"):
            for env in ("false", "true"):
                assert f"fbd::k_step_duo<{kin}, {x}, {env}>" in kernels
                for gnd in ("false", "true"):
                    assert f"fbd::k_step_air<{kin}, {x}, {gnd}, {env}>" in kernels
    assert "fbf::k_step_f32" in kernels and "fbd::k_trim<false>" in kernels and "fbd::k_trim<true>" in kernels and "fbd::k_scenario<0>" in kernels


def test_budgets(kernels):
    for name, k in kernels.items():
        assert k["lds"] <= 160 * 1024, (name, k)
        m = re.match(r"fbd::k_step_duo<(\d), (true|false), (true|false)>", name)
        if m:   # two waves per SIMD: 256 registers; no scratch to speak of (a few launch-level values in the environment-row instances)
            assert k["vgpr"] <= 256, (name, k)
            assert k["scratch"] <= (48 if m.group(3) == "true" else 8), (name, k)
            continue
        m = re.match(r"fbd::k_step_air<(\d), (true|false), (true|false), (true|false)>", name)
        if m:
            x, gnd = m.group(2) == "true", m.group(3) == "true"
            limit = 0 if not x else (400 if gnd else 64)   # Cessna172Sv0: none; Xv2 ground-capable: 0.3-0.4 KB, none of it per evaluation but 1-3 reloads
            assert k["scratch"] <= limit, (name, k, limit)
            continue
        if name.startswith("fbd::k_trim"):
            assert k["scratch"] <= 512, (name, k)   # 1 168 / 1 280 B up to round 6
            continue
        assert k["scratch"] == 0, (name, k)
    assert kernels["fbd::k_step_duo<0, false, false>"]["scratch"] == 0 and kernels["fbf::k_step_f32"]["scratch"] == 0, "the headline kernels"
