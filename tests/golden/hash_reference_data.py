#!/usr/bin/env python3
"""Records the sha256 of every DATA file of the reference that the repo ships a copy of.

Run in the build container, where /root/reference exists (it does not exist on the GPU box):

    python tests/golden/hash_reference_data.py

writes tests/golden/reference_data_sha256.json: {shipped path (relative to the repo root): {"sha256": ..., "bytes": ...,
"reference": path below the reference's root}}. The hashes are taken from the REFERENCE's files, not from the shipped copies:
tests/test_reference_trim_points.py and tests/test_reference_robot2d_linearization.py recompute the hash of the shipped copy
and compare, so that the numbers those tests read (the trim solutions and gain matrices the reference's design script stored,
lib/FlightApps/design/c172/c172x_design.jl:87-216,549-671) are known to be the reference's own.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("FLIGHT_JL_ROOT", "/root/reference")

CTL = "lib/FlightApps/src/c172/c172x/control/data/"
FILES = {   # shipped copy -> the reference's file (four gain files carry non-ASCII names there)
    "flight.jl_amd/data/ww15mgh_le.bin": "lib/FlightPhysics/src/data/ww15mgh_le.bin",
    "flight.jl_amd/data/robot2d.h5": "lib/FlightApps/src/robot2d/robot2d.h5",
    "flight.jl_amd/data/c172x_ctl/te2te.h5": CTL + "te2te.h5",
    "flight.jl_amd/data/c172x_ctl/tv2te.h5": CTL + "tv2te.h5",
    "flight.jl_amd/data/c172x_ctl/vh2te.h5": CTL + "vh2te.h5",
    "flight.jl_amd/data/c172x_ctl/q2e.h5": CTL + "q2e.h5",
    "flight.jl_amd/data/c172x_ctl/c2theta.h5": CTL + "c2θ.h5",
    "flight.jl_amd/data/c172x_ctl/v2t.h5": CTL + "v2t.h5",
    "flight.jl_amd/data/c172x_ctl/ar2ar.h5": CTL + "ar2ar.h5",
    "flight.jl_amd/data/c172x_ctl/phibeta2ar.h5": CTL + "φβ2ar.h5",
    "flight.jl_amd/data/c172x_ctl/p2phi.h5": CTL + "p2φ.h5",
    "flight.jl_amd/data/c172x_ctl/chi2phi.h5": CTL + "χ2φ.h5",
}


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit(f"{REFERENCE} not found: this script runs where the reference checkout is")
    out = {}
    for shipped, ref in FILES.items():
        p = os.path.join(REFERENCE, ref)
        out[shipped] = {"sha256": sha256(p), "bytes": os.path.getsize(p), "reference": ref}
    with open(os.path.join(HERE, "reference_data_sha256.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
        f.write("\n")
    print(f"hashed {len(out)} reference data files")


if __name__ == "__main__":
    main()
