"""Regenerates tests/golden/golden_v1.npz from the CPU oracle.

The reference holds no golden vectors, so these fixtures are the oracle's own
outputs on the synthetic cases, frozen so that any later change of the oracle or of the generators
is caught.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

CASES = {
    "box_sym_12x10x8": dict(dims=(12, 10, 8), symmetric=True),
    "box_asym_12x10x8": dict(dims=(12, 10, 8), symmetric=False),
    "cavity_32": dict(dims=(32, 32, 32), symmetric=True),  # BASELINE config 1 size (N = 32768, F = 95232)
}


def build(pkg, orc):
    out = {}
    for name, spec in CASES.items():
        case = pkg.synthetic.box_case(*spec["dims"], symmetric=spec["symmetric"])
        S = orc.System([case])
        x = pkg.synthetic.splitmix_uniform(2024, case.n_cells) - 0.5
        out[f"{name}/n"] = np.array([case.n_cells, case.n_faces])
        out[f"{name}/amul"] = S.amul(x)[:: max(1, case.n_cells // 257)]
        out[f"{name}/tmul"] = S.tmul(x)[:: max(1, case.n_cells // 257)]
        out[f"{name}/sumA"] = S.sumA()[:: max(1, case.n_cells // 257)]
        z = np.zeros(case.n_cells)
        if spec["symmetric"]:
            for pre in ("diagonal", "AINV", "DIC_upstream"):
                _, p = S.pcg(z, case.source, pre, tolerance=1e-8, maxIter=1000)
                out[f"{name}/pcg_{pre}"] = p["history"]
        else:
            for pre in ("diagonal", "AINV"):
                _, p = S.pbicg(z, case.source, pre, tolerance=1e-10, maxIter=300)
                out[f"{name}/pbicg_{pre}"] = p["history"]
                _, p = S.pbicgstab(z, case.source, pre, tolerance=1e-10, maxIter=300, replicate_quirk=True)
                out[f"{name}/pbicgstab_quirk_{pre}"] = p["history"]
                _, p = S.pbicgstab(z, case.source, pre, tolerance=1e-10, maxIter=300, replicate_quirk=False)
                out[f"{name}/pbicgstab_{pre}"] = p["history"]
    return out


if __name__ == "__main__":
    graft.build()
    pkg = graft.load_package()
    from oracle import oracle as orc
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz"), **build(pkg, orc))
    print("written")
