"""examples/c_abi_demo.c: the C ABI used from plain C99 (gcc; no C++, Python or torch in the process).  CPU: it compiles against
include/mi_ldu.h and links; GPU: it runs and its solverPerformance line equals the oracle's for the same matrix."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "examples", "c_abi_demo")


def test_demo_is_plain_c_and_links(pkg):
    assert os.path.exists(DEMO), "built by __graft_entry__.build()"
    src = open(DEMO + ".c").read()
    assert "extern \"C\"" not in src and "#include \"mi_ldu.h\"" in src
    out = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-isystem", "/opt/rocm/include",
                          "-D__HIP_PLATFORM_AMD__", DEMO + ".c"], capture_output=True, text=True)
    assert out.returncode == 0 and "warning" not in out.stderr, out.stderr
    needed = subprocess.run(["readelf", "-d", DEMO], capture_output=True, text=True).stdout
    assert "librapidcfd_amd.so" in needed and "libstdc++" not in needed and "libtorch" not in needed


@pytest.mark.gpu
def test_demo_matches_the_oracle(pkg, orc):
    dims = (24, 20, 16)
    out = subprocess.run([DEMO, *map(str, dims)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    syn = pkg.synthetic
    case0 = syn.box_case(*dims)                           # only for the addressing (same OpenFOAM face order)
    lo, up = case0.lower_addr, case0.upper_addr
    n, nf = case0.n_cells, case0.n_faces
    f = np.arange(nf); c = np.arange(n)
    upper = -(1.0 + 0.001 * (f % 97))
    diag = np.zeros(n); np.subtract.at(diag, lo, upper); np.subtract.at(diag, up, upper)
    diag += 0.05 + 0.001 * (c % 13)
    src = ((c.astype(np.uint64) * np.uint64(2654435761)) % np.uint64(2 ** 32) % np.uint64(1000)).astype(np.float64) / 1000.0 - 0.5
    case = syn.LduCase(n, lo, up, diag, upper, None, src)
    _, p = orc.System([case]).pcg(np.zeros(n), src, "AINV", tolerance=1e-9, maxIter=1000)
    m = re.search(r"AINVPCG:  Solving for p, Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(3)) == p["nIterations"]
    assert abs(float(m.group(1)) - p["initialResidual"]) < 1e-12 and abs(float(m.group(2)) - p["finalResidual"]) < 1e-10 * p["initialResidual"] + 1e-15
    chk = float(re.search(r"check: sum\|b - A psi\| / normFactor = (\S+)", out.stdout).group(1))
    assert abs(chk - float(m.group(2))) < 1e-6 * float(m.group(2)) + 1e-14          # the recursive residual equals the true one
    assert out.stdout.strip().endswith("End")
