"""CPU: tests/golden/full_size_v1.npz (the oracle's results at the BASELINE sizes, which the GPU tests compare the engine
against -- tests/full_size_ref.py) cannot drift from the oracle unseen:

  * every section carries the sha256 of what it was derived from -- the oracle's C sources and ctypes wrapper, its own generator
    function, the case generator's outputs (assembly section: also tests/assembly_full_size.py and the tile layout's outputs);
    they must be the current ones (a change => re-run tests/golden/make_full_size.py <section>);
  * a sample of the records is RE-DERIVED here with the live oracle, bit for bit: the 108^3 persistent-kernel reference solve
    (300 iterations, whole history and the solution sample) and, at 216^3, the Amul result (sha256 + sample) and the first
    iterations of the diagonal-PCG history (PCG.C:133-204).
"""
import numpy as np
import pytest

import full_size_ref as fs

SECTIONS = ("assembly216", "box216_sym", "box216_asym", "persist108", "config4", "config5")


@pytest.fixture(scope="module")
def rec():
    return fs.Records()


def test_every_section_was_derived_from_the_current_oracle_sources(rec):
    for name in SECTIONS:
        now = fs.source_hashes(name)
        for rel, h in now.items():
            assert str(rec.scalar(f"sources/{name}/{rel}")) == h, f"section {name} is older than {rel}: re-run tests/golden/make_full_size.py {name}"


def test_every_record_the_gpu_tests_read_is_there(rec):
    for key in ("box216_sym/pcg_diagonal_120", "box216_sym/pcg_AINV_120", "box216_sym/pcg_diagonal_to_1e-6", "box216_sym/pcg_diagonal_60",
                "box216_sym/gamg_to_1e-6", "box216_asym/pbicg_AINV_40", "box216_asym/pbicgstab_AINV_24", "box216_asym/pbicgstab_diagonal_zA_48",
                "config4/pcg_diagonal_20", "config4/gamg_5", "config5/bicg", "config5/gamg"):
        p = rec.perf(key)
        assert p["history"].shape[0] == p["nIterations"] + 1 or key.startswith(("box216_sym/gamg", "config4/gamg", "config5/gamg")), key
        assert np.all(np.isfinite(p["history"])) and p["history"][0] > 0
    assert rec.perf("box216_sym/pcg_diagonal_120")["nIterations"] == 121 and rec.perf("box216_sym/pcg_diagonal_to_1e-6")["converged"] == 1
    for name in ("amul", "tmul", "sumA", "residual", "H", "H1", "precondition_diagonal", "precondition_AINV", "jacobi_2_sweeps"):
        assert len(rec.sha("box216_sym/" + name)) == 64
    for name in ("amul", "tmul", "precondition_AINV_transpose0", "precondition_AINV_transpose1"):
        assert len(rec.sha("box216_asym/" + name)) == 64
    assert len(rec.sha("config4/amul")) == 64 and rec.scalar("config5/bicg/rank_sum").shape == (8,)


def test_rederive_the_108_cubed_reference_solve(pkg, orc, rec):
    case = pkg.synthetic.box_case(108, 108, 108)
    psi, perf = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=0.0, maxIter=299)
    key = "persist108/single_rank/300_iterations"
    ref = rec.perf(key)
    assert np.array_equal(perf["history"], ref["history"]) and perf["nIterations"] == ref["nIterations"] and perf["normFactor"] == ref["normFactor"]
    sol = rec.solution(key)
    assert np.array_equal(psi[fs.sample_idx(case.n_cells)], sol.sample) and sol.deviation(psi) == 0.0
    sol.check(psi, 1e-15)


def test_rederive_amul_and_the_first_pcg_iterations_at_216_cubed(pkg, orc, rec):
    case = pkg.synthetic.box_case(216, 216, 216)
    S = orc.System([case])
    x = pkg.synthetic.splitmix_uniform(99, case.n_cells) - 0.5
    y = S.amul(x)
    assert np.array_equal(y[fs.sample_idx(case.n_cells)], rec.scalar("box216_sym/amul_sample"))
    assert fs.sha(y) == rec.sha("box216_sym/amul")
    _, perf = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=0.0, maxIter=5)
    ref = rec.perf("box216_sym/pcg_diagonal_120")
    assert np.array_equal(perf["history"], ref["history"][:perf["history"].shape[0]]) and perf["normFactor"] == ref["normFactor"]


def test_rederive_assembly_records_on_a_slab_of_the_same_generators(pkg, orc, rec):
    """the assembly section's generators (tests/assembly_full_size.py) run here on a small box on both meshes: every operator name the
    216^3 record holds is produced, and the unfused oracle sequence of the fused assembly agrees with plain numpy on the diagonal"""
    import assembly_full_size as afs
    for variant in afs.VARIANTS:
        M = afs.mesh(pkg, variant, dims=(20, 12, 9))
        q = afs.inputs(pkg, M)
        res = afs.oracle_run(pkg, orc, M, q)
        assert sorted(res) == list(rec.scalar(f"assembly216/{variant}/names"))
        for name, a in res.items():
            assert np.all(np.isfinite(a)), name
            assert len(rec.sha(f"assembly216/{variant}/{name}")) == 64
        m = res["assemble_momentum/diag"]
        dA = (q["rdt"] * q["rho"]) * q["vol"]
        assert np.array_equal(m, ((dA + orc.fvm_div(M["n"], M["lo"], M["up"], orc.upwind_weights(q["flux"]), q["flux"])[2])
                                  - res["laplacian/diag"]) - q["vol"] * q["sp"])
