"""CPU: the PMC traffic figures bench.py quotes (profiles/traffic_latest.json) belong to the kernels and the layout of THIS tree.
bench.py reports `roofline.traffic` only while the kernel sources (by text) and the tile layout / GAMG hierarchy the host
builders produce (by output on fixed reference cases, tools/source_fingerprint.py) are what the figures were measured with; a
change of those without a new `tools/prof_round.sh` + `tools/summarize_prof.py` (and `tools/gpu_r03_g.sh` for the GAMG cycle) run would
silently turn the driver's bench line into `"traffic": null` -- this test makes that visible here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_traffic_figures_match_the_sources():
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert rec["layout_source_sha256_16"] == bench.layout_source_hash(), "kernels.hip.hpp / tiling.hpp changed or the tile layout came out different: re-run tools/prof_round.sh and tools/summarize_prof.py"
    t, src = bench.traffic_from_profile(216, 216, 216, 1)
    assert t is not None and 0.9 * (24 * 216 ** 3 + 16 * (3 * 216 ** 3 - 3 * 216 ** 2)) < t < 1.5 * (24 * 216 ** 3 + 16 * 3 * 216 ** 3)
    assert os.path.exists(os.path.join(ROOT, src)), src
    g = bench.gamg_traffic_from_profile(216, 216, 216)
    assert g is not None and g > 1e9, "gamg_engine.inc / kernels changed or the layout / hierarchy came out different: re-run tools/gpu_r03_g.sh and tools/summarize_gamg_traffic.py --update"
    assert bench.traffic_from_profile(64, 64, 64, 1)[0] is None      # never quoted for another workload
