#!/bin/bash
# round 3, GPU call C: GAMG (register inverse, graph replay), timing of the GAMG stage
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gamg.py -x -q -m gpu -k "register_resident" > gpurun_out/r03c/t_gamg.log 2>&1; echo "gamg rc=$?"
timeout 600 python tools/bench_gamg_setup.py > gpurun_out/r03c/gamg_setup.json 2> gpurun_out/r03c/gamg_setup.err; echo "setup rc=$?"
timeout 600 python tools/bench_timestep.py > gpurun_out/r03c/timestep.json 2> gpurun_out/r03c/timestep.err; echo "ts rc=$?"

tail -n 3 gpurun_out/r03c/t_gamg.log; cat gpurun_out/r03c/gamg_setup.json
grep -E "ms_per_time_step|PBiCG|GAMG" gpurun_out/r03c/timestep*.json
