#!/bin/bash
# round 3, GPU call A: the peer-window path (tests + the 108^3 self-communicator measurement) and a headline sanity run
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or halo_windows or self_exchange or attached_comm or distributed_matrix" > gpurun_out/r03a/t_parity.log 2>&1; echo "parity rc=$?"
timeout 600 python -m pytest tests/test_gamg.py -x -q -m gpu -k "cycle_graph or gamg_history" > gpurun_out/r03a/t_gamg.log 2>&1; echo "gamg rc=$?"
timeout 900 python -m pytest tests/test_distributed.py -x -q -m gpu -k "peer or several_engine_ranks" > gpurun_out/r03a/t_dist.log 2>&1; echo "dist rc=$?"
timeout 600 python tools/bench_selfcomm.py --out gpurun_out/r03a/selfcomm_108.json > gpurun_out/r03a/selfcomm.log 2>&1; echo "selfcomm rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r03a/t_parity.log gpurun_out/r03a/t_gamg.log gpurun_out/r03a/t_dist.log; tail -5 gpurun_out/r03a/selfcomm.log; cat gpurun_out/r03a/bench.json | cut -c1-400
