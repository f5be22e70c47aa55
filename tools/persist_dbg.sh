# timing experiments on the distributed persistent kernel (results invalid for dbg != 0): which piece costs what
for d in ${DBG_LIST:-0 1 8 15}; do echo "dbg=$d $(MI_PERSIST_DBG=$d timeout 200 python tools/bench_selfcomm.py --mode persist --iters 200 2>&1 | grep -E '^persist' | cut -c1-60)"; done
