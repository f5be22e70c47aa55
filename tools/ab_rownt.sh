for nt in 0 1 2; do echo "== MI_ROW_NT=$nt"; MI_ROW_NT=$nt timeout 300 python tools/bench_assembly.py 2>&1 | grep -E "laplacian|fvm::div"; done
