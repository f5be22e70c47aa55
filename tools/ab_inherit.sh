#!/bin/bash
# A/B of MI_GAMG_INHERIT_TILES (level layouts from inherited tiles: no clustering inside the level-layout tasks) on start-up and V-cycle, fresh process each
export TMPDIR=/tmp
for rep in 1 2; do for f in 0 1; do
  echo "== MI_GAMG_INHERIT_TILES=$f"
  MI_GAMG_INHERIT_TILES=$f GAMG_CYCLES=20 timeout 300 python tools/bench_gamg.py 2>&1 | grep -E "^addr|ms_per_cycle|solve to" | cut -c1-400
done; done
