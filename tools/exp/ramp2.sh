#!/bin/bash
# one box, 170 s: a fresh short process every ~1.5 s printing its streaming rate with the wall time since the start
cd "$(dirname "$0")"
t0=$(date +%s)
while true; do
  t=$(( $(date +%s) - t0 ))
  [ $t -gt 170 ] && break
  echo -n "t=${t}s "; ./exp_ramp 1 40 | tr '\n' ' '; echo
  sleep 1
done
