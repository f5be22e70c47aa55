#!/bin/bash
# fresh box: clocks while the bench loop runs cold, then with the performance level forced high
cd "$(dirname "$0")/../.."
D=$(ls -d /sys/class/drm/card*/device | head -1)
show() { for f in pp_dpm_fclk pp_dpm_mclk pp_dpm_sclk pp_dpm_socclk power_dpm_force_performance_level; do echo "$f: $(cat $D/$f 2>/dev/null | tr '\n' ' ')"; done; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "clock level|Power"; }
b() { timeout 100 python bench.py --steps 2000 --warmup 20 --no-cpu 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('bench', round(b['value'],1), 'it/s amul', round(b['roofline']['avg_launch_us'],1))"; }
echo "== idle"; show
echo "== cold run 1"; (sleep 4.0; show) & b; wait
echo "== cold run 2"; (sleep 4.0; show) & b; wait
echo "== setperflevel high"; rocm-smi --setperflevel high 2>&1 | tail -3
echo "== high run 1"; (sleep 4.0; show) & b; wait
echo "== high run 2"; (sleep 4.0; show) & b; wait
rocm-smi --setperflevel auto 2>&1 | tail -2
echo "== auto again"; (sleep 4.0; show) & b; wait
