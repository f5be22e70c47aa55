cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MI_BENCH_NO_SUPPLEMENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fused -o t -- python bench.py --steps 200 --warmup 10 --no-cpu > gpurun_out/prof_fused_bench.json 2> gpurun_out/prof_fused.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_fused/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf gpurun_out/prof_fused/*/*kernel_trace.csv
