# A/B of the 16-bit row entries (11 % fewer bytes per Amul) on tile sizes whose LDS image keeps four workgroups per CU
run() { echo "== $*"; env "$@" MI_BENCH_NO_SUPPLEMENTS=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu 2> gpurun_out/r03e/ab_last.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value'],1), 'amul_in_loop_us', round(d['roofline']['avg_launch_us'],2), 'amul_alone_us', round(d['config'].get('amul_alone_us_rotating_buffers'),2))" || tail -5 gpurun_out/r03e/ab_last.err; }
run MI_ENTRY16=0
run MI_ENTRY16=1 MI_TILE_CELLS=896
run MI_ENTRY16=1 MI_TILE_CELLS=960
run MI_ENTRY16=1 MI_TILE_CELLS=832
run MI_ENTRY16=1
run MI_ENTRY16=0 MI_TILE_CELLS=896
run MI_ENTRY16=0
run MI_ENTRY16=1 MI_TILE_CELLS=896
