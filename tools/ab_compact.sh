run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['config'].get('amul_alone_us_rotating_buffers'))"; env "$@" timeout 100 python tools/occupancy.py 2>/dev/null | tail -2; }
run MI_ENTRY16=0
run MI_ENTRY16=1 MI_TILE_CELLS=896
run MI_ENTRY16=1 MI_TILE_CELLS=960
run MI_ENTRY16=1 MI_TILE_CELLS=832
run MI_ENTRY16=0
run MI_ENTRY16=1 MI_TILE_CELLS=896
