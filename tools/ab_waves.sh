L=rapidcfd-dev_amd/librapidcfd_amd.so
cp $L /tmp/w8.so; cp tools/exp/ab/librapidcfd_amd_w0.so /tmp/w0.so
run() { python bench.py --steps 300 --warmup 20 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pcg', d['value'], 'amul in loop', d['roofline']['avg_launch_us'], 'alone', d['config'].get('amul_alone_us_rotating_buffers'))"; python bench.py --solver gamg --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gamg ms/cycle', d['ms_per_step'])"; }
for v in w8 w0 w8 w0; do echo "== $v"; cp /tmp/$v.so $L; touch $L rapidcfd-dev_amd/libmiFoam.so rapidcfd-dev_amd/pEqnFoam rapidcfd-dev_amd/pEqnFoamPar rapidcfd-dev_amd/polyMeshFoam rapidcfd-dev_amd/polyMeshFoamPar; run; done
