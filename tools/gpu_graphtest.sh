cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -x -q 2>&1 | tail -2
for g in 1 0; do for w in objects4 static; do
 CF_GN_GRAPH=$g timeout 150 python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('graph',$g,'$w','fps',d['value'],'ms',d['ms_per_step'])"
done; done
