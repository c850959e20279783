# developer tool (run via gpurun): nano + big bench one-liners
ulimit -c 0
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'tok/s', d['ms_per_step'], {k:v['us_per_launch'] for k,v in d['kernels'].items()})"
timeout 300 python bench.py --workload big:q4_0 --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'tok/s', d['ms_per_step'], {k:(v['us_per_launch'],v['GBps']) for k,v in d['kernels'].items()})"
