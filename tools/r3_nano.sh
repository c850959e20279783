#!/bin/bash
ulimit -c 0; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or golden or greedy or full_size or nano or graph or reset or streams or context or fallback" 2>&1 | tail -15) > gpurun_out/r3_nano_pytest.log; cat gpurun_out/r3_nano_pytest.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r3_nano_b20.json
(timeout 300 python bench.py --steps 512 --warmup 64 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r3_nano_b512.json
python3 -c "
import json
for f in ('gpurun_out/r3_nano_b20.json','gpurun_out/r3_nano_b512.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:v['us_per_launch'] for k,v in d['kernels'].items()})
"
(timeout 200 python3 tools/block_probe.py nano q8_0) 2>&1 | tail -8
