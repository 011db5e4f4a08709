#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_voxelize.py tests/test_gpu_ties_shell.py -m gpu -q -k "reference_volume_encoder or real_ranks or bench_self_launch" 2>&1 | tail -30 > gpurun_out/r4k_tests.log; grep -E "passed|failed|Error|assert|real VolumeEncoder" gpurun_out/r4k_tests.log | head -20
WHICH=adaptive,host timeout 100 python tools/time_adaptive.py 2>&1 | grep "^adaptive\|^host" | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4k_bench.log 2>&1; tail -1 gpurun_out/r4k_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step')}, c['stage_ms'], 'ref_sched', c.get('reference_schedule_ms_per_volume'), 'cold', {k:v for k,v in c.get('cold_image_ms',{}).items() if k!='note'}, 'reserve', c.get('reserve_cus_cost',{}).get('ms_per_step'), 'parity', c.get('parity',{}).get('max_abs'), 'frac', r['frac'], r.get('frac_of_sustained'))
"
