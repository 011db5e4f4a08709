#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4h_pytest.log; tail -6 gpurun_out/r4h_pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4h_bench.log 2>&1; tail -1 gpurun_out/r4h_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step')}, c['stage_ms'], 'prep', c['prep_ms'], 'ref_sched', c.get('reference_schedule_ms_per_volume'), c.get('reference_schedule_native'), c.get('reference_schedule_points'), 'cold', {k:v for k,v in c.get('cold_image_ms',{}).items() if k!='note'}, 'mesh', c.get('mesh',{}).get('chamfer_x100_dense_vs_reference_schedule'), 'parity', c.get('parity',{}).get('max_abs'), 'frac', r['frac'], 'sustained', r.get('sustained_peak'), r.get('sustained_peak_layer1_shape'), r.get('frac_of_sustained'))
"
