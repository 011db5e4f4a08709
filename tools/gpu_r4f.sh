#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_DUMP_DIR=$PWD/gpurun_out/dump
timeout 100 python tools/time_mesh_build.py 5 2>&1 | grep "^build"
timeout 200 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q 2>&1 | tail -30 > gpurun_out/r4f_meshbuild.log; grep -E "passed|failed|differs|Error|mismatch" gpurun_out/r4f_meshbuild.log | head
timeout 400 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -k "adaptive or native_schedule" 2>&1 | tail -40 > gpurun_out/r4f_adaptive.log; grep -E "passed|failed|Error|assert|queries" gpurun_out/r4f_adaptive.log | head -30
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r4f_bench.log 2>&1; tail -1 gpurun_out/r4f_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print({k:d[k] for k in ('value','ms_per_step')}, c['stage_ms'], 'prep', c['prep_ms'], 'ref_sched', c.get('reference_schedule_ms_per_volume'), c.get('reference_schedule_points'), 'cold', c.get('cold_image_ms'), 'mesh', c.get('mesh',{}).get('chamfer_x100_dense_vs_reference_schedule'), 'parity', c.get('parity',{}).get('max_abs'), d['roofline']['frac'])
"
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4f_prof -- python $R/tools/time_mesh_build.py 6 > $R/gpurun_out/r4f_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4f_prof -name "*.db" | head -1) > gpurun_out/r4f_kernel_stats.csv; head -16 gpurun_out/r4f_kernel_stats.csv
find gpurun_out -name "*.db" -delete
