#!/bin/bash
# One GPU call that regenerates everything profiles/ quotes: tests, bench lines, kernel stats, HBM traffic, PMC.
#   usage: gpurun -- 'bash tools/gpu_round.sh <tag>'      (outputs under gpurun_out/<tag>_*; summarise with tools/rocprof_summary.py)
T=${1:-rX}
R=$PWD
mkdir -p gpurun_out
if [ -z "$PROFILES_ONLY" ]; then     # PROFILES_ONLY=1: kernel stats + HBM traffic + PMC only
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; tail -3 gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log | cut -c1-400
timeout 600 python bench.py --prior pamir --no-cpu-baseline > gpurun_out/${T}_bench_pamir.log 2>&1; tail -1 gpurun_out/${T}_bench_pamir.log | cut -c1-300
timeout 600 python bench.py --precision f32 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/${T}_bench_f32.log 2>&1; tail -1 gpurun_out/${T}_bench_f32.log | cut -c1-300
timeout 600 python bench.py --res 513 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_513.log 2>&1; tail -1 gpurun_out/${T}_bench_513.log | cut -c1-300
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${T}_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${T}_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/${T}_pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_pmc.log 2>&1
if [ -n "$WITH_LDS" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $R/gpurun_out/${T}_lds -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_lds.log 2>&1
fi
# the reference's schedule as one native call: kernel stats + the launch timeline of the last volume; the per-image mesh build
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_adaptive -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/${T}_adaptive.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_meshbuild -- python $R/tools/time_mesh_build.py 20 > $R/gpurun_out/${T}_meshbuild.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/${T}_stats -name "*.db" | head -1) > gpurun_out/${T}_kernel_stats.csv; head -9 gpurun_out/${T}_kernel_stats.csv
# every launch of the dominant kernel (2 warm-up + 10 timed steps) beside the bench line of the SAME run (its roofline.wg_span_ms is the last step's)
(python tools/rocprof_summary.py launches $(find gpurun_out/${T}_stats -name "*.db" | head -1) k_fused_f16x3; grep '^{"metric"' gpurun_out/${T}_stats.log | python -c "import sys, json; r = json.loads(sys.stdin.read())['roofline']; print('# bench line of this run: avg_launch_ms', r['avg_launch_ms'], 'wg_span_ms', r.get('wg_span_ms'), 'tail_ms', r.get('tail_ms'), 'tail_inside_kernel_ms', r.get('tail_inside_kernel_ms'), 'per_xcd_clock_mhz', [round(x) for x in r.get('per_xcd_clock_mhz', [])], 'per_xcd_tiles', r.get('per_xcd_tiles'))") > gpurun_out/${T}_fused_launches.txt; tail -2 gpurun_out/${T}_fused_launches.txt | cut -c1-300
python tools/rocprof_summary.py traffic $(find gpurun_out/${T}_fetch -name "*.db" | head -1) $(find gpurun_out/${T}_write -name "*.db" | head -1) > gpurun_out/${T}_traffic.json
python tools/pmc_extract.py $(find gpurun_out/${T}_pmc -name "*.db" | head -1) | grep -A9 "k_fused\|k_nearest" > gpurun_out/${T}_pmc.txt
if [ -n "$WITH_LDS" ]; then
python tools/pmc_extract.py $(find gpurun_out/${T}_lds -name "*.db" | head -1) | grep -A9 "k_fused\|k_nearest" > gpurun_out/${T}_pmc_lds.txt
python tools/mlp_power_probe.py f16x3 5 > gpurun_out/${T}_power_probe.txt; cat gpurun_out/${T}_power_probe.txt
fi
DBA=$(find gpurun_out/${T}_adaptive -name "*.db" | head -1)
python tools/rocprof_summary.py stats $DBA > gpurun_out/${T}_adaptive_kernel_stats.csv
python tools/rocprof_summary.py timeline $DBA 23 > gpurun_out/${T}_adaptive_timeline.csv
grep "^adaptive" gpurun_out/${T}_adaptive.log | cut -c1-100
python tools/rocprof_summary.py stats $(find gpurun_out/${T}_meshbuild -name "*.db" | head -1) | grep "k_face_prep\|k_vertex_normals\|k_bvh\|k_tri_records\|k_scan_cells\|k_bin_\|kernel,calls" > gpurun_out/${T}_mesh_build_kernel_stats.csv
grep "^build 1[0-9]" gpurun_out/${T}_meshbuild.log | cut -c1-170 | head -3
python tools/time_coarse.py 2>/dev/null | grep "^slab" > gpurun_out/${T}_coarse_slabs.txt; cat gpurun_out/${T}_coarse_slabs.txt
python tools/trav_stats.py 2>/dev/null | grep "^33\|^65\|^129\|^257" | cut -c1-250 > gpurun_out/${T}_traversal_stats.txt
(echo "box $(cat /proc/sys/kernel/random/boot_id)"; N=${STRESS_N:-4000} python tools/stress_adaptive.py 2>&1 | tail -2) > gpurun_out/${T}_stress.txt; cat gpurun_out/${T}_stress.txt
python tools/time_mesh_extract.py 2>/dev/null | tail -8 > gpurun_out/${T}_mesh_extract.txt
# round 6: the fused kernel's tile partition, the search beside the MLP kernel, the schedule's latency distribution, the real-package harness's self-test
python tools/steal_probe.py > gpurun_out/${T}_steal_probe.txt 2>/dev/null
python tools/overlap_probe.py > gpurun_out/${T}_overlap_probe.txt 2>/dev/null
python tools/schedule_latency.py > gpurun_out/${T}_schedule_latency.txt 2>/dev/null; cat gpurun_out/${T}_schedule_latency.txt | cut -c1-200
python tools/parity_real_packages.py --stand-ins > gpurun_out/${T}_parity_harness_selftest.txt 2>&1; tail -1 gpurun_out/${T}_parity_harness_selftest.txt
find gpurun_out -name "*.db" -delete
