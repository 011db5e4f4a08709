set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_m.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python bench.py > gpurun_out/bench_m.log 2>&1; tail -1 gpurun_out/bench_m.log | cut -c1-900
for d in gpurun_out/pmc_fetch gpurun_out/pmc_write; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f | grep -A2 -E "k_mlp|k_nearest|k_features"; done
find gpurun_out/prof_m -name "*.csv" | head
