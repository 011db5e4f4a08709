R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sh in 8 4 1 16; do
  ICON_AMD_SHARE=$sh timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s15_$sh -- env REPEAT=1 ONLY=513 python $R/tools/time_mesh_extract.py > $R/gpurun_out/s15_$sh.log 2>&1
  python $R/tools/rocprof_summary.py timeline $(find $R/gpurun_out/s15_$sh -name "*.db" | head -1) 60 > $R/gpurun_out/s15_${sh}_tl.csv
  echo "== share $sh"; grep "k_ad_nearest\|k_nearest_shared" $R/gpurun_out/s15_${sh}_tl.csv | tail -4
done
find $R/gpurun_out -name "*.db" -delete
