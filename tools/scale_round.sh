#!/bin/bash
# ONE-SHOT script for the first session on an 8-GPU MI355X node (none has been in reach of a build session: every multi-GPU
# number in DESIGN.md section 6 is a projection until this has run).  Runs the headline step at N in {1,2,4,8} for
#   gather   : volume all-gather (north_star, what the driver's SCALE run measures)  |  --mesh-exchange
#   reserve  : CUs the persistent MLP grid leaves to RCCL: 0 | 16
#   overlap  : two slabs per rank, own sign exchange each, gathers straight into the result ('ab' layout, default)  |
#              --slab-layout contiguous (round 5: one slab in two halves, assembly copies)  |  --no-overlap-gather (one blocking
#              exchange, one gather)
#   receiver : every rank (all_gather)  |  --gather-to 0 (only the mesh consumer receives)
# and cfg 5 (one 513^3 image per GPU, no collective), and prints ONE table; the JSON lines are kept under $OUT.
#   usage: bash tools/scale_round.sh [outdir]          env: NS="1 2 4 8"  STEPS=20  WARMUP=3
#   on a box with fewer GPUs than N: ICON_AMD_DIST_BACKEND=gloo runs the control flow on the devices that are there (numbers mean nothing)
OUT=${1:-gpurun_out/scale}
NS=${NS:-"1 2 4 8"}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-3}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # tag, n, flags...
  tag=$1; n=$2; shift 2
  timeout 900 python bench.py --gpus $n --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-extras "$@" > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log > $OUT/$tag.json
}
for n in $NS; do
  if [ "$n" = "1" ]; then run n1 1; continue; fi
  for rc in 0 16; do
    run n${n}_vol_rc${rc}_ov $n --reserve-cus $rc
    run n${n}_vol_rc${rc}_ov_contig $n --reserve-cus $rc --slab-layout contiguous
    run n${n}_vol_rc${rc}_ov_to0 $n --reserve-cus $rc --gather-to 0
    run n${n}_vol_rc${rc}_blk $n --reserve-cus $rc --no-overlap-gather
    run n${n}_mesh_rc${rc} $n --reserve-cus $rc --mesh-exchange
  done
  run n${n}_cfg5_513 $n --replicas --res 513 --steps 5 --warmup 1
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads(open(f).read())
    except Exception:
        rows.append((os.path.basename(f)[:-5], "FAILED (see the .log)"))
        continue
    c = d["config"]
    dist = c.get("dist") or {}
    rs = c.get("rank_stage_ms") or []
    slow = max((r["step_ms"] for r in rs), default=d["ms_per_step"])
    rows.append((os.path.basename(f)[:-5], d["n_gpus"], d["scaling"], str(c.get("gather", "-")) + ("" if c.get("gather_to") is None else f" -> rank {c['gather_to']}")
                 + ("" if c.get("slab_layout") in (None, "contiguous") else " [ab]"), c.get("reserve_cus", "-"), c.get("overlap_gather", "-"),
                 c.get("split_features", "-"), f"{d['value'] / 1e6:.1f}", f"{d['ms_per_step']:.3f}", f"{slow:.3f}",
                 f"{max((r['mlp_ms'] for r in rs), default=c['stage_ms']['mlp']):.3f}", f"{max((r['features_ms'] for r in rs), default=c['stage_ms']['features']):.3f}",
                 dist.get("backend", "-"), dist.get("world_size_seen", "-"), dist.get("distinct_devices", "-")))
base = next((float(r[7]) for r in rows if len(r) > 2 and r[1] == 1), None)
print("| run | N | scaling | gather | reserve_cus | overlap | split | Mpts/s | ms/step | slowest rank ms | max mlp ms | max pre-pass ms | backend | ranks seen | devices | x N=1 |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if len(r) == 2:
        print(f"| {r[0]} | {r[1]} |")
        continue
    print("| " + " | ".join(str(v) for v in r) + (f" | {float(r[7]) / base:.2f} |" if base else " | - |"))
PY
