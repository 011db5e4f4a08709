"""Summaries of rocprofv3 result databases for profiles/ (the .db files themselves stay in gpurun_out/).
  kernel stats : python tools/rocprof_summary.py stats <results.db> > profiles/rNN_kernel_stats_X.csv
  timeline     : python tools/rocprof_summary.py timeline <results.db> <last N dispatches>
  launches     : python tools/rocprof_summary.py launches <results.db> <kernel name part>     (every launch's duration)
  HBM traffic  : python tools/rocprof_summary.py traffic <fetch.db> <write.db> > profiles/traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced streaming read; other access widths and WRITE_SIZE are
uncalibrated - "calibrate on a known byte count in your own access pattern".  Calibrations of this repo (known array
sizes against the counter, one MI355X):
  k_fused_f16x3   : icon minus pamir launch (the difference is the slot / code / d^2 / sign arrays): with 4-byte slots 89 MB known,
                    80.7 MB raw (1.10); with 2-byte slots 58 MB known, 63.7 MB raw (0.92) -> factor 1.0: the sparse 2-4 B/lane and
                    1 B/lane loads of 256 threads per tile are NOT halved
  k_outlier_compact: streams the 17.0 MB code array (1 B/lane): 12.7 MB raw -> factor 1.33
  k_sign (round 2, when it still streamed 68 MB of d^2 at 4 B/lane): 35-39 MB raw -> factor 2 (the guide's case)
Every kernel gets its raw value, the x2 upper bound and - where calibrated - the calibrated fetch; totals and the
`*_bytes_per_launch` entries use the calibrated fetch where there is one and the x2 upper bound otherwise."""
import collections
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha() -> str:
    """fingerprint of everything the kernels are compiled from: bench.py reports profiles/traffic.json only while it
    matches (a PMC pass is a separate run; a stale file must not pose as a measurement of the current kernels)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "icon_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "icon_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "icon_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "icon_amd", "csrc", "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# kernels of the per-image preparation (device mesh build, feature plane packing): not part of a step
PREP_KERNELS = ("pack_planes", "k_face_prep", "k_vertex_normals", "k_bvh_", "k_tri_records", "k_scan_cells", "k_bin_fill", "k_bin_sort")

# FETCH_SIZE calibration factors (true bytes / reported bytes) measured on known array sizes - see the module docstring
FETCH_CALIBRATION = {"k_fused_f16x3": 1.0, "k_outlier_compact": 1.33}


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("kernel,calls,total_us,avg_us,percent")
    for n, c, t, a, p in rows:
        print(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{p:.2f}")      # top_kernels reports microseconds


def timeline(path, last):
    """the last `last` kernel dispatches in launch order: start (us from the first of them), duration, gap to the one before"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    try:
        rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    except sqlite3.Error as e:
        names = [r[0] for r in cur.execute("select name from sqlite_master").fetchall()]
        print("no `kernels` view:", e, names)
        return
    rows = rows[-last:]
    t0 = rows[0][1]
    print("kernel,start_us,dur_us,gap_us")
    prev_end = None
    for n, s_, e_ in rows:
        gap = (s_ - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"\"{short(n)}\",{(s_ - t0) / 1e3:.1f},{(e_ - s_) / 1e3:.2f},{gap:.2f}")
        prev_end = e_


def launches(path, needle):
    """every dispatch of the kernels whose name contains `needle`, in launch order: duration (us).  The stats view averages over
    ALL launches of a run, the untimed warm-up steps included (cold clocks, first touch): this is the list behind the average"""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    sel = [(short(n), (e_ - s_) / 1e3) for n, s_, e_ in rows if needle in n]
    print("kernel,launch,dur_us")
    for k, (n, d) in enumerate(sel):
        print(f"\"{n}\",{k},{d:.2f}")
    if sel:
        ds = sorted(d for _, d in sel)
        print(f"# {len(sel)} launches: min {ds[0]:.2f} median {ds[len(ds) // 2]:.2f} max {ds[-1]:.2f} mean {sum(ds) / len(ds):.2f} us")


def counters(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name",
                    (counter,)).fetchall()
    return {short(k): (v, n) for k, v, n in q if "icon" in k}


def traffic(fetch_db, write_db):
    f, w = counters(fetch_db, "FETCH_SIZE"), counters(write_db, "WRITE_SIZE")
    out = collections.OrderedDict()
    tot_raw = tot_cor = tot_w = 0.0
    per = {}
    tot_best = prep = 0.0
    for k in sorted(set(f) | set(w)):
        fr = f.get(k, (0.0, 0))[0] * 1024.0
        wr = w.get(k, (0.0, 0))[0] * 1024.0
        per[k] = {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2.0 * fr, "write_bytes": wr}
        cal = next((c for name, c in FETCH_CALIBRATION.items() if name in k), None)
        if cal is not None:
            per[k]["fetch_bytes_calibrated"] = cal * fr
            per[k]["fetch_calibration_factor"] = cal
        per[k]["fetch_bytes_best"] = cal * fr if cal is not None else 2.0 * fr
        if any(n in k for n in PREP_KERNELS):      # per-image preparation, not part of a step: summed separately
            prep += per[k]["fetch_bytes_best"] + wr
            continue
        tot_raw += fr; tot_cor += 2.0 * fr; tot_w += wr; tot_best += per[k]["fetch_bytes_best"]
    out["kernel_sources_sha"] = kernel_sources_sha()
    out["note"] = ("per dispatch, one 257^3 step; fetch: raw FETCH_SIZE, its x2 upper bound (MI355X_MICROARCH.md: exact for 16 B/lane "
                   "streams) and, for the kernels calibrated on known byte counts (tools/rocprof_summary.py header), the calibrated value; "
                   "'best' = calibrated where available, x2 otherwise; WRITE_SIZE as reported")
    out["per_kernel"] = per
    out["step_total_bytes"] = {"fetch_raw": tot_raw, "fetch_x2": tot_cor, "fetch_best": tot_best, "write": tot_w,
                               "fetch_x2_plus_write": tot_cor + tot_w, "fetch_best_plus_write": tot_best + tot_w}
    out["prep_total_bytes"] = prep      # the per-image mesh build + plane packing (once per image, outside the step)
    for k, v in per.items():
        key = k.split("::")[-1].split("<")[0]
        out[key + "_bytes_per_launch"] = v["fetch_bytes_best"] + v["write_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "launches":
        launches(sys.argv[2], sys.argv[3])
    else:
        traffic(sys.argv[2], sys.argv[3])
