"""Summaries of rocprofv3 result databases for profiles/ (the .db files themselves stay in gpurun_out/).
  kernel stats : python tools/rocprof_summary.py stats <results.db> > profiles/rNN_kernel_stats_X.csv
  HBM traffic  : python tools/rocprof_summary.py traffic <fetch.db> <write.db> > profiles/traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced streaming read; other access widths and WRITE_SIZE are
uncalibrated.  Both the raw value and the x2-corrected fetch are written; totals use the corrected fetch
(an upper bound for the kernels whose loads are narrower than 16 B/lane)."""
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


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("kernel,calls,total_us,avg_us,percent")
    for n, c, t, a, p in rows:
        print(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{p:.2f}")      # top_kernels reports microseconds


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
    for k in sorted(set(f) | set(w)):
        fr = f.get(k, (0.0, 0))[0] * 1024.0
        wr = w.get(k, (0.0, 0))[0] * 1024.0
        per[k] = {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2.0 * fr, "write_bytes": wr}
        if "pack_planes" in k:          # per-image preparation, not part of a step
            continue
        tot_raw += fr; tot_cor += 2.0 * fr; tot_w += wr
    out["kernel_sources_sha"] = kernel_sources_sha()
    out["note"] = ("per dispatch, one 257^3 step; FETCH_SIZE doubled per MI355X_MICROARCH.md (exact for 16 B/lane streams, an upper "
                   "bound otherwise); WRITE_SIZE as reported")
    out["per_kernel"] = per
    out["step_total_bytes"] = {"fetch_raw": tot_raw, "fetch_x2": tot_cor, "write": tot_w, "fetch_x2_plus_write": tot_cor + tot_w}
    for k, v in per.items():
        key = k.split("::")[-1].split("<")[0]
        out[key + "_bytes_per_launch"] = v["fetch_bytes_x2"] + v["write_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        traffic(sys.argv[2], sys.argv[3])
