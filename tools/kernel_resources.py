#!/usr/bin/env python3
"""Registers / scratch / spills of every kernel of one translation unit of icon_amd/csrc (hipcc's resource remarks).
usage: tools/kernel_resources.py fused_f16x3.hip [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = {"fused_f16x3.hip", "query_kernels.hip", "adaptive.hip", "mesh_device.hip", "mc_device.hip", "vox_kernels.hip", "vis_kernels.hip"}


def remarks(src):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-ffp-contract=off"] if src in EXACT else []) + \
              ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "icon_amd", "csrc", src), "-o", os.path.join(d, "x.o")]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode:
            sys.exit(p.stdout[-3000:])
        out, name = {}, None
        for line in p.stdout.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1); out[name] = {}
            m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and name:
                out[name][m.group(1).strip()] = int(m.group(2))
        return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in remarks(sys.argv[1]).items():
        if flt in k:
            print(f"{k[:90]:90s} sgpr {v.get('TotalSGPRs')} vgpr {v.get('VGPRs')} scratch {v.get('ScratchSize')} "
                  f"sspill {v.get('SGPRs Spill')} vspill {v.get('VGPRs Spill')} occ {v.get('Occupancy')}")
