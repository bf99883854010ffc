#!/usr/bin/env python
"""Turn the FETCH_SIZE / WRITE_SIZE PMC summaries (scripts/pmc_summary.py output) of the calibration kernels and of
the product kernels into HBM bytes per launch: profiles/pmc_traffic.json.
usage: scripts/pmc_traffic.py <dir with pmc_*.txt> <out.json>"""
import json
import re
import sys

import hashlib
import os

d, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = {"conv": "pixie_amd/csrc/conv3d_f16x3.hip", "mpm": "pixie_amd/csrc/mpm.hip"}   # the file each kernel lives in: bench.py drops an entry
                                                                                         # whose source has changed since (source_sha16)


def sha16(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
CAL_BYTES = 2 << 30


def read(name):
    res = {}
    try:
        for line in open(f"{d}/pmc_{name}.txt"):
            m = re.match(r"(.*?) \[dispatches (\d+)\]: (.*)", line)
            if m:
                res[m.group(1).strip()] = {k: float(v) for k, v in (kv.split("=") for kv in m.group(3).split())}
    except FileNotFoundError:
        pass
    return res


def find(tab, sub, ctr):
    for k, v in tab.items():
        if sub in k and ctr in v:
            return v[ctr]
    return None


cal_f, cal_w = read("calib_fetch"), read("calib_write")
# counter units are KiB; factor = true bytes / (counter * 1024) for each access width
fac = {}
for nm, tab, ctr in (("read4", cal_f, "FETCH_SIZE"), ("read16", cal_f, "FETCH_SIZE"), ("write4", cal_w, "WRITE_SIZE"), ("write16", cal_w, "WRITE_SIZE")):
    c = find(tab, "calib_" + nm, ctr)
    fac[nm] = (CAL_BYTES / (c * 1024.0)) if c else None
res = {"calibration": {"bytes_moved_per_kernel": CAL_BYTES, "factor_true_over_counter": fac,
                       "note": "counter unit KiB; factors measured with scripts/microbench/hbm_calib.hip in the same rocprofv3 passes"}}
for tag, kernel, rd, wr in (("conv_64_64_128", "conv3d_f16x3_c64_fullres_kernel", "read4", "write4"),
                            ("mpm_100k_block", "mpm_block_kernel<true, true", "read4", "write4"),
                            ("mpm_100k_grid", "mpm_grid_block_kernel", "read16", "write16"),
                            ("mpm_1m_block", "mpm_block_kernel<true, true", "read4", "write4"),
                            ("mpm_1m_grid", "mpm_grid_block_kernel", "read16", "write16")):
    run = tag.split("_")[0] + "_" + tag.split("_")[1] if tag.startswith("mpm") else "conv"
    f = find(read(run + "_fetch"), kernel, "FETCH_SIZE")
    w = find(read(run + "_write"), kernel, "WRITE_SIZE")
    if f is None or w is None or not fac.get(rd) or not fac.get(wr):
        res[tag] = None
        continue
    src = SOURCE["mpm" if tag.startswith("mpm") else "conv"]
    res[tag] = {"source": src, "source_sha16": sha16(src), "fetch_counter_KiB": f, "write_counter_KiB": w, "read_bytes": f * 1024 * fac[rd], "write_bytes": w * 1024 * fac[wr],
                "hbm_bytes_per_launch": f * 1024 * fac[rd] + w * 1024 * fac[wr],
                "correction": f"reads x{fac[rd]:.3f} ({rd}), writes x{fac[wr]:.3f} ({wr})"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
