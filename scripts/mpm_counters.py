#!/usr/bin/env python
"""Collect, per MPM scene, what bench.py cannot measure from inside its own process: SQ_INSTS_VALU / SQ_WAVES of the block kernel
(a `rocprofv3 --pmc` pass, summarised by scripts/pmc_summary.py into <dir>/pmc_sq_<scene>.txt) and the block / grid kernels' mean
duration (a separate `rocprofv3 --kernel-trace --stats` pass, scripts/rocpd_stats.py -> <dir>/stats_<scene>.csv).
usage: scripts/mpm_counters.py <session dir> <out.json> [tag]"""
import csv
import glob
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.path.normpath(src))
res = {}
key_of = lambda scene: "100k_jelly" if scene == "jelly100k" else "1m_" + scene
for path in sorted(glob.glob(os.path.join(src, "pmc_sq_*.txt"))):
    scene = os.path.basename(path)[len("pmc_sq_"):-4]
    best = None
    for line in open(path):
        if "mpm_block_kernel" not in line:
            continue
        vals = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.e+-]+)", line)}
        n = int(re.search(r"dispatches (\d+)", line).group(1))
        if "SQ_INSTS_VALU" in vals and "SQ_WAVES" in vals and (best is None or n > best[0]):
            best = (n, vals, line.split(" [")[0])
    if best:
        n, v, name = best
        res.setdefault(key_of(scene), {}).update({
            "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1), "waves_per_launch": round(v["SQ_WAVES"]),
            "valu_active_frac": (round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"], 3) if "SQ_ACTIVE_INST_VALU" in v else None),
            "pmc_dispatches": n, "kernel": name, "pmc_source": f"profiles/{tag}_pmc_sq_{scene}.txt"})
for path in sorted(glob.glob(os.path.join(src, "stats_*.csv"))):
    scene = os.path.basename(path)[len("stats_"):-4]
    blk = grd = None
    for r in csv.DictReader(open(path)):
        if "mpm_block_kernel" in r["kernel"] and (blk is None or int(r["calls"]) > blk[0]):
            blk = (int(r["calls"]), float(r["avg_us"]))
        if "mpm_grid_block_kernel" in r["kernel"] and (grd is None or int(r["calls"]) > grd[0]):
            grd = (int(r["calls"]), float(r["avg_us"]))
    if blk:
        res.setdefault(key_of(scene), {}).update({"block_kernel_us": blk[1], "block_kernel_calls": blk[0], "grid_kernel_us": grd[1] if grd else None,
                                                  "stats_source": f"profiles/{tag}_stats_{scene}.csv"})
import hashlib
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_sha = hashlib.sha256(open(os.path.join(_root, "pixie_amd/csrc/mpm.hip"), "rb").read()).hexdigest()[:16]
_sha_math = hashlib.sha256(open(os.path.join(_root, "pixie_amd/csrc/mpm_math.h"), "rb").read()).hexdigest()[:16]
for v in res.values():      # bench.py drops an entry whose kernel source has changed since the pass (bench.py: _fresh)
    v.update({"source": "pixie_amd/csrc/mpm.hip", "source_sha16": _sha, "mpm_math_sha16": _sha_math})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
