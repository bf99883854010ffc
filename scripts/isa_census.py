#!/usr/bin/env python
"""Static instruction census of one kernel in a hipcc -S listing: counts per mnemonic class between s_barrier marks.
usage: isa_census.py file.s <substring of the kernel symbol>"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
body = []
for l in lines[start + 1:]:
    if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
        break
    body.append(l)


def klass(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("ds_"): return "lds:" + m
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem:" + m.split("_dword")[0]
    if m.startswith("s_"): return "salu" if not m.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_cbranch", "s_branch")) else m
    if m.endswith("_f64") or "_f64_" in m or m.startswith("v_cvt_f64") or "f64" in m: return "valu:f64"
    if m.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "valu:trans"
    if m.startswith("v_pk_"): return "valu:pk"
    if m.startswith("v_"): return "valu"
    return "other"


sec = 0
counts = collections.defaultdict(collections.Counter)
detail = collections.Counter()
for l in body:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    m = re.split(r"\s+", t)[0]
    if m == "s_barrier":
        sec += 1
    counts[sec][klass(m)] += 1
    if m.startswith("v_"):
        detail[re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", m)] += 1
tot = collections.Counter()
for s in sorted(counts):
    c = counts[s]
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    print(f"section {s}: valu {valu} (f64 {c['valu:f64']}, trans {c['valu:trans']}, pk {c['valu:pk']}), salu {c['salu']}, "
          + ", ".join(f"{k} {v}" for k, v in sorted(c.items()) if k.startswith(("lds", "vmem"))))
    tot.update(c)
print("total:", dict(tot))
print("top valu:", detail.most_common(40))
