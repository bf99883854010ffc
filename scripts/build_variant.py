#!/usr/bin/env python
"""Build libpixie_hip.so from the csrc/ + include/ of another git revision (A/B sessions on the GPU box swap the file in):
    python scripts/build_variant.py <git-rev> <out.so> [extra hipcc flags...]"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pixie_amd import build as B  # noqa: E402

rev, out, extra = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3:]
with tempfile.TemporaryDirectory() as tmp:
    subprocess.check_call(f"git -C {REPO} archive {rev} pixie_amd/csrc include | tar -x -C {tmp}", shell=True)
    csrc = os.path.join(tmp, "pixie_amd", "csrc")
    objs, procs = [], []
    for src in B.SOURCES:
        sp = os.path.join(csrc, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(tmp, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + extra + ["-c", sp, "-o", obj]))
    assert all(p.wait() == 0 for p in procs)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
print(out)
