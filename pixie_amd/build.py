"""Builds pixie_amd/libpixie_hip.so (gfx950) in-tree with hipcc.

`python -m pixie_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels with the working tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libpixie_hip.so")
LIB_DIAG = os.path.join(HERE, "libpixie_hip_diag.so")   # same sources + -DPIXIE_DIAG (tests, profilers)
ARCH = "gfx950"
SOURCES = ["common.hip", "mpm.hip", "unet_ops.hip", "conv3d_mfma.hip", "conv3d_f16x3.hip", "unet_exec.hip", "projector_fused.hip", "field_transfer.hip", "particle_filling.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]
# Per-file flags.  mpm.hip: hipcc's SLP vectoriser turns a third of the fused MPM kernel's fp32 arithmetic into packed
# v_pk_* instructions, each of which needs its operands in aligned register pairs: 168 VGPRs (3 waves per SIMD) and
# ~340 extra v_mov per wave to shuffle values into place.  A packed op issues in ~5.7 cycles against 4.5 for a scalar one
# (scripts/microbench/valu_rate.hip), so the packing saves little even before the moves.  Without it the same kernel needs
# 96 VGPRs (5 waves per SIMD) and no spills.
EXTRA_FLAGS = {"mpm.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _uses_diag(path: str) -> bool:
    with open(path) as f:
        return "PIXIE_DIAG" in f.read()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles csrc/*.hip into libpixie_hip.so (the product) and libpixie_hip_diag.so (the same sources with -DPIXIE_DIAG:
    + pixie_mpm_phase / pixie_mpm_kernel_times / pixie_conv_kernel_variant and the kernel trace buffer, for tests and
    profilers).  Only the sources that mention PIXIE_DIAG are compiled twice.  `force` (or PIXIE_FORCE_BUILD=1 in the
    environment) recompiles everything and prints hipcc's wall time per file, so that a cold build is observable."""
    force = force or os.environ.get("PIXIE_FORCE_BUILD", "") not in ("", "0")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "pixie_hip.h"))
    hipcc = _hipcc()
    objs = {False: [], True: []}
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        for diag in ((False, True) if _uses_diag(sp) else (False,)):
            obj = os.path.join(OBJ, src.replace(".hip", ".diag.o" if diag else ".o"))
            objs[diag].append(obj)
            if not force and _newer(obj, [sp, os.path.abspath(__file__)] + headers):
                continue
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DPIXIE_DIAG"] if diag else []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src + (" -DPIXIE_DIAG" if diag else ""), time.time(),
                          subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if not _uses_diag(sp):
            objs[True].append(objs[False][-1])
    failed = False
    for src, t0, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        else:
            if force or verbose:
                print(f"hipcc {src}: done {time.time() - t0:.0f} s after launch (all files compile in parallel)", flush=True)
            if verbose and out.strip():
                print(out)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    for diag, lib in ((False, LIB), (True, LIB_DIAG)):
        if force or procs or not _newer(lib, objs[diag]):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs[diag]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)} bytes) and {LIB_DIAG} ({os.path.getsize(LIB_DIAG)} bytes) from {len(procs)} compilations", flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
