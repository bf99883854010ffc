"""Loads tests/host_harness/libmpm_math_host.so (pixie_amd/csrc/mpm_math.h compiled for the host).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "mpm_math_host.cpp")
LIB = os.path.join(HERE, "host_harness", "libmpm_math_host.so")
HDR = os.path.join(HERE, "..", "pixie_amd", "csrc", "mpm_math.h")


def load():
    fresh = os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(HDR))
    if not fresh:
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    lib.hh_svd3.argtypes = [C.c_int] + [C.c_void_p] * 4
    lib.hh_left_stretch.argtypes = [C.c_int] + [C.c_void_p] * 4
    lib.hh_stress.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_float] * 6 + [C.c_void_p] * 2
    lib.hh_polar.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hh_stencil.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib
