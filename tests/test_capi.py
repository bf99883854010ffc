"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol that
include/pixie_hip.h declares, struct layouts match, and the product refuses to run without a device."""
import ctypes as C
import os
import re

import pytest
import torch

from pixie_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(diag=False):
    """entry points include/pixie_hip.h declares: without / with its #ifdef PIXIE_DIAG section"""
    text = open(os.path.join(REPO, "include", "pixie_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    head, sep, tail = text.partition("#ifdef PIXIE_DIAG")
    assert sep and "#endif" in tail
    diag_part, _, rest = tail.partition("#endif")
    names = lambda t: set(re.findall(r"\b(pixie_[A-Za-z0-9_]+)\s*\(", t))
    return sorted(names(head) | names(rest) | (names(diag_part) if diag else set()))


def exported(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("pixie_")}


def test_library_exports_every_declared_symbol():
    """header == ctypes table == exported symbols, for BOTH builds: libpixie_hip.so (the product: no diagnostic entry point, no
    trace buffer) and libpixie_hip_diag.so (-DPIXIE_DIAG: the product's symbols + the header's diagnostic section)."""
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 28
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in include/pixie_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert exported(_lib.LIB_PATH) == set(names), exported(_lib.LIB_PATH) ^ set(names)
    assert lib.pixie_build_arch() == b"gfx950"
    diag_names = header_symbols(diag=True)
    assert set(diag_names) - set(names) == set(_lib.DIAG_SIGNATURES) == {"pixie_mpm_phase", "pixie_mpm_kernel_times", "pixie_conv_kernel_variant"}
    dlib = _lib.load(diag=True)
    assert exported(_lib.DIAG_LIB_PATH) == set(diag_names), exported(_lib.DIAG_LIB_PATH) ^ set(diag_names)
    assert dlib.pixie_build_arch() == b"gfx950"
    import subprocess
    syms = lambda p: subprocess.check_output(["nm", "-D", p], text=True)     # the 2 MB device trace buffer's host shadow + its reader
    assert "mpm_trace" in syms(_lib.DIAG_LIB_PATH) and "mpm_trace" not in syms(_lib.LIB_PATH)


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof of every struct field, as gcc lays out include/pixie_hip.h, equal the ctypes mirrors."""
    import subprocess
    structs = {"pixie_bc_desc": _lib.BCDesc, "pixie_pmod_desc": _lib.PModDesc, "pixie_conv_desc": _lib.ConvDesc,
               "pixie_field_desc": _lib.FieldDesc, "pixie_unet_config": _lib.UNetConfigC}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(REPO, "include", "pixie_hip.h")}"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device failure mode")
def test_fails_loudly_without_a_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.pixie_mpm_create(C.byref(h), 8, 8, 1.0) != 0
    assert b"hipMalloc" in lib.pixie_last_error() or b"device" in lib.pixie_last_error()
    from pixie_amd.mpm_solver import MPM_Simulator_WARP
    with pytest.raises(_lib.PixieHipError):
        MPM_Simulator_WARP(8, n_grid=8)
    from pixie_amd.unet import RegressionUNet
    m = RegressionUNet(32, 32, 32, 1, (1, 2), (), 8)
    with pytest.raises(_lib.PixieHipError):
        m(torch.zeros(1, 32, 8, 8, 8))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "pixie_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("oracle/ on the CPU", ""), f


def test_reference_material_name_quirk():
    from pixie_amd.mpm_solver import NAME_TO_MATERIAL_ID, get_material_id, get_material_name
    assert get_material_name("sand") == 2 and get_material_name("visplas") == -1 and get_material_id("snow") == 5
    assert NAME_TO_MATERIAL_ID["stationary"] == 6 and "fluid" not in NAME_TO_MATERIAL_ID


def test_one_hip_runtime_whatever_the_import_order():
    """Loading the library before torch must not leave two HIP runtimes in the process (torch's bundled libamdhip64 and the
    system ROCm one): build() followed by smoke() in one process broke exactly that way."""
    import subprocess
    import sys
    code = ("from pixie_amd import _lib; _lib.load(); import torch;"
            "m = open('/proc/self/maps').read();"
            "libs = sorted({l.split()[-1] for l in m.splitlines() if 'libamdhip64' in l});"
            "print(len(libs)); print(libs)")
    out = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == "1", out.stdout
