import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build what is MISSING once -- the product library with
    hipcc (cross-compiles without a GPU; the product library and its -DPIXIE_DIAG twin) and the oracle's C restatement with gcc -- exactly as __graft_entry__.build() does.
    Never rebuilds something that exists (on the GPU box the prebuilt files travel with the tree)."""
    libs = [os.path.join(REPO, "pixie_amd", n) for n in ("libpixie_hip.so", "libpixie_hip_diag.so")]
    oracle_so = [os.path.join(REPO, "oracle", "build", n) for n in ("libmpm_oracle_f32.so", "libmpm_oracle_f64.so")]
    if all(os.path.exists(p) for p in libs + oracle_so):
        return
    try:
        import __graft_entry__
        __graft_entry__.build()
    except Exception as exc:   # the tests that need the artefacts will fail with the loader's own message
        print(f"conftest: building the missing artefacts failed: {exc}", file=sys.stderr)


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch.device("cuda:0")
