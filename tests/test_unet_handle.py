"""The U-Net handle of include/pixie_hip.h section (A') without a GPU: construction, the state_dict table and the
workspace sizing are host code (no launches).  The launches themselves are covered by tests/test_unet_hip.py (-m gpu)."""
import ctypes as C

import pytest

from pixie_amd import _lib
from pixie_amd.unet_plan import UNetConfig, param_shapes

CONFIGS = {
    "baseline_config2_128": UNetConfig(64, 32, 64, 3, (1, 1, 2, 4), (8,), 128, 8),
    "shipped_64x768": UNetConfig(768, 32, 64, 3, (1, 1, 2, 4), (8,), 64, 3),
    "odd9_attention": UNetConfig(16, 16, 16, 1, (1, 2), (2,), 9, 3),
    "light_projector": UNetConfig(8, 16, 16, 1, (1, 2), (), 8, 3),
}


def _create(cfg, precision=0):
    lib = _lib.load()
    c = _lib.UNetConfigC()
    c.feature_channels, c.cond_dim, c.model_channels, c.num_res_blocks = cfg.feature_channels, cfg.cond_dim, cfg.model_channels, cfg.num_res_blocks
    c.n_channel_mult = len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        c.channel_mult[i] = m
    c.n_attention_resolutions = len(cfg.attention_resolutions)
    for i, m in enumerate(cfg.attention_resolutions):
        c.attention_resolutions[i] = m
    c.grid_size, c.out_channels, c.precision = cfg.grid_size, cfg.out_channels, precision
    h = C.c_void_p()
    rc = lib.pixie_unet_create(C.byref(h), C.byref(c))
    return lib, h, rc


@pytest.mark.parametrize("name", list(CONFIGS))
def test_state_dict_table_equals_the_reference_key_order(name):
    """Keys, order and shapes of the handle's parameter table equal pixie_amd.unet_plan.param_shapes, which
    tests/test_unet_oracle.py pins to the reference modules' state_dict (strict=True load)."""
    cfg = CONFIGS[name]
    lib, h, rc = _create(cfg)
    assert rc == 0, lib.pixie_last_error()
    shapes = param_shapes(cfg)
    assert lib.pixie_unet_param_count(h) == len(shapes)
    for i, (key, shape) in enumerate(shapes.items()):
        k = C.c_char_p(); numel = C.c_int64(); nd = C.c_int32(); shp = (C.c_int64 * 5)()
        assert lib.pixie_unet_param_info(h, i, C.byref(k), C.byref(numel), C.byref(nd), shp) == 0
        assert k.value.decode() == key
        assert tuple(shp[: nd.value]) == tuple(shape)
        n = 1
        for s in shape:
            n *= s
        assert numel.value == n
    assert lib.pixie_unet_param_info(h, len(shapes), None, None, None, None) != 0
    lib.pixie_unet_destroy(h)


def test_workspace_sizing_is_a_dry_run():
    """The sizing pass walks the same plan without launching: it works without a device, grows with the grid, and the
    exact-fp32 executor (no epilogue statistics, no split-K scratch) sizes differently from the f16x3 one."""
    cfg = CONFIGS["baseline_config2_128"]
    lib, h, rc = _create(cfg)
    assert rc == 0
    big = lib.pixie_unet_workspace_bytes(h, 128, 128, 128)
    # the skip stack alone: 13 tensors, the first four of 64 x 128^3 floats
    assert 4 * 64 * 128 ** 3 * 4 < big < 12 * 2 ** 30
    assert lib.pixie_unet_workspace_bytes(h, 128, 128, 128) == big          # cached, deterministic
    lib.pixie_unet_destroy(h)
    small_cfg = UNetConfig(64, 32, 64, 3, (1, 1, 2, 4), (8,), 16, 8)
    lib, h16, _ = _create(small_cfg)
    lib, h32, _ = _create(small_cfg, precision=1)
    a, b = lib.pixie_unet_workspace_bytes(h16, 16, 16, 16), lib.pixie_unet_workspace_bytes(h32, 16, 16, 16)
    assert 0 < a < big and 0 < b < big and a != b
    lib.pixie_unet_destroy(h16); lib.pixie_unet_destroy(h32)


def test_bad_configurations_are_rejected():
    lib, h, rc = _create(UNetConfig(64, 32, 64, 3, (), (), 16, 8))
    assert rc != 0 and b"channel_mult" in lib.pixie_last_error()
    lib, h, rc = _create(UNetConfig(64, 32, 64, 3, (1, 2), (), 16, 8), precision=7)
    assert rc != 0 and b"precision" in lib.pixie_last_error()
    lib, h, rc = _create(UNetConfig(64, 32, 64, 3, (1, 2), (), 16, 8))
    assert rc == 0
    assert lib.pixie_unet_set_param(h, b"unet.out.2.bias", None, 8) != 0
    lib.pixie_unet_destroy(h)
