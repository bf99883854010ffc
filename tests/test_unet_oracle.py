"""CPU suite for the U-Net half: pins oracle/unet_oracle.py to the reference's golden vectors and
checks the product's plan + graph wiring without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_unet_golden import CASES, HEADS  # noqa: E402

from oracle import unet_oracle
from pixie_amd.synthetic import feature_grid
from pixie_amd.unet import UNetRunner
from pixie_amd.unet_plan import UNetConfig, build_plan, conv_flops, param_shapes, synthetic_state_dict
from tests._torch_ref_ops import TorchRefOps

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REFERENCE = "/root/reference/third_party/Wavelet-Generation"


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    """oracle/unet_oracle.py == outputs of the reference's own MyUNetModel/FeatureProjector (fixtures
    written by tests/golden/make_unet_golden.py).  Same torch CPU kernels => tolerance is roundoff only."""
    kw, wseed, iseed = CASES[name]
    g = np.load(os.path.join(GOLDEN, f"unet_{name}.npz"))
    feat = feature_grid(kw["grid_size"], kw["feature_channels"], seed=iseed)
    for head, oc, off in HEADS:
        cfg = UNetConfig(out_channels=oc, **kw)
        y = unet_oracle.unet_forward(synthetic_state_dict(cfg, wseed + off), cfg, feat).numpy()
        assert y.shape == g[head].shape
        assert rel_l2(y, g[head]) < 1e-6
        assert np.abs(g[head]).mean() > 0.05  # golden is not the vacuous all-zero output of fresh weights


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only exists in the build container")
def test_oracle_matches_reference_live():
    from make_unet_golden import run_reference
    kw = dict(feature_channels=64, cond_dim=32, model_channels=32, num_res_blocks=2, channel_mult=(1, 2, 2),
              attention_resolutions=(4,), grid_size=8)
    feat = feature_grid(8, 64, seed=11)
    cfg = UNetConfig(out_channels=5, **kw)
    ref = run_reference(cfg, 7, feat)  # load_state_dict(strict=True) inside also proves the key plan
    y = unet_oracle.unet_forward(synthetic_state_dict(cfg, 7), cfg, feat).numpy()
    assert rel_l2(y, ref) < 1e-6


def test_plan_structure_matches_survey_appendix():
    cfg = UNetConfig(grid_size=128, out_channels=8)
    plan = build_plan(cfg)
    assert len(plan.input_blocks) == 16 and len(plan.output_blocks) == 16
    assert [s[0].kind for s in plan.input_blocks][:6] == ["conv_in", "res", "res", "res", "down", "res"]
    assert plan.skip_channels == [64, 64, 64, 64, 64, 64, 64, 64, 64, 128, 128, 128, 128, 256, 256, 256]
    assert [b.kind for b in plan.middle] == ["res", "attn", "res"]
    assert (plan.output_blocks[0][0].cin, plan.output_blocks[0][0].cout) == (512, 256)
    assert [b.kind for b in plan.output_blocks[3]] == ["res", "up"]
    shapes = param_shapes(cfg)
    unet_keys = [k for k in shapes if k.startswith("unet.")]
    assert len(unet_keys) == 300
    assert shapes["unet.input_blocks.1.0.in_layers.0.weight"] == (128, 128, 128)
    assert shapes["unet.middle_block.1.qkv.weight"] == (768, 256, 1)
    assert shapes["projector.net.3.weight"] == (128, 128, 3, 3, 3)
    # algorithmic FLOPs agree with the survey's count to < 0.5 %
    assert abs(conv_flops(UNetConfig(grid_size=32, out_channels=8)) / 211.4e9 - 1) < 5e-3
    assert abs(conv_flops(cfg) / 13527.7e9 - 1) < 5e-3


def test_oracle_structure_is_its_own_and_agrees_with_the_product():
    """oracle/unet_oracle.py restates MyUNetModel.__init__ itself (it imports nothing from pixie_amd/), and that restatement, the
    product's Python plan and -- through tests/test_unet_handle.py -- the C handle's table name the same parameters in the
    same order, for every architecture variant the goldens cover plus the BASELINE ones."""
    import ast
    src = open(unet_oracle.__file__).read()
    imported = [n.module or "" for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ImportFrom)] + \
               [a.name for n in ast.walk(ast.parse(src)) if isinstance(n, ast.Import) for a in n.names]
    assert not any(m.startswith("pixie_amd") for m in imported), imported
    variants = [UNetConfig(out_channels=oc, **kw) for kw, _, _ in CASES.values() for _, oc, _ in HEADS]
    variants += [UNetConfig(grid_size=128, out_channels=8), UNetConfig(768, 32, 64, 3, (1, 1, 2, 4), (8,), 64, 3),
                 UNetConfig(128, 32, 64, 3, (1, 1, 2, 4), (), 256, 8), UNetConfig(16, 16, 16, 1, (1, 2), (1, 2), 9, 3)]
    for cfg in variants:
        assert unet_oracle.state_dict_keys(cfg) == list(param_shapes(cfg).keys()), cfg
        plan = build_plan(cfg)
        ins, mid, outs = unet_oracle.structure(cfg)
        flat = lambda seqs: [(b.kind, b.prefix, b.cin, b.cout) for seq in seqs for b in seq]
        assert flat(ins) == flat(plan.input_blocks) and flat([mid]) == flat([plan.middle]) and flat(outs) == flat(plan.output_blocks)


@pytest.mark.parametrize("name", ["full16", "noproj_attn8", "lightproj8"])
def test_runner_wiring_against_oracle(name):
    """pixie_amd.unet.UNetRunner with torch stand-ins for the five HIP operators reproduces the oracle:
    checks layer order, prologue/epilogue assignment, skip-stack order and statistics plumbing."""
    kw, wseed, iseed = CASES[name]
    feat = feature_grid(kw["grid_size"], kw["feature_channels"], seed=iseed)
    for head, oc, off in HEADS:
        cfg = UNetConfig(out_channels=oc, **kw)
        sd = synthetic_state_dict(cfg, wseed + off)
        params = {k: torch.from_numpy(v) for k, v in sd.items()}
        runner = UNetRunner(cfg, params, TorchRefOps())
        taps_r, taps_o = {}, {}
        y = runner.forward(torch.from_numpy(feat[0]), taps_r).numpy()
        ref = unet_oracle.unet_forward(sd, cfg, feat, taps=taps_o).numpy()[0]
        for key, val in taps_r.items():
            assert rel_l2(val.numpy(), taps_o[key].numpy()[0]) < 2e-5, key
        assert rel_l2(y, ref) < 2e-5


def test_f16x3_scheme_bounds_and_accuracy():
    """The f16x3 convolution scheme (fp16 hi/lo split of both operands, three products, fp32 accumulate), emulated
    on the CPU with the HIP kernel's scale selection: every f16x3 launch gets a magnitude bound that really holds
    (no fp16 overflow -- asserted inside the emulation), and the full 16^3 network stays within 1e-5 rel-L2 of the
    fp32 oracle (the GPU parity bar is 1e-4)."""
    from tests._torch_ref_ops import TorchRefOpsF16x3
    kw, wseed, iseed = CASES["full16"]
    feat = feature_grid(kw["grid_size"], kw["feature_channels"], seed=iseed)
    feat[0, :, 3, 4, 5] *= 300.0  # outliers in the raw input exercise the device-side amax scaling
    for head, oc, off in HEADS:
        cfg = UNetConfig(out_channels=oc, **kw)
        sd = synthetic_state_dict(cfg, wseed + off)
        params = {k: torch.from_numpy(v) for k, v in sd.items()}
        TorchRefOpsF16x3.n_f16x3 = TorchRefOpsF16x3.n_exact = 0
        runner = UNetRunner(cfg, params, TorchRefOpsF16x3(), precision="f16x3")
        y = runner.forward(torch.from_numpy(feat[0])).numpy()
        ref = unet_oracle.unet_forward(sd, cfg, feat).numpy()[0]
        err = rel_l2(y, ref)
        print(f"{head}: f16x3 emulation rel-L2 vs fp32 oracle {err:.3e}; {TorchRefOpsF16x3.n_f16x3} f16x3 + "
              f"{TorchRefOpsF16x3.n_exact} exact conv launches")
        assert TorchRefOpsF16x3.n_exact == 0       # every conv of the BASELINE architecture is on the f16x3 path (stride 2 included)
        assert TorchRefOpsF16x3.n_f16x3 >= 83
        assert err < 1e-5


def test_fresh_reference_style_init_is_zero_output():
    """SURVEY.md 'things to know' #5: zero_module makes a freshly built network output exactly 0; our
    module keeps that construction-time behaviour (and therefore needs synthetic weights for parity)."""
    from pixie_amd.unet import RegressionUNet
    m = RegressionUNet(32, 32, 32, 1, (1, 2), (), 8, out_channels=3)
    params = {k: v for k, v in m.named_parameters()}
    y = UNetRunner(m.cfg, params, TorchRefOps()).forward(torch.randn(32, 8, 8, 8))
    assert float(y.abs().max()) == 0.0
    assert set(m.state_dict().keys()) == set(param_shapes(m.cfg).keys())


def test_unscale_and_combine():
    rng = np.random.default_rng(0)
    pred = rng.normal(size=(11, 4, 4, 4)).astype(np.float32) * 2
    out = unet_oracle.unscale_prediction(pred)
    r = unet_oracle.NORMALIZATION_RANGES
    assert out[0].min() >= 10 ** r["density_min"] * (1 - 1e-6) and out[0].max() <= 10 ** r["density_max"] * (1 + 1e-6)
    assert out[2].min() >= r["nu_min"] - 1e-6 and out[2].max() <= r["nu_max"] + 1e-6
    np.testing.assert_array_equal(out[3:], pred[3:])
    logits = torch.tensor(rng.normal(size=(8, 3, 3, 3)).astype(np.float32))
    logits[2] = logits[5]  # ties resolve to the lowest index
    cmb = unet_oracle.combine_predictions(logits, torch.zeros(3, 3, 3, 3))
    assert torch.all(cmb[3:].sum(0) == 1)
    assert float(cmb[3 + 5].sum()) == 0 or not torch.any((cmb[3 + 5] == 1) & (logits[2] >= logits.max(0).values))
