"""Generates tests/golden/unet_*.npz by running the REFERENCE's own modules.

Run in the build container (needs /root/reference; it does not exist on the GPU box):
    python tests/golden/make_unet_golden.py
Imports MyUNetModel / FeatureProjector unmodified from
/root/reference/third_party/Wavelet-Generation/models/module/diffusion_network.py, wraps them as
SegmentationUNet / RegressionUNet do (trainer/training_discrete.py:50-88,
trainer/training_continuous_mse.py:48-89 -- those modules import hydra/wandb, absent here),
loads pixie_amd.unet_plan.synthetic_state_dict with strict=True (which also proves the key/shape
plan), and stores input seed + outputs.  Only outputs are stored: weights and inputs are
regenerated from seeds by pixie_amd.unet_plan / pixie_amd.synthetic.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
REF_WG = "/root/reference/third_party/Wavelet-Generation"

from pixie_amd.synthetic import feature_grid  # noqa: E402
from pixie_amd.unet_plan import UNetConfig, synthetic_state_dict  # noqa: E402

CASES = {
    # name: (cfg kwargs without out_channels, weight seed, input seed)
    "full16": (dict(feature_channels=64, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4),
                    attention_resolutions=(), grid_size=16), 0, 0),
    "full32": (dict(feature_channels=64, cond_dim=32, model_channels=64, num_res_blocks=3, channel_mult=(1, 1, 2, 4),
                    attention_resolutions=(), grid_size=32), 0, 0),
    "noproj_attn8": (dict(feature_channels=32, cond_dim=32, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
                          attention_resolutions=(2,), grid_size=8), 3, 4),
    # odd grid: Downsample 9 -> 5, Upsample 5 -> 10, cropped back to 9 (diffusion_network.py:925-930)
    "odd9": (dict(feature_channels=32, cond_dim=32, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
                  attention_resolutions=(), grid_size=9), 7, 8),
    # two odd levels with the projector: 13 -> 7 -> 4, back up 4 -> 8 (crop 7) -> 14 (crop 13)
    "odd13": (dict(feature_channels=64, cond_dim=32, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2),
                   attention_resolutions=(), grid_size=13), 9, 10),
    "lightproj8": (dict(feature_channels=3, cond_dim=32, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
                        attention_resolutions=(), grid_size=8), 5, 6),
}
HEADS = (("seg", 8, 0), ("cont", 3, 1000))  # (name, out_channels, weight-seed offset)


def reference_model(cfg: UNetConfig) -> nn.Module:
    if REF_WG not in sys.path:
        sys.path.insert(0, REF_WG)
    from models.module.diffusion_network import FeatureProjector, MyUNetModel  # the reference, unmodified

    class Wrapper(nn.Module):  # == SegmentationUNet / RegressionUNet
        def __init__(self):
            super().__init__()
            hidden = 128 if cfg.feature_channels > cfg.cond_dim else None
            self.projector = None if cfg.feature_channels == cfg.cond_dim else FeatureProjector(
                cfg.feature_channels, out_channels=cfg.cond_dim, hidden_channels=hidden)
            self.unet = MyUNetModel(in_channels=cfg.cond_dim, model_channels=cfg.model_channels,
                                    out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks,
                                    channel_mult=cfg.channel_mult, attention_resolutions=cfg.attention_resolutions,
                                    spatial_size=cfg.grid_size, dims=3, activation=nn.LeakyReLU(0.02))

        def forward(self, feat_grid):
            x = feat_grid
            if self.projector is not None:
                x = self.projector(feat_grid)
            return self.unet(x)

    return Wrapper().eval()


def run_reference(cfg: UNetConfig, wseed: int, feat: np.ndarray) -> np.ndarray:
    model = reference_model(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, wseed).items()}
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        return model(torch.from_numpy(feat)).numpy()


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])
    for name, (kw, wseed, iseed) in CASES.items():
        if only and name not in only:
            continue
        feat = feature_grid(kw["grid_size"], kw["feature_channels"], seed=iseed)
        res = {}
        for head, oc, off in HEADS:
            cfg = UNetConfig(out_channels=oc, **kw)
            res[head] = run_reference(cfg, wseed + off, feat)
            print(name, head, res[head].shape, float(np.abs(res[head]).mean()))
        np.savez_compressed(os.path.join(out_dir, f"unet_{name}.npz"), seg=res["seg"], cont=res["cont"],
                            weight_seed=wseed, input_seed=iseed)


if __name__ == "__main__":
    main()
