"""Structure of the reference U-Net as data: block list + state_dict keys and shapes.

Restates the constructor logic of MyUNetModel (third_party/Wavelet-Generation/models/module/
diffusion_network.py:712-873), MyResBlock (:639-710), AttentionBlock (:192-212), Upsample /
Downsample (:51-97), FeatureProjector (:534-589) and the two wrappers
(trainer/training_discrete.py:50-88, trainer/training_continuous_mse.py:48-89) as a flat plan
that both the HIP executor (pixie_amd/unet.py) and the synthetic-weight generator walk.
Key names equal the reference's nn.Module state_dict keys so reference checkpoints load.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclass
class UNetConfig:
    feature_channels: int = 64
    cond_dim: int = 32
    model_channels: int = 64
    num_res_blocks: int = 3
    channel_mult: Tuple[int, ...] = (1, 1, 2, 4)
    attention_resolutions: Tuple[int, ...] = ()
    grid_size: int = 32
    out_channels: int = 8  # num_classes for the segmentation net, 3 for the regression net

    @property
    def has_projector(self) -> bool:  # training_discrete.py:63-69
        return self.feature_channels != self.cond_dim

    @property
    def projector_hidden(self) -> Optional[int]:
        return 128 if self.feature_channels > self.cond_dim else None


@dataclass
class Block:
    kind: str                 # "conv_in" | "res" | "down" | "attn" | "up"
    prefix: str               # state_dict prefix, e.g. "unet.input_blocks.3.0"
    cin: int = 0
    cout: int = 0
    sp: int = 0               # spatial size the block runs at (input side)


@dataclass
class UNetPlan:
    cfg: UNetConfig
    input_blocks: List[List[Block]] = field(default_factory=list)   # one list per nn.Sequential
    middle: List[Block] = field(default_factory=list)
    output_blocks: List[List[Block]] = field(default_factory=list)
    skip_channels: List[int] = field(default_factory=list)          # channels pushed on hs, in push order
    out_sp: int = 0


def build_plan(cfg: UNetConfig) -> UNetPlan:
    """diffusion_network.py:760-873"""
    mc, nrb = cfg.model_channels, cfg.num_res_blocks
    plan = UNetPlan(cfg)
    plan.input_blocks.append([Block("conv_in", "unet.input_blocks.0.0", cfg.cond_dim, mc, cfg.grid_size)])
    chans = [mc]
    sizes = [cfg.grid_size]
    ch, ds, sp = mc, 1, cfg.grid_size
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(nrb):
            idx = len(plan.input_blocks)
            seq = [Block("res", f"unet.input_blocks.{idx}.0", ch, mult * mc, sp)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                seq.append(Block("attn", f"unet.input_blocks.{idx}.1", ch, ch, sp))
            plan.input_blocks.append(seq)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            idx = len(plan.input_blocks)
            plan.input_blocks.append([Block("down", f"unet.input_blocks.{idx}.0", ch, ch, sp)])
            chans.append(ch)
            sizes.append(sp)
            ds *= 2
            sp = (sp + 1) // 2
    plan.skip_channels = list(chans)
    plan.middle = [Block("res", "unet.middle_block.0", ch, ch, sp), Block("attn", "unet.middle_block.1", ch, ch, sp),
                   Block("res", "unet.middle_block.2", ch, ch, sp)]
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(nrb + 1):
            idx = len(plan.output_blocks)
            ich = chans.pop()
            seq = [Block("res", f"unet.output_blocks.{idx}.0", ch + ich, mc * mult, sp)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                seq.append(Block("attn", f"unet.output_blocks.{idx}.{len(seq)}", ch, ch, sp))
            if level and i == nrb:
                seq.append(Block("up", f"unet.output_blocks.{idx}.{len(seq)}", ch, ch, sp))
                ds //= 2
                sp = sizes.pop()
            plan.output_blocks.append(seq)
    plan.out_sp = sp
    return plan


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape, in the reference's registration order."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(prefix, cout, cin, k, dims=3):
        shapes[prefix + ".weight"] = (cout, cin) + (k,) * dims
        shapes[prefix + ".bias"] = (cout,)

    def norm(prefix, shape):
        shapes[prefix + ".weight"] = tuple(shape)
        shapes[prefix + ".bias"] = tuple(shape)

    if cfg.has_projector:  # diffusion_network.py:556-585
        hid = cfg.projector_hidden
        if hid is None:
            conv("projector.net.0", cfg.cond_dim, cfg.feature_channels, 1)
            norm("projector.net.1", (cfg.cond_dim,))
        else:
            conv("projector.net.0", hid, cfg.feature_channels, 1)
            norm("projector.net.1", (hid,))
            conv("projector.net.3", hid, hid, 3)
            norm("projector.net.4", (hid,))
            conv("projector.net.6", cfg.cond_dim, hid, 1)
            norm("projector.net.7", (cfg.cond_dim,))

    plan = build_plan(cfg)

    def emit(b: Block):
        if b.kind == "conv_in":
            conv(b.prefix, b.cout, b.cin, 3)
        elif b.kind == "res":  # :673-694
            norm(b.prefix + ".in_layers.0", (b.sp,) * 3)
            conv(b.prefix + ".in_layers.2", b.cout, b.cin, 3)
            norm(b.prefix + ".out_layers.0", (b.sp,) * 3)
            conv(b.prefix + ".out_layers.3", b.cout, b.cout, 3)
            if b.cin != b.cout:
                conv(b.prefix + ".skip_connection", b.cout, b.cin, 1)
        elif b.kind == "down":
            conv(b.prefix + ".op", b.cout, b.cin, 3)
        elif b.kind == "up":
            conv(b.prefix + ".conv", b.cout, b.cin, 3)
        elif b.kind == "attn":  # :199-208
            norm(b.prefix + ".norm", (b.cin,))
            conv(b.prefix + ".qkv", 3 * b.cin, b.cin, 1, dims=1)
            conv(b.prefix + ".proj_out", b.cin, b.cin, 1, dims=1)

    for seq in plan.input_blocks:
        for b in seq:
            emit(b)
    for b in plan.middle:
        emit(b)
    for seq in plan.output_blocks:
        for b in seq:
            emit(b)
    norm("unet.out.0", (plan.out_sp,) * 3)
    conv("unet.out.2", cfg.out_channels, cfg.model_channels, 3)
    return shapes


def is_norm_key(key: str) -> bool:
    parts = key.split(".")
    if key.startswith("projector.net."):
        return parts[2] in ("1", "4", "7")
    return (".in_layers.0." in key or ".out_layers.0." in key or ".norm." in key or key.startswith("unet.out.0."))


def synthetic_state_dict(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random-init weights of the reference architecture.

    Freshly constructed reference networks output exactly zero (zero_module on every ResBlock's second
    conv, the attention projection and the head, WG/models/module/nn.py:67-73), so every parameter --
    including those -- is drawn at random: conv weights/biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the
    torch default), normalisation weights 1 + 0.1 N(0,1), biases 0.1 N(0,1).  Each tensor has its own
    generator keyed by (seed, crc32(name)), so values do not depend on iteration order.
    """
    out: Dict[str, np.ndarray] = {}
    for key, shape in param_shapes(cfg).items():
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        if is_norm_key(key):
            if key.endswith(".weight"):
                arr = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
            else:
                arr = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        else:
            wkey = key[: key.rfind(".")] + ".weight"
            wshape = param_shapes_cache(cfg)[wkey]
            fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            arr = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        out[key] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


_SHAPE_CACHE: Dict[Tuple, Dict[str, Tuple[int, ...]]] = {}


def param_shapes_cache(cfg: UNetConfig):
    k = (cfg.feature_channels, cfg.cond_dim, cfg.model_channels, cfg.num_res_blocks, tuple(cfg.channel_mult),
         tuple(cfg.attention_resolutions), cfg.grid_size, cfg.out_channels)
    if k not in _SHAPE_CACHE:
        _SHAPE_CACHE[k] = param_shapes(cfg)
    return _SHAPE_CACHE[k]


def conv_flops(cfg: UNetConfig) -> int:
    """2*MACs of every convolution (+ 4*T^2*C per attention block) -- the algorithmic FLOPs of one
    forward pass used for the MFMA roofline."""
    plan = build_plan(cfg)
    total = 0
    D = cfg.grid_size
    if cfg.has_projector:
        hid = cfg.projector_hidden
        if hid is None:
            total += 2 * cfg.feature_channels * cfg.cond_dim * D ** 3
        else:
            total += 2 * cfg.feature_channels * hid * D ** 3 + 2 * 27 * hid * hid * D ** 3 + 2 * hid * cfg.cond_dim * D ** 3

    def blk(b: Block):
        v = b.sp ** 3
        if b.kind == "conv_in":
            return 2 * 27 * b.cin * b.cout * v
        if b.kind == "res":
            f = 2 * 27 * b.cin * b.cout * v + 2 * 27 * b.cout * b.cout * v
            if b.cin != b.cout:
                f += 2 * b.cin * b.cout * v
            return f
        if b.kind == "down":
            vo = ((b.sp + 1) // 2) ** 3
            return 2 * 27 * b.cin * b.cout * vo
        if b.kind == "up":
            return 2 * 27 * b.cin * b.cout * (2 * b.sp) ** 3
        if b.kind == "attn":
            return 2 * b.cin * 3 * b.cin * v + 4 * v * v * b.cin + 2 * b.cin * b.cin * v
        return 0

    for seq in plan.input_blocks + [plan.middle] + plan.output_blocks:
        for b in seq:
            total += blk(b)
    total += 2 * 27 * cfg.model_channels * cfg.out_channels * plan.out_sp ** 3
    return total
