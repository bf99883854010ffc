"""Drop-in `SegmentationUNet` / `RegressionUNet` running on libpixie_hip.so (MI355X).

Same constructor signatures, `forward(feat_grid)` contract and state_dict key names as the
reference wrappers (third_party/Wavelet-Generation/trainer/training_discrete.py:50-88,
trainer/training_continuous_mse.py:48-89) around FeatureProjector + MyUNetModel
(models/module/diffusion_network.py:534-589, :712-935), so `create_models` / `load_checkpoint` /
`process_batch` of trainer/inference_combined.py:81-126 work unchanged.

Execution model: the network is walked as a flat plan (pixie_amd/unet_plan.py) and every layer is
one call into the C ABI.  Each convolution consumes its normalisation + activation as a fused
prologue and its residual as a fused epilogue, `th.cat` and nearest-upsampling are index maths inside
the conv kernel, so the only tensors that touch HBM are conv outputs.  Host code (this file) only
sequences launches on the current stream; there is no CPU implementation of any operator here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ConvDesc, check
from .unet_plan import Block, UNetConfig, UNetPlan, build_plan, param_shapes, is_norm_key

ACT_NONE, ACT_LEAKY, ACT_SILU = 0, 1, 2
_ZERO_INIT_SUFFIXES = (".out_layers.3.weight", ".out_layers.3.bias", ".proj_out.weight", ".proj_out.bias",
                       "unet.out.2.weight", "unet.out.2.bias")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class HipOps:
    """Operator set backed by include/pixie_hip.h (section A).  All tensors are fp32 CUDA(HIP) tensors
    without batch dimension: activations (C, D, H, W), attention (C, T)."""

    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise _lib.PixieHipError("pixie_amd U-Net operators run on a HIP device only (no CPU fallback)")
        self.device = device
        self.lib = _lib.load()

    @property
    def stream(self):
        return _lib.current_stream_ptr()

    def pack_conv(self, weight: torch.Tensor) -> torch.Tensor:
        cout, cin = weight.shape[0], weight.shape[1]
        k = weight.shape[2]
        coutp = self.lib.pixie_conv_cout_padded(cout)
        w = weight.detach().to(self.device, torch.float32).contiguous()
        taps = 1
        for s in weight.shape[2:]:
            taps *= int(s)
        packed = torch.empty((taps, cin, coutp), device=self.device, dtype=torch.float32)
        check(self.lib.pixie_conv_pack_weights(_ptr(w), _ptr(packed), cout, cin, k, self.stream), "pixie_conv_pack_weights")
        return packed

    def conv(self, parts: Sequence[torch.Tensor], packed_w: torch.Tensor, bias: Optional[torch.Tensor], cout: int, ksize: int,
             stride: int = 1, upsample: bool = False, pro: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             affine: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, act: int = ACT_NONE,
             residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        x0 = parts[0]
        x1 = parts[1] if len(parts) > 1 else None
        cin0, d, h, w = x0.shape
        up = 2 if upsample else 1
        pad = 1 if ksize == 3 else 0
        od = (d * up + 2 * pad - ksize) // stride + 1
        oh = (h * up + 2 * pad - ksize) // stride + 1
        ow = (w * up + 2 * pad - ksize) // stride + 1
        out = torch.empty((cout, od, oh, ow), device=self.device, dtype=torch.float32)
        desc = ConvDesc()
        desc.d_in0 = x0.data_ptr(); desc.c0 = cin0
        desc.d_in1 = x1.data_ptr() if x1 is not None else None
        desc.c1 = x1.shape[0] if x1 is not None else 0
        desc.in_d, desc.in_h, desc.in_w = d, h, w
        desc.upsample = 1 if upsample else 0
        desc.stride = stride
        desc.ksize = ksize
        desc.d_pro_a = pro[0].data_ptr() if pro is not None else None
        desc.d_pro_b = pro[1].data_ptr() if pro is not None else None
        desc.d_gamma = affine[0].data_ptr() if affine is not None else None
        desc.d_beta = affine[1].data_ptr() if affine is not None else None
        desc.act = act
        desc.d_w = packed_w.data_ptr()
        desc.d_bias = bias.data_ptr() if bias is not None else None
        desc.c_out = cout
        desc.d_residual = residual.data_ptr() if residual is not None else None
        desc.d_out = out.data_ptr()
        check(self.lib.pixie_conv3d_forward(C.byref(desc), self.stream), "pixie_conv3d_forward")
        return out

    def channel_sums(self, x: torch.Tensor) -> torch.Tensor:
        c = x.shape[0]
        spatial = x.numel() // c
        sums = torch.empty((c, 2), device=self.device, dtype=torch.float64)
        check(self.lib.pixie_channel_sums(_ptr(x), c, spatial, _ptr(sums), self.stream), "pixie_channel_sums")
        return sums

    def norm_finalize(self, sums: torch.Tensor, spatial: int, mode: int, groups: int = 1, eps: float = 1e-5,
                      weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None):
        c = sums.shape[0]
        a = torch.empty(c, device=self.device, dtype=torch.float32)
        b = torch.empty(c, device=self.device, dtype=torch.float32)
        check(self.lib.pixie_norm_finalize(_ptr(sums), c, spatial, mode, groups, eps, _ptr(weight), _ptr(bias), _ptr(a), _ptr(b),
                                           self.stream), "pixie_norm_finalize")
        return a, b

    def attention(self, qkv: torch.Tensor, channels: int, tokens: int) -> torch.Tensor:
        out = torch.empty((channels, tokens), device=self.device, dtype=torch.float32)
        check(self.lib.pixie_attention_forward(_ptr(qkv), _ptr(out), channels, tokens, self.stream), "pixie_attention_forward")
        return out

    def combine(self, logits: torch.Tensor, cont: torch.Tensor):
        ncls = logits.shape[0]
        spatial = logits.numel() // ncls
        combined = torch.empty((3 + ncls,) + tuple(logits.shape[1:]), device=self.device, dtype=torch.float32)
        amax = torch.empty(tuple(logits.shape[1:]), device=self.device, dtype=torch.int32)
        check(self.lib.pixie_combine_predictions(_ptr(logits), ncls, _ptr(cont), spatial, _ptr(combined), _ptr(amax), self.stream),
              "pixie_combine_predictions")
        return combined, amax


class UNetRunner:
    """Walks the plan for one sample.  `ops` is HipOps in the product; tests inject a torch reference
    implementation of the same five operators to check the wiring on a CPU."""

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor], ops):
        self.cfg = cfg
        self.plan: UNetPlan = build_plan(cfg)
        self.p = params
        self.ops = ops
        self._packed: Dict[str, Tuple[int, int, torch.Tensor]] = {}

    # -- parameter helpers
    def _w(self, key: str) -> torch.Tensor:
        t = self.p[key + ".weight"]
        ent = self._packed.get(key)
        if ent is None or ent[0] != t.data_ptr() or ent[1] != t._version:
            self._packed[key] = (t.data_ptr(), t._version, self.ops.pack_conv(t))
        return self._packed[key][2]

    def _b(self, key: str) -> torch.Tensor:
        return self.p[key + ".bias"]

    def _sums(self, cache: dict, t: torch.Tensor) -> torch.Tensor:
        k = id(t)
        if k not in cache:
            cache[k] = (t, self.ops.channel_sums(t))  # keep t alive so id() stays unique
        return cache[k][1]

    # -- blocks
    def _res(self, b: Block, parts: List[torch.Tensor], cache: dict) -> torch.Tensor:
        """MyResBlock.forward, diffusion_network.py:696-705"""
        ops, p = self.ops, b.prefix
        spatial = parts[0][0].numel()
        sums = torch.cat([self._sums(cache, t) for t in parts], dim=0) if len(parts) > 1 else self._sums(cache, parts[0])
        pro = ops.norm_finalize(sums, spatial, 0)
        h = ops.conv(parts, self._w(p + ".in_layers.2"), self._b(p + ".in_layers.2"), b.cout, 3, pro=pro,
                     affine=(self.p[p + ".in_layers.0.weight"], self.p[p + ".in_layers.0.bias"]), act=ACT_LEAKY)
        pro2 = ops.norm_finalize(self._sums(cache, h), spatial, 0)
        if b.cin != b.cout:
            skip = ops.conv(parts, self._w(p + ".skip_connection"), self._b(p + ".skip_connection"), b.cout, 1)
        else:
            skip = parts[0]
        return ops.conv([h], self._w(p + ".out_layers.3"), self._b(p + ".out_layers.3"), b.cout, 3, pro=pro2,
                        affine=(self.p[p + ".out_layers.0.weight"], self.p[p + ".out_layers.0.bias"]), act=ACT_LEAKY,
                        residual=skip)

    def _attn(self, b: Block, x: torch.Tensor, cache: dict) -> torch.Tensor:
        """AttentionBlock._forward, diffusion_network.py:213-221"""
        ops, p = self.ops, b.prefix
        c = x.shape[0]
        spatial = x[0].numel()
        pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[p + ".norm.weight"],
                                bias=self.p[p + ".norm.bias"])
        qkv = ops.conv([x], self._w(p + ".qkv"), self._b(p + ".qkv"), 3 * c, 1, pro=pro)
        att = ops.attention(qkv.reshape(3 * c, spatial), c, spatial).reshape(x.shape)
        return ops.conv([att], self._w(p + ".proj_out"), self._b(p + ".proj_out"), c, 1, residual=x)

    def _block(self, b: Block, parts: List[torch.Tensor], cache: dict) -> torch.Tensor:
        ops = self.ops
        if b.kind == "res":
            return self._res(b, parts, cache)
        assert len(parts) == 1
        x = parts[0]
        if b.kind == "attn":
            return self._attn(b, x, cache)
        if b.kind == "down":
            return ops.conv([x], self._w(b.prefix + ".op"), self._b(b.prefix + ".op"), b.cout, 3, stride=2)
        if b.kind == "up":
            return ops.conv([x], self._w(b.prefix + ".conv"), self._b(b.prefix + ".conv"), b.cout, 3, upsample=True)
        raise ValueError(b.kind)

    def forward(self, feat: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """feat: (C_feat, D, H, W) -> (out_channels, D, H, W)"""
        cfg, ops = self.cfg, self.ops
        cache: dict = {}
        spatial = feat[0].numel()
        x = feat
        pro_in, act_in = None, ACT_NONE
        if cfg.has_projector:  # FeatureProjector.net, diffusion_network.py:556-585
            q = "projector.net."
            if cfg.projector_hidden is None:
                x = ops.conv([x], self._w(q + "0"), self._b(q + "0"), cfg.cond_dim, 1)
                pro_in = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=max(cfg.cond_dim // 2, 1),
                                           weight=self.p[q + "1.weight"], bias=self.p[q + "1.bias"])
                act_in = ACT_SILU
            else:
                hid = cfg.projector_hidden
                x = ops.conv([x], self._w(q + "0"), self._b(q + "0"), hid, 1)
                pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "1.weight"], bias=self.p[q + "1.bias"])
                x = ops.conv([x], self._w(q + "3"), self._b(q + "3"), hid, 3, pro=pro, act=ACT_SILU)
                pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "4.weight"], bias=self.p[q + "4.bias"])
                x = ops.conv([x], self._w(q + "6"), self._b(q + "6"), cfg.cond_dim, 1, pro=pro, act=ACT_SILU)
                pro_in = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "7.weight"], bias=self.p[q + "7.bias"])
        plan = self.plan
        hs: List[torch.Tensor] = []
        first = plan.input_blocks[0][0]
        h = ops.conv([x], self._w(first.prefix), self._b(first.prefix), first.cout, 3, pro=pro_in, act=act_in)
        hs.append(h)
        if taps is not None:
            taps["unet.input_blocks.0"] = h
        for seq in plan.input_blocks[1:]:
            for b in seq:
                h = self._block(b, [h], cache)
            hs.append(h)
            if taps is not None:
                taps[seq[0].prefix.rsplit(".", 1)[0]] = h
        for b in plan.middle:
            h = self._block(b, [h], cache)
            if taps is not None:
                taps[b.prefix] = h
        for seq in plan.output_blocks:
            skip = hs.pop()
            if tuple(skip.shape[1:]) != tuple(h.shape[1:]):
                raise NotImplementedError("odd grid sizes (the crop at diffusion_network.py:925-930) are not supported")
            parts = [h, skip]  # th.cat([h, hs.pop()], dim=1), :932 -- never materialised
            for b in seq:
                h = self._block(b, parts, cache)
                parts = [h]
            if taps is not None:
                taps[seq[0].prefix.rsplit(".", 1)[0]] = h
        pro = ops.norm_finalize(self._sums(cache, h), h[0].numel(), 0)
        return ops.conv([h], self._w("unet.out.2"), self._b("unet.out.2"), cfg.out_channels, 3, pro=pro,
                        affine=(self.p["unet.out.0.weight"], self.p["unet.out.0.bias"]), act=ACT_LEAKY)


class _Node(nn.Module):
    """Empty container used to reproduce the reference's state_dict key hierarchy."""


class _PixieUNet(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        shapes = param_shapes(cfg)
        gen = torch.Generator().manual_seed(0)
        for key, shape in shapes.items():
            parts = key.split(".")
            node = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            node.register_parameter(parts[-1], nn.Parameter(self._init(key, shape, shapes, gen), requires_grad=False))
        self._runner: Optional[UNetRunner] = None

    @staticmethod
    def _init(key, shape, shapes, gen) -> torch.Tensor:
        """Same distributions as the reference's construction: torch default conv init, identity norms,
        zero_module on the ResBlock second conv / attention projection / head (nn.py:67-73)."""
        if is_norm_key(key):
            return torch.ones(shape) if key.endswith(".weight") else torch.zeros(shape)
        if key.endswith(_ZERO_INIT_SUFFIXES):
            return torch.zeros(shape)
        wshape = shapes[key[: key.rfind(".")] + ".weight"]
        bound = 1.0 / math.sqrt(float(torch.tensor(wshape[1:]).prod()))
        return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound

    def _params(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in self.named_parameters()}

    def load_numpy_state(self, sd) -> None:
        """Load a {key: np.ndarray} dict such as pixie_amd.unet_plan.synthetic_state_dict."""
        self.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)

    @torch.no_grad()
    def forward(self, feat_grid: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """(N, feature_channels, D, H, W) float32 -> (N, out_channels, D, H, W)"""
        if feat_grid.dim() != 5 or feat_grid.shape[1] != self.cfg.feature_channels:
            raise ValueError(f"expected (N,{self.cfg.feature_channels},D,H,W), got {tuple(feat_grid.shape)}")
        dev = next(self.parameters()).device
        if dev.type != "cuda" or feat_grid.device != dev:
            raise _lib.PixieHipError("model and input must live on the same HIP device (no CPU fallback)")
        if self._runner is None or self._runner.ops.device != dev:
            self._runner = UNetRunner(self.cfg, self._params(), HipOps(dev))
        else:
            self._runner.p = self._params()
        x = feat_grid.detach().to(torch.float32).contiguous()
        outs = [self._runner.forward(x[n], taps if n == 0 else None) for n in range(x.shape[0])]
        return torch.stack(outs, dim=0)


class SegmentationUNet(_PixieUNet):
    """trainer/training_discrete.py:50-88"""

    def __init__(self, feature_channels: int, cond_dim: int, model_channels: int, num_res_blocks: int,
                 channel_mult: Tuple[int, ...], attention_resolutions: Tuple[int, ...], grid_size: int, num_classes: int):
        super().__init__(UNetConfig(feature_channels, cond_dim, model_channels, num_res_blocks, tuple(channel_mult),
                                    tuple(attention_resolutions), grid_size, num_classes))


class RegressionUNet(_PixieUNet):
    """trainer/training_continuous_mse.py:48-89"""

    def __init__(self, feature_channels: int, cond_dim: int, model_channels: int, num_res_blocks: int,
                 channel_mult: Tuple[int, ...], attention_resolutions: Tuple[int, ...], grid_size: int, out_channels: int = 3):
        super().__init__(UNetConfig(feature_channels, cond_dim, model_channels, num_res_blocks, tuple(channel_mult),
                                    tuple(attention_resolutions), grid_size, out_channels))


@torch.no_grad()
def predict_material_field(seg_network: SegmentationUNet, cont_network: RegressionUNet, feat_grid: torch.Tensor):
    """The compute of process_batch + save_predictions (trainer/inference_combined.py:122-126,186-195):
    returns (combined (N, 3+num_classes, D, H, W), seg_pred (N, D, H, W) int32, seg_logits, cont_pred)."""
    seg_logits = seg_network(feat_grid)
    cont_pred = cont_network(feat_grid)
    ops = seg_network._runner.ops
    combined, seg_pred = [], []
    for n in range(feat_grid.shape[0]):
        cmb, am = ops.combine(seg_logits[n].contiguous(), cont_pred[n].contiguous())
        combined.append(cmb)
        seg_pred.append(am)
    return torch.stack(combined), torch.stack(seg_pred), seg_logits, cont_pred
