"""Drop-in `SegmentationUNet` / `RegressionUNet` running on libpixie_hip.so (MI355X).

Same constructor signatures, `forward(feat_grid)` contract and state_dict key names as the
reference wrappers (third_party/Wavelet-Generation/trainer/training_discrete.py:50-88,
trainer/training_continuous_mse.py:48-89) around FeatureProjector + MyUNetModel
(models/module/diffusion_network.py:534-589, :712-935), so `create_models` / `load_checkpoint` /
`process_batch` of trainer/inference_combined.py:81-126 work unchanged.

Execution model: a network's forward pass is ONE call into the C ABI -- pixie_unet_forward (csrc/unet_exec.hip) owns the plan,
the workspace placement and the ~400-launch sequence -- replayed as a captured HIP graph by default (UNetHandle,
_PixieUNet._forward_graphed).  UNetRunner + HipOps below walk the same plan (pixie_amd/unet_plan.py) from Python with one
foreign call per operator: bit-identical, kept for the per-block `taps` hook and for the CPU wiring tests (which inject torch
stand-ins for the operators).  Either way each convolution consumes its normalisation + activation as a fused prologue and its
residual (and, where channel counts change, the block's 1x1x1 skip convolution) as a fused epilogue, `th.cat` and
nearest-upsampling are index maths inside the conv kernel, so the only tensors that touch HBM are conv outputs.  Host code
(this file) only sequences launches on the current stream; there is no CPU implementation of any operator here.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ConvDesc, check
from .unet_plan import Block, UNetConfig, UNetPlan, build_plan, param_shapes, is_norm_key

ACT_NONE, ACT_LEAKY, ACT_SILU = 0, 1, 2
# "f16x3": stride-1 convolutions with 16-aligned channel counts run on the f16 matrix cores with every fp32 operand
# split into fp16 hi+lo (three MFMAs per product, fp32 accumulate; csrc/conv3d_f16x3.hip).  "f32": every convolution
# on the exact-fp32 MFMA kernel (csrc/conv3d_mfma.hip).  Both are HIP paths; there is no CPU path.
DEFAULT_PRECISION = os.environ.get("PIXIE_CONV_PRECISION", "f16x3")
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
        self.split_k = os.environ.get("PIXIE_CONV_SPLIT_K", "1") != "0"
        self.record_variant = False      # set by profilers (bench.py): conv() then leaves the kernel instantiation in last_variant
        self.last_variant = (0, 1)

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

    def pack_conv16(self, weight: torch.Tensor) -> torch.Tensor:
        """fp16 hi/lo split + MFMA A-operand swizzle of a conv weight (device side, no host sync)."""
        cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        nbytes = self.lib.pixie_conv_packed16_bytes(cout, cin, k)
        if nbytes <= 0:
            raise _lib.PixieHipError(f"f16x3 packing needs c_in % 8 == 0 (got {cin})")
        w = weight.detach().to(self.device, torch.float32).contiguous()
        packed = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
        check(self.lib.pixie_conv_pack_weights_f16x2(_ptr(w), _ptr(packed), cout, cin, k, self.stream), "pixie_conv_pack_weights_f16x2")
        return packed

    @staticmethod
    def f16x3_ok(parts: Sequence[torch.Tensor], stride: int) -> bool:
        cin = sum(int(p.shape[0]) for p in parts)
        return stride in (1, 2) and cin % 16 == 0 and int(parts[0].shape[0]) % 8 == 0   # stride 2: the 3^3 Downsample convs

    def conv(self, parts: Sequence[torch.Tensor], packed_w: Optional[torch.Tensor], bias: Optional[torch.Tensor], cout: int, ksize: int,
             stride: int = 1, upsample: bool = False, pro: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             affine: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, act: int = ACT_NONE,
             residual: Optional[torch.Tensor] = None, w16: Optional[torch.Tensor] = None,
             in_amax: Optional[Sequence[torch.Tensor]] = None, in_bound: float = 0.0,
             out_amax: Optional[torch.Tensor] = None, out_size: Optional[Tuple[int, int, int]] = None,
             skip: Optional[dict] = None):
        """Returns the output tensor; with `out_amax` (f16x3 path) returns (output, channel sums float64 (c_out, 2))
        computed in the conv epilogue, and atomicMax'es |output|max into out_amax."""
        x0 = parts[0]
        x1 = parts[1] if len(parts) > 1 else None
        cin0, d, h, w = x0.shape
        up = 2 if upsample else 1
        pad = 1 if ksize == 3 else 0
        od = (d * up + 2 * pad - ksize) // stride + 1
        oh = (h * up + 2 * pad - ksize) // stride + 1
        ow = (w * up + 2 * pad - ksize) // stride + 1
        desc = ConvDesc()
        if out_size is not None:   # odd-grid crop (diffusion_network.py:925-930): the cropped voxels are never computed
            od, oh, ow = min(od, int(out_size[0])), min(oh, int(out_size[1])), min(ow, int(out_size[2]))
            desc.out_d, desc.out_h, desc.out_w = od, oh, ow
        out = torch.empty((cout, od, oh, ow), device=self.device, dtype=torch.float32)
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
        desc.d_w = packed_w.data_ptr() if packed_w is not None else None
        desc.d_w16 = w16.data_ptr() if w16 is not None else None
        desc.d_in_amax0 = in_amax[0].data_ptr() if in_amax else None
        desc.d_in_amax1 = in_amax[1].data_ptr() if in_amax and len(in_amax) > 1 else None
        desc.in_bound = float(in_bound)
        desc.d_bias = bias.data_ptr() if bias is not None else None
        desc.c_out = cout
        desc.d_residual = residual.data_ptr() if residual is not None else None
        desc.d_out = out.data_ptr()
        if skip is not None:   # folded 1x1x1 skip convolution: dict(parts, w16, bias, amax)
            sp = skip["parts"]
            desc.d_skip_in0 = sp[0].data_ptr(); desc.skip_c0 = sp[0].shape[0]
            desc.d_skip_in1 = sp[1].data_ptr() if len(sp) > 1 else None
            desc.skip_c1 = sp[1].shape[0] if len(sp) > 1 else 0
            desc.d_skip_w16 = skip["w16"].data_ptr()
            desc.d_skip_bias = skip["bias"].data_ptr() if skip["bias"] is not None else None
            desc.d_skip_amax0 = skip["amax"][0].data_ptr()
            desc.d_skip_amax1 = skip["amax"][1].data_ptr() if len(sp) > 1 else None
        sums = None
        if w16 is not None and self.split_k:
            wsb = self.lib.pixie_conv_workspace_bytes(C.byref(desc))
            if wsb > 0:   # small-output layer: split-K scratch (torch's caching allocator makes this a pointer bump)
                workspace = torch.empty(wsb, device=self.device, dtype=torch.uint8)
                desc.d_workspace = workspace.data_ptr()
        if out_amax is not None and w16 is not None:
            nfl = self.lib.pixie_conv_stats_floats(C.byref(desc))
            if nfl > 0:
                stats = torch.empty(nfl, device=self.device, dtype=torch.float32)
                desc.d_out_stats = stats.data_ptr()
                desc.d_out_amax = out_amax.data_ptr()
        if self.record_variant:   # profilers: which kernel instantiation this launch is (grouping key of rocprofv3)
            sl = C.c_int(1)
            self.last_variant = (int(_lib.load(diag=True).pixie_conv_kernel_variant(C.byref(desc), C.byref(sl))), int(sl.value))   # a pure function of the descriptor
        check(self.lib.pixie_conv3d_forward(C.byref(desc), self.stream), "pixie_conv3d_forward")
        if desc.d_out_stats:
            sums = torch.empty((cout, 2), device=self.device, dtype=torch.float64)
            check(self.lib.pixie_stats_finalize(_ptr(stats), C.byref(desc), _ptr(sums), self.stream), "pixie_stats_finalize")
        if out_amax is not None:
            return out, sums
        return out

    def skip_foldable(self, x: torch.Tensor, cout: int, ksize: int, skip_parts: Sequence[torch.Tensor]) -> bool:
        """Would the f16x3 launch of a stride-1 `ksize`^3 convolution x -> cout take a folded 1x1x1 skip convolution over
        skip_parts (pixie_conv_skip_foldable: channel alignment, and the layer must not be a split-K one)?"""
        desc = ConvDesc()
        desc.c0 = x.shape[0]
        desc.in_d, desc.in_h, desc.in_w = (int(v) for v in x.shape[1:])
        desc.stride, desc.ksize, desc.c_out = 1, ksize, cout
        desc.d_w16 = 1                       # "f16x3 weights will be given"
        desc.d_workspace = 1 if self.split_k else None
        desc.skip_c0 = skip_parts[0].shape[0]
        desc.skip_c1 = skip_parts[1].shape[0] if len(skip_parts) > 1 else 0
        return bool(self.lib.pixie_conv_skip_foldable(C.byref(desc)))

    def channel_sums(self, x: torch.Tensor) -> torch.Tensor:
        c = x.shape[0]
        spatial = x.numel() // c
        sums = torch.empty((c, 2), device=self.device, dtype=torch.float64)
        check(self.lib.pixie_channel_sums(_ptr(x), c, spatial, _ptr(sums), self.stream), "pixie_channel_sums")
        return sums

    def channel_stats(self, x: torch.Tensor, amax_slot: torch.Tensor) -> torch.Tensor:
        """channel_sums + the tensor's |x|max (float bits) atomicMax'ed into amax_slot (a zeroed int32[1] view)."""
        c = x.shape[0]
        spatial = x.numel() // c
        sums = torch.empty((c, 2), device=self.device, dtype=torch.float64)
        check(self.lib.pixie_channel_stats(_ptr(x), c, spatial, _ptr(sums), _ptr(amax_slot), self.stream), "pixie_channel_stats")
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

    def projector_conv0(self, grid_dhwc_f16: torch.Tensor, nets: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]], cout: int):
        """FeatureProjector.net[0] of one or two networks straight from the (D, H, W, C) float16 voxel grid (one read of the
        grid): nets = [(packed f16x2 weight, bias), ...] -> [(cout, D, H, W) float32, ...]."""
        d, h, w, c = (int(v) for v in grid_dhwc_f16.shape)
        g = grid_dhwc_f16.contiguous()
        outs = [torch.empty((cout, d, h, w), device=self.device, dtype=torch.float32) for _ in nets]
        n = len(nets)
        w_arr = (C.c_void_p * n)(*[t[0].data_ptr() for t in nets])
        b_arr = (C.c_void_p * n)(*[(t[1].data_ptr() if t[1] is not None else None) for t in nets])
        o_arr = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        check(self.lib.pixie_projector_conv0(_ptr(g), d * h * w, c, n, w_arr, b_arr, o_arr, cout, self.stream), "pixie_projector_conv0")
        return outs

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

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor], ops, precision: Optional[str] = None):
        self.cfg = cfg
        self.plan: UNetPlan = build_plan(cfg)
        self.p = params
        self.ops = ops
        self.precision = precision or DEFAULT_PRECISION
        if self.precision not in ("f16x3", "f32"):
            raise ValueError(f"unknown conv precision {self.precision!r}")
        self._packed: Dict[str, Tuple[int, int, torch.Tensor]] = {}
        self._packed16: Dict[str, Tuple[int, int, torch.Tensor]] = {}
        self._bounds: Dict[str, Tuple[int, int, float, float]] = {}
        self.fuse_stats = os.environ.get("PIXIE_FUSE_STATS", "1") != "0"   # channel statistics in the conv epilogue
        self.fold_skip = os.environ.get("PIXIE_FOLD_SKIP", "1") != "0"     # skip_connection 1x1x1 inside the block's second conv

    @property
    def _f16x3(self) -> bool:
        return self.precision == "f16x3" and hasattr(self.ops, "pack_conv16")

    # -- parameter helpers
    def _w(self, key: str) -> torch.Tensor:
        t = self.p[key + ".weight"]
        ent = self._packed.get(key)
        if ent is None or ent[0] != t.data_ptr() or ent[1] != t._version:
            self._packed[key] = (t.data_ptr(), t._version, self.ops.pack_conv(t))
        return self._packed[key][2]

    def _w16(self, key: str) -> torch.Tensor:
        t = self.p[key + ".weight"]
        ent = self._packed16.get(key)
        if ent is None or ent[0] != t.data_ptr() or ent[1] != t._version:
            self._packed16[key] = (t.data_ptr(), t._version, self.ops.pack_conv16(t))
        return self._packed16[key][2]

    def _b(self, key: str) -> torch.Tensor:
        return self.p[key + ".bias"]

    def _absmax(self, key: str) -> Tuple[float, float]:
        """(max|weight|, max|bias|) of a normalisation layer, cached per parameter version (one host sync each)."""
        w, bb = self.p[key + ".weight"], self.p[key + ".bias"]
        ent = self._bounds.get(key)
        if ent is None or ent[0] != w._version or ent[1] != bb._version:
            self._bounds[key] = (w._version, bb._version, float(w.abs().max()), float(bb.abs().max()))
            ent = self._bounds[key]
        return ent[2], ent[3]

    def _norm_bound(self, key: str, count: int) -> float:
        """Bound on |normalised * weight + bias| when `count` elements share the statistics: a standardised sample
        of n values cannot exceed sqrt(n-1) in magnitude.  LeakyReLU/SiLU do not increase magnitudes."""
        wmax, bmax = self._absmax(key)
        return math.sqrt(float(count)) * wmax + bmax + 1e-30

    _SLOT_POOL = 1024

    @classmethod
    def _new_slot(cls, cache: dict, device) -> torch.Tensor:
        """One zeroed int32 device word (an atomicMax target for a tensor's |x|max).  Words come from pools of
        _SLOT_POOL; a network deeper than one pool gets another pool, never an empty slice."""
        pool = cache.get("_slots")
        if pool is None or cache["_next"] >= pool.numel():
            pool = cache["_slots"] = torch.zeros(cls._SLOT_POOL, dtype=torch.int32, device=device)
            cache["_next"] = 0
        i = cache["_next"]
        cache["_next"] = i + 1
        return pool[i:i + 1]

    @staticmethod
    def _remember(cache: dict, t: torch.Tensor, sums, slot) -> None:
        """Statistics of `t`, keyed by id(t) WITHOUT keeping t alive: the entry disappears with the tensor, so only
        the tensors the graph still needs (the skip stack) stay resident during a forward pass."""
        k = id(t)
        cache[k] = (sums, slot)
        weakref.finalize(t, cache.pop, k, None)

    def _stats(self, cache: dict, t: torch.Tensor):
        k = id(t)
        if k not in cache:
            if hasattr(self.ops, "channel_stats"):
                slot = self._new_slot(cache, t.device)
                self._remember(cache, t, self.ops.channel_stats(t, slot), slot)
            else:
                self._remember(cache, t, self.ops.channel_sums(t), None)
        return cache[k]

    def _sums(self, cache: dict, t: torch.Tensor) -> torch.Tensor:
        return self._stats(cache, t)[0]

    def _amax(self, cache: dict, t: torch.Tensor) -> torch.Tensor:
        return self._stats(cache, t)[1]

    def _conv(self, cache: dict, parts: List[torch.Tensor], wkey: str, cout: int, ksize: int, *, stride: int = 1,
              upsample: bool = False, pro=None, affine_key: Optional[str] = None, act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None, bound: float = 0.0, out_size=None, skip=None) -> torch.Tensor:
        """One convolution launch.  `bound` bounds the magnitude of the prologue's output (needed by the f16x3
        kernel to place the tensor in the fp16 range); raw inputs use their device-side |x|max instead."""
        ops = self.ops
        affine = (self.p[affine_key + ".weight"], self.p[affine_key + ".bias"]) if affine_key else None
        if not (self._f16x3 and ops.f16x3_ok(parts, stride)):
            return ops.conv(parts, self._w(wkey), self._b(wkey), cout, ksize, stride=stride, upsample=upsample, pro=pro,
                            affine=affine, act=act, residual=residual, **({"out_size": out_size} if out_size is not None else {}))
        if pro is None and affine is None:
            kw = dict(in_amax=[self._amax(cache, t) for t in parts])
        else:
            kw = dict(in_bound=bound)
        if out_size is not None:
            kw["out_size"] = out_size
        if skip is not None:
            kw["skip"] = skip
        if self.fuse_stats:
            # the output's channel sums and |x|max come out of the conv epilogue: no separate pass over the tensor
            slot = self._new_slot(cache, parts[0].device)
            out, sums = ops.conv(parts, None, self._b(wkey), cout, ksize, stride=stride, upsample=upsample, pro=pro, affine=affine,
                                 act=act, residual=residual, w16=self._w16(wkey), out_amax=slot, **kw)
            if sums is not None:
                self._remember(cache, out, sums, slot)
            return out
        return ops.conv(parts, None, self._b(wkey), cout, ksize, stride=stride, upsample=upsample, pro=pro, affine=affine,
                        act=act, residual=residual, w16=self._w16(wkey), **kw)

    # -- blocks
    def _res(self, b: Block, parts: List[torch.Tensor], cache: dict) -> torch.Tensor:
        """MyResBlock.forward, diffusion_network.py:696-705"""
        ops, p = self.ops, b.prefix
        spatial = parts[0][0].numel()
        sums = torch.cat([self._sums(cache, t) for t in parts], dim=0) if len(parts) > 1 else self._sums(cache, parts[0])
        pro = ops.norm_finalize(sums, spatial, 0)
        h = self._conv(cache, parts, p + ".in_layers.2", b.cout, 3, pro=pro, affine_key=p + ".in_layers.0", act=ACT_LEAKY,
                       bound=self._norm_bound(p + ".in_layers.0", spatial))
        pro2 = ops.norm_finalize(self._sums(cache, h), spatial, 0)
        fold = None
        if b.cin != b.cout:
            if (self.fold_skip and self._f16x3 and ops.f16x3_ok([h], 1) and ops.f16x3_ok(parts, 1)
                    and ops.skip_foldable(h, b.cout, 3, parts)):
                # out = conv(h) + skip_connection(x) in ONE launch: the 1x1x1 convolution rides in the accumulators of the
                # second 3^3 convolution, the skip tensor never exists (csrc/conv3d_f16x3.hip, "folded skip")
                skip = None
                fold = dict(parts=parts, w16=self._w16(p + ".skip_connection"), bias=self._b(p + ".skip_connection"),
                            amax=[self._amax(cache, t) for t in parts])
            else:
                skip = self._conv(cache, parts, p + ".skip_connection", b.cout, 1)
        else:
            skip = parts[0]
        return self._conv(cache, [h], p + ".out_layers.3", b.cout, 3, pro=pro2, affine_key=p + ".out_layers.0", act=ACT_LEAKY,
                          residual=skip, bound=self._norm_bound(p + ".out_layers.0", spatial), skip=fold)

    def _attn(self, b: Block, x: torch.Tensor, cache: dict) -> torch.Tensor:
        """AttentionBlock._forward, diffusion_network.py:213-221"""
        ops, p = self.ops, b.prefix
        c = x.shape[0]
        spatial = x[0].numel()
        pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[p + ".norm.weight"],
                                bias=self.p[p + ".norm.bias"])
        qkv = self._conv(cache, [x], p + ".qkv", 3 * c, 1, pro=pro, bound=self._norm_bound(p + ".norm", spatial * (c // 32)))
        att = ops.attention(qkv.reshape(3 * c, spatial), c, spatial).reshape(x.shape)
        return self._conv(cache, [att], p + ".proj_out", c, 1, residual=x)

    def _block(self, b: Block, parts: List[torch.Tensor], cache: dict, up_size=None) -> torch.Tensor:
        if b.kind == "res":
            return self._res(b, parts, cache)
        assert len(parts) == 1
        x = parts[0]
        if b.kind == "attn":
            return self._attn(b, x, cache)
        if b.kind == "down":
            return self._conv(cache, [x], b.prefix + ".op", b.cout, 3, stride=2)
        if b.kind == "up":
            # `up_size`: spatial size of the skip tensor this output will be concatenated with; on odd grids it is one
            # less than 2 * sp and the reference crops h[..., :-1] per axis (diffusion_network.py:925-930)
            return self._conv(cache, [x], b.prefix + ".conv", b.cout, 3, upsample=True, out_size=up_size)
        raise ValueError(b.kind)

    def _stem(self, feat: Optional[torch.Tensor], cache: dict, proj0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """FeatureProjector (diffusion_network.py:556-585) + input_blocks[0] (:762): the projector's last normalisation is
        applied in the prologue of the first U-Net convolution, so its output is never materialised."""
        cfg, ops = self.cfg, self.ops
        spatial = (proj0 if proj0 is not None else feat)[0].numel()
        x = feat
        pro_in, act_in, bound_in = None, ACT_NONE, 0.0
        if cfg.has_projector:
            q = "projector.net."
            if cfg.projector_hidden is None:
                g = max(cfg.cond_dim // 2, 1)
                x = self._conv(cache, [x], q + "0", cfg.cond_dim, 1)
                pro_in = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=g,
                                           weight=self.p[q + "1.weight"], bias=self.p[q + "1.bias"])
                act_in = ACT_SILU
                bound_in = self._norm_bound(q + "1", spatial * (cfg.cond_dim // g))
            else:
                hid = cfg.projector_hidden
                x = proj0 if proj0 is not None else self._conv(cache, [x], q + "0", hid, 1)
                pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "1.weight"], bias=self.p[q + "1.bias"])
                x = self._conv(cache, [x], q + "3", hid, 3, pro=pro, act=ACT_SILU, bound=self._norm_bound(q + "1", spatial * (hid // 32)))
                pro = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "4.weight"], bias=self.p[q + "4.bias"])
                x = self._conv(cache, [x], q + "6", cfg.cond_dim, 1, pro=pro, act=ACT_SILU, bound=self._norm_bound(q + "4", spatial * (hid // 32)))
                pro_in = ops.norm_finalize(self._sums(cache, x), spatial, 1, groups=32, weight=self.p[q + "7.weight"], bias=self.p[q + "7.bias"])
                bound_in = self._norm_bound(q + "7", spatial * max(cfg.cond_dim // 32, 1))
        first = self.plan.input_blocks[0][0]
        return self._conv(cache, [x], first.prefix, first.cout, 3, pro=pro_in, act=act_in, bound=bound_in)

    def _head(self, h: torch.Tensor, cache: dict) -> torch.Tensor:
        """unet.out: LayerNorm -> LeakyReLU -> Conv3d (diffusion_network.py:869-873, :934)"""
        pro = self.ops.norm_finalize(self._sums(cache, h), h[0].numel(), 0)
        return self._conv(cache, [h], "unet.out.2", self.cfg.out_channels, 3, pro=pro, affine_key="unet.out.0", act=ACT_LEAKY,
                          bound=self._norm_bound("unet.out.0", h[0].numel()))

    def forward(self, feat: Optional[torch.Tensor], taps: Optional[dict] = None, proj0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat: (C_feat, D, H, W) -> (out_channels, D, H, W).  `proj0`: the output of projector.net[0] computed elsewhere
        (pixie_projector_conv0 on the channels-last grid, shared with the other network); the pass then starts behind it.
        `taps`, if a dict, receives the output of every block under the reference's module path (a dict that already holds
        keys receives only those: large grids keep few tensors alive)."""
        cache: dict = {}
        only = set(taps) if taps else None

        def tap(key, t):
            if taps is not None and (only is None or key in only):
                taps[key] = t

        plan = self.plan
        hs: List[torch.Tensor] = []
        h = self._stem(feat, cache, proj0)
        hs.append(h)
        tap("unet.input_blocks.0", h)
        for seq in plan.input_blocks[1:]:
            for b in seq:
                h = self._block(b, [h], cache)
            hs.append(h)
            tap(seq[0].prefix.rsplit(".", 1)[0], h)
        for b in plan.middle:
            h = self._block(b, [h], cache)
            tap(b.prefix, h)
        for seq in plan.output_blocks:
            skip = hs.pop()
            assert tuple(skip.shape[1:]) == tuple(h.shape[1:]), (skip.shape, h.shape)   # the up-conv already cropped
            parts = [h, skip]  # th.cat([h, hs.pop()], dim=1), :932 -- never materialised
            for b in seq:
                h = self._block(b, parts, cache, up_size=tuple(hs[-1].shape[1:]) if (b.kind == "up" and hs) else None)
                parts = [h]
            tap(seq[0].prefix.rsplit(".", 1)[0], h)
        return self._head(h, cache)


class UNetHandle:
    """One network behind the C handle of include/pixie_hip.h section (A'): pixie_unet_create / set_param / forward.  The
    whole forward pass is ONE foreign call; the plan walk, the temporaries (one workspace tensor) and the ~400 launches are
    the library's (csrc/unet_exec.hip).  Bit-identical to UNetRunner + HipOps, which it replaces as the default executor."""

    def __init__(self, cfg: UNetConfig, precision: str, device: torch.device):
        if device.type != "cuda":
            raise _lib.PixieHipError("pixie_amd U-Net runs on a HIP device only (no CPU fallback)")
        if precision not in ("f16x3", "f32"):
            raise ValueError(f"unknown conv precision {precision!r}")
        self.lib = _lib.load()
        self.cfg, self.precision, self.device = cfg, precision, device
        c = _lib.UNetConfigC()
        c.feature_channels, c.cond_dim, c.model_channels, c.num_res_blocks = cfg.feature_channels, cfg.cond_dim, cfg.model_channels, cfg.num_res_blocks
        c.n_channel_mult = len(cfg.channel_mult)
        for i, m in enumerate(cfg.channel_mult):
            c.channel_mult[i] = int(m)
        c.n_attention_resolutions = len(cfg.attention_resolutions)
        for i, m in enumerate(cfg.attention_resolutions):
            c.attention_resolutions[i] = int(m)
        c.grid_size, c.out_channels, c.precision = cfg.grid_size, cfg.out_channels, 0 if precision == "f16x3" else 1
        self._h = C.c_void_p()
        check(self.lib.pixie_unet_create(C.byref(self._h), C.byref(c)), "pixie_unet_create")
        weakref.finalize(self, self.lib.pixie_unet_destroy, self._h)
        self._seen: Dict[str, Tuple[int, int]] = {}
        self._keep: Dict[str, torch.Tensor] = {}      # the tensors the handle points at stay alive here

    def keys(self) -> List[str]:
        out = []
        for i in range(self.lib.pixie_unet_param_count(self._h)):
            key = C.c_char_p()
            check(self.lib.pixie_unet_param_info(self._h, i, C.byref(key), None, None, None), "pixie_unet_param_info")
            out.append(key.value.decode())
        return out

    def load(self, params: Dict[str, torch.Tensor]) -> None:
        """load_state_dict: hand every parameter that changed since the last call (by storage and version) to the handle."""
        for key, t in params.items():
            tag = (t.data_ptr(), t._version)
            if self._seen.get(key) == tag:
                continue
            v = t.detach()
            if v.device != self.device or v.dtype != torch.float32 or not v.is_contiguous():
                v = v.to(self.device, torch.float32).contiguous()
            self._keep[key] = v
            check(self.lib.pixie_unet_set_param(self._h, key.encode(), _ptr(v), v.numel()), "pixie_unet_set_param")
            self._seen[key] = tag

    def workspace_bytes(self, d: int, h: int, w: int) -> int:
        n = int(self.lib.pixie_unet_workspace_bytes(self._h, d, h, w))
        if n < 0:
            raise _lib.PixieHipError(self.lib.pixie_last_error().decode())
        return n

    def forward(self, feat: Optional[torch.Tensor], proj0: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat (C_feat, D, H, W) float32 (or None with proj0 = the output of projector.net[0]) -> (out_channels, D, H, W).
        `workspace`: a caller-owned uint8 device tensor of at least workspace_bytes(d, h, w) bytes (any content); default:
        allocated per call."""
        src = proj0 if proj0 is not None else feat
        d, h, w = (int(v) for v in src.shape[1:])
        out = torch.empty((self.cfg.out_channels, d, h, w), device=self.device, dtype=torch.float32)
        nbytes = self.workspace_bytes(d, h, w)
        if workspace is None:
            workspace = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
        elif workspace.dtype != torch.uint8 or workspace.device != self.device or workspace.numel() < nbytes or not workspace.is_contiguous():
            raise ValueError(f"workspace must be a contiguous uint8 tensor of >= {nbytes} bytes on {self.device}")
        check(self.lib.pixie_unet_forward(self._h, _ptr(feat), _ptr(proj0), d, h, w, _ptr(out), _ptr(workspace), nbytes,
                                          _lib.current_stream_ptr()), "pixie_unet_forward")
        return out


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
        self._handle: Optional[UNetHandle] = None
        # "c" (default): the whole pass is one call into the library's own executor (pixie_unet_forward);
        # "python": this file walks the plan and calls one operator at a time (the same launches; needed for `taps`)
        self.executor = os.environ.get("PIXIE_UNET_EXECUTOR", "c")
        self.conv_precision = DEFAULT_PRECISION  # "f16x3" (default) or "f32" (exact-fp32 MFMA everywhere)
        # replay a captured HIP graph per (shape, precision, parameter versions): the device runs the ~400 launches back to back
        # (128^3: 49.1 -> 45.5 ms per network) and the host queues ONE launch (16^3: 2.3 -> 0.7 ms).  PIXIE_UNET_GRAPH=0: eager.
        self.use_graph = os.environ.get("PIXIE_UNET_GRAPH", "1") == "1"
        self._graphs: dict = {}
        self._plist = None

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
        # the library launches on the CURRENT device's stream: make the model's device current for the call, as torch's own modules
        # effectively do (a caller that never called torch.cuda.set_device(rank) would otherwise launch on GPU 0 with GPU-k pointers)
        with torch.cuda.device(dev):
            self._prepare(dev)
            x = feat_grid.detach().to(torch.float32).contiguous()
            if self.use_graph and taps is None:
                return torch.stack([self._forward_graphed(x[n]) for n in range(x.shape[0])], dim=0)
            outs = [self._forward_one(x[n], taps=taps if n == 0 else None) for n in range(x.shape[0])]
            return torch.stack(outs, dim=0)

    def _prepare(self, dev) -> None:
        """(Re)bind the executors to the current parameters, device and precision."""
        params = self._params()
        if self._runner is None or self._runner.ops.device != dev or self._runner.precision != self.conv_precision:
            self._runner = UNetRunner(self.cfg, params, HipOps(dev), precision=self.conv_precision)
        else:
            self._runner.p = params
        if self.executor == "c":
            if self._handle is None or self._handle.device != dev or self._handle.precision != self.conv_precision:
                self._handle = UNetHandle(self.cfg, self.conv_precision, dev)
            self._handle.load(params)
        elif self.executor != "python":
            raise ValueError(f"unknown executor {self.executor!r} (PIXIE_UNET_EXECUTOR is 'c' or 'python')")

    def _forward_one(self, x: Optional[torch.Tensor], taps: Optional[dict] = None, proj0: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.executor == "c" and taps is None:
            return self._handle.forward(x, proj0)
        return self._runner.forward(x, taps, proj0)

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): parameter storage may move
        self._plist = None
        return super()._apply(fn, *args, **kwargs)

    def _param_list(self) -> List[torch.Tensor]:
        """The parameters in registration order.  Walking the module tree costs more than the rest of a 16^3 replay's host
        work, so what is cached is WHERE each parameter lives -- (the owning module's `_parameters` dict, name) -- and every
        call reads the current objects out of those dicts: a parameter that was REPLACED (load_state_dict(assign=True),
        `module.weight = nn.Parameter(...)`) is seen at once and changes the graph key (a cached list of the
        objects themselves would have replayed stale weights)."""
        refs = self.__dict__.get("_plist")
        if refs is None:
            refs = self._plist = [(m._parameters, name) for m in self.modules() for name, p in m._parameters.items() if p is not None]
        return [d[name] for d, name in refs]

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._plist = None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _forward_graphed(self, x: Optional[torch.Tensor], proj0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One sample through a captured HIP graph.  A forward pass is ~400 kernel launches; its launch sequence for a given
        (network, input shape, precision, parameter version) never changes, so it is captured once -- after one eager pass that
        packs the weights and takes the host-side parameter bounds -- and replayed with a single hipGraphLaunch: the device
        runs the kernels back to back and the host queues one launch.  The capture reads the CALLER's input tensor in place
        (keyed by its address: a serving loop that refills one buffer, as bench.py and inference_combined.py's batch loop
        do, never copies the 0.5 GB grid); a caller that brings a new address every call gets a graph that reads a private input
        buffer each call copies into.  Which of the two is decided at the SECOND call (the first runs eagerly), so nobody pays
        for two captures.  Returns a copy of the graph's static output."""
        src = proj0 if proj0 is not None else x
        # (layout and dtype are part of the key: a captured graph reads the caller's memory as laid out at capture time, and a
        # permuted / expanded view of the same address and shape must not replay it -- ADVICE r3)
        base = (tuple(src.shape), tuple(src.stride()), src.dtype, proj0 is not None, self.executor, self.conv_precision, src.device.index,
                tuple((id(p), p.data_ptr(), p._version) for p in self._param_list()))   # identity, storage AND version: `p.data = t` / `.to()` keep the version
        run = (lambda t: self._forward_one(None, proj0=t)) if proj0 is not None else (lambda t: self._forward_one(t))
        if self._graphs.get("base") != base:           # a new shape / parameter version invalidates every capture
            self._graphs = {"base": base, "first_ptr": None}
        ent = self._graphs.get(src.data_ptr()) or self._graphs.get("copy")
        if ent is None:
            out = run(src)                             # eager: weight packing, bounds, function attributes -- and this call's result
            if self._graphs["first_ptr"] is None:      # first call: remember the address, capture when it is known whether it comes back
                self._graphs["first_ptr"] = src.data_ptr()
                return out
            # second call: the same buffer again -> capture in place; another one -> capture on a private input buffer, for
            # good (at most two captures; each owns a workspace: 4 GB at 128^3, 30 GB at 256^3)
            # (a first-projector output handed in by the fused grid path is a fresh tensor every call: never in place)
            in_place = (proj0 is None and self._graphs["first_ptr"] == src.data_ptr() and "copy" not in self._graphs
                        and os.environ.get("PIXIE_UNET_GRAPH_INPLACE", "1") == "1")
            try:
                static_in = src if in_place else src.clone()
                side = torch.cuda.Stream(src.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):          # capture must not run on the legacy default stream
                    run(static_in)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                # thread_local: calls other threads make meanwhile (e.g. the RCCL watchdog of a multi-GPU job) do not void the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static_out = run(static_in)
            except RuntimeError as exc:                # same kernels either way: fall back to launching them one by one, loudly
                warnings.warn(f"pixie_amd: HIP-graph capture of the U-Net forward failed ({exc}); continuing with eager launches",
                              RuntimeWarning)
                self.use_graph = False
                return out
            if in_place:
                ent = self._graphs[src.data_ptr()] = (graph, None, static_out)
            else:
                ent = self._graphs["copy"] = (graph, static_in, static_out)
        graph, static_in, static_out = ent
        if static_in is not None:
            static_in.copy_(src)
        graph.replay()
        return static_out.clone()


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
def predict_material_field_from_voxel_grid(seg_network: "SegmentationUNet", cont_network: "RegressionUNet", grid_dhwc_f16: torch.Tensor,
                                           dual_stream: Optional[bool] = None):
    """The same as predict_material_field, fed with the voxel grid as the reference stores it -- (D, H, W, C) float16,
    `clip_features_features.npy` (pixie/voxel/voxelize.py:86,111) -- instead of the float32 (1, C, D, H, W) tensor
    my_data.py:160-224 makes of it.  The first projector convolution of BOTH networks reads the grid once
    (pixie_projector_conv0); the float32 NCDHW copy of the grid (4 B per feature written, 8 B read) never exists.
    Needs both networks to have the hidden-128 projector (feature_channels > cond_dim, as shipped) and the f16x3 path."""
    dev = next(seg_network.parameters()).device
    g = grid_dhwc_f16
    if g.dim() != 4 or g.dtype != torch.float16 or g.device != dev or dev.type != "cuda":
        raise ValueError("expected a (D, H, W, C) float16 tensor on the networks' HIP device")
    nets = (seg_network, cont_network)
    for net in nets:
        if net.cfg.projector_hidden is None or net.cfg.feature_channels != g.shape[3] or net.conv_precision != "f16x3":
            raise ValueError("the fused grid path needs the hidden-128 projector, matching feature_channels and conv_precision 'f16x3'")
        net._prepare(dev)
    ops = seg_network._runner.ops
    q = "projector.net.0"
    packed = [(net._runner._w16(q), net._runner._b(q)) for net in nets]
    x_seg, x_cont = ops.projector_conv0(g, packed, seg_network.cfg.projector_hidden)
    if dual_stream is None:
        dual_stream = _dual_stream_default(seg_network, cont_network, int(g.shape[0]) * int(g.shape[1]) * int(g.shape[2]))

    def fwd(net, p0):
        return net._forward_graphed(None, p0) if net.use_graph else net._forward_one(None, proj0=p0)

    if dual_stream:   # as in predict_material_field: the two networks on two HIP streams behind the shared first convolution
        cur = torch.cuda.current_stream()
        s1, s2 = _side_streams(dev)
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            seg_logits = fwd(seg_network, x_seg)[None]
        with torch.cuda.stream(s2):
            cont_pred = fwd(cont_network, x_cont)[None]
        x_seg.record_stream(s1); x_cont.record_stream(s2)
        cur.wait_stream(s1); cur.wait_stream(s2)
        seg_logits.record_stream(cur); cont_pred.record_stream(cur)
    else:
        seg_logits = fwd(seg_network, x_seg)[None]
        cont_pred = fwd(cont_network, x_cont)[None]
    combined, seg_pred = ops.combine(seg_logits[0].contiguous(), cont_pred[0].contiguous())
    return combined[None], seg_pred[None], seg_logits, cont_pred


_SIDE_STREAMS: Dict[str, Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}


def _dual_stream_default(seg_network, cont_network, voxels: int) -> bool:
    """The two networks are independent: on two HIP streams one network's small kernels and kernel tails fill the CUs the other
    leaves idle (eager launches at 128^3: 94.7 -> 89.6 ms per scene; graph replays at 64^3: 17.0 -> 15.2 ms).  The exception is
    large grids replayed as HIP graphs: every launch fills the chip, there are no launch gaps left to fill, and two graphs only
    contend (128^3, one box: 89.0 ms on two streams, 85.5-87.1 ms back to back).  PIXIE_DUAL_STREAM=0/1 overrides."""
    env = os.environ.get("PIXIE_DUAL_STREAM")
    if env is not None:
        return env == "1"
    return not (seg_network.use_graph and cont_network.use_graph and voxels >= 128 ** 3)


def _side_streams(device):
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _SIDE_STREAMS[key]


@torch.no_grad()
def predict_material_field(seg_network: SegmentationUNet, cont_network: RegressionUNet, feat_grid: torch.Tensor,
                           dual_stream: Optional[bool] = None):
    """The compute of process_batch + save_predictions (trainer/inference_combined.py:122-126,186-195):
    returns (combined (N, 3+num_classes, D, H, W), seg_pred (N, D, H, W) int32, seg_logits, cont_pred).
    `dual_stream`: the two networks on two HIP streams; default: see _dual_stream_default."""
    if feat_grid.is_cuda and feat_grid.device.index != torch.cuda.current_device():
        with torch.cuda.device(feat_grid.device):      # (launches go to the current device's stream: see forward())
            return predict_material_field(seg_network, cont_network, feat_grid, dual_stream)
    if dual_stream is None:
        dual_stream = _dual_stream_default(seg_network, cont_network, int(feat_grid[0, 0].numel()))
    if dual_stream and feat_grid.is_cuda:
        # the two networks are independent: run them on two HIP streams so that one network's small kernels and
        # kernel tails fill the CUs the other leaves idle
        cur = torch.cuda.current_stream()
        s1, s2 = _side_streams(feat_grid.device)
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            seg_logits = seg_network(feat_grid)
        with torch.cuda.stream(s2):
            cont_pred = cont_network(feat_grid)
        feat_grid.record_stream(s1); feat_grid.record_stream(s2)
        cur.wait_stream(s1); cur.wait_stream(s2)
        seg_logits.record_stream(cur); cont_pred.record_stream(cur)
    else:
        seg_logits = seg_network(feat_grid)
        cont_pred = cont_network(feat_grid)
    ops = seg_network._runner.ops
    combined, seg_pred = [], []
    for n in range(feat_grid.shape[0]):
        cmb, am = ops.combine(seg_logits[n].contiguous(), cont_pred[n].contiguous())
        combined.append(cmb)
        seg_pred.append(am)
    return torch.stack(combined), torch.stack(seg_pred), seg_logits, cont_pred
