"""Test double for pixie_amd.unet.HipOps: the same five operators written with torch CPU ops.

TEST INFRASTRUCTURE ONLY -- lets the CPU suite check the wiring of pixie_amd.unet.UNetRunner (which
layer feeds which, which statistics go with which conv, skip-stack order) against oracle/unet_oracle.py
without a GPU.  The product never constructs this class.
"""
import torch
import torch.nn.functional as F


class TorchRefOps:
    device = torch.device("cpu")

    def pack_conv(self, weight):
        return weight.detach().to(torch.float32)

    def conv(self, parts, packed_w, bias, cout, ksize, stride=1, upsample=False, pro=None, affine=None, act=0, residual=None,
             out_size=None):
        if out_size is not None:   # the odd-grid crop: compute, then drop the trailing planes (the HIP kernels never compute them)
            y = self.conv(parts, packed_w, bias, cout, ksize, stride, upsample, pro, affine, act, None)
            y = y[:, :out_size[0], :out_size[1], :out_size[2]].contiguous()
            return y + residual if residual is not None else y
        x = torch.cat(list(parts), dim=0) if len(parts) > 1 else parts[0]
        if pro is not None:
            x = x * pro[0][:, None, None, None] + pro[1][:, None, None, None]
        if affine is not None:
            x = x * affine[0][None] + affine[1][None]
        if act == 1:
            x = F.leaky_relu(x, 0.02)
        elif act == 2:
            x = F.silu(x)
        x = x[None]
        if upsample:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        w = packed_w
        if w.dim() == 3:  # Conv1d weight (cout, cin, 1)
            w = w[:, :, :, None, None]
        y = F.conv3d(x, w, bias, stride=stride, padding=1 if ksize == 3 else 0)[0]
        if residual is not None:
            y = y + residual
        assert y.shape[0] == cout
        return y

    def channel_sums(self, x):
        xd = x.reshape(x.shape[0], -1).double()
        return torch.stack([xd.sum(1), (xd * xd).sum(1)], dim=1)

    def norm_finalize(self, sums, spatial, mode, groups=1, eps=1e-5, weight=None, bias=None):
        c = sums.shape[0]
        if mode == 0:
            mean = sums[:, 0] / spatial
            var = sums[:, 1] / spatial - mean * mean
            rstd = 1.0 / torch.sqrt(var.clamp_min(0) + eps)
            return rstd.float(), (-mean * rstd).float()
        cpg = c // groups
        g = sums.reshape(groups, cpg, 2).sum(1)
        cnt = spatial * cpg
        mean = g[:, 0] / cnt
        var = g[:, 1] / cnt - mean * mean
        rstd = 1.0 / torch.sqrt(var.clamp_min(0) + eps)
        mean = mean.repeat_interleave(cpg)
        rstd = rstd.repeat_interleave(cpg)
        w = weight.double() if weight is not None else torch.ones(c, dtype=torch.float64)
        b = bias.double() if bias is not None else torch.zeros(c, dtype=torch.float64)
        return (rstd * w).float(), (b - mean * rstd * w).float()

    def attention(self, qkv, channels, tokens):
        q, k, v = torch.split(qkv, channels, dim=0)
        scale = channels ** -0.25
        w = torch.softmax((q * scale).t() @ (k * scale), dim=-1)
        return v @ w.t()

    def combine(self, logits, cont):
        am = torch.argmax(logits, dim=0)
        out = torch.zeros((3 + logits.shape[0],) + tuple(am.shape))
        out[:3] = cont
        for i in range(logits.shape[0]):
            out[3 + i] = (am == i).float()
        return out, am.int()


def _scale_exponent(bound: float) -> int:
    """Mirror of scale_exponent() in pixie_amd/csrc/conv3d_f16x3.hip."""
    import math
    if not (bound > 0.0) or math.isinf(bound) or math.isnan(bound):
        return 0
    eb = math.frexp(bound)[1] - 1  # bound in [2^eb, 2^(eb+1))
    return max(-100, min(100, 14 - eb))


def _split16(x: torch.Tensor, e: int):
    xs = x.float() * (2.0 ** e)
    hi = xs.half()
    assert torch.isfinite(hi).all(), "fp16 overflow: the scale bound was violated"
    lo = (xs - hi.float()).half()
    return hi.float(), lo.float()


class TorchRefOpsF16x3(TorchRefOps):
    """CPU emulation of the f16x3 convolution arithmetic (fp16 hi/lo split of both operands, three products,
    fp32 accumulation) with the same scale selection as the HIP kernel, so that the CPU suite can (a) check that
    UNetRunner hands every f16x3 launch a valid magnitude bound and (b) measure the end-to-end error of the
    scheme against the fp32 oracle.  TEST INFRASTRUCTURE ONLY."""

    n_f16x3 = 0
    n_exact = 0

    def pack_conv16(self, weight):
        return weight.detach().to(torch.float32)

    @staticmethod
    def f16x3_ok(parts, stride):
        cin = sum(int(p.shape[0]) for p in parts)
        return stride in (1, 2) and cin % 16 == 0 and int(parts[0].shape[0]) % 8 == 0

    def channel_stats(self, x, amax_slot):
        amax_slot.copy_(x.abs().max().reshape(1).float().view(torch.int32))
        return self.channel_sums(x)

    # the HIP launch folds a residual block's 1x1x1 skip convolution where the layer is not a split-K one; the emulation
    # takes every fold it is offered (same arithmetic: the skip products join the sum before bias / residual / statistics)
    fold_all = True

    def skip_foldable(self, x, cout, ksize, skip_parts):
        cin = sum(int(p.shape[0]) for p in skip_parts)
        return self.fold_all and cin % 16 == 0 and int(skip_parts[0].shape[0]) % 8 == 0

    def conv(self, parts, packed_w, bias, cout, ksize, stride=1, upsample=False, pro=None, affine=None, act=0, residual=None,
             w16=None, in_amax=None, in_bound=0.0, out_amax=None, out_size=None, skip=None):
        if skip is not None:   # out = conv(...) + skip_conv(raw skip parts) + skip bias, then the epilogue statistics
            assert stride == 1 and not upsample and out_size is None and w16 is not None
            s = self.conv(skip["parts"], None, skip["bias"], cout, 1, w16=skip["w16"], in_amax=skip["amax"])
            res = s if residual is None else s + residual
            return self.conv(parts, packed_w, bias, cout, ksize, stride, upsample, pro, affine, act, res, w16, in_amax, in_bound, out_amax)
        if out_size is not None:
            r = self.conv(parts, packed_w, bias, cout, ksize, stride, upsample, pro, affine, act, None, w16, in_amax, in_bound, None)
            y = r[:, :out_size[0], :out_size[1], :out_size[2]].contiguous()
            y = y + residual if residual is not None else y
            if out_amax is None:
                return y
            out_amax.copy_(torch.maximum(out_amax.view(torch.float32), y.abs().max().reshape(1).float()).view(torch.int32))
            yd = y.reshape(cout, -1).double()
            return y, torch.stack([yd.sum(1), (yd * yd).sum(1)], dim=1)
        if out_amax is not None:   # epilogue statistics of the f16x3 path
            y = self.conv(parts, packed_w, bias, cout, ksize, stride, upsample, pro, affine, act, residual, w16, in_amax, in_bound)
            out_amax.copy_(torch.maximum(out_amax.view(torch.float32), y.abs().max().reshape(1).float()).view(torch.int32))
            yd = y.reshape(cout, -1)
            # per-tile fp32 partial sums, fp64 across tiles -- as the HIP epilogue + pixie_stats_finalize do
            n = yd.shape[1]
            pad = (-n) % 512
            t = torch.nn.functional.pad(yd, (0, pad)).reshape(cout, -1, 512)
            return y, torch.stack([t.sum(2).double().sum(1), (t * t).sum(2).double().sum(1)], dim=1)
        if w16 is None:
            TorchRefOpsF16x3.n_exact += 1
            return super().conv(parts, packed_w, bias, cout, ksize, stride, upsample, pro, affine, act, residual)
        TorchRefOpsF16x3.n_f16x3 += 1
        x = torch.cat(list(parts), dim=0) if len(parts) > 1 else parts[0]
        if pro is not None:
            x = x * pro[0][:, None, None, None] + pro[1][:, None, None, None]
        if affine is not None:
            x = x * affine[0][None] + affine[1][None]
        if act == 1:
            x = F.leaky_relu(x, 0.02)
        elif act == 2:
            x = F.silu(x)
        if in_amax:
            assert pro is None and affine is None and len(in_amax) == len(parts)
            bound = max(float(s.view(torch.float32)[0]) for s in in_amax)
        else:
            bound = float(in_bound)
            assert bound > 0.0
        assert float(x.abs().max()) <= bound * (1 + 1e-6), (float(x.abs().max()), bound)
        ex = _scale_exponent(bound)
        w = w16 if w16.dim() == 5 else w16[:, :, :, None, None]
        ew = _scale_exponent(float(w.abs().max()))
        xh, xl = _split16(x, ex)
        wh, wl = _split16(w, ew)
        if upsample:
            xh = F.interpolate(xh[None], scale_factor=2, mode="nearest")[0]
            xl = F.interpolate(xl[None], scale_factor=2, mode="nearest")[0]
        pad = 1 if ksize == 3 else 0
        y = (F.conv3d(xl[None], wh, None, stride=stride, padding=pad) + F.conv3d(xh[None], wl, None, stride=stride, padding=pad)
             + F.conv3d(xh[None], wh, None, stride=stride, padding=pad))
        y = y[0] * (2.0 ** (-ex - ew))
        if bias is not None:
            y = y + bias[:, None, None, None]
        if residual is not None:
            y = y + residual
        assert y.shape[0] == cout
        return y
