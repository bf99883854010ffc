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

    def conv(self, parts, packed_w, bias, cout, ksize, stride=1, upsample=False, pro=None, affine=None, act=0, residual=None):
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
