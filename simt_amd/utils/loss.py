"""utils.loss of the reference (utils/loss.py:6-49) over libsimt_hip.so: CrossEntropy2d and EntropyLoss.

Same constructor / forward signatures, same asserts, same reductions (mean over valid pixels; NaN when no pixel is
valid).  Inputs must live on the GPU: there is no CPU fallback."""
import torch
import torch.nn as nn

from simt_amd import _lib as L
from simt_amd import ops


def _ws(dev):
    return torch.empty(L.load().simt_loss_ws_bytes(), device=dev, dtype=torch.uint8)


class _CE2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predict, target, weight, ignore_label, is_softmax):
        p = predict.detach().contiguous().float()
        t = target.contiguous().long()
        n, c, h, w = p.shape
        out = torch.empty(2, device=p.device)
        wt = None if weight is None else weight.detach().to(p.device, torch.float32).contiguous()
        L.call("simt_ce2d_fwd", ops._p(p), ops._p(t), ops._p(wt), n, c, h, w, int(ignore_label), int(is_softmax),
               ops._p(_ws(p.device)), ops._p(out), ops.stream_ptr())
        ctx.save_for_backward(p, t, out)
        ctx.wt, ctx.cfg = wt, (int(ignore_label), int(is_softmax))
        return out[0].clone()

    @staticmethod
    def backward(ctx, go):
        p, t, out = ctx.saved_tensors
        n, c, h, w = p.shape
        dp = torch.empty_like(p)
        g = go.detach().reshape(1).float().contiguous()
        L.call("simt_ce2d_bwd", ops._p(p), ops._p(t), ops._p(ctx.wt), n, c, h, w, ctx.cfg[0], ctx.cfg[1], ops._p(out),
               ops._p(g), ops._p(dp), ops.stream_ptr())
        return dp, None, None, None, None


class CrossEntropy2d(nn.Module):
    def __init__(self, size_average=True, ignore_label=255, is_softmax=True):
        super().__init__()
        self.size_average = size_average
        self.ignore_label = ignore_label
        self.is_softmax = is_softmax

    def forward(self, predict, target, weight=None):
        """predict (n, c, h, w) logits (is_softmax=True) or probabilities (False); target (n, h, w) int64."""
        assert not target.requires_grad
        assert predict.dim() == 4
        assert target.dim() == 3
        assert predict.size(0) == target.size(0), "{0} vs {1} ".format(predict.size(0), target.size(0))
        assert predict.size(2) == target.size(1), "{0} vs {1} ".format(predict.size(2), target.size(1))
        assert predict.size(3) == target.size(2), "{0} vs {1} ".format(predict.size(3), target.size(2))
        return _CE2dFn.apply(predict, target, weight, self.ignore_label, self.is_softmax)


class _EntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.detach().contiguous().float()
        n, c = xc.shape[:2]
        hw = xc.numel() // (n * c)
        out = torch.empty(2, device=xc.device)
        L.call("simt_entropy2d", ops._p(xc), n, c, hw, 1, ops._p(_ws(xc.device)), ops._p(out), None, None, ops.stream_ptr())
        ctx.save_for_backward(xc)
        return out[0].clone()

    @staticmethod
    def backward(ctx, go):
        (xc,) = ctx.saved_tensors
        n, c = xc.shape[:2]
        hw = xc.numel() // (n * c)
        dx = torch.empty_like(xc)
        g = go.detach().reshape(1).float().contiguous()
        L.call("simt_entropy2d", ops._p(xc), n, c, hw, 1, None, None, ops._p(g), ops._p(dx), ops.stream_ptr())
        return dx


class EntropyLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x):
        return _EntropyFn.apply(x)
