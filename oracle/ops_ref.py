"""Per-op CPU fp32 checkers (torch CPU) used by the kernel-level GPU parity tests.  Test infrastructure only."""
import torch
import torch.nn.functional as F


def conv2d(x, w, b=None, stride=1, pad=0, dil=1):
    return F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)


def conv2d_fwd_bwd(x, w, stride, pad, dil, seed, grad_dtype=torch.float32):
    """y, dx, dw for a seeded upstream gradient dy (dy is rounded to grad_dtype first, like the device path)."""
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)
    g = torch.Generator().manual_seed(seed)
    dy = torch.randn(y.shape, generator=g).to(grad_dtype).float()
    y.backward(dy)
    return y.detach(), x.grad.detach(), w.grad.detach(), dy
