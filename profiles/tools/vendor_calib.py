"""Calibration of the kernel ceiling against the vendor stack.  MEASUREMENT ONLY: nothing under simt_amd/ imports torch.matmul,
F.conv2d, MIOpen, hipBLASLt or rocBLAS, and nothing there imports this file.

    python profiles/tools/vendor_calib.py [--step] > profiles/r06_vendor_calibration.txt

Per production shape of the DeepLab-v2 trunk at BASELINE configs[1] (B = 4, 768 x 768 -> M = 4 * 97 * 97 = 37 636 pixels, bf16; reference shapes
model/deeplab_multi.py:62,68,73) it times, on ONE box, in ONE process, under the same protocol:
  own     simt_conv_fprop of the shipped library, the tile the production plan picks, bias + ReLU epilogue (the frozen net's flavour)
  vendor  torch.matmul (hipBLASLt / rocBLAS) on [M, K] x [K, N] for the 1x1 shapes; F.conv2d (MIOpen) on channels_last bf16 for the 3x3 shapes
Protocol (as profiles/tools/conv_modes.py / one_conv.py): 6 rotating operand sets (each launch's operands come from HBM, not from the previous launch's
cache state), 3 untimed rounds (clocks and MIOpen's find step warm), then `rounds` x 6 launches back to back between two HIP events -> us per launch;
the median of 5 such measurements, own and vendor interleaved.  `--step` adds one eager PyTorch-ROCm training step of a torch.nn restatement of the
two-head DeepLab-v2 ResNet-101 at configs[1] (frozen eval forward + train-mode forward + backward of a plain cross-entropy + SGD; bf16 channels_last):
what the vendor stack does with the whole iteration (WITHOUT the SimT loss block, which only makes its step shorter)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L          # noqa: E402
from simt_amd import ops                # noqa: E402

BF, dev = torch.bfloat16, torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
NSET = 6


CLOCKS = {}          # candidate name of the last timeit() -> (socket W, shader MHz) while it ran back to back for ~0.3 s (sysfs, bench.PowerWatch)


def timeit(fns, rounds=8, reps=5):
    """fns: {name: [callable per operand set]} -> {name: median us per launch}; the candidates' measurements alternate."""
    for f in fns.values():
        for _ in range(3):
            for g in f:
                g()
    torch.cuda.synchronize()
    out = {k: [] for k in fns}
    for _ in range(reps):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(rounds):
                for g in f:
                    g()
            e1.record()
            torch.cuda.synchronize()
            out[k].append(e0.elapsed_time(e1) * 1e3 / (rounds * len(f)))
    med = {k: float(np.median(v)) for k, v in out.items()}
    # power / shader clock under each candidate alone: enough launches for ~0.35 s so that the 0.1-s sysfs sampler sees the steady state
    CLOCKS.clear()
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
        from bench import PowerWatch
        for k, f in fns.items():
            n = max(1, int(0.35e6 / (med[k] * len(f))))
            with PowerWatch(0) as pw:
                for _ in range(n):
                    for g in f:
                        g()
                torch.cuda.synchronize()
            r = pw.report() or {}
            CLOCKS[k] = (r.get("avg_w"), r.get("sclk_mhz_avg"))
    except Exception:
        pass
    return med


def _clk(k):
    w, m = CLOCKS.get(k, (None, None))
    return f"{w:.0f} W {m:.0f} MHz" if w and m else "-"


def own_conv(lib, st, Cin, Cout, k, dil, relu=True, stats=False):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    tile = ops.pick_tile_n(Cout, BF)
    fs, keep = [], []
    for _ in range(NSET):
        x = torch.randn(M, Cin, device=dev).to(BF)
        npad = ops.round_up(Cout, tile)
        wp = (torch.randn(npad, len(taps) * Cin, device=dev) * 0.02).to(BF)
        y = torch.empty(M, Cout, device=dev, dtype=BF)
        kw = {}
        if stats:
            kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
        else:
            kw["bias"] = torch.zeros(Cout, device=dev)
            kw["relu"] = relu
        d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=tile, **kw)
        keep.append((d, x, wp, y, kw))
        fs.append(lambda d=d: lib.simt_conv_fprop(C.byref(d), st))
    return fs, keep


def vendor_mm(K, N, epilogue=False):
    fs, keep = [], []
    for _ in range(NSET):
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(K, N, device=dev) * 0.02).to(BF)
        y = torch.empty(M, N, device=dev, dtype=BF)
        b = torch.zeros(N, device=dev, dtype=BF)
        keep.append((x, w, y, b))
        if epilogue:
            fs.append(lambda x=x, w=w, y=y, b=b: torch.relu_(torch.addmm(b, x, w, out=y)))
        else:
            fs.append(lambda x=x, w=w, y=y: torch.matmul(x, w, out=y))
    return fs, keep


def vendor_conv(Cin, Cout, k, dil):
    fs, keep = [], []
    for _ in range(NSET):
        x = torch.randn(B, Cin, H, W, device=dev).to(BF).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.02).to(BF).contiguous(memory_format=torch.channels_last)
        keep.append((x, w))
        fs.append(lambda x=x, w=w: F.conv2d(x, w, None, 1, dil * (k // 2), dil))
    return fs, keep


class Bottleneck(nn.Module):
    def __init__(self, inp, planes, stride, dil, down):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(inp, planes, 1, stride, bias=False), nn.BatchNorm2d(planes)
        self.c2, self.b2 = nn.Conv2d(planes, planes, 3, 1, dil, dil, bias=False), nn.BatchNorm2d(planes)
        self.c3, self.b3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.down = nn.Sequential(nn.Conv2d(inp, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if down else None

    def forward(self, x):
        r = x if self.down is None else self.down(x)
        o = F.relu(self.b1(self.c1(x)))
        o = F.relu(self.b2(self.c2(o)))
        return F.relu(self.b3(self.c3(o)) + r)


class Head(nn.Module):
    def __init__(self, inp, n):
        super().__init__()
        self.a, self.b = nn.Conv2d(inp, n, 3, 1, 6, 6), nn.Conv2d(inp, n, 3, 1, 12, 12)

    def forward(self, x):
        return self.a(x) + self.b(x)


class V2(nn.Module):
    """torch.nn restatement of the shapes of model/deeplab_multi.py:122-192 (measurement only; weights random)."""

    def __init__(self, n):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1, ceil_mode=True))
        inp, layers = 64, []
        for planes, nb, stride, dil in ((64, 3, 1, 1), (128, 4, 2, 1), (256, 23, 1, 2), (512, 3, 1, 4)):
            blk = [Bottleneck(inp, planes, stride, dil, True)]
            inp = planes * 4
            blk += [Bottleneck(inp, planes, 1, dil, False) for _ in range(nb - 1)]
            layers.append(nn.Sequential(*blk))
        self.l1, self.l2, self.l3, self.l4 = layers
        self.h1, self.h2 = Head(1024, n), Head(2048, n)

    def forward(self, x):
        x = self.l3(self.l2(self.l1(self.stem(x))))
        return self.h1(x), self.h2(self.l4(x))


def eager_step():
    torch.manual_seed(0)
    net = V2(22).to(dev).to(BF).to(memory_format=torch.channels_last).train()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)
    fixed = V2(19).to(dev).to(BF).to(memory_format=torch.channels_last).eval()
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=6e-4, momentum=0.9, weight_decay=5e-4)
    x = torch.randn(4, 3, 768, 768, device=dev).to(BF).contiguous(memory_format=torch.channels_last)
    lab = torch.randint(0, 19, (4, 768, 768), device=dev)

    def step():
        with torch.no_grad():
            _ = fixed(x)
        p1, p2 = net(x)
        u1 = F.interpolate(p1.float(), (768, 768), mode="bilinear", align_corners=True)
        u2 = F.interpolate(p2.float(), (768, 768), mode="bilinear", align_corners=True)
        loss = F.cross_entropy(u2, lab) + 0.1 * F.cross_entropy(u1, lab)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    t0 = time.time()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(f"# eager step: 3 warm-up steps (MIOpen find included) took {time.time() - t0:.1f} s", flush=True)
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print(f"eager PyTorch-ROCm step (torch.nn DeepLab-v2 R-101 two heads, bf16 channels_last, B=4 768x768; frozen forward + forward + plain-CE backward + SGD, "
          f"NO SimT loss block): median {ms:.1f} ms/step = {4e3 / ms:.1f} images/s   (min {min(ts):.1f}, max {max(ts):.1f} over 8)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", action="store_true")
    ap.add_argument("--only-step", action="store_true")
    a = ap.parse_args()
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    print(f"# torch {torch.__version__}, hip {torch.version.hip}, device {torch.cuda.get_device_name(0)}; lib sha256 {__import__('hashlib').sha256(open(os.path.join(os.path.dirname(L.__file__), 'libsimt_hip.so'), 'rb').read()).hexdigest()[:16]}")
    print(f"# M = {M} pixels (B=4, 97x97), bf16, 6 rotating operand sets, median of 5 x (8 rounds x 6 launches); TF/s on algorithmic FLOPs 2 M N K")
    print(f"# {'shape':34s} {'own us':>8s} {'own TF/s':>9s} {'vendor us':>10s} {'vendor TF/s':>12s} {'own/vendor time':>16s}   vendor call")
    rows = [("1x1 1024->256  (K=1024,N=256)", 1024, 256, 1, 1), ("1x1 256->1024  (K=256,N=1024)", 256, 1024, 1, 1),
            ("3x3 d2 256->256 (K=2304,N=256)", 256, 256, 3, 2), ("3x3 d4 512->512 (K=4608,N=512)", 512, 512, 3, 4),
            ("1x1 2048->512 (K=2048,N=512)", 2048, 512, 1, 1), ("1x1 512->2048 (K=512,N=2048)", 512, 2048, 1, 1)]
    if not a.only_step:
        for name, Cin, Cout, k, dil in rows:
            fl = 2.0 * M * Cout * Cin * k * k
            own, keep1 = own_conv(lib, st, Cin, Cout, k, dil)
            cands = {"own": own}
            mm, keep2 = vendor_mm(Cin * k * k, Cout)
            cands["matmul"] = mm                      # the 3x3 shapes too: the im2col GEMM of the same FLOPs, operands ideal (dense [M, 9 Cin])
            mme, keep4 = vendor_mm(Cin * k * k, Cout, epilogue=True)
            cands["addmm+relu"] = mme
            if k == 3:
                cv, keep3 = vendor_conv(Cin, Cout, k, dil)
                cands["conv2d"] = cv
            try:
                r = timeit(cands)
            except Exception as e:                    # a vendor path that does not support the shape must not take the table down
                print(f"  {name:34s} FAILED: {type(e).__name__}: {str(e)[:120]}")
                continue
            for vk in [c for c in cands if c != "own"]:
                what = {"matmul": "torch.matmul [M,K]x[K,N] (dense operand" + (", NOT a conv: the GEMM alone)" if k == 3 else ")"),
                        "addmm+relu": "torch.addmm + relu_ (bias / ReLU as the frozen net's epilogue; 2 launches)",
                        "conv2d": "F.conv2d channels_last (MIOpen)"}[vk]
                print(f"  {name:34s} {r['own']:8.1f} {fl / r['own'] / 1e6:9.0f} {r[vk]:10.1f} {fl / r[vk] / 1e6:12.0f} {r['own'] / r[vk]:16.2f}   {what}   [own {_clk('own')} | vendor {_clk(vk)}]", flush=True)
            del own, keep1, mm, keep2, mme, keep4, cands
            torch.cuda.empty_cache()
        # the tap-expanded classifier GEMM (2048 -> 432, fp32 result in the product; the vendor leg writes bf16: less output traffic)
        name, K_, N_ = "head GEMM 2048->432 (K=2048,N=432)", 2048, 432
        mm, keep = vendor_mm(K_, N_)
        r = timeit({"matmul": mm})
        fl = 2.0 * M * N_ * K_
        print(f"  {name:34s} {'-':>8s} {'-':>9s} {r['matmul']:10.1f} {fl / r['matmul'] / 1e6:12.0f} {'-':>16s}   torch.matmul, bf16 result (own: bench.py aspp_t_step.gemm_by_shape)")
    if a.step or a.only_step:
        eager_step()


if __name__ == "__main__":
    main()
