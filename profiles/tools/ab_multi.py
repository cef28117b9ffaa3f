"""Time one conv case under N builds of the library in one process (interleaved rounds, rotating buffers).
usage: python scratch/ab_multi.py "<case substring>" lib1.so lib2.so ..."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv, libs_paths, sel = [sys.argv[0]], sys.argv[2:], sys.argv[1]
import importlib.util
src = open(os.path.join(os.path.dirname(__file__), "ab_conv.py")).read().split("libs = [load(p)")[0].replace('if len(sys.argv) > 3:', 'if False:')
exec(src)
CASES = [c for c in CASES if sel in c[0]]
NSETS = int(os.environ.get("NSETS", NSETS))
libs = [load(p) for p in libs_paths]
st = torch.cuda.current_stream().cuda_stream
for case in CASES:
    name, sets, flops = make(case)
    res = [[] for _ in libs]
    for rnd in range(5):
        for li, lib in enumerate(libs):
            for d, _ in sets: lib.simt_conv_fprop(C.byref(d), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for d, _ in sets: lib.simt_conv_fprop(C.byref(d), st)
            e1.record(); torch.cuda.synchronize()
            res[li].append(e0.elapsed_time(e1) / (4 * len(sets)) * 1e3)
    print(name)
    for p, r in zip(libs_paths, res): print(f"   {os.path.basename(p):28s} {np.median(r):7.1f} us   min {min(r):.1f}", flush=True)
