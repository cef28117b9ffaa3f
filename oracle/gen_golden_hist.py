#!/usr/bin/env python3
"""Golden vectors for the offline NTM utilities: the reference's own `fast_hist` / `label_mapping` functions of
tools/compute_ClassDistribution.py:49-63 and tools/compute_ConfusionMatrix.py:54-66 are parsed out of the reference files at run
time (the modules themselves import ttach / torchvision / modules that do not exist and cannot be imported) and exec()'d on seeded
label images -> tests/golden/g14_hist.npz.  Build container only; nothing of the reference is stored."""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/tools"


def reference_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    cd = reference_functions(os.path.join(REF, "compute_ClassDistribution.py"), {"fast_hist", "label_mapping"})
    cm = reference_functions(os.path.join(REF, "compute_ConfusionMatrix.py"), {"fast_hist", "label_mapping"})
    rng = np.random.default_rng(14)
    out = {}
    # class distribution: CM += fast_hist(pred.flatten(), 19) over the pseudo-label images; Class_dist / (sum + 10e-10)   (:81-94)
    preds = [rng.integers(0, 19, (37, 53), dtype=np.uint8) for _ in range(3)]
    for p in preds:
        p[rng.random(p.shape) < 0.15] = 255
    CM = np.zeros(19)
    for p in preds:
        CM += cd["fast_hist"](p.flatten(), 19)
    out["cd_preds"], out["cd_counts"], out["cd_norm"] = np.stack(preds), CM, CM / (np.sum(CM) + 10e-10)
    # 34 x 19 confusion matrix: label ids 0..33 (+ some 255) through a label_mapping table, pseudo labels 0..18   (:87-97)
    mapping = np.array([[i, (i * 7 + 3) % 34] for i in range(34)] + [[200, 255]])
    gts = [rng.integers(0, 34, (29, 41), dtype=np.uint8) for _ in range(3)]
    for g_ in gts:
        g_[rng.random(g_.shape) < 0.1] = 200
    prs = [rng.integers(0, 19, (29, 41), dtype=np.uint8) for _ in range(3)]
    M = np.zeros((34, 19))
    for g_, p in zip(gts, prs):
        lab = cm["label_mapping"](g_, mapping)
        M += cm["fast_hist"](lab.flatten(), p.flatten(), 34, 19)
    out["cm_gts"], out["cm_preds"], out["cm_mapping"], out["cm_counts"] = np.stack(gts), np.stack(prs), mapping, M
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g14_hist.npz"), **out)
    print("wrote g14_hist.npz", CM.sum(), M.sum())


if __name__ == "__main__":
    main()
