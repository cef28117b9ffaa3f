"""CPU-only checks of the drop-in modules' structure (no compute): state_dict keys/shapes equal the reference's
(golden g8 lists the reference's 552 floating tensors; oracle.state_shapes the full 656), optimiser listing
multiplicity, frozen BN affine, missing-GPU behaviour."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import simt_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simt_amd"))      # what tools/_init_paths.py does for the reference tree


def test_import_surface_like_the_reference():
    from model.deeplab_multi import DeeplabMulti, sig_NTM, sig_W  # noqa: F401
    from model.deeplab import Res_Deeplab  # noqa: F401
    from utils.loss import CrossEntropy2d, EntropyLoss  # noqa: F401


def test_state_dict_contract():
    from model.deeplab_multi import DeeplabMulti
    m = DeeplabMulti(num_classes=19, open_classes=3, openset=True)
    sd = m.state_dict()
    ref = so.state_shapes(19, 3, True)
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 656
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k]), k
    d = np.load(os.path.join(ROOT, "tests", "golden", "g8_iteration.npz"))
    assert set(str(n) for n in d["param_names"]) <= set(sd.keys())       # names recorded from the reference itself
    m0 = DeeplabMulti(num_classes=19)
    assert list(m0.state_dict().keys()) == list(so.state_shapes(19, 0, False).keys())


def test_single_head_state_dict():
    from model.deeplab import Res_Deeplab
    m = Res_Deeplab(num_classes=19)
    assert list(m.state_dict().keys()) == list(so.state_shapes(19, single_head=True).keys())
    x_groups = m.optim_parameters(type("A", (), {"lr": 1e-3}))
    assert x_groups[1]["lr"] == pytest.approx(1e-2)


def test_optim_parameters_duplicates_and_frozen_bn():
    from model.deeplab_multi import DeeplabMulti
    m = DeeplabMulti(num_classes=19, open_classes=3, openset=True)
    args = type("A", (), {"learning_rate": 6e-4})
    g = m.optim_parameters(args)
    g0, g1 = list(g[0]["params"]), list(g[1]["params"])
    assert len(g0) == 726 and len({id(p) for p in g0}) == 240         # SURVEY quirk 4 (probe numbers)
    assert len(g1) == 32 and g[1]["lr"] == pytest.approx(6e-3)
    names = {id(p): n for n, p in m.named_parameters()}
    og0, og1 = so.optim_param_names(so.state_shapes(19, 3, True))
    assert [names[id(p)] for p in g0] == og0 and [names[id(p)] for p in g1] == og1
    for n, p in m.named_parameters():
        is_bn = ".bn" in n or n.startswith("bn1") or "downsample.1" in n
        assert p.requires_grad == (not is_bn), n
    gw = m.optim_parameters(args, warmup=True)
    assert len(list(gw[0]["params"])) > 726


def test_modules_refuse_cpu():
    from model.deeplab_multi import DeeplabMulti, sig_NTM
    from utils.loss import CrossEntropy2d
    m = DeeplabMulti(num_classes=19, open_classes=3, openset=True)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 3, 65, 65))
    with pytest.raises(AssertionError):
        sig_NTM(19, 3)()
    with pytest.raises(AssertionError):
        CrossEntropy2d()(torch.zeros(1, 19, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_deeplab_vgg_state_dict_keys_cpu():
    from model.deeplab_vgg import DeeplabVGG
    from simt_amd.engine_vgg import vgg_state_shapes
    m = DeeplabVGG(num_classes=19)
    sd = m.state_dict()
    ref = vgg_state_shapes(19)
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k]), k
    # torchvision vgg16.features with pool4/pool5 dropped: convs sit at these Sequential indices, fc6/fc7 at 29/31
    assert [i for i, mod in enumerate(m.features) if isinstance(mod, torch.nn.Conv2d)] == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 23, 25, 27, 29, 31]
    assert m.features[23].dilation == (2, 2) and m.features[29].dilation == (4, 4) and m.features[29].out_channels == 1024
