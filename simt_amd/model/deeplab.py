"""model.deeplab of the reference over the HIP engine: the single-head DeepLab-v2 ResNet-101 (`Res_Deeplab`).

Mirrors model/deeplab.py:101-116 (Classifier_Module with all FOUR dilated branches summed -- the `return` sits outside
the loop here, unlike deeplab_multi.py), :118-177 ResNet (forward returns `(x, x)`), :179-220 parameter groups (all of
conv1..layer4, filtered by requires_grad, + the head at 10x lr, reading `args.lr`) and :223-238 Res_Deeplab."""
import torch

from simt_amd.engine import LAYERS, single_head
from simt_amd.model.deeplab_multi import Bottleneck, Classifier_Module, ResNetMulti

affine_par = True


class ResNet(ResNetMulti):
    def __init__(self, block, layers, num_classes):
        super().__init__(block, layers, num_classes, 0, False)
        del self.layer6                        # deeplab.py has ONE head, on layer4, called layer5
        dil = [6, 12, 18, 24]
        self.layer5 = Classifier_Module(2048, dil, dil, num_classes)

    def _heads(self):
        return single_head(self.num_classes)

    def forward(self, x):
        (y,) = self._run(x)
        return y, y

    def get_1x_lr_params_NOscale(self):
        for root in (self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4):
            for sub in root.modules():
                for p in sub.parameters():
                    if p.requires_grad:
                        yield p

    def get_10x_lr_params(self):
        for p in self.layer5.parameters():
            yield p

    def optim_parameters(self, args):
        return [{"params": self.get_1x_lr_params_NOscale(), "lr": args.lr},
                {"params": self.get_10x_lr_params(), "lr": 10 * args.lr}]


def Res_Deeplab(num_classes=21, pretrained=False):
    model = ResNet(Bottleneck, list(LAYERS), num_classes)
    if pretrained:
        restore_from = "checkpoints/DeepLab_init.pth"
        saved_state_dict = torch.load(restore_from)
        new_params = model.state_dict().copy()
        for i in saved_state_dict:
            i_parts = i.split(".")
            if not i_parts[1] == "layer5":
                new_params[".".join(i_parts[1:])] = saved_state_dict[i]
        model.load_state_dict(new_params)
        print("ImageNet pretrained weights loaded")
    return model
