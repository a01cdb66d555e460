"""Synthetic stand-ins for two of the reference's mel-spectrogram classifiers -- test / bench material, NOT a product
component (the product lowers ANY ``nn.Module`` the scripts un-pickle: ``audiopure_amd.convnet.NativeConvNet``).

They reproduce the attribute tree, hence the state-dict keys, of

* VGG19-BN      (audio_models/ConvNets_SpeechCommands/models/vgg.py:30-201), and
* ResNeXt-29 8x64d, the scripts' default classifier (models/resnext.py:23-142; train_speech_commands.py:43),

so that weights keyed on those names (``audiopure_amd.synth.synth_init``) reproduce the golden logits generated from the
reference's own classes (tests/golden/make_golden_convnets.py).  Both are assembled from small shape tables.
"""
import torch.nn as nn
import torch.nn.functional as F

from audiopure_amd.synth import synth_init  # noqa: F401  (re-exported for the tests / benches)

_VGG19_PLAN = (2, 64), (2, 128), (4, 256), (4, 512), (4, 512)            # (convs, channels) per pooling stage


class VGG(nn.Module):
    def __init__(self, plan=_VGG19_PLAN, num_classes=10, in_channels=1, width_div=1):
        super().__init__()
        mods, c = [], in_channels
        for reps, ch in plan:
            ch = max(ch // width_div, 8)
            for _ in range(reps):
                mods += [nn.Conv2d(c, ch, 3, padding=1), nn.BatchNorm2d(ch), nn.ReLU(inplace=True)]
                c = ch
            mods.append(nn.MaxPool2d(2, 2))
        hid = 4096 // width_div
        self.features = nn.Sequential(*mods)
        self.classifier = nn.Sequential(nn.Linear(c, hid), nn.ReLU(True), nn.Dropout(), nn.Linear(hid, hid), nn.ReLU(True),
                                        nn.Dropout(), nn.Linear(hid, num_classes))

    def forward(self, x):
        return self.classifier(self.features(x).flatten(1))


def vgg19_bn(num_classes=10, in_channels=1, width_div=1):
    return VGG(_VGG19_PLAN, num_classes, in_channels, width_div)


def _conv(cin, cout, k, stride=1, groups=1):
    return nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)


class ResNeXtBottleneck(nn.Module):
    """1x1 reduce -> grouped 3x3 -> 1x1 expand, projection shortcut when the width changes."""

    def __init__(self, cin, cout, stride, cardinality, base_width, widen_factor):
        super().__init__()
        D = cardinality * int(base_width * cout / (widen_factor * 64.))
        members = [("conv_reduce", _conv(cin, D, 1)), ("bn_reduce", nn.BatchNorm2d(D)),
                   ("conv_conv", _conv(D, D, 3, stride, cardinality)), ("bn", nn.BatchNorm2d(D)),
                   ("conv_expand", _conv(D, cout, 1)), ("bn_expand", nn.BatchNorm2d(cout)), ("shortcut", nn.Sequential())]
        for name, m in members:
            self.add_module(name, m)
        if cin != cout:
            self.shortcut.add_module("shortcut_conv", _conv(cin, cout, 1, stride))
            self.shortcut.add_module("shortcut_bn", nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn_reduce(self.conv_reduce(x)))
        y = F.relu(self.bn(self.conv_conv(y)))
        return F.relu(self.shortcut(x) + self.bn_expand(self.conv_expand(y)))


class CifarResNeXt(nn.Module):
    def __init__(self, nlabels, cardinality=8, depth=29, base_width=64, widen_factor=4, in_channels=1):
        super().__init__()
        per_stage = (depth - 2) // 9
        widths = [64] + [w * widen_factor for w in (64, 128, 256)]
        self.conv_1_3x3 = _conv(in_channels, 64, 3)
        self.bn_1 = nn.BatchNorm2d(64)
        for s in (1, 2, 3):
            stage = nn.Sequential()
            for i in range(per_stage):
                stage.add_module(f"stage_{s}_bottleneck_{i}",
                                 ResNeXtBottleneck(widths[s - 1] if i == 0 else widths[s], widths[s],
                                                   (1 if s == 1 else 2) if i == 0 else 1, cardinality, base_width, widen_factor))
            self.add_module(f"stage_{s}", stage)
        self.classifier = nn.Linear(widths[3], nlabels)

    def forward(self, x):
        x = F.relu(self.bn_1(self.conv_1_3x3(x)))
        x = self.stage_3(self.stage_2(self.stage_1(x)))
        return self.classifier(F.avg_pool2d(x, 8, 1).flatten(1))
