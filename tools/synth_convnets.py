"""Synthetic stand-ins for the six families of the reference's mel-spectrogram classifiers -- test / bench material, NOT a product
component (the product lowers ANY ``nn.Module`` the scripts un-pickle: ``audiopure_amd.convnet.NativeConvNet``).

They reproduce the attribute tree, hence the state-dict keys, of

* VGG19-BN      (audio_models/ConvNets_SpeechCommands/models/vgg.py:30-201), and
* ResNeXt-29 8x64d, the scripts' default classifier (models/resnext.py:23-142; train_speech_commands.py:43),

so that weights keyed on those names (``audiopure_amd.synth.synth_init``) reproduce the golden logits generated from the
reference's own classes (tests/golden/make_golden_convnets.py).  Both are assembled from small shape tables.
"""
import torch
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


# ---- the other four families of models/__init__.py:8-45: attribute trees only (state-dict keys = the reference's), assembled
# from stage tables; the forward of each is the family's published dataflow, written against those attribute names.

def _bn(c):
    return nn.BatchNorm2d(c)


class _ResBottle(nn.Module):
    """models/resnet.py:66-103 -- 1x1 / 3x3 (strided) / 1x1 x4, post-activation, optional projection `downsample`."""

    def __init__(self, cin, width, stride, project):
        super().__init__()
        for i, (ci, co, k, s) in enumerate([(cin, width, 1, 1), (width, width, 3, stride), (width, 4 * width, 1, 1)], 1):
            setattr(self, f"conv{i}", _conv(ci, co, k, s))
            setattr(self, f"bn{i}", _bn(co))
        self.downsample = nn.Sequential(_conv(cin, 4 * width, 1, stride), _bn(4 * width)) if project else None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """models/resnet.py:106-165 with Bottleneck [3, 4, 6, 3]: 7x7/2 stem, 3x3/2 max-pool, four stages, AvgPool2d(1) (a no-op on the
    1 x 1 map a 32 x 32 input leaves), fc."""

    def __init__(self, num_classes=10, in_channels=1, reps=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = _bn(64)
        cin = 64
        for s, (n, width) in enumerate(zip(reps, (64, 128, 256, 512)), 1):
            stride = 1 if s == 1 else 2
            blocks = [_ResBottle(cin, width, stride, stride != 1 or cin != 4 * width)]
            blocks += [_ResBottle(4 * width, width, 1, False) for _ in range(n - 1)]
            setattr(self, f"layer{s}", nn.Sequential(*blocks))
            cin = 4 * width
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(x.flatten(1))


class _WideBlock(nn.Module):
    """models/wideresnet.py:16-41 -- pre-activation pair of 3x3s; when the width changes the activated input also feeds the 1x1
    `convShortcut`, otherwise the raw input is the identity branch."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.bn1, self.conv1 = _bn(cin), _conv(cin, cout, 3, stride)
        self.bn2, self.conv2 = _bn(cout), _conv(cout, cout, 3)
        if cin != cout:
            self.convShortcut = nn.Conv2d(cin, cout, 1, stride, 0, bias=False)
        self.project = cin != cout

    def forward(self, x):
        a = F.relu(self.bn1(x))
        y = self.conv2(F.relu(self.bn2(self.conv1(a))))
        return (self.convShortcut(a) if self.project else x) + y


class _Stack(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.layer = nn.Sequential(*blocks)

    def forward(self, x):
        return self.layer(x)


class WideResNet28_10(nn.Module):
    """models/wideresnet.py:55-92, depth 28 / widen 10 / no dropout: 3x3 stem (16), three stacks of four blocks (160, 320, 640),
    BN-ReLU, 8 x 8 average pool, fc."""

    def __init__(self, num_classes=10, in_channels=1, depth=28, widen=10):
        super().__init__()
        n, w = (depth - 4) // 6, [16, 16 * widen, 32 * widen, 64 * widen]
        self.conv1 = _conv(in_channels, w[0], 3)
        for s in (1, 2, 3):
            setattr(self, f"block{s}", _Stack([_WideBlock(w[s - 1] if i == 0 else w[s], w[s], (1 if s == 1 else 2) if i == 0 else 1)
                                               for i in range(n)]))
        self.bn1 = _bn(w[3])
        self.fc = nn.Linear(w[3], num_classes)

    def forward(self, x):
        x = self.block3(self.block2(self.block1(self.conv1(x))))
        return self.fc(F.avg_pool2d(F.relu(self.bn1(x)), 8).flatten(1))


class _DualPath(nn.Module):
    """models/dpn.py:15-45 -- 1x1 / grouped 3x3 (32 groups) / 1x1 to `res + dense` channels; the first `res` channels add to the
    shortcut's, the rest of both are concatenated behind them."""

    def __init__(self, cin, mid, res, dense, stride, first):
        super().__init__()
        self.res = res
        for i, (ci, co, k, s, g) in enumerate([(cin, mid, 1, 1, 1), (mid, mid, 3, stride, 32), (mid, res + dense, 1, 1, 1)], 1):
            setattr(self, f"conv{i}", _conv(ci, co, k, s, g))
            setattr(self, f"bn{i}", _bn(co))
        self.shortcut = nn.Sequential(*([_conv(cin, res + dense, 1, stride), _bn(res + dense)] if first else []))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        s, r = self.shortcut(x), self.res
        return F.relu(torch.cat([s[:, :r] + y[:, :r], s[:, r:], y[:, r:]], 1))


class DPN92(nn.Module):
    """models/dpn.py:48-101: (mid, res, blocks, dense) per stage from the DPN-92 table; 4 x 4 average pool on the last 4 x 4 map."""
    TABLE = ((96, 256, 3, 16), (192, 512, 4, 32), (384, 1024, 20, 24), (768, 2048, 3, 128))

    def __init__(self, num_classes=10, in_channels=1):
        super().__init__()
        self.conv1, self.bn1 = _conv(in_channels, 64, 3), _bn(64)
        cin = 64
        for s, (mid, res, n, dense) in enumerate(self.TABLE, 1):
            blocks = []
            for i in range(n):
                blocks.append(_DualPath(cin, mid, res, dense, (1 if s == 1 else 2) if i == 0 else 1, i == 0))
                cin = res + (i + 2) * dense
            setattr(self, f"layer{s}", nn.Sequential(*blocks))
        self.linear = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.linear(F.avg_pool2d(x, 4).flatten(1))


class _DenseBottle(nn.Module):
    """models/densenet.py:19-44 -- BN-ReLU-1x1 (4 k) -> BN-ReLU-3x3 (k), concatenated behind the input."""

    def __init__(self, cin, growth):
        super().__init__()
        self.bn1, self.conv1 = _bn(cin), _conv(cin, 4 * growth, 1)
        self.bn2, self.conv2 = _bn(4 * growth), _conv(4 * growth, growth, 3)

    def forward(self, x):
        y = self.conv1(F.relu(self.bn1(x)))
        return torch.cat([x, self.conv2(F.relu(self.bn2(y)))], 1)


class _Squeeze(nn.Module):
    """models/densenet.py:70-84 -- BN-ReLU-1x1 to half the channels, 2 x 2 average pool."""

    def __init__(self, cin, cout):
        super().__init__()
        self.bn1, self.conv1 = _bn(cin), _conv(cin, cout, 1)

    def forward(self, x):
        return F.avg_pool2d(self.conv1(F.relu(self.bn1(x))), 2)


class DenseNetBC100(nn.Module):
    """models/densenet.py:87-147 with depth 100, growth 12, compression 2: three dense blocks of 16 bottlenecks."""

    def __init__(self, num_classes=10, in_channels=1, depth=100, growth=12):
        super().__init__()
        n, c = (depth - 4) // 6, 2 * growth
        self.conv1 = _conv(in_channels, c, 3)
        for s in (1, 2, 3):
            setattr(self, f"dense{s}", nn.Sequential(*[_DenseBottle(c + i * growth, growth) for i in range(n)]))
            c += n * growth
            if s < 3:
                setattr(self, f"trans{s}", _Squeeze(c, c // 2))
                c //= 2
        self.bn = _bn(c)
        self.fc = nn.Linear(c, num_classes)

    def forward(self, x):
        x = self.trans1(self.dense1(self.conv1(x)))
        x = self.dense3(self.trans2(self.dense2(x)))
        return self.fc(F.avg_pool2d(F.relu(self.bn(x)), 8).flatten(1))


FAMILIES = {"vgg19_bn": lambda: vgg19_bn(10, 1), "resnext29_8_64": lambda: CifarResNeXt(10), "resnet50": lambda: ResNet50(10, 1),
            "wideresnet28_10": lambda: WideResNet28_10(10, 1), "dpn92": lambda: DPN92(10, 1), "densenet_bc_100_12": lambda: DenseNetBC100(10, 1)}
