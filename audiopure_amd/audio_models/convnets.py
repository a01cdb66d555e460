"""Restatements of two of the reference's mel-spectrogram classifiers, with the reference's attribute names and
state-dict keys so its checkpoints load unchanged:

* ``vgg19_bn``  (audio_models/ConvNets_SpeechCommands/models/vgg.py:30-201; also what ``create_model('resnet18')``
  silently returns, models/__init__.py:18-21,44-45)
* ``CifarResNeXt`` ResNeXt-29 8x64d, the default classifier (models/resnext.py:23-142;
  train_speech_commands.py:43)

They are ordinary ``nn.Module`` definitions (parameter containers + the graph); inference runs through
``audiopure_amd.convnet.NativeConvNet``, which lowers ANY of the reference's ConvNet families (also the reference's own
pickled modules) onto the HIP primitives.  Synthetic deterministic weights for tests / benches: ``synth_init``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import synth

_VGG19 = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG(nn.Module):
    def __init__(self, cfg=_VGG19, num_classes=10, in_channels=1, width_div=1):
        super().__init__()
        layers, c = [], in_channels
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                v = max(v // width_div, 8)
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        hid = 4096 // width_div
        self.classifier = nn.Sequential(nn.Linear(c, hid), nn.ReLU(True), nn.Dropout(), nn.Linear(hid, hid), nn.ReLU(True),
                                        nn.Dropout(), nn.Linear(hid, num_classes))

    def forward(self, x):
        x = self.features(x)
        x = x.view(x.size(0), -1)
        return self.classifier(x)


def vgg19_bn(num_classes=10, in_channels=1, width_div=1):
    return VGG(_VGG19, num_classes, in_channels, width_div)


class ResNeXtBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, stride, cardinality, base_width, widen_factor):
        super().__init__()
        width_ratio = out_channels / (widen_factor * 64.)
        D = cardinality * int(base_width * width_ratio)
        self.conv_reduce = nn.Conv2d(in_channels, D, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_reduce = nn.BatchNorm2d(D)
        self.conv_conv = nn.Conv2d(D, D, kernel_size=3, stride=stride, padding=1, groups=cardinality, bias=False)
        self.bn = nn.BatchNorm2d(D)
        self.conv_expand = nn.Conv2d(D, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_expand = nn.BatchNorm2d(out_channels)
        self.shortcut = nn.Sequential()
        if in_channels != out_channels:
            self.shortcut.add_module('shortcut_conv', nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride,
                                                                padding=0, bias=False))
            self.shortcut.add_module('shortcut_bn', nn.BatchNorm2d(out_channels))

    def forward(self, x):
        b = F.relu(self.bn_reduce(self.conv_reduce(x)), inplace=True)
        b = F.relu(self.bn(self.conv_conv(b)), inplace=True)
        b = self.bn_expand(self.conv_expand(b))
        return F.relu(self.shortcut(x) + b, inplace=True)


class CifarResNeXt(nn.Module):
    def __init__(self, nlabels, cardinality=8, depth=29, base_width=64, widen_factor=4, in_channels=1):
        super().__init__()
        self.cardinality, self.depth, self.base_width, self.widen_factor = cardinality, depth, base_width, widen_factor
        self.block_depth = (depth - 2) // 9
        self.stages = [64, 64 * widen_factor, 128 * widen_factor, 256 * widen_factor]
        self.conv_1_3x3 = nn.Conv2d(in_channels, 64, 3, 1, 1, bias=False)
        self.bn_1 = nn.BatchNorm2d(64)
        self.stage_1 = self.block('stage_1', self.stages[0], self.stages[1], 1)
        self.stage_2 = self.block('stage_2', self.stages[1], self.stages[2], 2)
        self.stage_3 = self.block('stage_3', self.stages[2], self.stages[3], 2)
        self.classifier = nn.Linear(self.stages[3], nlabels)

    def block(self, name, cin, cout, pool_stride=2):
        blk = nn.Sequential()
        for i in range(self.block_depth):
            blk.add_module('%s_bottleneck_%d' % (name, i),
                           ResNeXtBottleneck(cin if i == 0 else cout, cout, pool_stride if i == 0 else 1, self.cardinality,
                                             self.base_width, self.widen_factor))
        return blk

    def forward(self, x):
        x = F.relu(self.bn_1(self.conv_1_3x3(x)), inplace=True)
        x = self.stage_3(self.stage_2(self.stage_1(x)))
        x = F.avg_pool2d(x, 8, 1)
        return self.classifier(x.view(-1, self.stages[3]))


def synth_init(model: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic, torch-RNG-independent weights keyed on the state-dict names (He-scaled convs, BN statistics away
    from the identity so the fold is exercised)."""
    sd = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(100)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(synth.uniform("cn/" + k, shape, seed, 0.5, 1.5))
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(synth.uniform("cn/" + k, shape, seed, -0.2, 0.2))
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            a = (6.0 / fan_in) ** 0.5
            sd[k] = torch.from_numpy(synth.uniform("cn/" + k, shape, seed, -a, a))
        elif k.endswith("weight"):                       # BN gamma
            sd[k] = torch.from_numpy(synth.uniform("cn/" + k, shape, seed, 0.7, 1.3))
        else:                                            # biases / BN beta
            sd[k] = torch.from_numpy(synth.uniform("cn/" + k, shape, seed, -0.1, 0.1))
    model.load_state_dict(sd)
    return model.eval()
