"""Lowering of the 2-D ConvNet classifiers (ATen tape -> fused plan) checked on CPU: the plan, interpreted with plain
torch ops, must reproduce the original module.  Structures mirror the reference's families
(audio_models/ConvNets_SpeechCommands/models/{vgg,resnext,wideresnet,densenet,dpn}.py)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from synth_convnets import CifarResNeXt, synth_init, vgg19_bn
from audiopure_amd.convnet import lower
from audiopure_amd import synth
from oracle.convnet_plan_oracle import run_plan_torch


class PreActBlock(nn.Module):      # wideresnet.py:30-39 (BN-ReLU before conv, conv shortcut when shapes change)
    def __init__(s, cin, cout, stride):
        super().__init__()
        s.bn1, s.conv1 = nn.BatchNorm2d(cin), nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        s.bn2, s.conv2 = nn.BatchNorm2d(cout), nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        s.equal = cin == cout
        s.short = None if s.equal else nn.Conv2d(cin, cout, 1, stride, 0, bias=False)

    def forward(s, x):
        if not s.equal:
            x = F.relu(s.bn1(x))
            out = x
        else:
            out = F.relu(s.bn1(x))
        out = s.conv2(F.relu(s.bn2(s.conv1(out))))
        return torch.add(x if s.equal else s.short(x), out)


class DenseDPN(nn.Module):         # densenet.py:29-41 (cat) + dpn.py:36-44 (slices, partial add, cat)
    def __init__(s):
        super().__init__()
        s.conv0 = nn.Conv2d(1, 16, 3, 1, 1, bias=False)
        s.bn1, s.c1 = nn.BatchNorm2d(16), nn.Conv2d(16, 12, 1, bias=False)
        s.bn2, s.c2 = nn.BatchNorm2d(28), nn.Conv2d(28, 28, 3, 1, 1, groups=4, bias=False)
        s.bn3 = nn.BatchNorm2d(28)
        s.pre = PreActBlock(36, 48, 2)
        s.pre2 = PreActBlock(48, 48, 1)
        s.fc = nn.Linear(48, 7)

    def forward(s, x):
        x = s.conv0(x)
        x = torch.cat((x, s.c1(F.relu(s.bn1(x)))), 1)                       # 28 ch
        out = s.bn3(s.c2(F.relu(s.bn2(x))))
        d = 20
        x = torch.cat([x[:, :d, :, :] + out[:, :d, :, :], x[:, d:, :, :], out[:, d:, :, :]], 1)   # 36 ch
        x = F.relu(x)
        x = s.pre2(s.pre(x))
        x = F.avg_pool2d(x, 16)
        return s.fc(x.view(x.size(0), -1))


def _x(B=3):
    return torch.from_numpy(synth.uniform("mel", (B, 1, 32, 32), 3, -2.0, 2.0))


@pytest.mark.parametrize("make", [lambda: vgg19_bn(10, 1, width_div=8), lambda: CifarResNeXt(10, cardinality=4, base_width=8),
                                  DenseDPN])
def test_plan_reproduces_module(make):
    m = synth_init(make(), 1)
    plan = lower(m)
    x = _x()
    with torch.no_grad():
        ref = m(x)
    got = run_plan_torch(plan, x)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    kinds = [s.kind for s in plan.steps]
    assert "conv" in kinds


def test_fusions_happen():
    plan = lower(synth_init(vgg19_bn(10, 1, width_div=8), 1))
    kinds = [s.kind for s in plan.steps]
    assert kinds.count("conv") == 16 + 3 and kinds.count("affine") == 0 and kinds.count("pool") == 5   # BN+ReLU folded
    plan = lower(synth_init(CifarResNeXt(10, cardinality=4, base_width=8), 1))
    convs = [s for s in plan.steps if s.kind == "conv"]
    assert len(convs) == 1 + 9 * 3 + 3 + 1 and sum(1 for s in convs if s.p["res"] is not None) == 9     # residuals fused
    assert not any(s.kind in ("add", "affine") for s in plan.steps)


def test_unsupported_graph_is_loud():
    class Bad(nn.Module):
        def forward(s, x):
            return torch.sigmoid(x).mean((2, 3))
    with pytest.raises(NotImplementedError):
        lower(Bad())


def test_restated_models_have_reference_state_dict_keys():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_convnets_v1.npz"))
    assert list(vgg19_bn(10, 1).state_dict().keys()) == list(g["vgg19_bn/keys"])
    assert list(CifarResNeXt(10).state_dict().keys()) == list(g["resnext29_8_64/keys"])
    assert sum(p.numel() for p in CifarResNeXt(10).parameters()) == 34425546
    from synth_convnets import FAMILIES
    for name, make in FAMILIES.items():                                          # all six of models/__init__.py:8-45
        assert set(make().state_dict()) == set(str(k) for k in g[f"{name}/keys"]), name


@pytest.mark.parametrize("name", ["resnet50", "wideresnet28_10", "dpn92", "densenet_bc_100_12"])
def test_restated_families_reproduce_reference_logits_and_lower(name):
    """The containers the GPU tests run (tools/synth_convnets.py) ARE the reference's networks: with weights keyed on the
    state-dict names they reproduce logits of the reference's own classes (golden_convnets_v1.npz) on CPU, and their
    lowered plan replays to the same logits through the plan oracle."""
    import os
    from audiopure_amd import synth
    from synth_convnets import FAMILIES
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_convnets_v1.npz"))
    m = synth_init(FAMILIES[name](), 0)
    x = torch.from_numpy(synth.uniform("mel", (2, 1, 32, 32), 3, -2.0, 2.0))
    with torch.no_grad():
        y = m(x)
    gold = g[f"{name}/logits"]
    assert np.abs(y.numpy() - gold).max() <= 1e-6 * np.abs(gold).max()
    yp = run_plan_torch(lower(m), x)
    assert float((y - yp).abs().max() / y.abs().max()) < 1e-4


def test_mel_front_end_with_other_band_or_framing_is_left_alone():
    """lower_transform only replaces the exact pipelines the kernels implement (adaptive_attack_eval.py:83-85,
    kws_adaptive_attack_eval.py:64-66): a different band, framing or normalisation stays the caller's module."""
    import fake_torchaudio as ta
    from audiopure_amd.lowering import lower_transform
    from audiopure_amd.transforms.melspec import MelSpecDB

    def pipe(**kw):
        a = dict(n_fft=2048, hop_length=512, n_mels=32, norm="slaney", pad_mode="constant", mel_scale="slaney")
        a.update(kw)
        return ta.Compose([ta.MelSpectrogram(**a), ta.AmplitudeToDB(stype="power")])

    assert type(lower_transform(pipe())) is MelSpecDB
    assert type(lower_transform(pipe(f_max=8000.0))) is MelSpecDB
    for kw in (dict(f_max=4000.0), dict(f_min=50.0), dict(center=False), dict(normalized=True), dict(onesided=False),
               dict(pad=16)):
        t = pipe(**kw)
        assert lower_transform(t) is t, kw


def test_lazily_wrapped_module_keeps_the_callers_semantics_off_the_native_path(monkeypatch):
    """lower_classifier wraps any Conv2d-bearing module on sight; in train() mode and on CPU (CPU parameters, CPU input)
    the wrapper runs the module itself, as the reference's scripts would have (acoustic_system.py:35-51) -- outside
    AUDIOPURE_STRICT, and said with a RuntimeWarning."""
    import warnings
    import torch
    import torch.nn as nn
    from audiopure_amd.convnet import NativeConvNet
    from audiopure_amd.lowering import lower_classifier
    monkeypatch.setenv("AUDIOPURE_STRICT", "0")
    warnings.filterwarnings("ignore", message=".*PyTorch operators.*", category=RuntimeWarning)
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                      nn.Linear(4, 3))
    x = torch.randn(2, 1, 8, 8)
    w = lower_classifier(m.train())
    assert isinstance(w, NativeConvNet) and w.training
    assert w(x).shape == (2, 3)                                               # train mode: the module itself (BN batch stats)
    w.eval()
    with torch.no_grad():
        assert torch.equal(w(x), m(x))                                        # CPU module + CPU input: the module itself


def test_strict_mode_turns_every_route_off_the_native_path_into_an_error(monkeypatch):
    """under AUDIOPURE_STRICT=1 (the suite's and bench.py's default) train() mode and a CPU module raise instead
    of running the caller's module on PyTorch operators, and a lowering failure that is not "no kernel for this operator"
    is never caught."""
    import torch
    import torch.nn as nn
    from audiopure_amd import _native as N
    from audiopure_amd import convnet
    from audiopure_amd.lowering import lower_classifier
    monkeypatch.setenv("AUDIOPURE_STRICT", "1")
    assert convnet.strict()
    m = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1), nn.BatchNorm2d(4), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                      nn.Linear(4, 3))
    x = torch.randn(2, 1, 8, 8)
    w = lower_classifier(m.train())
    with pytest.raises((N.NativeError, RuntimeWarning)):
        w(x)
    w.eval()
    with pytest.raises(N.NativeError, match="AUDIOPURE_STRICT"):
        w(x)
    # patterns without a kernel are NotImplementedError (the only failure the lazy wrapper may answer with the caller's module)
    with pytest.raises(NotImplementedError):
        convnet.lower(nn.Sequential(nn.Conv2d(1, 4, 3, dilation=2), nn.AdaptiveAvgPool2d(1), nn.Flatten()), (1, 8, 8))
    with pytest.raises(NotImplementedError):
        convnet.lower(nn.Sequential(nn.Conv2d(1, 4, 3), nn.GELU(), nn.AdaptiveAvgPool2d(1), nn.Flatten()), (1, 8, 8))
