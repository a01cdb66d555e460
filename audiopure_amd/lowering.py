"""Lower whatever the reference's eval scripts hand to ``AcousticSystem`` / ``RobustCertificate`` onto the native modules.

The scripts obtain the classifier by un-pickling a whole module (audio_models/create_model.py:8-17) and build the mel
front-end themselves from torchaudio (adaptive_attack_eval.py:83-85, certified_robustness_eval.py:80-82,
kws_adaptive_attack_eval.py:64-66), so "scripts unchanged" means those objects arrive here as foreign ``nn.Module``s.
``lower_classifier`` / ``lower_transform`` recognise them structurally and return the HIP-backed equivalent:

* a module whose class is called ``M5`` with the reference's attribute tree (M5Net.py:4-20)   -> native ``M5``
* a module called ``KWSModel`` with ``CRNN_model / attn_layer / apply_attn`` (RCNN_KWS/model.py:66-90) -> native ``KWSModel``
* any other module that contains ``nn.Conv2d`` layers (VGG / ResNet / WRN / ResNeXt / DPN / DenseNet pickles)
                                                                                              -> ``NativeConvNet(module)``
  (lowered at its first input; a module whose trace meets an operator without a kernel, one left in ``train()`` mode, or a
  CPU module given CPU input keeps running as the caller built it -- ``NativeConvNet.forward`` says so once)
* ``Compose([MelSpectrogram(n_fft=2048, hop_length=512, n_mels, norm='slaney', pad_mode='constant',
  mel_scale='slaney'), AmplitudeToDB('power')])``                                             -> ``MelSpecDB(n_mels)``
* ``Sequential(MelSpectrogram(sample_rate=16000, n_mels), AmplitudeToDB('power'))`` (torchaudio defaults: n_fft 400,
  hop 200, reflect, HTK)                                                                      -> ``MelSpecDBHTK(n_mels)``

Objects that are already native are returned as they are.  Anything else is the caller's own module and is left alone
(a mel pipeline with other parameters has no kernel here; it is not silently approximated).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _is_native(m) -> bool:
    return type(m).__module__.startswith("audiopure_amd.")


def lower_classifier(clf, input_chw=None):
    from .audio_models.M5.M5Net import M5
    from .audio_models.RCNN_KWS.model import KWSModel
    from .convnet import NativeConvNet
    if clf is None or _is_native(clf) or not isinstance(clf, nn.Module):
        return clf
    if isinstance(clf, nn.DataParallel):                                     # create_model.py:11-12
        clf = clf.module
    name = type(clf).__name__
    if name == "M5" and all(hasattr(clf, a) for a in ("conv1", "bn1", "conv4", "bn4", "fc1")):
        nat = M5(n_input=clf.conv1.in_channels, first_kernel_size=clf.conv1.kernel_size[0],
                 n_output=clf.fc1.out_features, stride=clf.conv1.stride[0], n_channel=clf.conv1.out_channels)
        nat.load_state_dict(clf.state_dict())
        return _like(nat, clf)
    if name == "KWSModel" and all(hasattr(clf, a) for a in ("CRNN_model", "attn_layer", "apply_attn")):
        nat = KWSModel(in_size=clf.in_size, hidden_size=clf.hidden_size, kernel_size=tuple(clf.kernel_size),
                       stride=tuple(clf.stride), gru_num_layers=clf.gru_num_layers, num_dirs=clf.num_dirs,
                       num_classes=clf.num_classes)
        nat.load_state_dict(clf.state_dict())
        return _like(nat, clf)
    if any(isinstance(s, nn.Conv2d) for s in clf.modules()):
        was_training = clf.training
        clf.eval()
        net = NativeConvNet(clf, input_chw=input_chw)                        # None: traced at the first forward's [C,H,W]
        net.train(was_training)
        return net
    return clf


def _like(nat: nn.Module, src: nn.Module) -> nn.Module:
    p = next(src.parameters(), None)
    if p is not None:
        nat = nat.to(p.device)
    return nat.train(src.training)


def _stages(t):
    """The stages of a ``torchvision``-style ``Compose`` (``.transforms``) or an ``nn.Sequential``; else None."""
    if isinstance(t, nn.Sequential):
        return list(t)
    ts = getattr(t, "transforms", None)
    if isinstance(ts, (list, tuple)):
        return list(ts)
    return None


def _mel_signature(mel):
    """(n_fft, hop, n_mels, norm, mel_scale, pad_mode, power) of a torchaudio-shaped ``MelSpectrogram``; None if not one."""
    try:
        spec, scale = getattr(mel, "spectrogram", None), getattr(mel, "mel_scale", None)
        n_fft, hop, n_mels = int(mel.n_fft), int(mel.hop_length), int(mel.n_mels)
        norm = getattr(scale, "norm", getattr(mel, "norm", None))
        mscale = getattr(scale, "mel_scale", None)
        if not isinstance(mscale, str):
            mscale = getattr(mel, "mel_scale_name", "htk")
        pad_mode = getattr(spec, "pad_mode", getattr(mel, "pad_mode", "reflect"))
        power = float(getattr(mel, "power", 2.0) or 0.0)
        sr = int(getattr(mel, "sample_rate", 16000))
        win = int(getattr(mel, "win_length", n_fft) or n_fft)
        # everything else the kernels fix: the full band [0, sr/2], centred one-sided un-normalised frames, no extra pad,
        # a periodic Hann window (torchaudio's default window_fn; a registered ``window`` buffer is compared to it)
        f_min = float(getattr(scale, "f_min", getattr(mel, "f_min", 0.0)) or 0.0)
        f_max = getattr(scale, "f_max", getattr(mel, "f_max", None))
        full_band = f_min == 0.0 and (f_max is None or abs(float(f_max) - sr / 2.0) < 1e-6)
        plain = (bool(getattr(spec, "center", getattr(mel, "center", True)))
                 and not bool(getattr(spec, "normalized", getattr(mel, "normalized", False)))
                 and bool(getattr(spec, "onesided", getattr(mel, "onesided", True)))
                 and int(getattr(spec, "pad", getattr(mel, "pad", 0)) or 0) == 0)
        window = getattr(spec, "window", None)
        if isinstance(window, torch.Tensor):
            plain = plain and window.numel() == win and torch.allclose(window.detach().float().cpu(),
                                                                         torch.hann_window(win, periodic=True), atol=1e-6)
        return dict(n_fft=n_fft, hop=hop, n_mels=n_mels, norm=norm, mel_scale=mscale, pad_mode=pad_mode, power=power,
                    sample_rate=sr, win_length=win, standard=bool(full_band and plain))
    except (AttributeError, TypeError, ValueError):
        return None


def _is_power_to_db(a) -> bool:
    """torchaudio ``AmplitudeToDB(stype='power')``: multiplier 10, amin 1e-10, ref 1, no top_db."""
    try:
        return (float(a.multiplier) == 10.0 and getattr(a, "top_db", None) is None
                and abs(float(getattr(a, "amin", 1e-10)) - 1e-10) < 1e-20 and float(getattr(a, "ref_value", 1.0)) == 1.0)
    except (AttributeError, TypeError, ValueError):
        return False


def lower_transform(t):
    from .transforms.melspec import MelSpecDB, MelSpecDBHTK
    if t is None or _is_native(t):
        return t
    st = _stages(t)
    if st is None or len(st) != 2 or not _is_power_to_db(st[1]):
        return t
    sig = _mel_signature(st[0])
    if (sig is None or sig["power"] != 2.0 or sig["sample_rate"] != 16000 or sig["win_length"] != sig["n_fft"]
            or not sig["standard"]):                                         # f_max = 4000, center = False, ...: not ours
        return t
    if (sig["n_fft"], sig["hop"], sig["norm"], sig["mel_scale"], sig["pad_mode"]) == (2048, 512, "slaney", "slaney", "constant"):
        return MelSpecDB(n_mels=sig["n_mels"])                               # adaptive_attack_eval.py:83-85
    if (sig["n_fft"], sig["hop"], sig["norm"], sig["mel_scale"], sig["pad_mode"]) == (400, 200, None, "htk", "reflect"):
        return MelSpecDBHTK(n_mels=sig["n_mels"])                            # kws_adaptive_attack_eval.py:64-66
    return t
