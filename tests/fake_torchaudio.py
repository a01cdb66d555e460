"""Attribute-level stand-ins for ``torchaudio.transforms.MelSpectrogram`` / ``AmplitudeToDB`` and ``torchvision``-style
``Compose`` (torchaudio / torchvision are not installed in the build or the GPU image).  They carry the attribute layout
torchaudio 0.11 gives those modules -- ``MelSpectrogram.{sample_rate,n_fft,win_length,hop_length,n_mels,power,spectrogram,
mel_scale}``, ``Spectrogram.pad_mode``, ``MelScale.{norm,mel_scale}``, ``AmplitudeToDB.{multiplier,amin,ref_value,top_db}``
-- which is what ``audiopure_amd.lowering.lower_transform`` recognises the scripts' front-ends by
(adaptive_attack_eval.py:83-85, kws_adaptive_attack_eval.py:64-66).  Calling them raises: a test that reaches their
``forward`` has fallen off the native path."""
import torch.nn as nn


class Spectrogram(nn.Module):
    def __init__(self, n_fft, win_length, hop_length, pad, power, normalized, center, pad_mode, onesided):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length, self.pad, self.power = n_fft, win_length, hop_length, pad, power
        self.normalized, self.center, self.pad_mode, self.onesided = normalized, center, pad_mode, onesided


class MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft, norm, mel_scale):
        super().__init__()
        self.n_mels, self.sample_rate, self.f_min, self.norm, self.mel_scale = n_mels, sample_rate, f_min, norm, mel_scale
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)


class MelSpectrogram(nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, pad=0,
                 n_mels=128, power=2.0, normalized=False, center=True, pad_mode="reflect", onesided=True, norm=None,
                 mel_scale="htk"):
        super().__init__()
        self.sample_rate, self.n_fft = sample_rate, n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.pad, self.power, self.normalized, self.n_mels, self.f_max, self.f_min = pad, power, normalized, n_mels, f_max, f_min
        self.spectrogram = Spectrogram(n_fft, self.win_length, self.hop_length, pad, power, normalized, center, pad_mode, onesided)
        self.mel_scale = MelScale(n_mels, sample_rate, f_min, f_max, n_fft // 2 + 1, norm, mel_scale)

    def forward(self, x):
        raise AssertionError("torchaudio stand-in called: the front-end was not lowered onto the native mel kernel")


class AmplitudeToDB(nn.Module):
    def __init__(self, stype="power", top_db=None):
        super().__init__()
        self.stype, self.top_db = stype, top_db
        self.multiplier = 10.0 if stype == "power" else 20.0
        self.amin, self.ref_value, self.db_multiplier = 1e-10, 1.0, 0.0

    def forward(self, x):
        raise AssertionError("torchaudio stand-in called: the front-end was not lowered onto the native mel kernel")


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def script_wave2spect(n_mels=32):
    """adaptive_attack_eval.py:83-85 / certified_robustness_eval.py:80-82."""
    return Compose([MelSpectrogram(n_fft=2048, hop_length=512, n_mels=n_mels, norm="slaney", pad_mode="constant",
                                   mel_scale="slaney"), AmplitudeToDB(stype="power")])


def kws_wave2spect(n_mels=40):
    """kws_adaptive_attack_eval.py:64-66."""
    return nn.Sequential(MelSpectrogram(sample_rate=16000, n_mels=n_mels), AmplitudeToDB(stype="power"))
