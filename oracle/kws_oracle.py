"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the KWS route: ``KWSModel.forward``
(audio_models/RCNN_KWS/model.py:66-114: depthwise + grouped-pointwise Conv1d, 2-layer bidirectional GRU written out gate
by gate, additive attention, linear, log-softmax) and the default-parameter torchaudio front-end the KWS script builds
(kws_adaptive_attack_eval.py:65-67: MelSpectrogram(sample_rate=16000, n_mels) = n_fft 400, hop 200, periodic Hann,
centre + reflect padding, power 2, HTK mel scale, no filter normalisation; AmplitudeToDB('power')).

Pinned: ``kws_forward`` against golden vectors produced by the reference's own model class
(tests/golden/make_golden_kws.py).  Unpinned: the mel front-end (torchaudio is not importable here), restated from
torchaudio 0.11's documented defaults and cross-checked against an explicit float64 DFT in the tests.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def kws_forward(sd: dict, x: torch.Tensor, hidden: int = 64, stride=(8, 2), kernel=(20, 5)) -> torch.Tensor:
    """x: [B,1,n_mels,T] or [B,n_mels,T] mel-dB -> log-probabilities [B,num_classes] (model.py:92-114)."""
    t = {k: (torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v).to(x.dtype) for k, v in sd.items()}
    x = x.squeeze(1) if x.ndim == 4 else x
    n_in = x.shape[1]
    h = F.conv1d(x, t["CRNN_model.sepconv.0.weight"], t["CRNN_model.sepconv.0.bias"], stride=stride[1], groups=n_in)   # :7-9
    h = F.conv1d(h, t["CRNN_model.sepconv.1.weight"], t["CRNN_model.sepconv.1.bias"], stride=stride[0],
                 groups=int(n_in / kernel[0]))                                                                           # :10-11
    seq = h.permute(2, 0, 1)                                            # (T2, B, H)   :28
    H = hidden
    for layer in (0, 1):
        outs = []
        for suffix, order in (("", range(seq.shape[0])), ("_reverse", range(seq.shape[0] - 1, -1, -1))):
            wi, wh = t[f"CRNN_model.gru.weight_ih_l{layer}{suffix}"], t[f"CRNN_model.gru.weight_hh_l{layer}{suffix}"]
            bi, bh = t[f"CRNN_model.gru.bias_ih_l{layer}{suffix}"], t[f"CRNN_model.gru.bias_hh_l{layer}{suffix}"]
            hs = torch.zeros(seq.shape[1], H, dtype=x.dtype)                         # hidden=None -> zeros (:99-100)
            out = [None] * seq.shape[0]
            for s in order:
                gi, gh = seq[s] @ wi.t() + bi, hs @ wh.t() + bh
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                hs = (1 - z) * n + z * hs
                out[s] = hs
            outs.append(torch.stack(out))
        seq = torch.cat(outs, dim=2)                                    # (T2, B, 2H)
    e = torch.tanh(seq @ t["attn_layer.Wx_b.weight"].t() + t["attn_layer.Wx_b.bias"]) @ t["attn_layer.Vt.weight"].t()   # :40-43
    e = e.squeeze(2).t()                                                # (B, T2)      :104-108
    a = torch.softmax(e, dim=-1).unsqueeze(1)                           # :55-57
    c = torch.bmm(a, seq.transpose(0, 1)).squeeze(1)
    return F.log_softmax(c @ t["apply_attn.U.weight"].t(), dim=-1)


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank_htk(n_mels: int, n_freqs: int = 201, f_max: float = 8000.0) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, f_max, n_mels, 16000, norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = np.linspace(0.0, f_max, n_freqs)
    f_pts = mel_to_hz_htk(np.linspace(hz_to_mel_htk(0.0), hz_to_mel_htk(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def melspec_db_htk(x: np.ndarray, n_mels: int) -> np.ndarray:
    """x [B,1,L] -> [B,1,n_mels,1 + L // 200] in float64 (explicit rfft per frame)."""
    x = np.asarray(x, dtype=np.float64).reshape(x.shape[0], -1)
    n_fft, hop = 400, 200
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)            # periodic Hann
    xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    n_frames = 1 + x.shape[1] // hop
    fb = mel_filterbank_htk(n_mels)
    out = np.empty((x.shape[0], 1, n_mels, n_frames))
    for f in range(n_frames):
        seg = xp[:, f * hop:f * hop + n_fft] * win
        p = np.abs(np.fft.rfft(seg, axis=1)) ** 2
        out[:, 0, :, f] = 10.0 * np.log10(np.maximum(p @ fb, 1e-10))
    return out
