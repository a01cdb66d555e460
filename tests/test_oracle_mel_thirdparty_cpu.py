"""The mel front-end oracles against an INDEPENDENT third-party implementation of the same documented algorithms:
``transformers.audio_utils`` (HuggingFace; its ``mel_filter_bank`` / ``spectrogram`` / ``power_to_db`` are written to reproduce
librosa and torchaudio).  torchaudio 0.11 / librosa 0.9.1 -- what the reference's scripts call (adaptive_attack_eval.py:83-85,
kws_adaptive_attack_eval.py:64-66, transforms/transforms_stft.py:14-28,101-114) -- are not installable here, so by the
strict rule these oracles stay "parity unpinned"; this test removes the weaker worry that they are only consistent with
themselves: filterbanks, framing / padding, power spectrum and dB conversion are all checked against code we did not write."""
import numpy as np
import pytest
import torch

A = pytest.importorskip("transformers.audio_utils")

from audiopure_amd import synth
from oracle import diffwave_oracle as O
from oracle import kws_oracle as K


def _hann(n):
    return np.hanning(n + 1)[:-1]                       # periodic Hann, as torch.hann_window(n, periodic=True)


@pytest.mark.parametrize("n_mels", [32, 40])
def test_slaney_filterbank_matches_third_party(n_mels):
    ours = O.mel_filterbank(1025, 0.0, 8000.0, n_mels, 16000)                        # torchaudio melscale_fbanks(slaney, slaney)
    theirs = A.mel_filter_bank(num_frequency_bins=1025, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=8000.0,
                               sampling_rate=16000, norm="slaney", mel_scale="slaney")
    assert ours.shape == theirs.shape == (1025, n_mels)
    assert np.abs(ours - theirs).max() < 1e-6 * np.abs(theirs).max()


@pytest.mark.parametrize("n_mels", [40, 32])
def test_htk_filterbank_matches_third_party(n_mels):
    ours = K.mel_filterbank_htk(n_mels, 201, 8000.0)                                 # torchaudio defaults: HTK scale, no norm
    theirs = A.mel_filter_bank(num_frequency_bins=201, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=8000.0,
                               sampling_rate=16000, norm=None, mel_scale="htk")
    assert np.abs(np.asarray(ours).reshape(theirs.shape) - theirs).max() < 1e-6


def test_mel_db_pipeline_of_the_eval_scripts_matches_third_party():
    """MelSpectrogram(n_fft=2048, hop=512, n_mels=32, norm='slaney', pad_mode='constant', mel_scale='slaney') -> AmplitudeToDB
    ('power') -- adaptive_attack_eval.py:83-85."""
    x = synth.waveforms(2, 16000, seed=77)
    ours = O.melspec_db(torch.from_numpy(x), n_mels=32).numpy()[:, 0]
    fb = A.mel_filter_bank(1025, 32, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")
    for b in range(2):
        theirs = A.spectrogram(x[b, 0].astype(np.float64), _hann(2048), frame_length=2048, hop_length=512, fft_length=2048, power=2.0,
                               center=True, pad_mode="constant", onesided=True, mel_filters=fb, mel_floor=1e-10, log_mel="dB",
                               reference=1.0, min_value=1e-10, db_range=None, dtype=np.float64)
        assert theirs.shape == ours[b].shape == (32, 32)
        assert np.abs(ours[b] - theirs).max() < 2e-3                                 # dB; fp32 STFT against float64


def test_librosa_style_power_to_db_matches_third_party():
    """ToSTFT + ToMelSpectrogramFromSTFT (transforms/transforms_stft.py:14-28,101-114): slaney mel of |STFT|^2, then
    power_to_db(ref=np.max) with top_db = 80."""
    x = synth.waveforms(2, 16000, seed=78)
    ours = O.melspec_db(torch.from_numpy(x), n_mels=32, ref_max=True, top_db=80.0).numpy()[:, 0]
    fb = A.mel_filter_bank(1025, 32, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")
    for b in range(2):
        p = A.spectrogram(x[b, 0].astype(np.float64), _hann(2048), frame_length=2048, hop_length=512, fft_length=2048, power=2.0,
                          center=True, pad_mode="constant", onesided=True, mel_filters=fb, mel_floor=1e-10, dtype=np.float64)
        theirs = A.power_to_db(p, reference=float(p.max()), min_value=1e-10, db_range=80.0)
        assert np.abs(ours[b] - theirs).max() < 2e-3


def test_kws_htk_mel_db_matches_third_party():
    """Sequential(MelSpectrogram(sample_rate=16000, n_mels=40), AmplitudeToDB('power')) -- kws_adaptive_attack_eval.py:64-66:
    n_fft 400, hop 200, reflect padding, HTK scale."""
    x = synth.waveforms(2, 12000, seed=79)
    ours = np.asarray(K.melspec_db_htk(x, 40))
    ours = ours.reshape(2, 40, -1)
    fb = A.mel_filter_bank(201, 40, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    for b in range(2):
        theirs = A.spectrogram(x[b, 0].astype(np.float64), _hann(400), frame_length=400, hop_length=200, fft_length=400, power=2.0,
                               center=True, pad_mode="reflect", onesided=True, mel_filters=fb, mel_floor=1e-10, log_mel="dB",
                               reference=1.0, min_value=1e-10, db_range=None, dtype=np.float64)
        assert theirs.shape == ours[b].shape
        assert np.abs(ours[b] - theirs).max() < 2e-3
