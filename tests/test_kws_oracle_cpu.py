"""KWS route, CPU side: the oracle against golden vectors from the reference's own model class, the
reference's state-dict keys, the HTK mel front-end against an explicit float64 DFT."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiopure_amd import synth  # noqa: E402
from oracle import kws_oracle as K  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "golden_kws_v1.npz"))


def _sd(n_mels):
    return {k.split("/sd/")[1]: G[k] for k in G.files if k.startswith(f"m{n_mels}/sd/")}


def test_oracle_matches_reference_model_golden():
    for n_mels in (40, 32):
        sd = _sd(n_mels)
        for T in (81, 161, 47):
            x = torch.from_numpy(synth.uniform(f"kwsx/{n_mels}/{T}", (3, 1, n_mels, T), 1, -80.0, 20.0))
            assert np.abs(K.kws_forward(sd, x).numpy() - G[f"m{n_mels}/logp_T{T}"]).max() < 2e-6
        x1 = torch.from_numpy(synth.uniform(f"kwsx/{n_mels}/81", (3, 1, n_mels, 81), 1, -80.0, 20.0))[:1]
        assert np.abs(K.kws_forward(sd, x1).numpy() - G[f"m{n_mels}/logp_T81_b1"]).max() < 2e-6


def test_native_model_has_the_references_state_dict():
    from audiopure_amd.audio_models.RCNN_KWS import KWSModel
    for n_mels in (40, 32):
        m = KWSModel(in_size=n_mels)
        sd = _sd(n_mels)
        assert list(m.state_dict().keys()) == list(sd.keys())
        assert all(tuple(v.shape) == sd[k].shape for k, v in m.state_dict().items())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


def test_htk_mel_front_end_restates_torchaudio_defaults():
    fb = K.mel_filterbank_htk(40)
    assert fb.shape == (201, 40) and (fb >= 0).all() and fb.max() <= 1.0 + 1e-12
    pk = fb.argmax(0)
    assert (np.diff(pk) > 0).all()                                  # one triangle per filter, peaks increasing
    assert abs(K.mel_to_hz_htk(K.hz_to_mel_htk(1234.5)) - 1234.5) < 1e-9
    # a pure tone lands in the filter whose triangle covers it, 10 log10 of the power
    L, f0 = 3200, 1000.0
    x = (0.5 * np.sin(2 * np.pi * f0 * np.arange(L) / 16000.0))[None, None, :]
    m = K.melspec_db_htk(x, 40)
    assert m.shape == (1, 1, 40, 1 + L // 200)
    centre = m[0, 0, :, 5]
    f_pts = K.mel_to_hz_htk(np.linspace(0, K.hz_to_mel_htk(8000.0), 42))
    assert f_pts[centre.argmax()] <= f0 <= f_pts[centre.argmax() + 2]
