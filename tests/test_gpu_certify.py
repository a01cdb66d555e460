"""the certification sampling loop on the device (RobustCertificate.smooth_predict / certify)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _setup(dev):
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.audio_models.M5.M5Net import M5
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 1).items()})
    dw = DiffWave(model=net.to(dev), diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=5)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.m5_state_dict(10).items()})
    return dw, m5.to(dev).eval()


def test_smooth_predict_equals_the_reference_composition():
    from audiopure_amd import _native as N
    from audiopure_amd.robustness_eval.certified_robust import RobustCertificate
    dev = torch.device("cuda:0")
    dw, m5 = _setup(dev)
    rc = RobustCertificate(classifier=m5, transform=None, denoiser=dw)
    rc.seed, rc.native_batch = 77, 128
    x = torch.from_numpy(synth.waveforms(1, 16000, seed=5))[0:1].to(dev)
    n, sigma = 300, 0.25
    counts = rc.smooth_predict(x, num_sampling=n, sigma=sigma, batch_size=64)
    assert counts.dtype == torch.int64 and int(counts.sum()) == n
    assert torch.equal(counts, rc.smooth_predict(x, num_sampling=n, sigma=sigma, batch_size=64))      # same key, same votes
    # the reference's steps (certified_robust.py:45-55) with the same perturbations materialised sample by sample
    z = torch.empty((n, 16000), device=dev)
    for i in range(0, n, 100):
        N.check(N.lib().ap_philox_normal(N.ptr(z[i:i + 100]), 77, 0, i, 100, 16000, N.stream()))
    ab_star = 1 / (1 + sigma ** 2)
    dw.reverse_timestep = rc.compute_t_star(ab_star)
    x_in = ab_star ** 0.5 * (x.expand(n, 1, 16000) + sigma * z.reshape(n, 1, 16000))
    pred = m5(dw.one_shot_denoise(x_in)).argmax(1).cpu()
    ref = torch.bincount(pred, minlength=10)
    assert int((ref - counts).abs().sum()) <= 4            # an arg-max can flip on a 1e-7 difference of the two orderings
    rc.seed = 78
    assert not torch.equal(rc.smooth_predict(x, num_sampling=n, sigma=sigma), counts) or int(counts.max()) == n


def test_certify_and_randomised_smoothing_paths():
    from audiopure_amd.robustness_eval.certified_robust import RobustCertificate
    dev = torch.device("cuda:0")
    dw, m5 = _setup(dev)
    x = torch.from_numpy(synth.waveforms(2, 16000, seed=6)).to(dev)
    y = torch.tensor([1, 4])
    for den in (dw, None):
        rc = RobustCertificate(classifier=m5, transform=None, denoiser=den)
        y_pred, radius = rc.certify(x=x, y=y, sigma=0.25, n_0=64, n=400, batch_size=64)
        assert y_pred.shape == y.shape and radius.shape == y.shape
        for i in range(2):
            assert (y_pred[i] == -1 and radius[i] == 0) or (0 <= y_pred[i] < 10 and radius[i] > 0 and math.isfinite(float(radius[i])))


def test_nes_queries_and_gradient_estimate_match_the_reference_formula():
    """robustness_eval/_NES.py:14-55 restated with the SAME noise materialised (Philox draws): antithetic copies, the
    unperturbed lead copy of the first batch, mean(loss * noise) / sigma / num_batches."""
    from audiopure_amd import _native as N
    from audiopure_amd.robustness_eval._NES import NES
    dev = torch.device("cuda:0")
    A, L, S, spd, sigma = 3, 4000, 10, 30, 0.001
    x = torch.from_numpy(synth.waveforms(A, L, seed=8)).to(dev)
    wgt = torch.from_numpy(synth.uniform("nesw", (1, 1, L), 1, -1.0, 1.0)).to(dev)
    seen = []

    class Wrap:                                         # stands in for robustness_eval/_EOT.py:EOT (use_grad=False)
        EOT_size, EOT_batch_size = 1, 1

        def __call__(self, xb, yb):
            seen.append(xb.clone())
            s = (xb * wgt).sum(dim=(1, 2))
            scores = torch.stack([s, -s, 0.5 * s], 1)
            loss = s + 0.1 * s * s + 0.01 * yb.float()
            dec = [[int(v)] for v in scores.argmax(1).cpu().reshape(-1)]
            return scores, loss, None, [sum((dec[a * (xb.shape[0] // A) + k] for k in range(xb.shape[0] // A)), []) for a in range(A)]

    nes = NES(spd, S, sigma, Wrap())
    nes.seed = 5
    y = [1, 2, 0]
    mean_loss, grad, adver_loss, adver_score, predict = nes(x, y)
    assert grad.shape == x.shape and adver_loss.shape == (A,) and adver_score.shape == (A, 3) and len(predict) == A
    # reference arithmetic on the materialised noise
    g_ref = torch.zeros_like(x)
    ml_ref = torch.zeros(A, device=dev)
    for i in range(spd // S):
        z = torch.empty((A * (S // 2), L), device=dev)
        N.check(N.lib().ap_philox_normal(N.ptr(z), 5, i, 0, A * (S // 2), L, N.stream()))
        noise = z.reshape(A, S // 2, 1, L)
        noise = torch.cat((noise, -noise), 1)
        if i == 0:
            noise = torch.cat((torch.zeros_like(x).unsqueeze(1), noise), 1)
        ev = (noise * sigma + x.unsqueeze(1)).view(-1, 1, L)
        assert torch.allclose(seen[i], ev, rtol=0, atol=1e-7)
        yb = torch.tensor(y, device=dev).repeat_interleave(noise.shape[1])
        _, loss, _, _ = Wrap()(ev, yb)
        loss = loss.view(A, -1)
        if i == 0:
            assert torch.allclose(adver_loss, loss[:, 0])
            loss, noise = loss[:, 1:], noise[:, 1:]
        g_ref += torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
        ml_ref += loss.mean(1)
    g_ref = g_ref / sigma / (spd // S)
    assert torch.allclose(mean_loss, ml_ref / (spd // S), atol=1e-6)
    err = float((grad - g_ref).abs().max() / g_ref.abs().max())
    assert err < 1e-3, err          # the estimate is a difference of nearly equal losses: fp32 cancellation on both sides
