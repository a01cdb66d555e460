#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8 f-2 (query-batch fusion), produced by running the REFERENCE's own
``robustness_eval/certified_robust.py`` (``RobustCertificate.smooth_predict``, :33-65) and ``robustness_eval/_NES.py``
(``NES.forward``, :14-55, through the reference's ``_EOT.EOT``) -- imported from /root/reference, build container only, CPU.
Nothing of the reference is stored: inputs are seed recipes of ``audiopure_amd.synth``, outputs are the reference's numbers.

The reference draws its perturbations from torch's global generator (``torch.normal`` at certified_robust.py:47, ``torch.randn`` at
_NES.py:19).  Both are replaced here by the counter-based draws the HIP path generates in-kernel -- Philox4x32-10 + Box-Muller as
restated in ``oracle/philox.py`` -- with the keys the build documents (certification: seed, draw 0, utterance index = sample index;
NES: seed, draw = batch index, utterance index = audio * S/2 + copy), so the reference and the build see the same noise.
``statsmodels`` (absent from this image; only ``lower_conf_bound`` uses it) is stubbed.

    python tests/golden/make_golden_f2.py          # ~2 min on 8 cores

Contents of golden_f2_v1.npz
  cert/pred        per-sample arg-max of the 300 noisy copies (mini DiffWave one-shot denoise + M5), in sample order
  cert/counts      RobustCertificate.smooth_predict's return value (class histogram)
  cert/scores      the [300, 10] log-probabilities behind them (to see how close to a tie a vote is)
  cert/t_star      compute_t_star(1 / (1 + sigma^2))
  nes/mean_loss, nes/grad, nes/adver_loss, nes/adver_score, nes/predict     NES.forward's five return values
"""
from __future__ import annotations

import os
import sys
from unittest.mock import MagicMock

sys.dont_write_bytecode = True                # nothing is written under /root/reference

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.modules.setdefault("statsmodels", MagicMock())
sys.modules.setdefault("statsmodels.stats", MagicMock())
sys.modules.setdefault("statsmodels.stats.proportion", MagicMock())
import make_golden as G  # noqa: E402  (sys.path, third-party mocks, no-op .cuda(), torch.normal injector)

from audiopure_amd import synth  # noqa: E402
from oracle.philox import philox_normal  # noqa: E402

from robustness_eval.certified_robust import RobustCertificate  # noqa: E402  (the reference's)
from robustness_eval._NES import NES  # noqa: E402
from robustness_eval._EOT import EOT  # noqa: E402

torch.set_grad_enabled(False)

# ---- the workloads (tests/test_gpu_certify.py builds the same ones on the device) -------------------------------------------
CERT = dict(net_seed=1, x_seed=5, n=300, sigma=0.25, batch_size=64, philox_seed=77, L=16000, m5_seed=11)
NESW = dict(A=3, L=4000, S=10, spd=30, sigma=0.001, philox_seed=5, K=4, x_seed=8, y=[1, 2, 0], eot_size=2, eot_batch=1)


class ToyScores(torch.nn.Module):
    """scores = 4 tanh(x W^T): a stand-in for the defended system under NES (the estimator only needs per-copy losses)."""

    def __init__(self, K, L):
        super().__init__()
        self.register_buffer("w", torch.from_numpy(synth.uniform("nes_toy_w", (K, L), 1, -1.0, 1.0)) / 8.0)

    def forward(self, x):
        return 4.0 * torch.tanh(x.reshape(x.shape[0], -1) @ self.w.t())


def main():
    out = {}
    # ---- certified_robust.py:33-65 ------------------------------------------------------------------------------------------
    c = CERT
    dh = G.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    net = G.build_ref_net(synth.mini_wavenet_config(64, 12, 12), seed=c["net_seed"])
    dw = G.DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=5)
    m5 = G.build_ref_m5(10, seed=c["m5_seed"])
    rc = RobustCertificate(classifier=m5, transform=None, denoiser=dw)
    x = torch.from_numpy(synth.waveforms(1, c["L"], seed=c["x_seed"]))[0:1]           # [1, 1, L]
    done = [0]

    def normal(mean, std, size=None, **kw):                      # torch.normal(0, sigma, size=[batch, 1, L])  (:47)
        b = size[0]
        z = torch.from_numpy(philox_normal(c["philox_seed"], 0, done[0], b, c["L"])).reshape(tuple(size))
        done[0] += b
        return z * std + mean

    scores = []
    fwd = rc.forward
    rc.forward = lambda x_in: scores.append(fwd(x_in)) or scores[-1]
    torch.normal = normal
    try:
        counts = rc.smooth_predict(x, num_sampling=c["n"], sigma=c["sigma"], batch_size=c["batch_size"])
    finally:
        torch.normal = G.INJ
    assert done[0] == c["n"]
    sc = torch.cat(scores, 0)
    out["cert/scores"] = sc.numpy().copy()
    out["cert/pred"] = sc.max(1)[1].numpy().astype(np.int64)
    out["cert/counts"] = counts.numpy().astype(np.int64)
    out["cert/t_star"] = np.array([rc.compute_t_star(1 / (1 + c["sigma"] ** 2))], dtype=np.int64)
    srt = np.sort(out["cert/scores"], axis=1)
    print("cert: counts", out["cert/counts"].tolist(), " t*", int(out["cert/t_star"][0]), " smallest top-2 margin",
          float((srt[:, -1] - srt[:, -2]).min()))

    # ---- _NES.py:14-55 through _EOT.py ------------------------------------------------------------------------------------
    w = NESW
    model = ToyScores(w["K"], w["L"])
    eot = EOT(model, torch.nn.CrossEntropyLoss(reduction="none"), EOT_size=w["eot_size"], EOT_batch_size=w["eot_batch"], use_grad=False)
    nes = NES(w["spd"], w["S"], w["sigma"], eot)
    xa = torch.from_numpy(synth.waveforms(w["A"], w["L"], seed=w["x_seed"]))
    batch = [0]
    real_randn = torch.randn

    def randn(size, **kw):                                       # torch.randn([n_audios, S/2, 1, N])  (:19)
        a, h, ch, n = size
        z = philox_normal(w["philox_seed"], batch[0], 0, a * h, n).reshape(a, h, ch, n)
        batch[0] += 1
        return torch.from_numpy(z)

    torch.randn = randn
    try:
        mean_loss, grad, adver_loss, adver_score, predict = nes(xa, w["y"])
    finally:
        torch.randn = real_randn
    assert batch[0] == w["spd"] // w["S"]
    out["nes/mean_loss"] = mean_loss.numpy().copy()
    out["nes/grad"] = grad.numpy().copy()
    out["nes/adver_loss"] = adver_loss.numpy().copy()
    out["nes/adver_score"] = adver_score.numpy().copy()
    out["nes/predict"] = np.asarray(predict).astype(np.int64)
    print("nes: mean_loss", out["nes/mean_loss"], " |grad|max", float(np.abs(out["nes/grad"]).max()), " predict", out["nes/predict"])

    path = os.path.join(HERE, "golden_f2_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
