"""``NES`` with the reference's call surface (robustness_eval/_NES.py:5-55) and its query batch built on the device
(SURVEY.md section 8 f-2).

The reference draws ``torch.randn([n_audios, S/2, 1, N])`` per batch, concatenates its negation, broadcasts, evaluates,
and averages ``loss * noise`` over that tensor.  Here the antithetic copies are written by ``ap_nes_perturb`` from
counter-based Philox noise and the gradient estimate is formed by ``ap_nes_grad`` from the per-copy losses, regenerating
the same noise -- the [n_audios][S][N] noise tensor never exists, nothing comes from the host generator.
The model evaluation itself is the caller's ``EOT_wrapper`` (robustness_eval/_EOT.py), unchanged.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N


def resolve_prediction(decisions):                                           # robustness_eval/_utils.py:129-136
    from collections import Counter
    return np.array([Counter(d).most_common(1)[0][0] for d in decisions])


class NES(nn.Module):

    def __init__(self, samples_per_draw, samples_per_draw_batch, sigma, EOT_wrapper):
        super().__init__()
        self.samples_per_draw = samples_per_draw
        self.samples_per_draw_batch_size = samples_per_draw_batch
        self.sigma = sigma
        self.EOT_wrapper = EOT_wrapper
        # Philox key of this estimator's perturbations.  FAKEBOB builds a fresh NES on every attack iteration
        # (black_box_attack.py:181) and the reference draws fresh torch.randn each time, so the key must differ from
        # instance to instance: it is drawn from torch's global generator ("seed torch, get reproducible output").
        self.seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        self._draw = 0

    @N.on_device
    def forward(self, x, y):
        n_audios, n_channels, Nn = x.shape
        assert n_channels == 1
        S = self.samples_per_draw_batch_size
        assert S % 2 == 0
        num_batches = self.samples_per_draw // S
        lib, st = N.lib(), N.stream
        xd = x.detach().float().contiguous()
        grad = torch.empty_like(xd)
        y_t = torch.as_tensor(y, dtype=torch.long, device=x.device).reshape(-1)
        for i in range(num_batches):
            lead = 1 if i == 0 else 0
            draw = self._draw
            self._draw += 1
            eval_input = torch.empty((n_audios * (S + lead), 1, Nn), device=x.device, dtype=torch.float32)
            N.check(lib.ap_nes_perturb(N.ptr(xd), N.ptr(eval_input), float(self.sigma), self.seed, draw, n_audios, S, lead, Nn,
                                       st()), "ap_nes_perturb")
            eval_y = y_t.repeat_interleave(S + lead)                         # :25-31
            scores, loss, _, decisions = self.EOT_wrapper(eval_input, eval_y)
            EOT_num_batches = int(self.EOT_wrapper.EOT_size // self.EOT_wrapper.EOT_batch_size)
            loss = loss.detach().float() / EOT_num_batches                   # :35-36
            scores = scores.detach().float() / EOT_num_batches
            loss = loss.view(n_audios, -1)
            scores = scores.view(n_audios, -1, scores.shape[1])
            if i == 0:
                adver_loss = loss[..., 0]
                loss = loss[..., 1:]
                adver_score = scores[:, 0, :]
                mean_loss = loss.mean(1)
                predicts = resolve_prediction(decisions).reshape(n_audios, -1)
                predict = predicts[:, 0]
            else:
                mean_loss = mean_loss + loss.mean(1)
            lc = loss.contiguous()
            N.check(lib.ap_nes_grad(N.ptr(lc), N.ptr(grad), self.seed, draw, n_audios, S, Nn, 0 if i == 0 else 1, st()),
                    "ap_nes_grad")
        grad = grad / self.sigma / num_batches                               # :53
        mean_loss = mean_loss / num_batches
        return mean_loss, grad, adver_loss, adver_score, predict
