"""Host logic of the native RobustCertificate (reference: robustness_eval/certified_robust.py)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd.robustness_eval.certified_robust import RobustCertificate  # noqa: E402


class _Den:
    def __init__(self):
        beta = torch.linspace(1e-4, 0.02, 200)
        self.diffusion_hyperparams = {"T": 200, "Alpha_bar": torch.cumprod(1 - beta, 0)}


def test_clopper_pearson_lower_bound_and_radius_logic():
    rc = RobustCertificate(classifier=torch.nn.Linear(1, 1), denoiser=_Den())
    # closed forms of the one-sided Clopper-Pearson bound: k = n -> alpha^(1/n); k = 0 -> 0
    assert abs(rc.lower_conf_bound(100, 100, alpha=0.001) - 0.001 ** (1 / 100)) < 1e-12
    assert rc.lower_conf_bound(0, 100) == 0.0
    lo = rc.lower_conf_bound(990, 1000, alpha=0.001)
    assert 0.97 < lo < 0.99 and lo < 990 / 1000
    # monotone in k, and it is the beta quantile the reference gets from statsmodels (method='beta', alpha doubled)
    assert rc.lower_conf_bound(600, 1000) < rc.lower_conf_bound(700, 1000)
    y_pred, y, r_c = torch.tensor([1, 2, -1, 3]), torch.tensor([1, 2, 2, 0]), torch.tensor([0.5, 0.1, 0.0, 0.9])
    assert rc.certified_robust_correct(y_pred, y, r_c, r=0.25) == 1


def test_t_star_follows_the_reference_rule():
    rc = RobustCertificate(classifier=torch.nn.Linear(1, 1), denoiser=_Den())
    ab = rc.denoiser.diffusion_hyperparams["Alpha_bar"]
    for sigma in (0.1, 0.25, 0.5, 1.0):
        target = 1 / (1 + sigma ** 2)
        t = rc.compute_t_star(target)
        assert 1 <= t <= 200
        assert t - 1 == int(torch.argmin(torch.abs(ab - target)))                       # certified_robust.py:104
