"""``RobustCertificate`` with the reference's call surface (robustness_eval/certified_robust.py:6-127) and its sampling
loop on the device.

The reference replicates ONE utterance n times on the host side of each batch, draws the Gaussian perturbation with
``torch.normal`` on the CPU generator, copies it over, scales, denoises, classifies, concatenates all n score rows and
counts arg-maxes class by class with ``.item()`` round trips (certified_robust.py:34-65).  Here one batch is a single
chain call -- perturbation, the ``sqrt(alpha_bar*)`` scaling and the one-shot denoise are the chain's q-sample
(``qa = sqrt(ab*)``, ``qs = sigma sqrt(ab*)``, counter-based Philox noise keyed on the sample index, so no noise tensor
exists) followed by one link -- then the classifier, then ``ap_argmax_hist`` accumulates the votes into a device
histogram; the host reads 10 integers per ``smooth_predict``.
"""
from __future__ import annotations

import math

import torch
from scipy.stats import beta, norm

from .. import _native as N
from ..diffusion_models.diffwave_ddpm import DiffWave
from ..lowering import lower_classifier, lower_transform


class RobustCertificate():

    def __init__(self, classifier: torch.nn.Module, transform=None, denoiser=None, one_shot_rev: bool = False,
                 num_classes=10) -> None:
        self.classifier = lower_classifier(classifier)       # un-pickled reference modules / torchaudio Compose -> native
        self.transform = lower_transform(transform)
        self.denoiser = denoiser
        self.num_classes = num_classes
        self.one_shot_rev = one_shot_rev
        self.seed = 0                      # Philox key of the perturbations; sample i of a call uses counter i
        self.native_batch = 512            # samples per device batch when the caller's batch_size is smaller

    @torch.no_grad()
    @N.on_device
    def forward(self, x: torch.Tensor):                                      # certified_robust.py:17-31
        x_in = x
        if self.denoiser is not None:
            x_in = self.denoiser.one_shot_denoise(x_in)
        if self.transform is not None:
            x_in = self.transform(x_in)
        return self.classifier(x_in)

    def _perturb_and_denoise(self, xb, sigma, first_sample):
        """x_in = sqrt(ab*) (x + sigma z) then one_shot_denoise, z ~ Philox(seed, sample index) -- one chain call."""
        den = self.denoiser
        if isinstance(den, DiffWave):
            alpha_bar_star = 1 / (1 + sigma ** 2)                            # :48-51
            t_star = self.compute_t_star(alpha_bar_star)
            den.reverse_timestep = t_star
            t = t_star - 1
            ab = float(den.diffusion_hyperparams["Alpha_bar"][t].double())
            saved = den._noise
            den.set_noise_source(("philox", self.seed, first_sample))
            try:
                return den._chain(xb, [(float(t), math.sqrt(1.0 / ab), -math.sqrt(1.0 / ab - 1.0), 0.0, 0)],
                                  math.sqrt(alpha_bar_star), sigma * math.sqrt(alpha_bar_star), n_draws=1)
            finally:
                den._noise = saved
        # no denoiser (randomised smoothing, certified_robustness_eval.py:90-91) or a foreign one: perturb natively
        out = torch.empty_like(xb)
        N.check(N.lib().ap_affine_noise(N.ptr(xb), N.ptr(out), 1.0, float(sigma), None, self.seed, 0, first_sample,
                                        xb.shape[0], xb.shape[2], N.stream()), "ap_affine_noise")
        if den is not None:
            alpha_bar_star = 1 / (1 + sigma ** 2)
            den.reverse_timestep = self.compute_t_star(alpha_bar_star)
            out = den.one_shot_denoise(alpha_bar_star ** 0.5 * out)
        return out

    @torch.no_grad()
    def smooth_predict(self, x: torch.Tensor, num_sampling: int = 100, sigma=0.25, batch_size=64):
        """(RobustCertificate is no nn.Module and accepts CPU input, so the classifier's device is made current here: streams,
        allocations and every native launch of the body bind to it.)"""
        dev = next(self.classifier.parameters()).device
        if dev.type != "cuda":
            raise N.NativeError("RobustCertificate needs its classifier on a HIP device (.cuda()); there is no CPU path")
        with torch.cuda.device(dev):
            return self._smooth_predict(x, num_sampling, sigma, batch_size)

    def _smooth_predict(self, x: torch.Tensor, num_sampling: int = 100, sigma=0.25, batch_size=64):
        assert (x.shape[0] == 1)                                             # :36
        dev = next(self.classifier.parameters()).device
        x = x.to(dev).float().reshape(1, 1, -1)
        step = max(int(batch_size), int(self.native_batch))
        counts = torch.zeros(self.num_classes + 1, dtype=torch.int64, device=dev)      # last slot: non-finite score rows
        done = 0
        while done < num_sampling:
            nb = min(step, num_sampling - done)
            xb = x.expand(nb, 1, x.shape[2]).contiguous()
            x_in = self._perturb_and_denoise(xb, float(sigma), done)
            if self.transform is not None:
                x_in = self.transform(x_in)
            scores = self.classifier(x_in).float().contiguous()
            assert scores.shape[1] == self.num_classes
            N.check(N.lib().ap_argmax_hist(N.ptr(scores), counts.data_ptr(), nb, self.num_classes, N.stream()),   # (int64 tensor, allocated on the device made current above)
                    "ap_argmax_hist")
            done += nb
        counts = counts.cpu()
        if int(counts[-1]) != 0:
            raise FloatingPointError(f"smooth_predict: {int(counts[-1])} of {num_sampling} score rows are not finite "
                                     "(classifier or denoiser produced NaN/inf); refusing to count them as votes")
        return counts[:-1]

    @torch.no_grad()
    def certify(self, x: torch.Tensor, y: torch.Tensor, sigma: float = 0.25, n_0: int = 100, n: int = 100000,
                alpha: float = 0.001, batch_size: int = 64):                 # :67-97
        y_pred, radius = -torch.ones_like(y), torch.zeros_like(y, dtype=torch.float32)
        for i in range(x.shape[0]):
            x_in = x[i]
            if x_in.dim() == 2:
                x_in = x_in.unsqueeze(0)
            counts_0 = self.smooth_predict(x_in, num_sampling=n_0, sigma=sigma, batch_size=batch_size)
            c_A = counts_0.max(0, keepdim=True)[1].item()
            self.seed += 1                                                   # fresh draws for the estimation sample
            counts = self.smooth_predict(x_in, num_sampling=n, sigma=sigma, batch_size=batch_size)
            self.seed += 1
            pa = self.lower_conf_bound(k=int(counts[c_A]), n=n, alpha=alpha)
            if pa > 0.5:
                y_pred[i] = c_A
                radius[i] = sigma * norm.ppf(pa)
            else:
                y_pred[i] = -1
                radius[i] = 0
        return y_pred, radius

    def compute_t_star(self, alpha_bar_star):                                # :99-107
        Alpha_bar = self.denoiser.diffusion_hyperparams['Alpha_bar']
        return torch.abs(Alpha_bar - alpha_bar_star).min(0, keepdim=True)[1].item() + 1

    def lower_conf_bound(self, k, n, alpha=0.001):
        """statsmodels' proportion_confint(k, n, alpha=2 alpha, method='beta')[0] (:110-114): Clopper-Pearson."""
        return 0.0 if k <= 0 else float(beta.ppf(alpha, k, n - k + 1))

    def certified_robust_correct(self, y_pred: torch.Tensor, y_target: torch.Tensor, r_c: torch.Tensor, r: float = 1.):
        correct = 0
        for i in range(len(y_pred)):
            if y_pred[i] == y_target[i] and r_c[i] >= r:
                correct += 1
        return correct
