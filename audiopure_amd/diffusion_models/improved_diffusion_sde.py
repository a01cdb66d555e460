"""``RevImprovedDiffusion`` — the spectrogram purifier of the ``DiffSpec`` defense (``defense_type='spec'``,
acoustic_system.py:45-46) with the reference's call surface (diffusion_models/improved_diffusion_sde.py:140-226).
mel-dB [B,1,32,32] -> standardise -> q-sample -> continuous-beta reverse VP-SDE, fixed-step Euler-Maruyama with
torchsde's default dt = 1e-3 restated (one native UNet evaluation per step) -> de-standardise.
Per step at reference time tau:  x <- x (1 + beta h / 2) - beta h eps / sqrt(1 - abar(tau)) + sqrt(beta h) z,
beta = 0.1 + 19.9 tau, abar(tau) = exp(-9.95 tau^2 - 0.1 tau), model timestep floor(1000 tau)  (:74-75,:82-116).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _native as N
from .improved_diffusion_unet import UNetModel, create_model, model_and_diffusion_defaults

MEL_UPPER_BOUND, MEL_LOWER_BOUND = 38.22, -100.0     # sc09_spectrogram_dataset.py:62-63


def sde_step_table(t_star: int, dt: float = 1e-3, beta_min=0.1, beta_max=20.0, N_=1000):
    """[(model_timestep, ca, cb, cs)] of torchsde's fixed-step Euler grid on ts = linspace(1 - t/1000, 1 - 1e-5, 2)
    (:194-196).  The grid, tau = 1 - s and the model timestep floor(tau * 1000) follow the reference's float32 time
    arithmetic, which decides the integer timestep each step lands on; the coefficients are then formed in double."""
    t0, t1, d = np.float32(1 - t_star * 1. / 1000), np.float32(1 - 1e-5), np.float32(dt)
    steps, s = [], t0
    while s < t1:
        nxt = min(np.float32(s + d), t1)
        h, tau = float(np.float32(nxt - s)), np.float32(1) - s
        beta = float(np.float32(beta_min) + tau * np.float32(beta_max - beta_min))            # :86 (float32 tensor math)
        abar = math.exp(-0.5 * (beta_max - beta_min) * float(tau) ** 2 - beta_min * float(tau))   # :74
        disc = int(tau * np.float32(N_))                                                          # :82-83
        steps.append((float(disc), 1.0 + 0.5 * beta * h, -beta * h / math.sqrt(1.0 - abar), math.sqrt(beta * h)))
        s = nxt
    return steps


class RevVPSDE(torch.nn.Module):
    """The reverse VP-SDE object of the spectrogram purifier with the reference's constructor and ``f`` / ``g`` /
    ``vpsde_fn`` / ``rvpsde_fn`` (improved_diffusion_sde.py:48-137); the score is one native UNet evaluation.
    ``RevImprovedDiffusion`` itself integrates the same drift / diffusion through its coefficient table
    (``sde_step_table``) instead of calling back into Python per step."""

    def __init__(self, model, score_type='guided_diffusion', beta_min=0.1, beta_max=20, N=1000, img_shape=(1, 32, 32),
                 model_kwargs=None):
        super().__init__()
        self.model = model
        self.score_type = score_type
        self.model_kwargs = model_kwargs
        self.img_shape = img_shape
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)
        self.alphas_cumprod_cont = lambda t: torch.exp(-0.5 * (beta_max - beta_min) * t ** 2 - beta_min * t)
        self.sqrt_1m_alphas_cumprod_neg_recip_cont = lambda t: -1. / torch.sqrt(1. - self.alphas_cumprod_cont(t))
        self.noise_type = "diagonal"
        self.sde_type = "ito"

    def _scale_timesteps(self, t):
        assert torch.all(t <= 1) and torch.all(t >= 0), f't has to be in [0, 1], but get {t} with shape {t.shape}'
        return (t.float() * self.N).long()                                   # :80-82

    def vpsde_fn(self, t, x):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)               # :84-88
        return -0.5 * beta_t[:, None] * x, torch.sqrt(beta_t)

    def rvpsde_fn(self, t, x, return_type='drift'):
        drift, diffusion = self.vpsde_fn(t, x)                               # :90-116
        if return_type != 'drift':
            return diffusion
        assert x.ndim == 2 and np.prod(self.img_shape) == x.shape[1], x.shape
        if self.score_type != 'guided_diffusion':
            raise NotImplementedError(f'Unknown score type in RevVPSDE: {self.score_type}!')
        if self.model_kwargs:
            raise NotImplementedError("audiopure_amd RevVPSDE: model_kwargs (class conditioning) are not built")
        eps = self.model(x.view(-1, *self.img_shape), self._scale_timesteps(t)).view(x.shape[0], -1)
        score = self.sqrt_1m_alphas_cumprod_neg_recip_cont(t.float())[:, None].to(x.device) * eps
        return drift - diffusion[:, None] ** 2 * score

    def f(self, t, x):
        t = t.expand(x.shape[0]).to(x.device)                               # :118-126
        drift = self.rvpsde_fn(1 - t, x, return_type='drift')
        assert drift.shape == x.shape
        return -drift

    def g(self, t, x):
        t = t.expand(x.shape[0]).to(x.device)                               # :128-136
        diffusion = self.rvpsde_fn(1 - t, x, return_type='diffusion')
        assert diffusion.shape == (x.shape[0],)
        return diffusion[:, None].expand(x.shape)


class RevImprovedDiffusion(torch.nn.Module):
    def __init__(self, args, config=None, device=None):
        super().__init__()
        self.args = args
        self.config = config
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        model = create_model(**model_and_diffusion_defaults())
        model.load_state_dict(torch.load(args.ddpm_path, map_location="cpu"))
        self.model = model.eval().to(self.device)
        self.rev_vpsde = RevVPSDE(model=self.model, score_type=getattr(args, "score_type", "guided_diffusion"))   # :157-158
        print(f't: {args.t}, rand_t: {args.rand_t}, t_delta: {args.t_delta}')
        print(f'use_bm: {args.use_bm}')
        self._noise = None

    @classmethod
    def from_model(cls, model: UNetModel, args):
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.args, self.config, self.model = args, None, model
        self.rev_vpsde = RevVPSDE(model=model, score_type=getattr(args, "score_type", "guided_diffusion"))
        self.device = next(model.parameters()).device
        self._noise = None
        return self

    def set_noise_source(self, src=None):
        """None: torch.randn on the device | list of tensors consumed in draw order (tests)."""
        self._noise = list(src) if src is not None else None

    def _z(self, like):
        if self._noise is not None:
            return self._noise.pop(0).to(like.device).float().reshape(like.shape)
        return torch.randn_like(like)

    @N.on_device
    def image_editing_sample(self, img):
        assert isinstance(img, torch.Tensor)
        assert img.ndim == 4, img.ndim
        if torch.is_grad_enabled() and img.requires_grad:
            return self._differentiable_sample(img)
        lib = N.lib()
        img = img.detach().to(self.device).float().contiguous()
        B, n = img.shape[0], img.numel()
        st = N.stream
        k = 2.0 / (MEL_UPPER_BOUND - MEL_LOWER_BOUND)
        x0 = torch.empty_like(img)                      # melspec_standardize: 2 (x + 100) / 138.22 - 1  (:182)
        N.check(lib.ap_axpbyc(N.ptr(img), None, N.ptr(x0), k, 0.0, -MEL_LOWER_BOUND * k - 1.0, n, st()))
        xs = []
        with torch.no_grad():
            for it in range(self.args.sample_step):
                total = self.args.t
                if self.args.rand_t:
                    total = self.args.t + np.random.randint(-self.args.t_delta, self.args.t_delta)
                    print(f'total_noise_levels: {total}')
                betas = torch.linspace(0.1 / 1000, 20.0 / 1000, 1000)
                a = float((1 - betas).cumprod(dim=0)[total - 1].double())                           # :188
                x = torch.empty_like(x0)
                N.check(lib.ap_axpbyc(N.ptr(x0), N.ptr(self._z(x0)), N.ptr(x), math.sqrt(a), math.sqrt(1.0 - a), 0.0, n, st()))
                for (disc, ca, cb, cs) in sde_step_table(self.args.t):                             # :194-204
                    eps = self.model(x, torch.full((B,), disc, device=self.device))
                    t1 = torch.empty_like(x)
                    N.check(lib.ap_axpbyc(N.ptr(x), N.ptr(eps), N.ptr(t1), ca, cb, 0.0, n, st()))
                    x2 = torch.empty_like(x)
                    N.check(lib.ap_axpbyc(N.ptr(t1), N.ptr(self._z(x)), N.ptr(x2), 1.0, cs, 0.0, n, st()))
                    x = x2
                out = torch.empty_like(x)               # melspec_inv_standardize  (:207)
                N.check(lib.ap_axpbyc(N.ptr(x), None, N.ptr(out), 1.0 / k, 0.0, 1.0 / k + MEL_LOWER_BOUND, n, st()))
                xs.append(out)
                x0 = out                                # the reference feeds the de-standardised result back (:206-209)
        return torch.cat(xs, dim=0)

    def _differentiable_sample(self, img):
        """The same sampler as an autograd graph (white-box attack through the DiffSpec defense): the chain of Euler
        links is one node whose backward recomputes each link's UNet evaluation and applies its J^T on the HIP path
        (``_grad._ChainFn`` + ``UNetModel.input_grad``); the affine (de)standardisation is ordinary autograd."""
        from ._grad import _ChainFn
        from .improved_diffusion_unet import UNetEpsGrad
        if not hasattr(self.model, "_eps_grad"):
            self.model._eps_grad = UNetEpsGrad(self.model)
        k = 2.0 / (MEL_UPPER_BOUND - MEL_LOWER_BOUND)
        x0 = img.to(self.device).float() * k + (-MEL_LOWER_BOUND * k - 1.0)
        xs = []
        for it in range(self.args.sample_step):
            total = self.args.t
            if self.args.rand_t:
                total = self.args.t + np.random.randint(-self.args.t_delta, self.args.t_delta)
                print(f'total_noise_levels: {total}')
            betas = torch.linspace(0.1 / 1000, 20.0 / 1000, 1000)
            a = float((1 - betas).cumprod(dim=0)[total - 1].double())
            table = sde_step_table(self.args.t)
            zs = [self._z(x0.detach())]
            steps = []
            for i, (disc, ca, cb, cs) in enumerate(table):
                zs.append(self._z(x0.detach()))
                steps.append((disc, ca, cb, cs, i + 1))
            x = _ChainFn.apply(x0, self.model._eps_grad, steps, math.sqrt(a), math.sqrt(1.0 - a), zs)
            out = x * (1.0 / k) + (1.0 / k + MEL_LOWER_BOUND)
            xs.append(out)
            x0 = out
        return torch.cat(xs, dim=0)

    def forward(self, x):
        return self.image_editing_sample(x)
