"""``RevVPSDE`` / ``RevDiffWave`` with the reference's call surface
(diffusion_models/diffwave_sde.py:36-219).  The reference hands the reverse VP-SDE to
``torchsde.sdeint_adjoint(method='euler', dt=1/T)``; here the same fixed-step Euler-Maruyama
scheme runs as one native sampling chain (ap_purify_chain): per k = t*-1..0

    x <- x (1 + b_k/2) - b_k eps(x,k)/sqrt(1-ac_k) + sqrt(b_k) sqrt((1-ac_{k-1})/(1-ac_k)) z   (0 at k=0)

(SURVEY.md Appendix A.3).  torchsde is not needed.  Deviations, documented: exactly t* epsilon
evaluations (the reference's fp32 time stepping adds a spurious ~6e-8-long extra step for t* >= 10).
``rand_t`` follows the reference (diffwave_sde.py:186-194): the q-sample level is ``t + randint(-t_delta, t_delta)``
while the reverse integration always runs ``args.t`` Euler steps (``t0 = 1 - args.t/T``).  ``use_bm`` is accepted and
has no effect: the reference only uses it to hand torchsde a pre-built ``BrownianInterval`` (:198-200), which changes how
the increments are drawn, not their law; here the increments come from the noise source of the wrapped ``DiffWave``.
With ``x.requires_grad`` the chain is an autograd node (``_grad.py``): gradients reach the audio, as through the
reference's ``sdeint_adjoint``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _native as N
from .diffwave_ddpm import DiffWave, create_diffwave_model


class RevVPSDE(torch.nn.Module):
    def __init__(self, model: DiffWave, score_type='ddpm', beta_min=0.02, beta_max=4, N=200,
                 audio_shape=(1, 16000), model_kwargs=None):
        super().__init__()
        self.model = model
        self.score_type = score_type
        self.model_kwargs = model_kwargs
        self.audio_shape = audio_shape
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N
        # host tables exactly as diffwave_sde.py:56-60
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)
        self.noise_type = "diagonal"
        self.sde_type = "ito"

    def _scale_timesteps(self, t):
        assert torch.all(t <= 1) and torch.all(t >= 0), f't has to be in [0, 1], but get {t} with shape {t.shape}'
        return (t.float() * self.N).long()                                   # :69-71

    def _disc(self, t):
        return int(self._scale_timesteps(t.reshape(-1)[:1].cpu())[0]) - 1    # :76

    def vpsde_fn(self, t, x):
        """Forward VP-SDE drift / diffusion at reference time t (diffwave_sde.py:73-80)."""
        k = self._disc(t)
        beta_t = float(self.discrete_betas[k]) * self.N
        return -0.5 * beta_t * x, torch.full((x.shape[0],), math.sqrt(beta_t), device=x.device)

    def rvpsde_fn(self, t, x, return_type='drift'):
        """Reverse-SDE drift or diffusion at reference time t (diffwave_sde.py:82-116): f and g without the time flip."""
        return -self.f(1 - t, x) if return_type == 'drift' else self.g(1 - t, x)[:, 0]

    def f(self, t, x):
        """Drift of the reverse SDE in torchsde time (diffwave_sde.py:118-125) — one native eps evaluation."""
        if self.score_type != 'guided_diffusion':
            raise NotImplementedError(f'Unknown score type in RevVPSDE: {self.score_type}!')     # :101-102
        assert x.ndim == 2 and np.prod(self.audio_shape) == x.shape[1], x.shape
        k = self._disc(1 - t)
        beta_t = float(self.discrete_betas[k]) * self.N
        eps = self.model.compute_eps_t(x.view(-1, *self.audio_shape), k).view(x.shape[0], -1)
        score = -eps / float(self.sqrt_1m_alphas_cumprod[k])
        drift = -0.5 * beta_t * x - beta_t * score
        return -drift

    def g(self, t, x):
        k = self._disc(1 - t)                                                # :107-115
        beta_t = float(self.discrete_betas[k]) * self.N
        scale = (math.sqrt(1 - float(self.alphas_cumprod[k - 1])) / math.sqrt(1 - float(self.alphas_cumprod[k]))
                 if k > 0 else 0.0)
        return torch.full_like(x, scale * math.sqrt(beta_t))

    def euler_steps(self, t_star: int):
        """Coefficient table of the t* Euler-Maruyama links (h = 1/N)."""
        b = self.discrete_betas.double()
        ac = self.alphas_cumprod.double()
        steps = []
        for i, k in enumerate(range(t_star - 1, -1, -1)):
            beta, ack = float(b[k]), float(ac[k])
            cs = math.sqrt(beta) * math.sqrt(1.0 - float(ac[k - 1])) / math.sqrt(1.0 - ack) if k > 0 else 0.0
            steps.append((float(k), 1.0 + 0.5 * beta, -beta / math.sqrt(1.0 - ack), cs, 1 + i))
        return steps


class RevDiffWave(torch.nn.Module):
    def __init__(self, args, device=None):
        super().__init__()
        self.args = args
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        audio_shape = (1, 16000)
        print(f'model_config: {args.ddpm_config}')
        model = create_diffwave_model(model_path=args.ddpm_path, config_path=args.ddpm_config,
                                      reverse_timestep=args.t, device=self.device)
        model.eval().to(self.device)
        self.T = 200
        self.model = model
        self.rev_vpsde = RevVPSDE(model=model, score_type=args.score_type,
                                  beta_min=0.0001 * self.T, beta_max=0.02 * self.T,
                                  N=self.T, audio_shape=audio_shape, model_kwargs=None)
        self.betas = self.rev_vpsde.discrete_betas.float()
        print(f't: {args.t}, rand_t: {args.rand_t}, t_delta: {args.t_delta}')
        print(f'use_bm: {args.use_bm}')

    @classmethod
    def from_model(cls, model: DiffWave, args):
        """Build around an already constructed DiffWave (tests / synthetic weights; no checkpoint file)."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self.args = args
        self.device = next(model.model.parameters()).device
        self.T = 200
        self.model = model
        self.rev_vpsde = RevVPSDE(model=model, score_type=args.score_type, beta_min=0.0001 * self.T,
                                  beta_max=0.02 * self.T, N=self.T, audio_shape=(1, 16000))
        self.betas = self.rev_vpsde.discrete_betas.float()
        return self

    @N.on_device
    def audio_editing_sample(self, audio):
        assert isinstance(audio, torch.Tensor)
        assert audio.ndim == 3, audio.ndim
        if self.rev_vpsde.score_type != 'guided_diffusion':
            raise NotImplementedError(f'Unknown score type in RevVPSDE: {self.rev_vpsde.score_type}!')
        if torch.is_grad_enabled() and audio.requires_grad:
            return self._differentiable_sample(audio)
        x0 = self.model._prep(audio)
        xs = []
        with torch.no_grad():
            for it in range(self.args.sample_step):
                total_noise_levels = self.args.t
                if self.args.rand_t:
                    total_noise_levels = self.args.t + np.random.randint(-self.args.t_delta, self.args.t_delta)
                    print(f'total_noise_levels: {total_noise_levels}')
                a = float(self.rev_vpsde.alphas_cumprod[total_noise_levels - 1].double())     # :189-190
                steps = self.rev_vpsde.euler_steps(self.args.t)       # :192-193: args.t steps whatever level was drawn
                x0 = self.model._chain(x0, steps, math.sqrt(a), math.sqrt(1.0 - a), n_draws=self.args.t + 1)
                xs.append(x0)
        return torch.cat(xs, dim=0)

    def _differentiable_sample(self, audio):
        """The same Euler chain as an autograd node: the white-box attack's ``loss.backward()`` reaches the audio
        (white_box_attack.py:392,437-439; the reference gets there through sdeint_adjoint, diffwave_sde.py:200-204).
        States are check-pointed per step and each step's eps-evaluation is recomputed in the backward pass."""
        if audio.dim() != 3 or audio.shape[1] != 1:
            raise ValueError(f"expected audio of shape [B,1,L], got {tuple(audio.shape)}")
        dw = self.model
        x = audio.to(next(dw.model.parameters()).device).float()
        xs = []
        for it in range(self.args.sample_step):
            total_noise_levels = self.args.t
            if self.args.rand_t:
                total_noise_levels = self.args.t + np.random.randint(-self.args.t_delta, self.args.t_delta)
                print(f'total_noise_levels: {total_noise_levels}')
            a = float(self.rev_vpsde.alphas_cumprod[total_noise_levels - 1].double())
            steps = self.rev_vpsde.euler_steps(self.args.t)
            x = dw._chain_grad(x, steps, math.sqrt(a), math.sqrt(1.0 - a), n_draws=self.args.t + 1)
            xs.append(x)
        return torch.cat(xs, dim=0)

    def forward(self, x):
        return self.audio_editing_sample(x)
