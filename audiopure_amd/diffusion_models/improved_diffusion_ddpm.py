"""DDPM purification of mel spectrograms with the Improved-Diffusion UNet (BASELINE configs[4] "UNet DDPM n=5").

The reference's own ``diffusion_models/improved_diffusion_ddpm.py::ImprovedDiffusion`` is broken (``_reverse`` ignores
its input and returns None, :53-59), so this class follows the arithmetic of the sampler that
file wraps: ``GaussianDiffusion.q_sample`` (gaussian_diffusion.py:188-206) then ``p_sample`` for i = t*-1 .. 0
(:356-387) with epsilon prediction, ``clip_denoised`` and the FIXED_LARGE variance of script_util.py:15-35,231-269
(T = 200, linear betas 1e-4..0.02 in float64, identity ``SpacedDiffusion`` map).  mel-dB in, mel-dB out
(standardised like RevImprovedDiffusion, improved_diffusion_sde.py:182,207).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _native as N
from .improved_diffusion_sde import MEL_LOWER_BOUND, MEL_UPPER_BOUND
from .improved_diffusion_unet import UNetModel


class GaussianTables:
    """float64 tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:119-163)."""

    def __init__(self, T=200, beta_start=0.0001, beta_end=0.02):
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)            # :27-35
        ac = np.cumprod(1.0 - betas)
        ac_prev = np.append(1.0, ac[:-1])
        self.T = T
        self.sqrt_ac, self.sqrt_1m_ac = np.sqrt(ac), np.sqrt(1.0 - ac)
        self.sqrt_recip_ac, self.sqrt_recipm1_ac = np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1)
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)
        self.log_var_large = np.log(np.append(post_var[1], betas[1:]))             # FIXED_LARGE, :280-284


class ImprovedDiffusionDDPM(torch.nn.Module):
    def __init__(self, model: UNetModel, reverse_timestep: int = 5, T: int = 200, clip_denoised: bool = True):
        super().__init__()
        self.model, self.reverse_timestep, self.clip_denoised = model, reverse_timestep, clip_denoised
        self.tables = GaussianTables(T)
        self._noise = None

    def set_noise_source(self, src=None):
        self._noise = list(src) if src is not None else None

    def _z(self, like):
        if self._noise is not None:
            return self._noise.pop(0).to(like.device).float().reshape(like.shape).contiguous()
        return torch.randn_like(like)

    @N.on_device
    def forward(self, img):
        assert isinstance(img, torch.Tensor) and img.ndim == 4
        if torch.is_grad_enabled() and img.requires_grad:
            raise NotImplementedError("audiopure_amd ImprovedDiffusionDDPM: forward-only HIP path")
        lib, tb, f32 = N.lib(), self.tables, (lambda v: float(np.float32(v)))
        dev = next(self.model.parameters()).device
        img = img.detach().to(dev).float().contiguous()
        B, n, ts = img.shape[0], img.numel(), self.reverse_timestep
        if not (1 <= ts <= tb.T):
            raise ValueError(f"reverse_timestep {ts} outside [1, {tb.T}]")
        k = 2.0 / (MEL_UPPER_BOUND - MEL_LOWER_BOUND)
        x0 = torch.empty_like(img)
        N.check(lib.ap_axpbyc(N.ptr(img), None, N.ptr(x0), k, 0.0, -MEL_LOWER_BOUND * k - 1.0, n, N.stream()))
        with torch.no_grad():
            x = torch.empty_like(x0)                                                 # q_sample at t = t*-1  (:200-205)
            z0 = self._z(x0)
            N.check(lib.ap_axpbyc(N.ptr(x0), N.ptr(z0), N.ptr(x), f32(tb.sqrt_ac[ts - 1]), f32(tb.sqrt_1m_ac[ts - 1]), 0.0, n,
                                  N.stream()))
            for i in range(ts - 1, -1, -1):                                          # p_sample  (:356-387)
                eps = self.model(x, torch.full((B,), float(i), device=dev))
                z = self._z(x) if i > 0 else None
                out = torch.empty_like(x)
                N.check(lib.ap_psample_update(N.ptr(x), N.ptr(eps), N.ptr(z), N.ptr(out), f32(tb.sqrt_recip_ac[i]),
                                              f32(tb.sqrt_recipm1_ac[i]), f32(tb.coef1[i]), f32(tb.coef2[i]),
                                              f32(np.exp(0.5 * np.float32(tb.log_var_large[i]))), int(self.clip_denoised), n,
                                              N.stream()), "ap_psample_update")
                x = out
            res = torch.empty_like(x)
            N.check(lib.ap_axpbyc(N.ptr(x), None, N.ptr(res), 1.0 / k, 0.0, 1.0 / k + MEL_LOWER_BOUND, n, N.stream()))
        return res
