"""``DiffWave`` / ``create_diffwave_model`` with the reference's call surface
(diffusion_models/diffwave_ddpm.py:15-226,395-411); the sampling chains run in the HIP library.

Noise.  The reference draws ``torch.normal`` on the global CPU RNG and copies to the GPU
(diffwave_ddpm.py:66,100).  Here the default draws ``torch.randn`` on the device from torch's global
device generator (same "seed torch, get reproducible output" contract, no H2D copy); tests inject
explicit tensors with ``set_noise_source([...])``; throughput runs use the in-kernel counter-based
Philox stream with ``set_noise_source(("philox", seed, utt_offset))``.
"""
from __future__ import annotations

import ctypes as C
import json
import math
from typing import Union

import numpy as np
import torch

from .. import _native as N
from .DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
from .DiffWave_Unconditional.util import calc_diffusion_hyperparams


class DiffWave(torch.nn.Module):

    def __init__(self, model: WaveNet_Speech_Commands, diffusion_hyperparams: dict,
                 reverse_timestep: int = 200, grad_enable=True):
        super().__init__()
        if not isinstance(model, WaveNet_Speech_Commands):
            raise TypeError("audiopure_amd.DiffWave needs the native WaveNet_Speech_Commands "
                            "(audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet); "
                            f"got {type(model).__name__} and there is no generic fallback")
        self.model = model
        self.diffusion_hyperparams = diffusion_hyperparams
        self.reverse_timestep = reverse_timestep
        self.freeze = False
        self.grad_enable = grad_enable
        self._noise = None
        self._graphs = {}

    # ---- small batches: the chain as a replayed HIP graph ------------------------------------------------------
    # The callers' loops run B = 1 ... 10 clips hundreds of times (adaptive_attack_eval.py:47,156-160): a chain is then
    # 36 n + 5 n dependent launches of tens of microseconds each, and the gaps between them are a tenth of the step.
    # The launch functions neither allocate nor synchronise (include/audiopure.h), so the whole ap_purify_chain call is
    # captured once per (shape, coefficients, noise key) into a HIP graph on static buffers and replayed: same kernels,
    # same arguments, same results bit for bit (tests/test_gpu_parity.py).  Off: ``dw.graph_replay = False``.
    graph_replay = True
    GRAPH_MAX_SAMPLES = 16 * 16000                   # clips x samples per call up to which a chain is replayed
    GRAPH_CACHE = 8

    def __getstate__(self):                          # captured graphs are rebuilt on demand, never pickled
        d = dict(self.__dict__)
        d["_graphs"] = {}
        return d

    def _chain_replayed(self, eng, x, steps, qa, qs, n_draws):
        """-> purified batch through a cached graph of ONE ap_purify_chain call, or None where the call is not replayable
        (explicit noise tensors, profiling hooks on, a capture already in progress, a batch that needs several calls)."""
        B, _, L = x.shape
        src = self._noise
        if (not self.graph_replay or not steps or B < 1 or B * L > self.GRAPH_MAX_SAMPLES or B > eng.max_chunk or isinstance(src, list)
                or torch.cuda.is_current_stream_capturing() or eng.lib.ap_profile_is_enabled(eng.ctx)):
            return None
        dev = x.device
        ws = eng.workspace(B, L, dev)
        seed, off = (src[1], src[2]) if isinstance(src, tuple) else (0, 0)
        # hygiene: a graph holds raw pointers into the engine's weights and into the workspace it was captured on.  Entries of another
        # engine (set_precision / a rebuilt engine: new serial) or of a workspace the engine has since replaced (a larger batch came
        # by) are dropped here, so the cache never replays into freed memory and never pins more than the ONE current workspace.
        stale = [k for k, e in self._graphs.items() if k[7] != eng.serial or e[5] is not ws]
        for k in stale:
            del self._graphs[k]
        key = (dev.index, B, L, tuple(steps), float(qa), float(qs), n_draws, eng.serial, int(eng.skip_group or 0), ws.data_ptr(),
               isinstance(src, tuple), seed, off, int(eng.lib.ap_ctx_get_f32_form(eng.ctx)))
        ent = self._graphs.get(key)
        self._graph_misses = 0 if ent is not None else getattr(self, "_graph_misses", 0) + 1
        if self._graph_misses > 2 * self.GRAPH_CACHE:    # a caller whose every call differs (rand_t over a wide range): capturing
            self.graph_replay = False                    # costs two eager runs per miss -- stay eager for this object
            return None
        if ent is None:
            x_in, out = torch.empty_like(x), torch.empty_like(x)
            z = torch.empty((n_draws, B, L), device=dev, dtype=torch.float32) if (n_draws and src is None) else None
            arr = (N.ApStep * len(steps))(*[N.ApStep(*s) for s in steps])

            def call():
                N.check(eng.lib.ap_purify_chain(eng.ctx, N.ptr(x_in), float(qa), float(qs), arr, len(steps), N.ptr(z), seed, off,
                                                N.ptr(out), B, L, ws.data_ptr(), ws.numel(), N.stream()), "ap_purify_chain")

            try:
                x_in.copy_(x)
                if z is not None:
                    z.zero_()
                cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    call()                           # warm run outside the capture (first-use allocations inside torch)
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # thread_local: another host thread's HIP calls (a DataLoader's pin-memory thread) do not invalidate this capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    call()
            except Exception as e:                   # a failed capture must never fail a plain dw(x): this object stays eager
                import warnings
                self.graph_replay = False
                self._graphs.clear()
                warnings.warn(f"DiffWave: HIP-graph capture of the chain failed ({type(e).__name__}: {e}); this object runs its "
                              "chains eagerly from now on (same kernels, same results)", stacklevel=3)
                return None
            if len(self._graphs) >= self.GRAPH_CACHE:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (g, x_in, out, z, arr, ws)
        g, x_in, out, z = ent[:4]
        x_in.copy_(x)
        if z is not None:
            z.normal_()                              # the device generator's stream, as torch.randn in the eager path
        g.replay()
        return out.clone()

    # ---- noise plumbing ---------------------------------------------------------------------
    def set_noise_source(self, src=None):
        """None: torch.randn on the device | list of [B,1,L] tensors consumed in draw order |
        ("philox", seed, utt_offset): in-kernel counter-based stream."""
        if isinstance(src, (list, tuple)) and len(src) and isinstance(src[0], str):
            assert src[0] == "philox"
            src = ("philox", int(src[1]), int(src[2]) if len(src) > 2 else 0)
        elif isinstance(src, (list, tuple)):
            src = list(src)
        self._noise = src

    def _draws(self, n: int, like: torch.Tensor):
        """-> (z_all [n,B,L] tensor or None, seed, utt_offset)"""
        src = self._noise
        if n == 0:
            return None, 0, 0
        if isinstance(src, tuple):
            return None, src[1], src[2]
        B, _, L = like.shape
        if isinstance(src, list):
            if len(src) < n:
                raise ValueError(f"noise source exhausted: need {n} draws, have {len(src)}")
            zs = [src.pop(0).to(like.device).float().reshape(B, L) for _ in range(n)]
            return torch.stack(zs).contiguous(), 0, 0
        return torch.randn((n, B, L), device=like.device, dtype=torch.float32), 0, 0

    def _prep(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        assert x.ndim == 3                                                  # diffwave_ddpm.py:62,89
        dev = next(self.model.parameters()).device
        self.model._check_input(x)
        if torch.is_grad_enabled() and x.requires_grad:
            return x.to(dev).float()         # stays in the graph: _chain makes the sampling chain an autograd node
        return x.detach().to(dev).float().contiguous()

    def _tables(self):
        dh = self.diffusion_hyperparams
        T, Alpha, Alpha_bar, Sigma = dh["T"], dh["Alpha"], dh["Alpha_bar"], dh["Sigma"]
        assert len(Alpha) == T and len(Alpha_bar) == T and len(Sigma) == T   # :58-61
        eng = self.model.engine()
        eng.set_schedule(dh)
        return eng

    @N.on_device
    def _chain(self, x, steps, qa=1.0, qs=0.0, n_draws=0):
        """Run ap_purify_chain over the batch in workspace-sized chunks.  An input that requires grad (outside
        ``forward``, which is ``no_grad`` like the reference's) goes through the recompute-based autograd node instead."""
        if torch.is_grad_enabled() and x.requires_grad:
            return self._chain_grad(x, steps, qa, qs, n_draws)
        eng = self._tables()
        B, _, L = x.shape
        replayed = self._chain_replayed(eng, x, steps, qa, qs, n_draws)
        if replayed is not None:
            return replayed
        z_all, seed, off = self._draws(n_draws, x)
        out = torch.empty_like(x)
        arr = (N.ApStep * len(steps))(*[N.ApStep(*s) for s in steps]) if steps else None
        for s, e in eng.chunks(B, L, x.device):
            ws = eng.workspace(e - s, L, x.device)
            zc = z_all[:, s:e].contiguous() if z_all is not None else None
            N.check(eng.lib.ap_purify_chain(eng.ctx, N.ptr(x[s:e]), float(qa), float(qs), arr, len(steps), N.ptr(zc),
                                            seed, off + s, N.ptr(out[s:e]), e - s, L, ws.data_ptr(), ws.numel(),
                                            N.stream()), "ap_purify_chain")
        return out

    @N.on_device
    def _chain_grad(self, x, steps, qa=1.0, qs=0.0, n_draws=0):
        """The same chain as an autograd node (gradient with respect to the audio; ``_grad.py``).  The reference's
        ``_reverse`` / ``one_shot_denoise`` / ... are plain differentiable torch code (diffwave_ddpm.py:75-104,174-194) and
        its SDE runner re-integrates backwards (diffwave_sde.py:200-204); here states are check-pointed per link and each
        link's eps-evaluation is recomputed in the backward pass."""
        from ._grad import differentiable_chain
        self._tables()
        B, _, L = x.shape
        z_all, seed, off = self._draws(n_draws, x.detach())
        if z_all is None and n_draws:                            # in-kernel Philox stream: materialise the same draws
            z_all = torch.empty((n_draws, B, L), device=x.device)
            for k in range(n_draws):
                N.check(N.lib().ap_philox_normal(N.ptr(z_all[k]), seed, k, off, B, L, N.stream()), "ap_philox_normal")
        return differentiable_chain(self.model, x, steps, qa, qs, z_all)

    def _ddpm_steps(self, t_star):
        dh = self.diffusion_hyperparams
        A, Ab, Sg = dh["Alpha"].double(), dh["Alpha_bar"].double(), dh["Sigma"]
        steps, draw = [], 1
        for t in range(t_star - 1, -1, -1):                                  # :95
            a, ab = float(A[t]), float(Ab[t])
            ca = 1.0 / math.sqrt(a)                                          # :159
            cb = -(1.0 - a) / math.sqrt(1.0 - ab) / math.sqrt(a)
            cs = float(Sg[t]) if t > 0 else 0.0                              # :99-102,160
            steps.append((float(t), ca, cb, cs, draw if t > 0 else 0))
            draw += 1 if t > 0 else 0
        return steps

    # ---- reference surface ------------------------------------------------------------------
    def forward(self, waveforms: Union[torch.Tensor, np.ndarray]):
        t_star = int(self.reverse_timestep)
        ab = float(self.diffusion_hyperparams["Alpha_bar"][t_star - 1].double())
        with torch.no_grad():                                                # :41-43 (the output is detached there too)
            x0 = self._prep(waveforms)
            return self._chain(x0, self._ddpm_steps(t_star), math.sqrt(ab), math.sqrt(1.0 - ab), n_draws=t_star)

    def _diffusion(self, x_0):
        x0 = self._prep(x_0)
        t_star = int(self.reverse_timestep)
        ab = float(self.diffusion_hyperparams["Alpha_bar"][t_star - 1].double())
        return self._chain(x0, [], math.sqrt(ab), math.sqrt(1.0 - ab), n_draws=1)     # :66-67

    def _reverse(self, x_t):
        x = self._prep(x_t)
        t_star = int(self.reverse_timestep)
        # chain draws are numbered from 1 (0 is the q-sample draw); shift an injected list accordingly
        if isinstance(self._noise, list):
            self._noise.insert(0, torch.zeros_like(x))
            return self._chain(x, self._ddpm_steps(t_star), n_draws=t_star)
        return self._chain(x, self._ddpm_steps(t_star), n_draws=t_star)

    @N.on_device
    def compute_coefficients(self, x_t, t: int):
        """-> (eps_theta, mu_theta, sigma_theta) at timestep t (diffwave_ddpm.py:143-164)."""
        x = self._prep(x_t)
        eng = self._tables()
        dh = self.diffusion_hyperparams
        a, ab = float(dh["Alpha"][t].double()), float(dh["Alpha_bar"][t].double())
        ca, cb = 1.0 / math.sqrt(a), -(1.0 - a) / math.sqrt(1.0 - ab) / math.sqrt(a)
        if torch.is_grad_enabled() and x.requires_grad:          # differentiable in the reference (:143-164)
            return self.model.eps(x, float(t)), self._chain(x, [(float(t), ca, cb, 0.0, 0)]), dh["Sigma"][t]
        eps, mu = torch.empty_like(x), torch.empty_like(x)
        B, _, L = x.shape
        for s, e in eng.chunks(B, L, x.device):
            ws = eng.workspace(e - s, L, x.device)
            N.check(eng.lib.ap_eps_affine(eng.ctx, N.ptr(x[s:e]), float(t), ca, cb, N.ptr(eps[s:e]), N.ptr(mu[s:e]),
                                          e - s, L, ws.data_ptr(), ws.numel(), N.stream()), "ap_eps_affine")
        return eps, mu, dh["Sigma"][t]

    @torch.no_grad()
    def compute_eps_t(self, x_t, t):
        return self.model.eps(self._prep(x_t), float(t))                     # :166-172

    def one_shot_denoise(self, x_t):
        x = self._prep(x_t)
        t = int(self.reverse_timestep) - 1                                   # :176
        ab = float(self.diffusion_hyperparams["Alpha_bar"][t].double())
        return self._chain(x, [(float(t), math.sqrt(1.0 / ab), -math.sqrt(1.0 / ab - 1.0), 0.0, 0)])   # :197-203

    def two_shot_denoise(self, x_t):
        x = self._prep(x_t)
        dh = self.diffusion_hyperparams
        t = int(self.reverse_timestep) - 1
        A, Ab, Bt = dh["Alpha"].double(), dh["Alpha_bar"].double(), dh["Beta"].double()
        mu = math.sqrt(float(Ab[t] / A[0]))                                  # :211
        sigma = math.sqrt(float(1 - Ab[t] - (Ab[t] / A[0]) * Bt[0] ** 2))    # :212
        a0, ab0 = float(A[0]), float(Ab[0])
        return self._chain(x, [(float(t), 1.0 / mu, -sigma / mu, 0.0, 0),                      # :214
                               (0.0, 1.0 / math.sqrt(a0), -(1.0 - a0) / math.sqrt(1.0 - ab0) / math.sqrt(a0), 0.0, 0)])

    def fast_reverse(self, x_t):
        """K = 3 respaced reverse steps (diffwave_ddpm.py:106-141), including the reference's use of the
        variance Beta_tilde_new as a standard deviation (:138)."""
        x = self._prep(x_t)
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        K = 3
        S = torch.round(torch.linspace(1, self.reverse_timestep, K)).int() - 1
        Beta_new, Beta_tilde_new = torch.zeros(size=(K,)), torch.zeros(size=(K,))
        for i in range(K):
            if i > 0:
                Beta_new[i] = 1 - Alpha_bar[S[i]] / Alpha_bar[S[i - 1]]
                Beta_tilde_new[i] = (1 - Alpha_bar[S[i - 1]]) / (1 - Alpha_bar[S[i]]) * Beta_new[i]
            else:
                Beta_new[i] = 1 - Alpha_bar[S[i]]
                Beta_tilde_new[i] = 0
        Alpha_new = 1 - Beta_new
        Alpha_bar_new = torch.cumprod(Alpha_new, dim=0)
        steps = []
        for i, t in enumerate(range(K - 1, -1, -1)):
            a, ab = float(Alpha_new[t].double()), float(Alpha_bar_new[t].double())
            steps.append((float(S[t]), 1.0 / math.sqrt(a), -(1.0 - a) / math.sqrt(1.0 - ab) / math.sqrt(a),
                          float(Beta_tilde_new[t]), i + 1))
        if isinstance(self._noise, list):
            self._noise.insert(0, torch.zeros_like(x))
        return self._chain(x, steps, n_draws=K + 1)

    def _predict_x0_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        Alpha_bar = self.diffusion_hyperparams["Alpha_bar"]
        r1 = (1 / Alpha_bar).sqrt()[t].item()
        r2 = (1 / Alpha_bar - 1).sqrt()[t].item()
        return r1 * x_t - r2 * eps                                           # :197-203

    def _predict_x1_from_eps(self, x_t, t, eps):
        dh = self.diffusion_hyperparams
        Alpha, Alpha_bar, Beta = dh["Alpha"], dh["Alpha_bar"], dh["Beta"]
        mu = (Alpha_bar[t] / Alpha[0]).sqrt().item()
        sigma = (1 - Alpha_bar[t] - (Alpha_bar[t] / Alpha[0]) * Beta[0] ** 2).sqrt().item()
        return (x_t - sigma * eps) / mu                                      # :207-216

    def _predict_x0_from_x1(self, x_1):
        _, mu_0, _ = self.compute_coefficients(x_1, 0)                       # :218-224
        return mu_0


def create_diffwave_model(model_path, config_path, reverse_timestep=25, device=None):
    """diffwave_ddpm.py:395-411: JSON config -> schedule -> net -> checkpoint['model_state_dict'] -> DiffWave."""
    with open(config_path) as f:
        cfg = json.loads(f.read())
    wavenet_config = cfg["wavenet_config"]
    diffusion_hyperparams = calc_diffusion_hyperparams(**cfg["diffusion_config"])
    if device is None:
        device = torch.device("cuda")
    net = WaveNet_Speech_Commands(**wavenet_config)
    checkpoint = torch.load(model_path, map_location="cpu")
    net.load_state_dict(checkpoint["model_state_dict"])
    net = net.to(device)
    return DiffWave(model=net, diffusion_hyperparams=diffusion_hyperparams, reverse_timestep=reverse_timestep)


class ReffWave(DiffWave):
    """``ReffWave`` (diffwave_ddpm.py:251-348; not used by any script, kept for the module's surface): ``num_re`` rounds of
    diffuse-to-t* + one-shot denoise.  One round is a single native chain call (the q-sample plus one link)."""

    def __init__(self, model: WaveNet_Speech_Commands, diffusion_hyperparams: dict, reverse_timestep: int = 200,
                 num_re: int = 5):
        super().__init__(model=model, diffusion_hyperparams=diffusion_hyperparams, reverse_timestep=reverse_timestep)
        self.num_re = num_re

    def forward(self, waveforms: Union[torch.Tensor, np.ndarray]):
        x = self._prep(waveforms)
        t = int(self.reverse_timestep) - 1                                   # :280-283,:306-313
        ab = float(self.diffusion_hyperparams["Alpha_bar"][t].double())
        link = [(float(t), math.sqrt(1.0 / ab), -math.sqrt(1.0 / ab - 1.0), 0.0, 0)]
        src = self._noise
        try:
            for i in range(self.num_re):                                     # :276-278
                if isinstance(src, tuple):                                   # counter-based stream: a fresh key per round
                    self._noise = ("philox", src[1] + i, src[2])
                x = self._chain(x, link, math.sqrt(ab), math.sqrt(1.0 - ab), n_draws=1)
        finally:
            if isinstance(src, tuple):
                self._noise = src
        return x

    def diffusion(self, x_0):
        return self._diffusion(x_0)                                          # :284-304

