"""``WaveNet_Speech_Commands`` with the reference's constructor, ``forward((audio, steps))`` contract
and state-dict key names (diffusion_models/DiffWave_Unconditional/WaveNet.py:138-172), executed by
the gfx950 HIP library through the C-ABI of include/audiopure.h.

The module only *holds* parameters (so ``load_state_dict`` of a reference checkpoint with its
``weight_g`` / ``weight_v`` entries works unchanged); every FLOP of the forward pass runs in
``libaudiopure_hip.so``.  There is no PyTorch fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _native as N
from .util import embedding_frequencies


class _WNConv(nn.Module):
    """Parameter holder named like ``nn.utils.weight_norm(nn.Conv1d)``: bias, weight_g, weight_v."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_g = nn.Parameter(torch.ones(cout, 1, 1))
        w = torch.empty(cout, cin, k)
        nn.init.kaiming_normal_(w)
        self.weight_v = nn.Parameter(w)
        with torch.no_grad():
            self.weight_g.copy_(w.reshape(cout, -1).norm(dim=1).view(cout, 1, 1))


class _PlainConv(nn.Module):
    def __init__(self, cin, cout, k, zero=False):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, k))
        self.bias = nn.Parameter(torch.zeros(cout))


class _ConvBox(nn.Module):
    """Gives the ``.conv`` level of the reference's ``Conv`` / ``ZeroConv1d`` wrappers (WaveNet.py:23-48)."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv


class _Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        l = nn.Linear(cin, cout)
        self.weight = nn.Parameter(l.weight.detach().clone())
        self.bias = nn.Parameter(l.bias.detach().clone())


class _ResidualBlockParams(nn.Module):
    def __init__(self, C, S, E):
        super().__init__()
        self.fc_t = _Linear(E, C)
        self.dilated_conv_layer = _ConvBox(_WNConv(C, 2 * C, 3))
        self.res_conv = _WNConv(C, C, 1)
        self.skip_conv = _WNConv(C, S, 1)


class _ResidualGroupParams(nn.Module):
    def __init__(self, C, S, N, Ein, Emid, Eout):
        super().__init__()
        self.fc_t1 = _Linear(Ein, Emid)
        self.fc_t2 = _Linear(Emid, Eout)
        self.residual_blocks = nn.ModuleList([_ResidualBlockParams(C, S, Eout) for _ in range(N)])


import itertools

_ENGINE_SERIAL = itertools.count(1)


class NativeEngine:
    """Owns the ap_ctx, the packed weights on the device and a cached workspace."""

    def __init__(self, cfg: dict, precision: int = N.AP_PREC_F32):
        self.lib = N.lib()
        self.cfg = N.ApConfig(cfg["res_channels"], cfg["skip_channels"], cfg["num_res_layers"], cfg["dilation_cycle"],
                              cfg["diffusion_step_embed_dim_in"], cfg["diffusion_step_embed_dim_mid"],
                              cfg["diffusion_step_embed_dim_out"], 200, 1e-4, 0.02, precision)
        h = C.c_void_p()
        N.check(self.lib.ap_ctx_create(C.byref(self.cfg), C.byref(h)), "ap_ctx_create")
        self.ctx = h
        self.serial = next(_ENGINE_SERIAL)           # identity for caches that must not confuse a new engine with a collected one
        self.ws = None
        self.max_chunk = 512
        self.loaded_key = None
        # AP_PREC_BF16, deferred-skip form (include/audiopure.h, ap_ctx_set_skip_group): layers per skip GEMM.  A FIXED number
        # (never derived from the batch or from free memory): the grouping sets the fp32 summation order of skip, and a clip's
        # result must not depend on the batch it travels in (tests: a 512-clip run equals the 2-clip runs bit for bit).  Each
        # layer of a group keeps a [B][L][C] bf16 image (8.2 MB per clip-second): 36 layers = 295 MB per clip on top of the
        # 49 MB of activations (B = 512: 176 GB of the 288) -- where that does not fit, `chunks` walks the batch in smaller calls.  0 = the fused block.
        self.skip_group = min(self.SKIP_GROUP, cfg["num_res_layers"])
        self._ds_ok = (precision in (N.AP_PREC_BF16, N.AP_PREC_BF16_STORE) and cfg["res_channels"] == 256 and cfg["skip_channels"] == 256)

    BIG_WS = 4 << 30

    def _drop_ws(self):
        """Let go of the workspace.  One of tens of GB (B = 512 in a bf16 mode: 150-165 GB) goes straight back to the driver: parked in
        torch's caching allocator, the next small allocation would be carved out of it and pin the whole segment, and a later
        workspace of a different size would not fit beside it."""
        big = self.ws is not None and self.ws.numel() >= self.BIG_WS
        self.ws = None
        if big:
            torch.cuda.empty_cache()

    def __del__(self):
        try:
            self._drop_ws()
        except Exception:
            pass
        try:
            if getattr(self, "ctx", None):
                self.lib.ap_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def load(self, blob: torch.Tensor, freq: torch.Tensor):
        n = self.lib.ap_wavenet_blob_elems(C.byref(self.cfg))
        if blob.numel() != n:
            raise N.NativeError(f"weight blob has {blob.numel()} elements, config needs {n}")
        N.check(self.lib.ap_ctx_load_wavenet(self.ctx, N.ptr(blob), blob.numel(), N.ptr(freq), N.stream()),
                "ap_ctx_load_wavenet")

    def set_schedule(self, dh: dict):
        T = int(dh["T"])
        arrs = [N.farr(dh[k].detach().cpu().float().tolist()) for k in ("Beta", "Alpha", "Alpha_bar", "Sigma")]
        N.check(self.lib.ap_ctx_set_schedule(self.ctx, arrs[0], arrs[1], arrs[2], arrs[3], T), "ap_ctx_set_schedule")

    def set_sde_schedule(self, betas: torch.Tensor, ac: torch.Tensor):
        a, b = N.farr(betas.detach().cpu().float().tolist()), N.farr(ac.detach().cpu().float().tolist())
        N.check(self.lib.ap_ctx_set_sde_schedule(self.ctx, a, b, len(a)), "ap_ctx_set_sde_schedule")

    SKIP_GROUP = 36                                  # one skip GEMM per evaluation of the shipped net: skip is written once, never re-read
    WS_FRACTION = 0.7                                # of the device memory that is free (the current workspace counted as free)

    def _sync_group(self):
        if self._ds_ok:                              # the context's group size and the workspace handed over go together
            N.check(self.lib.ap_ctx_set_skip_group(self.ctx, int(self.skip_group or 0)), "ap_ctx_set_skip_group")

    def workspace(self, B: int, L: int, device) -> torch.Tensor:
        self._sync_group()
        need = self.lib.ap_workspace_bytes(self.ctx, B, L)
        if self.ws is None or self.ws.numel() < need or self.ws.device != device:
            self._drop_ws()
            try:
                self.ws = torch.empty(need, dtype=torch.uint8, device=device)
            except torch.OutOfMemoryError:
                torch.cuda.empty_cache()                     # (free blocks of other sizes the cache still holds)
                self.ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self.ws

    def clips_that_fit(self, L: int, device) -> int:
        """How many clips of length L one call can take with the memory that is free now (>= 1)."""
        self._sync_group()
        per_clip = max(self.lib.ap_workspace_bytes(self.ctx, 2, L) - self.lib.ap_workspace_bytes(self.ctx, 1, L), 1)
        try:
            free, _ = torch.cuda.mem_get_info(device)
            free += max(torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device), 0)
        except (RuntimeError, AssertionError):
            return self.max_chunk
        free += self.ws.numel() if self.ws is not None and self.ws.device == device else 0
        return max(1, int(self.WS_FRACTION * free) // per_clip)

    def chunks(self, B: int, L: int = 0, device=None):
        """Clip ranges of one native call each: at most ``max_chunk`` clips and -- when the clip length is given -- no more than
        the workspace of which fits the free memory.  Results do not depend on the split (noise is keyed on the global index)."""
        step = self.max_chunk
        if L and device is not None:
            self._sync_group()
            if self.ws is None or self.ws.device != device or self.ws.numel() < self.lib.ap_workspace_bytes(self.ctx, min(B, step), L):
                step = min(step, self.clips_that_fit(L, device))
        for s in range(0, B, step):
            yield s, min(B, s + step)


class WaveNet_Speech_Commands(nn.Module):
    _engine = None                             # class-level defaults: survive copy / un-pickling without __init__
    _precision = N.AP_PREC_F32
    _f32_form = 1                              # AP_PREC_F32: 1 = minimal-filtering (F(2,3)) form of the dilated conv where built, 0 = direct

    def __getstate__(self):                    # the native context (device weights, workspace) is rebuilt on demand
        d = dict(self.__dict__)
        d.pop("_engine", None)
        return d

    def __init__(self, in_channels=1, res_channels=256, skip_channels=128, out_channels=1,
                 num_res_layers=30, dilation_cycle=10,
                 diffusion_step_embed_dim_in=128,
                 diffusion_step_embed_dim_mid=512,
                 diffusion_step_embed_dim_out=512):
        super().__init__()
        if in_channels != 1 or out_channels != 1:
            raise NotImplementedError("audiopure_amd: only in_channels = out_channels = 1 (raw waveform) is built")
        self.config = dict(in_channels=in_channels, res_channels=res_channels, skip_channels=skip_channels,
                           out_channels=out_channels, num_res_layers=num_res_layers, dilation_cycle=dilation_cycle,
                           diffusion_step_embed_dim_in=diffusion_step_embed_dim_in,
                           diffusion_step_embed_dim_mid=diffusion_step_embed_dim_mid,
                           diffusion_step_embed_dim_out=diffusion_step_embed_dim_out)
        C_, S_ = res_channels, skip_channels
        # same tree / key names as the reference (WaveNet.py:147-162): init_conv.0.conv.*, residual_layer.*, final_conv.{0,2}.conv.*
        self.init_conv = nn.Sequential(_ConvBox(_WNConv(in_channels, C_, 1)), nn.Identity())
        self.residual_layer = _ResidualGroupParams(C_, S_, num_res_layers, diffusion_step_embed_dim_in,
                                                   diffusion_step_embed_dim_mid, diffusion_step_embed_dim_out)
        self.final_conv = nn.Sequential(_ConvBox(_WNConv(S_, S_, 1)), nn.Identity(), _ConvBox(_PlainConv(S_, out_channels, 1)))
        self._engine = None
        self._precision = N.AP_PREC_F32

    # ---- native plumbing ------------------------------------------------------------------
    def set_precision(self, mode: str):
        """"f32": exact fp32 MFMA (default, the reference's arithmetic); at res = skip = 256 channels the dilated conv runs in its
        F(2,3) minimal-filtering form (include/audiopure.h: ap_ctx_set_f32_form), "f32d" keeps the direct form.  "f32s": fp32 operands split exactly into three
        bf16 parts, six partial products per product on the bf16 MFMA, fp32 accumulate -- fp32-class results (held to the fp32
        tolerances and, on adversarial operands, to twice the direct fp32 kernel's error against fp64), ~1.3x the F(2,3) form's rate.
        "bf16": bf16 MFMA operands, fp32 accumulate and storage (BASELINE configs[3]).  "bf16s": the same arithmetic with the
        residual stream stored as bf16 between layers (SURVEY.md 8d "bf16 MFMA, bf16 storage"; one more rounding per layer;
        input gradients through the bf16 backward kernels).  All but "f32" need res_channels = 256."""
        modes = {"f32": N.AP_PREC_F32, "fp32": N.AP_PREC_F32, "f32d": N.AP_PREC_F32, "bf16": N.AP_PREC_BF16,
                 "f32s": N.AP_PREC_F32_SPLIT, "f32_split": N.AP_PREC_F32_SPLIT, "f32sw": N.AP_PREC_F32_SPLIT, "bf16s": N.AP_PREC_BF16_STORE,
                 "bf16_store": N.AP_PREC_BF16_STORE}
        if mode not in modes:
            raise ValueError(f"set_precision: unknown mode {mode!r} (one of {sorted(modes)}; 'f32h' was removed in round 5)")
        prec = modes[mode]
        # "f32sw": the split mode with the dilated conv in F(2,3) form (two launches per block, ap_resblock_f32s2.hip): opt-in -- 3-4 %
        # faster than "f32s" (both power-capped) and with the F(2,3) form's looser bound on a 2^40 dynamic range
        self._f32_form = 0 if mode in ("f32d", "f32s", "f32_split") else 1
        if prec != self._precision:
            self._precision = prec
            self._engine = None
        return self

    def _blob_tensors(self):
        r = self.residual_layer
        ic, f0, f2 = self.init_conv[0].conv, self.final_conv[0].conv, self.final_conv[2].conv
        ts = [ic.bias, ic.weight_g, ic.weight_v, r.fc_t1.weight, r.fc_t1.bias, r.fc_t2.weight, r.fc_t2.bias]
        for b in r.residual_blocks:
            d = b.dilated_conv_layer.conv
            ts += [b.fc_t.weight, b.fc_t.bias, d.bias, d.weight_g, d.weight_v,
                   b.res_conv.bias, b.res_conv.weight_g, b.res_conv.weight_v,
                   b.skip_conv.bias, b.skip_conv.weight_g, b.skip_conv.weight_v]
        ts += [f0.bias, f0.weight_g, f0.weight_v, f2.weight, f2.bias]
        return ts

    @N.on_device
    def engine(self) -> NativeEngine:
        """Fold + pack the current parameters into the native context (re-done when any parameter changed)."""
        ts = self._blob_tensors()
        dev = ts[0].device
        if dev.type != "cuda":
            raise N.NativeError("audiopure_amd WaveNet needs its parameters on a HIP device (.cuda()); no CPU path")
        key = (dev, tuple((t._version, t.data_ptr()) for t in ts))
        if self._engine is None:
            self._engine = NativeEngine(self.config, self._precision)
        if self._engine.loaded_key != key:
            with torch.no_grad():
                blob = torch.cat([t.detach().reshape(-1).float() for t in ts]).contiguous()
                freq = embedding_frequencies(self.config["diffusion_step_embed_dim_in"]).to(dev).contiguous()
                self._engine.load(blob, freq)
            self._engine.loaded_key = key
        if self._precision in (N.AP_PREC_F32, N.AP_PREC_F32_SPLIT):
            N.check(self._engine.lib.ap_ctx_set_f32_form(self._engine.ctx, int(self._f32_form)), "ap_ctx_set_f32_form")
        return self._engine

    @staticmethod
    def _check_input(x: torch.Tensor):
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"expected audio of shape [B,1,L], got {tuple(x.shape)}")

    @N.on_device
    def eps(self, x: torch.Tensor, step: float) -> torch.Tensor:
        """eps_theta(x, step) for a step shared by the batch."""
        self._check_input(x)
        if torch.is_grad_enabled() and x.requires_grad:          # d eps / d audio on the HIP library (parameters frozen)
            from .._grad import differentiable_eps
            return differentiable_eps(self, x, float(step))
        eng = self.engine()
        x = x.detach().float().contiguous()
        B, _, L = x.shape
        out = torch.empty_like(x)
        for s, e in eng.chunks(B, L, x.device):
            ws = eng.workspace(e - s, L, x.device)
            N.check(eng.lib.ap_eps_fwd(eng.ctx, N.ptr(x[s:e]), float(step), N.ptr(out[s:e]), e - s, L, ws.data_ptr(),
                                       ws.numel(), N.stream()), "ap_eps_fwd")
        return out

    def forward(self, input_data):
        audio, diffusion_steps = input_data          # tuple argument, as the reference (WaveNet.py:164-165)
        if torch.is_tensor(diffusion_steps):
            steps = diffusion_steps.detach().reshape(-1).float().cpu()
            if steps.numel() not in (1, audio.shape[0]):
                raise ValueError("diffusion_steps must have one entry per clip")
            uniq = torch.unique(steps)
            if uniq.numel() == 1:
                return self.eps(audio, float(uniq[0]))
            out = torch.empty_like(audio, dtype=torch.float32)
            for u in uniq.tolist():                    # per-clip steps: one native call per distinct value
                idx = (steps == u).nonzero().reshape(-1).to(audio.device)
                out[idx] = self.eps(audio[idx], u)
            return out
        return self.eps(audio, float(diffusion_steps))
