"""Input gradient of the purification chain.

The reference's white-box adaptive attack back-propagates a loss through the defender
(``robustness_eval/white_box_attack.py:392,437-439``; the scripts build ``RevDiffWave`` whose
``torchsde.sdeint_adjoint`` re-integrates backwards, ``diffusion_models/diffwave_sde.py:200-204``).
Every sampler here is a chain of links ``x <- ca x + cb eps(x, t) + cs z`` (``ap_step``), so the gradient is

    dL/dx_t = ca dL/dx_{t-1} + cb J_eps(x_t, t)^T dL/dx_{t-1}

with the states x_t check-pointed in the forward pass.  A link's 37 layer inputs are kept for the backward pass while the
chain fits a memory budget (``SAVE_BUDGET_BYTES``; 288 GB of HBM hold a PGD batch comfortably) and recomputed otherwise
(the adjoint's memory/compute trade).

``J_eps^T v`` runs on the HIP library.  Shipped shape (res = skip = 256 channels), fp32: the forward keeps the pre-gate
activations (``ap_resblock_fwd_save``) and a block's backward is ``ap_resblock_bwd`` -- two fused launches (the gate's
derivative behind ``W2^T [dh'; dskip]``, then the transposed dilated conv in F(2,3) form with the residual path in its
epilogue): the forward block's flops, no elementwise glue.  bf16 mode at the same shape: ``ap_resblock_bwd_bf16`` -- two launches per
layer on the bf16 matrix pipe from the layer inputs (the dilated conv recomputed inside), or -- the default -- ``ap_resblock_bwd_bf16_saved``
from gate derivative factors the forward kept; bf16-storage mode: the same backward behind ``ap_resblock_fwd_u_save``.  Every other shape / arithmetic mode: the residual blocks'
forward is the fused kernel (``ap_resblock_fwd``), the three GEMM-shaped backward terms of a block -- the recomputed dilated conv, ``W2^T [dh'; dskip]`` and the transposed
dilated conv -- are ``ap_conv2d_fwd`` calls in ``AP_CONV_1D`` mode (MFMA conv-as-GEMM, weights streamed as
fragments), with ``ap_gate_bwd`` / ``ap_relu_outer_bwd`` / ``ap_init_conv_bwd`` between them.  Gradients with respect
to the network's parameters are not formed (the attack differentiates with respect to the audio only; parameters
are frozen at evaluation, ``adaptive_attack_eval.py:98-101``).
"""
from __future__ import annotations

import math

import torch

from .. import _native as N

_RS = 0.707106781186547524
_F1D = 0x200                      # AP_CONV_1D


class EpsGrad:
    """Backward weights of one WaveNet_Speech_Commands plus eps forward-with-save / backward."""

    def __init__(self, net):
        self.net = net
        self._key = None
        self._gimg = None               # bf16 mode: the group's gate images of forward_save, one buffer reused by every link
        self.keep_gate_factors = True   # bf16 mode: keep the gate's derivative factors in the forward pass (False: recompute the dilated conv in the backward)
        self.fused_bf16 = True          # tools/check_bwd_bf16.py turns it off to time / compare the composed fp32 backward in bf16 mode

    # ---- weights ---------------------------------------------------------------------------------------
    def _prepare(self):
        net = self.net
        eng = net.engine()
        key = (eng.serial, eng.loaded_key, net._precision)   # (serial, not id(): a precision switch builds a new engine, possibly at the old one's address)
        if self._key == key:
            return eng
        lib, dev = eng.lib, next(net.parameters()).device
        C_, S_, NL = eng.cfg.res_channels, eng.cfg.skip_channels, eng.cfg.num_res_layers
        cyc = eng.cfg.dilation_cycle
        if net._precision == N.AP_PREC_BF16_STORE and not (C_ == 256 and S_ == 256):
            raise N.NativeError("set_precision('bf16s') (AP_PREC_BF16_STORE) needs res = skip = 256 channels")
        if C_ == 256 and S_ == 256 and net._precision in (N.AP_PREC_F32, N.AP_PREC_BF16, N.AP_PREC_BF16_STORE):
            # the fused backward kernels' own weight images: built here, once per load, outside any stream capture (the launch
            # functions allocate nothing: include/audiopure.h, ap_ctx_prepare_backward)
            N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()), "ap_ctx_prepare_backward")

        def folded(which, layer, n):
            t = torch.empty(n, device=dev, dtype=torch.float32)
            N.check(lib.ap_ctx_get_folded(eng.ctx, which, layer, N.ptr(t), n, N.stream()), "ap_ctx_get_folded")
            return t

        def pack(w4):                                           # [Cout][Cin][kh][kw] -> packed images
            w4 = w4.contiguous()
            co, ci, kh, kw = w4.shape
            out = torch.empty(lib.ap_conv2d_packed_elems(co, ci, kh, kw, 1), device=dev, dtype=torch.float32)
            N.check(lib.ap_conv2d_pack(N.ptr(w4), None, N.ptr(out), co, ci, kh, kw, 1, N.stream()), "ap_conv2d_pack")
            return out

        self.layers = []
        blocks = net.residual_layer.residual_blocks
        for n in range(NL):
            w1 = folded(0, n, 2 * C_ * C_ * 3).reshape(2 * C_, C_, 1, 3)
            w2 = torch.cat([folded(1, n, C_ * C_).reshape(C_, C_), folded(2, n, S_ * C_).reshape(S_, C_)], 0)   # [(C+S)][C]
            self.layers.append(dict(
                d=2 ** (n % cyc),
                a=pack(w1),                                                           # u -> pre-gate a   (C -> 2C, k=3, dil d)
                g=pack(w2.t().reshape(C_, C_ + S_, 1, 1)),                            # [RS dh'; dskip] -> dg  (C+S -> C, 1x1)
                u=pack(w1.flip(3).permute(1, 0, 2, 3)),                               # da -> du   (2C -> C, k=3 flipped, dil d)
                b1=blocks[n].dilated_conv_layer.conv.bias.detach().float().contiguous()))
        s = float(torch.tensor(math.sqrt(1.0 / NL), dtype=torch.float32))             # WaveNet.py:135
        wf1 = folded(3, 0, S_ * S_).reshape(S_, S_) * s
        self.wf1 = pack(wf1.reshape(S_, S_, 1, 1))                                    # skip_sum -> r (scale folded in)
        self.wf1_t = pack(wf1.t().reshape(S_, S_, 1, 1))                              # dr -> dskip
        self.bf1 = net.final_conv[0].conv.bias.detach().float().contiguous()
        self.wf2 = net.final_conv[2].conv.weight.detach().float().reshape(-1).contiguous()
        self.w0 = folded(4, 0, C_)
        self.ones = torch.ones(C_, device=dev)
        self.C, self.S, self.NL = C_, S_, NL
        torch.cuda.synchronize(dev)
        self._key = key
        return eng

    def _conv(self, lib, x, packed, bias, res, out, B, Cin, L, Cout, kw, pad, dil, flags=0):
        if self.net._precision in (N.AP_PREC_F32_SPLIT, N.AP_PREC_BF16, N.AP_PREC_BF16_STORE):   # follow the network's arithmetic mode: the split GEMM (fp32-class
            flags |= 0x100                                       # results from the bf16 matrix pipe, AP_CONV_SPLIT) where the forward ran on that pipe too
        fl = flags | _F1D | ((dil << 16) if dil > 1 else 0)
        N.use_conv_workspace(x.device)                           # short clips meet the split-K condition: this device's buffer
        N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(packed), N.ptr(bias), N.ptr(res), N.ptr(out), B, Cin, 1, L, Cout, 1, kw, 1,
                                  pad, 1, fl, Cin, 0, N.stream()), "ap_conv2d_fwd")

    def saved_bytes(self, x: torch.Tensor, acts: bool = True) -> int:
        """Bytes ``forward_save(x, ., acts)`` keeps, computed without allocating: NL + 1 layer inputs [B][C][L], the skip sum,
        the FiLM vectors and -- ``acts`` in fp32 arithmetic -- NL pre-gate tensors [B][2C][L]; in bf16 mode also the gate-image
        buffer of the deferred-skip forward while this object does not hold one of that size yet (it is allocated once and
        reused by every link, so only the first link is charged for it)."""
        eng = self._prepare()
        B, _, L = x.shape
        C_, S_, NL = self.C, self.S, self.NL
        bstore = self.net._precision == N.AP_PREC_BF16_STORE
        keeps = bstore or (acts and self._keeps_factors(eng, B, L))
        # (kept factors: layer 0's input + a ping-pong pair -- a pair of bf16 u images, half the bytes, with bf16 storage)
        n = ((2 if bstore else 3) if keeps else NL + 1) * B * C_ * L + B * S_ * L + NL * C_ + eng.cfg.embed_dim_out
        if acts and self.net._precision == N.AP_PREC_F32 and C_ in (64, 256):
            n += NL * B * 2 * C_ * L
        extra = 0
        if keeps:
            extra += NL * int(eng.lib.ap_gate_factor_bytes(B, L))
        G = self._group(eng)
        if G > 0:
            need = min(G, NL) * B * L * C_ * 2
            if self._gimg is None or self._gimg.numel() * 2 < need or self._gimg.device != x.device:
                extra = need
        return 4 * n + extra

    def _keeps_factors(self, eng, B, L) -> bool:
        """bf16 mode at the shipped shape: the forward pass keeps the gate's derivative factors (ap_resblock_fwd_gate_save: 16.4 MB per
        clip-second and layer) and the backward reads them instead of recomputing the dilated conv (ap_resblock_bwd_bf16_saved)."""
        return bool(self.fused_bf16 and self.keep_gate_factors and self._group(eng) > 0 and
                    eng.lib.ap_resblock_bwd_bf16_available(eng.ctx, B, L))

    def _group(self, eng) -> int:
        """Layers per skip GEMM of forward_save's deferred-skip form (bf16 mode; 0: the fused block per layer)."""
        return (int(eng.skip_group or 0) if getattr(eng, "_ds_ok", False) and
                self.net._precision in (N.AP_PREC_BF16, N.AP_PREC_BF16_STORE) else 0)

    def eps_only(self, x: torch.Tensor, step: float):
        """The plain fused forward (``ap_eps_fwd``): what the chain's forward pass calls -- nothing is kept."""
        with torch.no_grad():
            return self.net.eps(x.detach(), step)

    # ---- one eps evaluation, keeping what its backward needs ---------------------------------------------------
    def forward_save(self, x: torch.Tensor, step: float, acts: bool = True):
        """One eps evaluation that keeps what its backward needs: every layer's input and -- ``acts`` and fp32 arithmetic --
        every layer's pre-gate activations (``ap_resblock_fwd_save``; 3x the memory, and the backward skips the dilated
        conv's recomputation)."""
        eng = self._prepare()
        lib, dev = eng.lib, x.device
        B, _, L = x.shape
        C_, S_, NL = self.C, self.S, self.NL
        part = torch.empty(NL * C_ + eng.cfg.embed_dim_out, device=dev)
        N.check(lib.ap_embed(eng.ctx, float(step), N.ptr(part), N.stream()), "ap_embed")
        if self.net._precision == N.AP_PREC_BF16_STORE:
            return self._forward_save_bstore(eng, x, step, part)
        keeps = acts and self._keeps_factors(eng, B, L)
        # with kept gate factors the backward needs no layer input but the first (ap_init_conv_bwd): hs = [h_0, ping, pong]
        hs = torch.empty((3 if keeps else NL + 1, B, C_, L), device=dev)
        src = (lambda n: 0 if n == 0 else 1 + ((n - 1) & 1)) if keeps else (lambda n: n)
        dst = (lambda n: 1 + (n & 1)) if keeps else (lambda n: n + 1)
        skip = torch.empty((B, S_, L), device=dev)
        pre = torch.empty((NL, B, 2 * C_, L), device=dev) if acts and self.net._precision == N.AP_PREC_F32 and C_ in (64, 256) else None
        N.check(lib.ap_init_conv(eng.ctx, N.ptr(x), N.ptr(hs[0]), B, L, N.stream()), "ap_init_conv")
        G = self._group(eng)
        if G > 0:
            # bf16 mode, the deferred-skip form the chain's own forward runs (ap_resblock_fwd_gate + one ap_skip_gemm per group of G
            # layers: same grouping, so eps equals ap_eps_fwd's bit for bit); the last layer's h' is not computed (nobody reads it:
            # hs[NL] stays unwritten and backward() starts from dh = 0).  The gate images live in ONE buffer held by this object
            # and reused by every link (295 MB per clip-second at G = 36; saved_bytes charges it to the chain's budget once).
            need = min(G, NL) * B * L * C_
            if self._gimg is None or self._gimg.numel() < need or self._gimg.device != dev:
                self._gimg = None
                self._gimg = torch.empty(need, device=dev, dtype=torch.bfloat16)
            gimg = self._gimg[:need].view(min(G, NL), B, L, C_)
            if keeps:
                pre = torch.empty((NL, int(lib.ap_gate_factor_bytes(B, L))), device=dev, dtype=torch.uint8)   # (rides in the `pre` slot of `saved`)
            for n0 in range(0, NL, G):
                nl = min(G, NL - n0)
                for n in range(n0, n0 + nl):
                    if pre is not None:
                        N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, n, N.ptr(hs[src(n)]), N.ptr(part[n * C_:(n + 1) * C_]),
                                                              N.ptr(hs[dst(n)]) if n + 1 < NL else None, gimg[n - n0].data_ptr(),
                                                              pre[n].data_ptr(), B, L, N.stream()), "ap_resblock_fwd_gate_save")
                        continue
                    N.check(lib.ap_resblock_fwd_gate(eng.ctx, n, N.ptr(hs[n]), N.ptr(part[n * C_:(n + 1) * C_]),
                                                     N.ptr(hs[n + 1]) if n + 1 < NL else None, gimg[n - n0].data_ptr(), B, L, N.stream()),
                            "ap_resblock_fwd_gate")
                N.check(lib.ap_skip_gemm(eng.ctx, n0, nl, gimg.data_ptr(), N.ptr(skip), 1 if n0 else 0, B, L, N.stream()), "ap_skip_gemm")
        for n in range(NL if G == 0 else 0):
            pt = part[n * C_:(n + 1) * C_]
            if pre is not None:
                N.check(lib.ap_resblock_fwd_save(eng.ctx, n, N.ptr(hs[n]), N.ptr(pt), N.ptr(hs[n + 1]), N.ptr(skip), N.ptr(pre[n]),
                                                 1 if n else 0, B, L, N.stream()), "ap_resblock_fwd_save")
            else:
                N.check(lib.ap_resblock_fwd(eng.ctx, n, N.ptr(hs[n]), N.ptr(pt), N.ptr(hs[n + 1]), N.ptr(skip),
                                            1 if n else 0, B, L, N.stream()), "ap_resblock_fwd")
        eps = torch.empty((B, 1, L), device=dev)
        N.check(lib.ap_final_affine(eng.ctx, N.ptr(skip), None, N.ptr(eps), None, 0.0, 0.0, 0.0, None, 0, 0, 0, B, L,
                                    N.stream()), "ap_final_affine")
        return eps, (hs, skip, part, pre)

    def _forward_save_bstore(self, eng, x, step, part):
        """AP_PREC_BF16_STORE: the sweep ap_eps_fwd runs in this mode (ap_init_conv_u, ap_resblock_fwd_u per layer, one ap_skip_gemm per
        group: eps equals ap_eps_fwd's bit for bit) with every block also keeping its gate's derivative factors
        (ap_resblock_fwd_u_save).  The backward is the bf16 mode's (ap_resblock_bwd_bf16_saved): the rounding of the stored residual
        passes the gradient straight through, everything else about the two forwards is the same arithmetic.  There is no lean
        form (no fp32 layer inputs exist to recompute from): a link either keeps its factors or is recomputed whole."""
        lib, dev = eng.lib, x.device
        B, _, L = x.shape
        C_, S_, NL = self.C, self.S, self.NL
        G = self._group(eng)
        if G <= 0 or not lib.ap_resblock_bwd_bf16_available(eng.ctx, B, L):
            raise N.NativeError("set_precision('bf16s'): no backward for this shape (res = skip = 256 channels, the deferred-skip form)")
        hs = torch.empty((1, B, C_, L), device=dev)                         # h_0 in fp32: what ap_init_conv_bwd reads
        N.check(lib.ap_init_conv(eng.ctx, N.ptr(x), N.ptr(hs[0]), B, L, N.stream()), "ap_init_conv")
        u = torch.empty((2, B * C_ * L), device=dev, dtype=torch.bfloat16)  # the residual stream's ping-pong pair of u images
        N.check(lib.ap_init_conv_u(eng.ctx, N.ptr(x), N.ptr(part[:C_]), u[0].data_ptr(), B, L, N.stream()), "ap_init_conv_u")
        skip = torch.empty((B, S_, L), device=dev)
        need = min(G, NL) * B * L * C_
        if self._gimg is None or self._gimg.numel() < need or self._gimg.device != dev:
            self._gimg = None
            self._gimg = torch.empty(need, device=dev, dtype=torch.bfloat16)
        gimg = self._gimg[:need].view(min(G, NL), B, L, C_)
        fac = torch.empty((NL, int(lib.ap_gate_factor_bytes(B, L))), device=dev, dtype=torch.uint8)
        for n0 in range(0, NL, G):
            nl = min(G, NL - n0)
            for n in range(n0, n0 + nl):
                last = n + 1 == NL
                N.check(lib.ap_resblock_fwd_u_save(eng.ctx, n, u[n & 1].data_ptr(), None if last else N.ptr(part[(n + 1) * C_:(n + 2) * C_]),
                                                   None if last else u[(n + 1) & 1].data_ptr(), gimg[n - n0].data_ptr(), fac[n].data_ptr(),
                                                   B, L, N.stream()), "ap_resblock_fwd_u_save")
            N.check(lib.ap_skip_gemm(eng.ctx, n0, nl, gimg.data_ptr(), N.ptr(skip), 1 if n0 else 0, B, L, N.stream()), "ap_skip_gemm")
        eps = torch.empty((B, 1, L), device=dev)
        N.check(lib.ap_final_affine(eng.ctx, N.ptr(skip), None, N.ptr(eps), None, 0.0, 0.0, 0.0, None, 0, 0, 0, B, L,
                                    N.stream()), "ap_final_affine")
        return eps, (hs, skip, part, fac)

    def backward(self, saved, d_eps: torch.Tensor) -> torch.Tensor:
        """J_eps(x, t)^T d_eps for the evaluation ``saved`` came from."""
        eng = self._prepare()
        lib = eng.lib
        hs, skip, part, pre = saved
        NL, C_, S_ = self.NL, self.C, self.S
        B, L, dev = hs.shape[1], hs.shape[3], hs.device
        d_eps = d_eps.detach().float().contiguous()
        st = N.stream()
        # final_conv: r = W_f1 (skip s) + b_f1; eps = w_f2 . relu(r) + b_f2
        r = torch.empty((B, S_, L), device=dev)
        self._conv(lib, skip, self.wf1, self.bf1, None, r, B, S_, L, S_, 1, 0, 1)
        dr = torch.empty_like(r)
        N.check(lib.ap_relu_outer_bwd(N.ptr(r), N.ptr(self.wf2), N.ptr(d_eps), N.ptr(dr), B, S_, L, st), "ap_relu_outer_bwd")
        dskip = r                                                 # reuse
        self._conv(lib, dr, self.wf1_t, None, None, dskip, B, S_, L, S_, 1, 0, 1)
        dh = torch.zeros((B, C_, L), device=dev)                  # the last block's h' output is not used (WaveNet.py:133)
        if pre is not None and self.net._precision == N.AP_PREC_F32 and lib.ap_resblock_bwd_available(eng.ctx, B, L):
            # the shipped shape in fp32: two fused launches per layer (ap_resblock_bwd.hip) -- the gate's derivative as the epilogue of
            # W2^T [dh'; dskip], then the transposed dilated conv in its F(2,3) form with the residual path added in its epilogue
            dy = torch.empty((B, 2 * C_, L), device=dev)
            dh2 = torch.empty_like(dh)
            for n in range(NL - 1, -1, -1):
                N.check(lib.ap_resblock_bwd(eng.ctx, n, N.ptr(dh), N.ptr(dskip), N.ptr(pre[n]), N.ptr(dy), N.ptr(dh2), B, L, st),
                        "ap_resblock_bwd")
                dh, dh2 = dh2, dh
            dx = torch.empty((B, 1, L), device=dev)
            N.check(lib.ap_init_conv_bwd(N.ptr(hs[0]), N.ptr(self.w0), N.ptr(dh), N.ptr(dx), B, C_, L, st), "ap_init_conv_bwd")
            return dx
        if pre is not None and pre.dtype == torch.uint8:
            # bf16 mode with kept gate factors: dg = W2^T [dh'; dskip], dy = factor . dg, then the transposed dilated conv -- no recomputation
            dy = torch.empty((B, L, 2 * C_), device=dev, dtype=torch.bfloat16)
            dh2 = torch.empty_like(dh)
            dsk = torch.empty((B, L, S_), device=dev, dtype=torch.bfloat16)      # dskip once as the bf16 image every layer's kernel stages
            N.check(lib.ap_bwd_bf16_rows_image(N.ptr(dskip), dsk.data_ptr(), B, S_, L, st), "ap_bwd_bf16_rows_image")
            for n in range(NL - 1, -1, -1):
                N.check(lib.ap_resblock_bwd_bf16_saved(eng.ctx, n, pre[n].data_ptr(), N.ptr(dh), dsk.data_ptr(), 1, dy.data_ptr(), N.ptr(dh2), B, L, st),
                        "ap_resblock_bwd_bf16_saved")
                dh, dh2 = dh2, dh
            dx = torch.empty((B, 1, L), device=dev)
            N.check(lib.ap_init_conv_bwd(N.ptr(hs[0]), N.ptr(self.w0), N.ptr(dh), N.ptr(dx), B, C_, L, st), "ap_init_conv_bwd")
            return dx
        if pre is None and self.fused_bf16 and self.net._precision == N.AP_PREC_BF16 and lib.ap_resblock_bwd_bf16_available(eng.ctx, B, L):
            # bf16 mode at the shipped shape: two launches per layer on the bf16 matrix pipe (ap_resblock_bwd_bf16.hip) from the layer
            # INPUTS the forward pass wrote anyway -- the dilated conv is recomputed inside the first kernel
            dy = torch.empty((B, L, 2 * C_), device=dev, dtype=torch.bfloat16)
            dh2 = torch.empty_like(dh)
            for n in range(NL - 1, -1, -1):
                N.check(lib.ap_resblock_bwd_bf16(eng.ctx, n, N.ptr(hs[n]), N.ptr(part[n * C_:(n + 1) * C_]), N.ptr(dh), N.ptr(dskip),
                                                 dy.data_ptr(), N.ptr(dh2), B, L, st), "ap_resblock_bwd_bf16")
                dh, dh2 = dh2, dh
            dx = torch.empty((B, 1, L), device=dev)
            N.check(lib.ap_init_conv_bwd(N.ptr(hs[0]), N.ptr(self.w0), N.ptr(dh), N.ptr(dx), B, C_, L, st), "ap_init_conv_bwd")
            return dx
        z = torch.empty((B, C_ + S_, L), device=dev)             # [RS dh' ; dskip], dskip is the same for every block (the composed path only)
        N.check(lib.ap_copy_channels(N.ptr(dskip), N.ptr(z), B, S_, L, S_, 0, C_ + S_, C_, st), "ap_copy_channels")
        t1 = torch.empty_like(dh)
        dg = torch.empty_like(dh)
        u = torch.empty_like(dh) if pre is None else None
        a = torch.empty((B, 2 * C_, L), device=dev) if pre is None else None
        da = torch.empty((B, 2 * C_, L), device=dev)
        nel = dh.numel()
        for n in range(NL - 1, -1, -1):
            lay = self.layers[n]
            d = lay["d"]
            N.check(lib.ap_axpbyc(N.ptr(dh), None, N.ptr(t1), _RS, 0.0, 0.0, nel, st), "ap_axpbyc")
            N.check(lib.ap_copy_channels(N.ptr(t1), N.ptr(z), B, C_, L, C_, 0, C_ + S_, 0, st), "ap_copy_channels")
            self._conv(lib, z, lay["g"], None, None, dg, B, C_ + S_, L, C_, 1, 0, 1)
            if pre is None:                                      # not kept: recompute y = DilConv(h + part_t) + b
                pt = part[n * C_:(n + 1) * C_]
                N.check(lib.ap_affine_nchw(N.ptr(hs[n]), N.ptr(self.ones), N.ptr(pt), N.ptr(u), B, C_, L, C_, 0, 0, st),
                        "ap_affine_nchw")
                self._conv(lib, u, lay["a"], lay["b1"], None, a, B, C_, L, 2 * C_, 3, d, d)
            N.check(lib.ap_gate_bwd(N.ptr(a if pre is None else pre[n]), N.ptr(dg), N.ptr(da), B, C_, L, st), "ap_gate_bwd")
            self._conv(lib, da, lay["u"], None, t1, dh, B, 2 * C_, L, C_, 3, d, d)
        dx = torch.empty((B, 1, L), device=dev)
        N.check(lib.ap_init_conv_bwd(N.ptr(hs[0]), N.ptr(self.w0), N.ptr(dh), N.ptr(dx), B, C_, L, st), "ap_init_conv_bwd")
        return dx


def _axpby(x, y, a, b):
    """a x + b y on the library (``ap_axpbyc``); y None -> a x.  Shapes may differ as long as the element counts agree."""
    x = x if x.is_contiguous() else x.contiguous()
    if y is not None:
        y = y.float()
        y = y if y.is_contiguous() else y.contiguous()
        assert y.numel() == x.numel()
    out = torch.empty_like(x)
    N.check(N.lib().ap_axpbyc(N.ptr(x), N.ptr(y), N.ptr(out), float(a), float(b) if y is not None else 0.0, 0.0, x.numel(),
                              N.stream()), "ap_axpbyc")
    return out


SAVE_BUDGET_BYTES = 96 << 30      # ceiling per chain of what the links keep for the backward pass (288 GB of HBM)
SAVE_FREE_FRACTION = 0.6          # ... and never more than this share of the device memory that is free when the chain starts


def _chain_budget(device) -> int:
    """What one chain may keep: the ceiling, capped by a share of the memory that is actually free (several chains -- EOT
    samples, sample_step > 1 -- start one after the other, each seeing what the earlier ones hold)."""
    try:
        free, _ = torch.cuda.mem_get_info(device)
        # blocks torch's caching allocator holds but has free are as reusable as driver-free memory: after the first chain of a
        # PGD loop tens of GB sit there, and counting only the driver's figure would shrink every later chain's budget
        free += max(torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device), 0)
    except (RuntimeError, AssertionError):
        return SAVE_BUDGET_BYTES
    return int(min(SAVE_BUDGET_BYTES, SAVE_FREE_FRACTION * free))


def _saved_bytes(saved) -> int:
    n = 0
    stack = [saved]
    while stack:
        o = stack.pop()
        if isinstance(o, torch.Tensor):
            n += o.numel() * o.element_size()
        elif isinstance(o, (tuple, list)):
            stack.extend(o)
        elif isinstance(o, dict):
            stack.extend(o.values())
    return n


class _ChainFn(torch.autograd.Function):
    """x_out = chain(x_in): q-sample then the links (step, ca, cb, cs); noise tensors given explicitly.

    Every elementwise update is an ``ap_axpbyc`` call.  A link's eps-evaluation keeps its per-layer inputs and (fp32
    arithmetic) pre-gate activations for the backward pass while the chain's total stays under ``SAVE_BUDGET_BYTES`` (one
    evaluation of the shipped net is 1.8 GB per clip that way, 0.6 GB with the layer inputs only: a PGD batch of 8 clips x
    5 links is 72 GB of the 288); past the budget a link keeps the layer inputs only (its backward recomputes the dilated
    conv), then only the state entering it (the evaluation is recomputed in the backward pass: the adjoint's trade)."""

    @staticmethod
    def forward(ctx, x, grad, steps, qa, qs, zs):
        """zs[k] = draw k of the chain ([B,1,L] or [B,L]); draw 0 is the q-sample's (the numbering of ap_purify_chain)."""
        xs, saves, held = [], [], 0
        cur = x.detach().float().contiguous()
        eps_only = getattr(grad, "eps_only", None)
        with torch.no_grad():
            if qs != 0.0:
                cur = _axpby(cur, zs[0], qa, qs)
            elif qa != 1.0:
                cur = _axpby(cur, None, qa, 0.0)
            budget = _chain_budget(cur.device)
            sizes = getattr(grad, "saved_bytes", None)           # analytic sizes where the gradient object knows them: nothing is
            full = sizes(cur, True) if sizes else None           # allocated to find out that it does not fit
            lean = sizes(cur, False) if sizes else None
            for (t, ca, cb, cs, draw) in steps:
                xs.append(cur)
                if full is None:                                 # (a gradient object without sizes: measure its first link)
                    eps, saved = grad.forward_save(cur, t)
                    full = _saved_bytes(saved)
                    lean = _saved_bytes(saved[:3]) if isinstance(saved, tuple) and len(saved) == 4 else full
                    if full > budget:
                        saved = None if lean > budget else saved[:3] + (None,)
                    if saved is not None:
                        held += _saved_bytes(saved)
                    saves.append(saved)
                elif held + full <= budget:
                    eps, saved = grad.forward_save(cur, t)
                    held += full
                    saves.append(saved)
                elif lean < full and held + lean <= budget:
                    eps, saved = grad.forward_save(cur, t, acts=False)    # layer inputs only: the backward recomputes the dilated conv
                    held += lean
                    saves.append(saved)
                else:
                    eps = eps_only(cur, t) if eps_only is not None else grad.forward_save(cur, t)[0]
                    saves.append(None)
                nxt = _axpby(cur, eps, ca, cb)
                if cs != 0.0 and draw:
                    nxt = _axpby(nxt, zs[draw], 1.0, cs)
                cur = nxt
        ctx.grad, ctx.steps, ctx.qa, ctx.xs, ctx.saves = grad, steps, qa, xs, saves
        return cur

    @staticmethod
    def backward(ctx, g):
        g = g.detach().float().contiguous()
        with torch.no_grad():
            for k in range(len(ctx.steps) - 1, -1, -1):
                (t, ca, cb, cs, draw), xt = ctx.steps[k], ctx.xs[k]
                saved = ctx.saves[k]
                ctx.saves[k] = None
                if saved is None:                                # over budget in the forward pass: recompute this link -- with the
                    sizes = getattr(ctx.grad, "saved_bytes", None)   # pre-gate activations only if they fit what is free NOW
                    lean_only = sizes is not None and sizes(xt, True) > _chain_budget(xt.device)
                    _, saved = ctx.grad.forward_save(xt, t, acts=False) if lean_only else ctx.grad.forward_save(xt, t)
                g = _axpby(g, ctx.grad.backward(saved, g), ca, cb)
                del saved
            if ctx.qa != 1.0:
                g = _axpby(g, None, ctx.qa, 0.0)
        return g, None, None, None, None, None


class _EpsFn(torch.autograd.Function):
    """eps_theta(x, t) with its input gradient (the reference's network is plain differentiable torch, WaveNet.py:164-172)."""

    @staticmethod
    def forward(ctx, x, grad, step):
        with torch.no_grad():
            eps, saved = grad.forward_save(x.detach().float().contiguous(), step)
        ctx.grad, ctx.saved = grad, saved
        return eps

    @staticmethod
    def backward(ctx, g):
        with torch.no_grad():
            dx = ctx.grad.backward(ctx.saved, g.detach().float().contiguous())
        ctx.saved = None
        return dx, None, None


def _eps_grad_of(net):
    if getattr(net, "_eps_grad", None) is None or net._eps_grad.net is not net:
        net._eps_grad = EpsGrad(net)
    return net._eps_grad


def differentiable_chain(net, x, steps, qa, qs, zs):
    """The sampling chain as an autograd node (gradient with respect to ``x`` only)."""
    return _ChainFn.apply(x, _eps_grad_of(net), list(steps), float(qa), float(qs), zs)


def differentiable_eps(net, x, step):
    return _EpsFn.apply(x, _eps_grad_of(net), float(step))
