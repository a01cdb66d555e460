"""Improved-Diffusion ``UNetModel`` (the spectrogram epsilon-network of the ``DiffSpec`` defense) with the reference's
constructor, state-dict keys and ``forward(x, timesteps)`` contract
(diffusion_models/Improved_Diffusion_Unconditional/improved_diffusion/unet.py:278-491, nn.py, script_util.py:15-35,
99-126), executed on the HIP primitives of include/audiopure.h: conv-as-GEMM on the fp32 MFMA (3x3 / 1x1 / strided),
GroupNorm32 fused with the scale-shift FiLM and SiLU, QKV attention, nearest up-sampling.

The modules below only HOLD parameters under the reference's names (``input_blocks.4.0.in_layers.2.weight`` ...), so a
reference checkpoint (``model*.pt`` / ``ema_*.pt`` bare state dicts, train_util.py:274-297) loads unchanged.
Not built (raises): class conditioning, ``use_scale_shift_norm=False``, ``conv_resample=False``, dims != 2.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _native as N


# ---- parameter holders with the reference's attribute tree -----------------------------------------------------
class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(0.0),
                                        nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads):
        super().__init__()
        self.channels, self.num_heads = channels, num_heads
        self.norm = nn.GroupNorm(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class UNetModel(nn.Module):
    _packed = None                             # class-level defaults: survive copy / un-pickling without __init__
    _key = None
    _tape = None
    _conv_flags = 0

    def __getstate__(self):                    # packed device images and the tape are rebuilt on demand, never pickled
        d = dict(self.__dict__)
        for k in ("_packed", "_key", "_tape"):
            d.pop(k, None)
        d["_packed_t"] = {}
        return d

    def __init__(self, in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                 attention_resolutions=(2, 4), dropout=0, channel_mult=(1, 2, 2, 2), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, num_heads=4, num_heads_upsample=-1,
                 use_scale_shift_norm=True):
        super().__init__()
        if num_classes is not None or not use_scale_shift_norm or not conv_resample or dims != 2:
            raise NotImplementedError("audiopure_amd UNetModel: only the unconditional 2-D scale-shift-norm conv-resample "
                                      "configuration of script_util.py:15-35 is built")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_heads, self.num_heads_upsample = num_heads, num_heads_upsample
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):                       # unet.py:351-380
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(ResBlock(ch, ted, ch), AttentionBlock(ch, num_heads), ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:           # unet.py:401-430
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + chans.pop(), ted, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._packed, self._key = None, None
        self._packed_t, self._tape = {}, None
        half = model_channels // 2                                        # nn.py:113-116, as the reference computes them
        self.register_buffer("_freqs", torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half),
                             persistent=False)

    # ---- native execution --------------------------------------------------------------------------------------
    def _prepare(self):
        ps = list(self.parameters())
        dev = ps[0].device
        if dev.type != "cuda":
            raise N.NativeError("audiopure_amd UNetModel needs its parameters on a HIP device (.cuda()); no CPU path")
        key = (dev, tuple((p._version, p.data_ptr()) for p in ps))
        if key == self._key:
            return
        lib, packed = N.lib(), {}
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d, nn.Linear)):
                w = m.weight.detach().float().contiguous()
                cout = w.shape[0]
                cin_g = w.shape[1]
                kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
                wT = torch.empty(lib.ap_conv2d_packed_elems(cout, cin_g, kh, kw, 1), device=dev, dtype=torch.float32)
                N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), cout, cin_g, kh, kw, 1, N.stream()), "ap_conv2d_pack")
                packed[m] = (wT, m.bias.detach().float().contiguous() if m.bias is not None else None, cout, kh, kw,
                             (m.stride[0] if hasattr(m, "stride") else 1), (m.padding[0] if hasattr(m, "padding") else 0))
        # every ResBlock projects the SAME embedding vector (unet.py:182, emb_layers = SiLU -> Linear): one GEMM over the
        # concatenated rows of all those Linear layers replaces one 256-column launch per block (24 of them per evaluation,
        # each latency-bound at 58 us); a block then takes its own rows out of the result
        rbs = [m for m in self.modules() if isinstance(m, ResBlock)]
        wcat = torch.cat([rb.emb_layers[1].weight.detach().float() for rb in rbs], 0).contiguous()
        bcat = torch.cat([rb.emb_layers[1].bias.detach().float() for rb in rbs], 0).contiguous()
        wT = torch.empty(lib.ap_conv2d_packed_elems(wcat.shape[0], wcat.shape[1], 1, 1, 1), device=dev, dtype=torch.float32)
        N.check(lib.ap_conv2d_pack(N.ptr(wcat), None, N.ptr(wT), wcat.shape[0], wcat.shape[1], 1, 1, 1, N.stream()), "ap_conv2d_pack")
        self._emb_cat = (wT, bcat, wcat.shape[0])
        self._emb_rows, off = {}, 0
        for rb in rbs:
            n = rb.emb_layers[1].weight.shape[0]
            self._emb_rows[rb] = (off, n)
            off += n
        torch.cuda.synchronize(dev)
        self._packed, self._key = packed, key
        self._packed_t = {}                                # transposed images for the input gradient, built on first use

    def set_precision(self, mode: str):
        """"f32": fp32 MFMA convolutions (default); "f32s": the bf16 MFMA with exactly 3-way-split fp32 operands
        (AP_CONV_SPLIT): fp32-class results, faster; "f16x2" (round-3/4 name: "f32h"): two fp16 parts per operand on the fp16
        MFMA (AP_CONV_SPLIT_F16), faster again and NOT fp32-class -- fine for this net's GroupNorm-ed activations only."""
        self._conv_flags = {"f32": 0, "fp32": 0, "f32s": 0x100, "f32_split": 0x100, "f16x2": 0x400, "f32h": 0x400}[mode]
        return self

    def _conv(self, m, x, B, Cin, H, W, res=None, track=True, dest=None):
        wT, bias, cout, kh, kw, stride, pad = self._packed[m]
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        if dest is not None and Cin % 16 == 0 and cout >= 64 and getattr(self, "_conv_flags", 0) == 0 and self._tape is None:
            # the layer's output is the first half of the next torch.cat (unet.py:490-491): written in place (ap_conv2d_fwd_slice)
            cat = dest
            assert cat.shape[0] == B and cat.shape[1] >= cout and tuple(cat.shape[2:]) == (Ho, Wo)
            N.check(N.lib().ap_conv2d_fwd_slice(N.ptr(x), N.ptr(wT), N.ptr(bias), N.ptr(res), N.ptr(cat), B, Cin, H, W, cout, kh, kw,
                                                stride, pad, 1, 0, Cin, 0, cat.shape[1], 0, N.stream()), "ap_conv2d_fwd_slice")
            return cat[:, :cout]
        out = torch.empty((B, cout, Ho, Wo), device=x.device, dtype=torch.float32)
        N.check(N.lib().ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(bias), N.ptr(res), N.ptr(out), B, Cin, H, W, cout, kh, kw, stride,
                                      pad, 1, getattr(self, "_conv_flags", 0), Cin, 0, N.stream()), "ap_conv2d_fwd")
        if track and self._tape is not None:
            self._tape.append(("conv", m, x, res, out, (B, Cin, H, W)))
        return out

    def _gn(self, gn, x, ss=None, act=2):
        B, C_, H, W = x.shape
        y = torch.empty_like(x)
        N.check(N.lib().ap_groupnorm_nchw(N.ptr(x), N.ptr(gn.weight.detach()), N.ptr(gn.bias.detach()), N.ptr(ss), N.ptr(y), B, C_,
                                          H * W, gn.num_groups, float(gn.eps), act, N.stream()), "ap_groupnorm_nchw")
        if self._tape is not None:
            self._tape.append(("gn", gn, x, ss, act, y))
        return y

    def _resblock(self, rb, x, emb_proj, dest=None):
        B, C_, H, W = x.shape
        h = self._conv(rb.in_layers[2], self._gn(rb.in_layers[0], x), B, C_, H, W)                 # unet.py:181
        off, n = self._emb_rows[rb]                                                                 # :182: this block's rows of the
        ss = torch.empty((B, n), device=x.device, dtype=torch.float32)                              # one projection of the step
        N.check(N.lib().ap_copy_channels(N.ptr(emb_proj), N.ptr(ss), B, n, 1, emb_proj.shape[1], off, n, 0, N.stream()))
        h = self._gn(rb.out_layers[0], h, ss=ss)                                                   # :186-190 (+ SiLU)
        skip = x if isinstance(rb.skip_connection, nn.Identity) else self._conv(rb.skip_connection, x, B, C_, H, W)
        return self._conv(rb.out_layers[3], h, B, rb.out_channels, H, W, res=skip, dest=dest)      # :194

    def _attention(self, ab, x, dest=None):
        B, C_, H, W = x.shape
        qkv = self._conv(ab.qkv, self._gn(ab.norm, x, act=0), B, C_, H, W)                         # unet.py:229-230
        att = torch.empty_like(x)
        N.check(N.lib().ap_attention_qkv(N.ptr(qkv), N.ptr(att), B, C_, H * W, ab.num_heads, N.stream()), "ap_attention_qkv")
        if self._tape is not None:
            self._tape.append(("attn", qkv, att, ab.num_heads))
        return self._conv(ab.proj_out, att, B, C_, H, W, res=x, dest=dest)                         # :234-235

    def _out_shape(self, seq, C_, H, W):
        """(channels, H, W) of a block's output, from its layers (to allocate the concatenation buffer its last layer writes into)."""
        for layer in seq:
            if isinstance(layer, ResBlock):
                C_ = layer.out_channels
            elif isinstance(layer, Downsample):
                H, W = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            elif isinstance(layer, Upsample):
                H, W = 2 * H, 2 * W
            elif isinstance(layer, nn.Conv2d):
                _, _, C_, kh, kw, stride, pad = self._packed[layer]
                H, W = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        return C_, H, W

    def _run(self, seq, h, emb_proj, dest=None):
        n = len(seq)
        for i, layer in enumerate(seq):
            B, C_, H, W = h.shape
            d = dest if i == n - 1 else None                    # only a block's last layer writes into the next concatenation
            if isinstance(layer, ResBlock):
                h = self._resblock(layer, h, emb_proj, dest=d)
            elif isinstance(layer, AttentionBlock):
                h = self._attention(layer, h, dest=d)
            elif isinstance(layer, Upsample) and d is not None:
                up = torch.empty((B, C_, 2 * H, 2 * W), device=h.device, dtype=torch.float32)
                N.check(N.lib().ap_upsample_nearest2x(N.ptr(h), N.ptr(up), B * C_, H, W, N.stream()), "ap_upsample_nearest2x")
                h = self._conv(layer.conv, up, B, C_, 2 * H, 2 * W, dest=d)
            elif isinstance(layer, Downsample):
                h = self._conv(layer.op, h, B, C_, H, W)
            elif isinstance(layer, Upsample):
                up = torch.empty((B, C_, 2 * H, 2 * W), device=h.device, dtype=torch.float32)
                N.check(N.lib().ap_upsample_nearest2x(N.ptr(h), N.ptr(up), B * C_, H, W, N.stream()), "ap_upsample_nearest2x")
                if self._tape is not None:
                    self._tape.append(("up", h, up))
                h = self._conv(layer.conv, up, B, C_, 2 * H, 2 * W)
            elif isinstance(layer, nn.Conv2d):
                h = self._conv(layer, h, B, C_, H, W)
            else:
                raise NotImplementedError(type(layer).__name__)
        return h

    @N.on_device
    def forward(self, x, timesteps, y=None):
        assert y is None, "class conditioning is not built"
        if torch.is_grad_enabled() and x.requires_grad:
            return _UNetInputGrad.apply(x, self, timesteps)       # white-box attack: d/dx on the HIP path (input_grad)
        return self._forward(x, timesteps)

    @N.on_device
    def forward_save(self, x, timesteps):
        """One evaluation that also records what its input gradient needs: (eps, tape)."""
        self._tape = []
        try:
            out = self._forward(x, timesteps)
            return out, self._tape
        finally:
            self._tape = None

    def _forward(self, x, timesteps):
        self._prepare()
        lib = N.lib()
        N.use_conv_workspace(x.device)                                      # split-K partial sums of the 4 x 4 / 8 x 8 maps
        x = x.detach().float().contiguous()
        B, dev = x.shape[0], x.device
        if self._tape is not None:
            self._tape.append(("in", x))
        t = timesteps.detach().to(dev).float().reshape(-1).contiguous()
        if t.numel() == 1 and B > 1:
            t = t.expand(B).contiguous()
        temb = torch.empty((B, self.model_channels), device=dev, dtype=torch.float32)
        N.check(lib.ap_timestep_embedding(N.ptr(t), N.ptr(self._freqs), N.ptr(temb), B, self.model_channels, N.stream()))
        e = self._conv(self.time_embed[0], temb, B, self.model_channels, 1, 1, track=False).view(B, -1)
        e_s = torch.empty_like(e)
        N.check(lib.ap_silu(N.ptr(e), N.ptr(e_s), e.numel(), N.stream()))
        emb = self._conv(self.time_embed[2], e_s, B, e.shape[1], 1, 1, track=False).view(B, -1)    # unet.py:479
        emb_s = torch.empty_like(emb)                                      # every ResBlock starts emb_layers with SiLU
        N.check(lib.ap_silu(N.ptr(emb), N.ptr(emb_s), emb.numel(), N.stream()))
        wT, bcat, total = self._emb_cat                                    # all blocks' emb_layers Linear in one GEMM: [B, total]
        emb_proj = torch.empty((B, total), device=dev, dtype=torch.float32)
        N.check(lib.ap_conv2d_fwd(N.ptr(emb_s), N.ptr(wT), N.ptr(bcat), None, N.ptr(emb_proj), B, emb.shape[1], 1, 1, total, 1, 1, 1,
                                  0, 1, getattr(self, "_conv_flags", 0), emb.shape[1], 0, N.stream()), "ap_conv2d_fwd")
        hs, h = [], x
        for blk in self.input_blocks:                                       # :486-488
            h = self._run(blk, h, emb_proj)
            hs.append(h)
        # h = th.cat([h, hs.pop()], dim=1) (:490-491): the block that produces h writes it straight into the concatenation's
        # buffer (inference, fp32 arithmetic: ap_conv2d_fwd_slice); the skip half is copied; otherwise both halves are
        direct = self._tape is None and getattr(self, "_conv_flags", 0) == 0

        def next_cat(seq, hin):
            if not direct or not hs:
                return None
            C1, H1, W1 = self._out_shape(seq, *hin.shape[1:])
            skip = hs[-1]
            if tuple(skip.shape[2:]) != (H1, W1):
                return None
            return torch.empty((B, C1 + skip.shape[1], H1, W1), device=dev, dtype=torch.float32)
        cat = next_cat(self.middle_block, h)
        h = self._run(self.middle_block, h, emb_proj, dest=cat)
        for i, blk in enumerate(self.output_blocks):                        # :490-492
            skip = hs.pop()
            Bc, C1, H, W = h.shape
            C2 = skip.shape[1]
            if cat is not None and h.data_ptr() == cat.data_ptr():      # h is the view cat[:, :C1] its producer wrote in place
                assert cat.shape[1] == C1 + C2
            else:
                cat = torch.empty((Bc, C1 + C2, H, W), device=dev, dtype=torch.float32)
                N.check(lib.ap_copy_channels(N.ptr(h), N.ptr(cat), Bc, C1, H * W, C1, 0, C1 + C2, 0, N.stream()))
            N.check(lib.ap_copy_channels(N.ptr(skip), N.ptr(cat), Bc, C2, H * W, C2, 0, C1 + C2, C1, N.stream()))
            if self._tape is not None:
                self._tape.append(("cat", h, skip, cat))
            ncat = next_cat(blk, cat) if i + 1 < len(self.output_blocks) else None
            h = self._run(blk, cat, emb_proj, dest=ncat)
            cat = ncat
        Bc, C_, H, W = h.shape
        return self._conv(self.out[2], self._gn(self.out[0], h), Bc, C_, H, W)                     # :494

    # ---- input gradient: reverse sweep over the tape of one evaluation -----------------------------------------------
    def _conv_t(self, m):
        """Packed image of the transposed convolution of ``m`` (spatially flipped, in/out swapped)."""
        if m not in self._packed_t:
            w = m.weight.detach().float()
            if w.dim() == 3:
                w = w[..., None]
            wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
            co, ci, kh, kw = wt.shape
            img = torch.empty(N.lib().ap_conv2d_packed_elems(co, ci, kh, kw, 1), device=wt.device, dtype=torch.float32)
            N.check(N.lib().ap_conv2d_pack(N.ptr(wt), None, N.ptr(img), co, ci, kh, kw, 1, N.stream()), "ap_conv2d_pack")
            self._packed_t[m] = img
        return self._packed_t[m]

    @N.on_device
    def input_grad(self, tape, dout):
        """J^T dout of the evaluation ``tape`` came from (parameters and the timestep embedding are constants)."""
        lib, st = N.lib(), N.stream
        flags = getattr(self, "_conv_flags", 0)
        out_t = tape[-1][4]
        grads = {id(out_t): dout.detach().float().contiguous()}

        def acc(t, g):
            k = id(t)
            if k in grads:
                N.check(lib.ap_axpbyc(N.ptr(grads[k]), N.ptr(g), N.ptr(grads[k]), 1.0, 1.0, 0.0, g.numel(), st()), "ap_axpbyc")
            else:
                grads[k] = g

        for op in reversed(tape):
            kind = op[0]
            if kind == "conv":
                _, m, x, res, out, (B, Cin, H, W) = op
                g = grads.pop(id(out), None)
                if g is None:
                    continue
                _, _, cout, kh, kw, stride, pad = self._packed[m]
                Ho, Wo = out.shape[2], out.shape[3]
                if stride != 1:                                   # Downsample: spread dy onto the input grid first
                    z = torch.empty((B, cout, H, W), device=g.device, dtype=torch.float32)
                    N.check(lib.ap_zero_insert2d(N.ptr(g), N.ptr(z), B * cout, Ho, Wo, H, W, stride, st()), "ap_zero_insert2d")
                    src, Hs, Ws = z, H, W
                else:
                    src, Hs, Ws = g, Ho, Wo
                dx = torch.empty((B, Cin, H, W), device=g.device, dtype=torch.float32)
                N.check(lib.ap_conv2d_fwd(N.ptr(src), N.ptr(self._conv_t(m)), None, None, N.ptr(dx), B, cout, Hs, Ws, Cin, kh, kw, 1,
                                          kh - 1 - pad, 1, flags, cout, 0, st()), "ap_conv2d_fwd")
                acc(x, dx)
                if res is not None:
                    acc(res, g)
            elif kind == "gn":
                _, gn, x, ss, act, y = op
                g = grads.pop(id(y), None)
                if g is None:
                    continue
                B, C_, H, W = x.shape
                dx = torch.empty_like(x)
                N.check(lib.ap_groupnorm_bwd(N.ptr(x), N.ptr(gn.weight.detach()), N.ptr(gn.bias.detach()), N.ptr(ss), N.ptr(g),
                                             N.ptr(dx), B, C_, H * W, gn.num_groups, float(gn.eps), act, st()), "ap_groupnorm_bwd")
                acc(x, dx)
            elif kind == "attn":
                _, qkv, att, heads = op
                g = grads.pop(id(att), None)
                if g is None:
                    continue
                B, C_, H, W = att.shape
                dqkv = torch.empty_like(qkv)
                stats = torch.empty(B * heads * H * W * 3, device=g.device, dtype=torch.float32)
                N.check(lib.ap_attention_qkv_bwd(N.ptr(qkv), N.ptr(att), N.ptr(g), N.ptr(dqkv), N.ptr(stats), B, C_, H * W, heads,
                                                 st()), "ap_attention_qkv_bwd")
                acc(qkv, dqkv)
            elif kind == "up":
                _, h, up = op
                g = grads.pop(id(up), None)
                if g is None:
                    continue
                B, C_, H, W = h.shape
                dh = torch.empty_like(h)
                N.check(lib.ap_upsample_nearest2x_bwd(N.ptr(g), N.ptr(dh), B * C_, H, W, st()), "ap_upsample_nearest2x_bwd")
                acc(h, dh)
            elif kind == "cat":
                _, h, skip, cat = op
                g = grads.pop(id(cat), None)
                if g is None:
                    continue
                B, C1, H, W = h.shape
                C2 = skip.shape[1]
                dh, dsk = torch.empty_like(h), torch.empty_like(skip)
                N.check(lib.ap_copy_channels(N.ptr(g), N.ptr(dh), B, C1, H * W, C1 + C2, 0, C1, 0, st()), "ap_copy_channels")
                N.check(lib.ap_copy_channels(N.ptr(g), N.ptr(dsk), B, C2, H * W, C1 + C2, C1, C2, 0, st()), "ap_copy_channels")
                acc(h, dh)
                acc(skip, dsk)
            elif kind == "in":
                return grads[id(op[1])]
        raise RuntimeError("tape without an input record")


class _UNetInputGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, timesteps):
        out, tape = model.forward_save(x, timesteps)
        ctx.model, ctx.tape = model, tape
        return out

    @staticmethod
    def backward(ctx, g):
        with torch.no_grad():
            dx = ctx.model.input_grad(ctx.tape, g)
        ctx.tape = None
        return dx, None, None


class UNetEpsGrad:
    """The chain's view of the UNet (same interface as ``_grad.EpsGrad``): eps with a tape, and J^T v."""

    def __init__(self, model):
        self.model = model

    def forward_save(self, x, step):
        t = torch.full((x.shape[0],), float(step), device=x.device)
        return self.model.forward_save(x, t)

    def backward(self, saved, d_eps):
        return self.model.input_grad(saved, d_eps)


def model_and_diffusion_defaults():
    """script_util.py:15-35 (model part)."""
    return dict(image_size=32, num_channels=128, num_res_blocks=3, num_heads=4, num_heads_upsample=-1,
                attention_resolutions="16,8", dropout=0.3, learn_sigma=False, class_cond=False, diffusion_steps=200,
                noise_schedule="linear", use_checkpoint=False, use_scale_shift_norm=True)


def create_model(image_size=32, num_channels=128, num_res_blocks=3, learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16,8", num_heads=4, num_heads_upsample=-1,
                 use_scale_shift_norm=True, dropout=0.3, **_):
    """script_util.py:86-126."""
    channel_mult = {256: (1, 1, 2, 2, 4, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}[image_size]
    attention_ds = tuple(image_size // int(r) for r in attention_resolutions.split(","))
    return UNetModel(in_channels=1, model_channels=num_channels, out_channels=(1 if not learn_sigma else 2),
                     num_res_blocks=num_res_blocks, attention_resolutions=attention_ds, dropout=dropout,
                     channel_mult=channel_mult, num_classes=(1000 if class_cond else None), use_checkpoint=use_checkpoint,
                     num_heads=num_heads, num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm)
