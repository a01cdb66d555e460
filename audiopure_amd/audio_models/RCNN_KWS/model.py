"""``KWSModel`` with the reference's constructor, state-dict keys and call surface
(audio_models/RCNN_KWS/model.py:66-114; kws_adaptive_attack_eval.py:74-75 ``KWSModel(in_size=n_mels)`` +
``load_state_dict``); the forward pass is one HIP kernel per batch (``ap_kws_fwd``).

The submodules below exist only to own parameters under the reference's names (``CRNN_model.sepconv.{0,1}``,
``CRNN_model.gru`` incl. the ``_reverse`` tensors, ``attn_layer.{Wx_b,Vt}``, ``apply_attn.U``); their torch forward is
never called.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _native as N


class _CRNN(nn.Module):
    def __init__(self, in_size, hidden_size, kernel_size, stride, gru_nl):
        super().__init__()
        self.sepconv = nn.Sequential(
            nn.Conv1d(in_size, in_size, kernel_size[1], stride=stride[1], groups=in_size),
            nn.Conv1d(in_size, hidden_size, kernel_size=1, stride=stride[0], groups=int(in_size / kernel_size[0])))
        self.gru = nn.GRU(input_size=hidden_size, hidden_size=hidden_size, num_layers=gru_nl, bidirectional=True)


class _AttnMech(nn.Module):
    def __init__(self, lin_size):
        super().__init__()
        self.Wx_b = nn.Linear(lin_size, lin_size)
        self.Vt = nn.Linear(lin_size, 1, bias=False)


class _ApplyAttn(nn.Module):
    def __init__(self, in_size, num_classes):
        super().__init__()
        self.U = nn.Linear(in_size, num_classes, bias=False)


class _KwsInputGrad(torch.autograd.Function):
    """log-probabilities with a gradient with respect to the mel input only (parameters are frozen at evaluation)."""

    @staticmethod
    def forward(ctx, x, model):
        h = model._handle()
        xc = x.detach().float().contiguous()
        out = torch.empty((xc.shape[0], model.num_classes), device=xc.device, dtype=torch.float32)
        N.check(N.lib().ap_kws_fwd(h, N.ptr(xc), N.ptr(out), xc.shape[0], xc.shape[2], N.stream()), "ap_kws_fwd")
        ctx.model, ctx.x = model, xc
        return out

    @staticmethod
    def backward(ctx, g):
        model, xc = ctx.model, ctx.x
        h = model._handle()
        B, _, T = xc.shape
        g = g.detach().float().contiguous()
        dx = torch.empty_like(xc)
        scr = torch.empty(N.lib().ap_kws_bwd_scratch_elems(h, B, T), device=xc.device, dtype=torch.float32)
        N.check(N.lib().ap_kws_bwd(h, N.ptr(xc), N.ptr(g), N.ptr(dx), N.ptr(scr), B, T, N.stream()), "ap_kws_bwd")
        return dx, None


class KWSModel(nn.Module):
    _h = None                                  # class-level defaults: survive un-pickling without __init__
    _key = None

    def __getstate__(self):                    # the device handle is rebuilt on demand, never pickled
        d = dict(self.__dict__)
        d.pop("_h", None)
        d.pop("_key", None)
        return d

    def __init__(self, in_size=40, hidden_size=64, kernel_size=(20, 5), stride=(8, 2), gru_num_layers=2, num_dirs=2,
                 num_classes=4):
        super().__init__()
        if tuple(kernel_size) != (20, 5) or tuple(stride) != (8, 2) or gru_num_layers != 2 or num_dirs != 2:
            raise NotImplementedError("audiopure_amd KWSModel: only the reference's default separable-conv / 2-layer "
                                      "bidirectional GRU configuration is built")
        self.in_size, self.hidden_size, self.kernel_size, self.stride = in_size, hidden_size, kernel_size, stride
        self.gru_num_layers, self.num_dirs, self.num_classes = gru_num_layers, num_dirs, num_classes
        self.CRNN_model = _CRNN(in_size, hidden_size, kernel_size, stride, gru_num_layers)
        self.attn_layer = _AttnMech(hidden_size * num_dirs)
        self.apply_attn = _ApplyAttn(hidden_size * 2, num_classes)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().ap_kws_destroy(self._h)
        except Exception:
            pass

    def _handle(self):
        ts = [v for v in self.state_dict().values()]
        dev = ts[0].device
        if dev.type != "cuda":
            raise N.NativeError("audiopure_amd KWSModel needs its parameters on a HIP device (.cuda()); no CPU path")
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._h is None or key != self._key:
            if self._h is not None:
                N.lib().ap_kws_destroy(self._h)
                self._h = None
            blob = torch.cat([t.detach().reshape(-1).float() for t in ts]).contiguous()
            h = C.c_void_p()
            N.check(N.lib().ap_kws_create(self.in_size, self.hidden_size, self.num_classes, N.ptr(blob), blob.numel(), N.stream(),
                                          C.byref(h)), "ap_kws_create")
            self._h, self._key = h, key
        return self._h

    @N.on_device
    def forward(self, batch, hidden=None):
        """batch: mel-dB spectrogram [B,1,n_mels,T] (or [B,n_mels,T]) -> log-probabilities [B,num_classes]."""
        if self.training:
            raise NotImplementedError("audiopure_amd KWSModel: inference only; call .eval()")
        if hidden is not None:
            raise NotImplementedError("audiopure_amd KWSModel: a caller-supplied initial GRU state is not built "
                                      "(the scripts pass none, model.py:99-100)")
        x = batch.squeeze(1) if batch.ndim == 4 else batch
        if x.ndim != 3 or x.shape[1] != self.in_size:
            raise ValueError(f"expected [B,1,{self.in_size},T], got {tuple(batch.shape)}")
        if torch.is_grad_enabled() and x.requires_grad and x.shape[0] > 0:
            return _KwsInputGrad.apply(x.float(), self)          # white-box attack: d/d(mel) on the device (ap_kws_bwd)
        h = self._handle()
        x = x.detach().float().contiguous()
        out = torch.empty((x.shape[0], self.num_classes), device=x.device, dtype=torch.float32)
        if x.shape[0] == 0:
            return out
        N.check(N.lib().ap_kws_fwd(h, N.ptr(x), N.ptr(out), x.shape[0], x.shape[2], N.stream()), "ap_kws_fwd")
        return out
