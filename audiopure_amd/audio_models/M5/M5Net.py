"""``M5`` raw-waveform classifier with the reference's constructor, attribute names and state dict
(audio_models/M5/M5Net.py:4-38); eval-mode forward runs as one fused HIP kernel (ap_m5_fwd).
The class keeps the name ``M5`` because the eval scripts select ``transform=None`` by
``Classifier._get_name() == 'M5'`` (adaptive_attack_eval.py:90-93)."""
import ctypes as C

import torch
import torch.nn as nn

from ... import _native as N


class _M5InputGrad(torch.autograd.Function):
    """log-probabilities with the gradient w.r.t. the waveform formed by ap_m5_bwd (white_box_attack.py:392,437-439)."""

    @staticmethod
    def forward(ctx, x, mod):
        with torch.no_grad():
            out = mod.forward(x.detach())
        ctx.mod = mod
        ctx.save_for_backward(x.detach().float().contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.detach().float().contiguous()
        dx = torch.empty_like(x)
        N.check(N.lib().ap_m5_bwd(ctx.mod._handle(), N.ptr(x), N.ptr(g), N.ptr(dx), x.shape[0], x.shape[2], N.stream()),
                "ap_m5_bwd")
        return dx, None


class M5(nn.Module):
    # Class-level defaults: the scripts obtain the classifier by un-pickling a whole module (audio_models/create_model.py:8-17),
    # which restores __dict__ without running __init__ -- a reference-pickled ``M5Net.M5`` lands here with the reference's
    # attributes only.
    _native = None
    _key = None

    def __getstate__(self):                    # the device handle is rebuilt on demand, never pickled
        d = dict(self.__dict__)
        d.pop("_native", None)
        d.pop("_key", None)
        return d

    def __init__(self, n_input=1, first_kernel_size=80, n_output=35, stride=16, n_channel=32):
        super().__init__()
        if n_input != 1:
            raise NotImplementedError("audiopure_amd M5: n_input must be 1 (mono waveform)")
        self.conv1 = nn.Conv1d(n_input, n_channel, kernel_size=first_kernel_size, stride=stride)
        self.bn1 = nn.BatchNorm1d(n_channel)
        self.pool1 = nn.MaxPool1d(4)
        self.conv2 = nn.Conv1d(n_channel, n_channel, kernel_size=3)
        self.bn2 = nn.BatchNorm1d(n_channel)
        self.pool2 = nn.MaxPool1d(4)
        self.conv3 = nn.Conv1d(n_channel, 2 * n_channel, kernel_size=3)
        self.bn3 = nn.BatchNorm1d(2 * n_channel)
        self.pool3 = nn.MaxPool1d(4)
        self.conv4 = nn.Conv1d(2 * n_channel, 2 * n_channel, kernel_size=3)
        self.bn4 = nn.BatchNorm1d(2 * n_channel)
        self.pool4 = nn.MaxPool1d(4)
        self.fc1 = nn.Linear(2 * n_channel, n_output)

    def _tensors(self):
        ts = []
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4)):
            ts += [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return ts + [self.fc1.weight, self.fc1.bias]

    def __del__(self):
        try:
            if self._native:
                N.lib().ap_m5_destroy(self._native)
        except Exception:
            pass

    def _handle(self):
        ts = self._tensors()
        dev = ts[0].device
        if dev.type != "cuda":
            raise N.NativeError("audiopure_amd M5 needs its parameters on a HIP device (.cuda()); no CPU path")
        key = (dev, tuple((t._version, t.data_ptr()) for t in ts))
        if key != self._key:
            lib = N.lib()
            if self._native:
                lib.ap_m5_destroy(self._native)
                self._native = None
            blob = torch.cat([t.detach().reshape(-1).float() for t in ts]).contiguous()
            h = C.c_void_p()
            N.check(lib.ap_m5_create(self.fc1.out_features, self.conv1.out_channels, self.conv1.kernel_size[0],
                                     self.conv1.stride[0], float(self.bn1.eps), N.ptr(blob), blob.numel(), N.stream(),
                                     C.byref(h)), "ap_m5_create")
            self._native, self._key = h, key
        return self._native

    @N.on_device
    def forward(self, x):
        if self.training:
            raise NotImplementedError("audiopure_amd M5: inference only (BatchNorm folded); call .eval()")
        if torch.is_grad_enabled() and x.requires_grad:
            return _M5InputGrad.apply(x, self)                   # white-box attack: dL/dx (parameters frozen)
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"expected [B,1,L], got {tuple(x.shape)}")
        h = self._handle()
        x = x.detach().float().contiguous()
        out = torch.empty((x.shape[0], self.fc1.out_features), device=x.device, dtype=torch.float32)
        if x.shape[0] == 0:                                      # empty batch: empty scores, like the torch modules
            return out
        N.check(N.lib().ap_m5_fwd(h, N.ptr(x), N.ptr(out), x.shape[0], x.shape[2], N.stream()), "ap_m5_fwd")
        return out
