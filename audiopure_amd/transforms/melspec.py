"""Native mel-dB front-ends.

``MelSpecDB`` replaces the eval scripts' ``Compose([torchaudio.transforms.MelSpectrogram(n_fft=2048,
hop_length=512, n_mels=n_mels, norm='slaney', pad_mode='constant', mel_scale='slaney'),
AmplitudeToDB(stype='power')])`` (adaptive_attack_eval.py:83-85).
``ToMelSpectrogramDB`` is the device restatement of the dataset-time ``ToSTFT`` +
``ToMelSpectrogramFromSTFT`` pair (transforms/transforms_stft.py:14-28,101-114; librosa
``power_to_db(ref=np.max)``, zero-padded centre frames — SURVEY.md section 8 a12')."""
import torch

from .. import _native as N


class _MelGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        with torch.no_grad():
            out = mod.forward(x.detach())
        ctx.n_mels = mod.n_mels
        ctx.save_for_backward(x.detach().float().contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        L = x.shape[-1]
        xf = x.reshape(-1, L)
        B = xf.shape[0]
        frames = 1 + L // 512
        gf = g.detach().float().reshape(B, ctx.n_mels, frames).contiguous()
        dx = torch.empty_like(xf)
        scratch = torch.empty((B, frames, 2048), device=x.device, dtype=torch.float32)
        N.check(N.lib().ap_melspec_db_bwd(N.ptr(xf), N.ptr(gf), N.ptr(dx), N.ptr(scratch), ctx.n_mels, B, L, N.stream()),
                "ap_melspec_db_bwd")
        return dx.reshape(x.shape), None


class MelSpecDB(torch.nn.Module):
    mode = 0

    def __init__(self, n_mels: int = 32):
        super().__init__()
        self.n_mels = n_mels

    @N.on_device
    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            if self.mode != 0:
                raise NotImplementedError("audiopure_amd ToMelSpectrogramDB: forward-only HIP path")
            return _MelGrad.apply(x, self)                       # white-box attack: d/dx by ap_melspec_db_bwd
        lead = x.shape[:-1]
        L = x.shape[-1]
        xf = x.detach().float().reshape(-1, L).contiguous()
        B = xf.shape[0]
        frames = 1 + L // 512
        out = torch.empty((B, self.n_mels, frames), device=xf.device, dtype=torch.float32)
        N.check(N.lib().ap_melspec_db(N.ptr(xf), N.ptr(out), self.n_mels, self.mode, B, L, N.stream()), "ap_melspec_db")
        return out.reshape(*lead, self.n_mels, frames)     # [B,1,L] -> [B,1,n_mels,frames] like torchaudio


class ToMelSpectrogramDB(MelSpecDB):
    mode = 1


class _MelHTKGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xf, n_mels):
        xc = xf.detach().float().contiguous()
        B, L = xc.shape
        out = torch.empty((B, n_mels, 1 + L // 200), device=xc.device, dtype=torch.float32)
        N.check(N.lib().ap_melspec_db_htk(N.ptr(xc), N.ptr(out), n_mels, B, L, N.stream()), "ap_melspec_db_htk")
        ctx.x, ctx.n_mels = xc, n_mels
        return out

    @staticmethod
    def backward(ctx, g):
        xc = ctx.x
        B, L = xc.shape
        g = g.detach().float().contiguous()
        dx = torch.empty_like(xc)
        scr = torch.empty(B * (1 + L // 200) * 400, device=xc.device, dtype=torch.float32)
        N.check(N.lib().ap_melspec_db_htk_bwd(N.ptr(xc), N.ptr(g), N.ptr(dx), N.ptr(scr), ctx.n_mels, B, L, N.stream()),
                "ap_melspec_db_htk_bwd")
        return dx, None


class MelSpecDBHTK(torch.nn.Module):
    """The KWS script's front-end (kws_adaptive_attack_eval.py:65-67): torchaudio ``MelSpectrogram(sample_rate=16000,
    n_mels)`` with its defaults (n_fft = win = 400, hop = 200, reflect padding, HTK mel scale, no filter norm) followed
    by ``AmplitudeToDB('power')``; clips of any length >= 201 samples."""

    def __init__(self, n_mels: int = 40):
        super().__init__()
        self.n_mels = n_mels

    @N.on_device
    def forward(self, x):
        lead = x.shape[:-1]
        L = x.shape[-1]
        if torch.is_grad_enabled() and x.requires_grad and x.numel() > 0:
            return _MelHTKGrad.apply(x.float().reshape(-1, L), self.n_mels).reshape(*lead, self.n_mels, 1 + L // 200)
        xf = x.detach().float().reshape(-1, L).contiguous()
        B = xf.shape[0]
        frames = 1 + L // 200
        out = torch.empty((B, self.n_mels, frames), device=xf.device, dtype=torch.float32)
        N.check(N.lib().ap_melspec_db_htk(N.ptr(xf), N.ptr(out), self.n_mels, B, L, N.stream()), "ap_melspec_db_htk")
        return out.reshape(*lead, self.n_mels, frames)
