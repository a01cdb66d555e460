"""ORACLE — test infrastructure only.  PyTorch-CPU restatement of the Improved-Diffusion UNet forward
(improved_diffusion/unet.py:462-491 and the blocks it calls) and of the continuous-beta VP-SDE Euler chain of
``RevImprovedDiffusion`` (diffusion_models/improved_diffusion_sde.py:48-137,173-221).  It walks a parameter-holder
``UNetModel`` (same attribute tree / state-dict keys as the reference) with plain torch ops.
Pinned by tests/test_unet_oracle_golden.py against outputs of the reference's own UNetModel / RevVPSDE
(tests/golden/make_golden_unet.py).  The torchsde Euler loop itself is restated (parity unpinned, as for DiffWave)."""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):                     # nn.py:103-121
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(gn, x):
    return F.group_norm(x.float(), gn.num_groups, gn.weight, gn.bias, gn.eps)      # GroupNorm32, nn.py:17-19


def _silu(x):
    return x * torch.sigmoid(x)                                        # nn.py:12-14


def _resblock(rb, x, emb):                                             # unet.py:180-194
    h = rb.in_layers[2](_silu(_gn(rb.in_layers[0], x)))
    emb_out = rb.emb_layers[1](_silu(emb))[..., None, None]
    scale, shift = torch.chunk(emb_out, 2, dim=1)
    h = _gn(rb.out_layers[0], h) * (1 + scale) + shift
    h = rb.out_layers[3](_silu(h))
    return rb.skip_connection(x) + h


def _attention(ab, x):                                                 # unet.py:226-252
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = ab.qkv(_gn(ab.norm, xf))
    qkv = qkv.reshape(b * ab.num_heads, -1, qkv.shape[2])
    ch = qkv.shape[1] // 3
    q, k, v = torch.split(qkv, ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale).float(), dim=-1)
    h = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, qkv.shape[2])
    return (xf + ab.proj_out(h)).reshape(b, c, *spatial)


def _run(seq, h, emb):
    for layer in seq:
        name = type(layer).__name__
        if name == "ResBlock":
            h = _resblock(layer, h, emb)
        elif name == "AttentionBlock":
            h = _attention(layer, h)
        elif name == "Downsample":
            h = layer.op(h)
        elif name == "Upsample":
            h = layer.conv(F.interpolate(h, scale_factor=2, mode="nearest"))
        else:
            h = layer(h)
    return h


def unet_forward(model, x, timesteps, grad=False):
    """grad=True keeps the autograd graph (tests take the reference input gradient from it)."""
    with torch.set_grad_enabled(grad):
        emb = model.time_embed[2](_silu(model.time_embed[0](timestep_embedding(timesteps, model.model_channels))))
        hs, h = [], x.float()
        for blk in model.input_blocks:
            h = _run(blk, h, emb)
            hs.append(h)
        h = _run(model.middle_block, h, emb)
        for blk in model.output_blocks:
            h = _run(blk, torch.cat([h, hs.pop()], dim=1), emb)
        return model.out[2](_silu(_gn(model.out[0], h)))


# ---- continuous-beta reverse VP-SDE (improved_diffusion_sde.py:48-137) -------------------------------------------
MEL_UPPER_BOUND, MEL_LOWER_BOUND = 38.22, -100.0                       # sc09_spectrogram_dataset.py:62-63


def melspec_standardize(x):
    return 2 * (x - MEL_LOWER_BOUND) / (MEL_UPPER_BOUND - MEL_LOWER_BOUND) - 1          # :65-72


def melspec_inv_standardize(x):
    return (x + 1) * (MEL_UPPER_BOUND - MEL_LOWER_BOUND) / 2 + MEL_LOWER_BOUND          # :74-81


def sde_f_g(model, x, tau, beta_min=0.1, beta_max=20.0, N=1000, grad=False):
    """f, g of the reverse SDE in torchsde time at reference time tau = 1 - s (scalar float tensor)."""
    beta_t = beta_min + tau * (beta_max - beta_min)                                      # :86
    drift = -0.5 * beta_t * x
    diffusion = torch.sqrt(beta_t)
    disc = (tau.float() * N).long()                                                      # :82-83
    eps = unet_forward(model, x, disc.expand(x.shape[0]), grad)                          # :106
    ac = torch.exp(-0.5 * (beta_max - beta_min) * tau ** 2 - beta_min * tau)             # :74
    score = (-1.0 / torch.sqrt(1.0 - ac)) * eps                                          # :75,:111
    drift = drift - diffusion ** 2 * score                                               # :116
    return -drift, diffusion                                                             # :126,:136


def sde_step_times(t_star: int, dt: float = 1e-3):
    """torchsde's fixed-step Euler grid on ts = linspace(1 - t/1000, 1 - 1e-5, 2) (improved_diffusion_sde.py:194-196),
    restated with the float32 time arithmetic the reference runs in (ts is a float32 tensor; curr_t + dt, 1 - t and the
    model timestep floor((1 - t) * 1000) are all float32): t steps, the last one shortened by 1e-5; which integer
    timestep a step lands on is decided by that float32 rounding (t = 5 gives 4,4,3,2,1; t = 12 gives 12..1).
    Returns [(tau_i, h_i)] as float32."""
    import numpy as np
    t0, t1, d = np.float32(1 - t_star * 1. / 1000), np.float32(1 - 1e-5), np.float32(dt)
    out, s = [], t0
    while s < t1:
        nxt = min(np.float32(s + d), t1)
        out.append((np.float32(1) - s, np.float32(nxt - s)))
        s = nxt
    return out


def spec_sde_purify(model, img_db, t_star: int, noises, grad=False):
    """RevImprovedDiffusion.image_editing_sample, sample_step = 1 (:173-221): mel-dB in, mel-dB out."""
    with torch.set_grad_enabled(grad):
        x0 = melspec_standardize(img_db.float())
        betas = torch.linspace(0.1 / 1000, 20.0 / 1000, 1000)                            # RevVPSDE defaults, :49,:66
        a = (1 - betas).cumprod(dim=0)
        x = x0 * a[t_star - 1].sqrt() + noises[0] * (1.0 - a[t_star - 1]).sqrt()         # :188-189
        for i, (tau, h) in enumerate(sde_step_times(t_star)):
            f, g = sde_f_g(model, x, torch.tensor(float(tau), dtype=torch.float32), grad=grad)
            x = x + f * float(h) + g * math.sqrt(float(h)) * noises[1 + i]
        return melspec_inv_standardize(x)


# ---- GaussianDiffusion DDPM chain (gaussian_diffusion.py:119-163,188-206,232-387), eps-prediction, FIXED_LARGE ------
def ddpm_spec_purify(model, img_db, t_star: int, noises, T=200, clip=True):
    import numpy as np
    betas = np.linspace(0.0001, 0.02, T, dtype=np.float64)
    ac = np.cumprod(1.0 - betas)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
    coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)
    logvar = np.log(np.append(post_var[1], betas[1:]))
    f = lambda a, i: torch.tensor(a[i]).float()          # _extract_into_tensor: float64 table -> .float()
    with torch.no_grad():
        x0 = melspec_standardize(img_db.float())
        x = f(np.sqrt(ac), t_star - 1) * x0 + f(np.sqrt(1.0 - ac), t_star - 1) * noises[0]
        k = 1
        for i in range(t_star - 1, -1, -1):
            eps = unet_forward(model, x, torch.full((x.shape[0],), float(i)))
            px0 = f(np.sqrt(1.0 / ac), i) * x - f(np.sqrt(1.0 / ac - 1), i) * eps
            if clip:
                px0 = px0.clamp(-1, 1)
            mean = f(coef1, i) * px0 + f(coef2, i) * x
            if i > 0:
                x = mean + torch.exp(0.5 * f(logvar, i)) * noises[k]
                k += 1
            else:
                x = mean
        return melspec_inv_standardize(x)


def conv3x3_minimal_filtering(x, w, b=None):
    """nn.Conv2d(k = 3, stride 1, padding 1) of improved_diffusion/unet.py:60-104,150-197 in the F(2,3)-along-W form the HIP conv
    kernel uses for these layers (audiopure_amd/csrc/ap_conv_w3.hip), same operation order: transformed weights computed in double
    and rounded to fp32 once, input differences in fp32, fp32 products summed per product over (ky, ci), then
    out[x0] = (m1 + m2) + m3, out[x0 + 1] = (m2 - m3) + m4'.  x: [B, Cin, H, W] with W even."""
    B, Cin, H, W = x.shape
    w64 = w.double()
    G = [w64[..., 0], (w64[..., 0] + w64[..., 1] + w64[..., 2]) / 2, (w64[..., 0] - w64[..., 1] + w64[..., 2]) / 2, w64[..., 2]]
    G = [g.float() for g in G]                                           # [Cout, Cin, 3 (ky)]
    xp = F.pad(x, (1, 1, 1, 1))                                          # zero padding
    rows = torch.stack([xp[:, :, ky:ky + H, :] for ky in range(3)], 2)   # [B, Cin, ky, H, W + 2]
    d0, d1, d2, d3 = (rows[..., k:k + W:2] for k in range(4))            # columns x0 - 1 .. x0 + 2 of every pair (x0 even)
    m1 = torch.einsum("ock,bckhp->bohp", G[0], d0 - d2)
    m2 = torch.einsum("ock,bckhp->bohp", G[1], d1 + d2)
    m3 = torch.einsum("ock,bckhp->bohp", G[2], d2 - d1)
    m4 = torch.einsum("ock,bckhp->bohp", G[3], d3 - d1)
    out = torch.empty(B, w.shape[0], H, W, dtype=x.dtype)
    out[..., 0::2] = (m1 + m2) + m3
    out[..., 1::2] = (m2 - m3) + m4
    return out if b is None else out + b.view(1, -1, 1, 1)
