"""Host-side schedule / embedding helpers with the reference's names and semantics
(diffusion_models/DiffWave_Unconditional/util.py:68-123), minus its hard-coded ``.cuda()``.

These run once at construction on the host; the per-step device work is in the HIP library.
"""
import numpy as np
import torch


def calc_diffusion_hyperparams(T, beta_0, beta_T):
    """Same sequential fp32 products as the reference (util.py:111-118) -> dict of CPU tensors
    with keys T, Beta, Alpha, Alpha_bar, Sigma (util.py:120-122)."""
    Beta = torch.linspace(beta_0, beta_T, T)
    Alpha = 1 - Beta
    Alpha_bar = Alpha + 0
    Beta_tilde = Beta + 0
    for t in range(1, T):
        Alpha_bar[t] *= Alpha_bar[t - 1]
        Beta_tilde[t] *= (1 - Alpha_bar[t - 1]) / (1 - Alpha_bar[t])
    Sigma = torch.sqrt(Beta_tilde)
    return {"T": T, "Beta": Beta, "Alpha": Alpha, "Alpha_bar": Alpha_bar, "Sigma": Sigma}


def embedding_frequencies(diffusion_step_embed_dim_in):
    """exp(-j ln(1e4)/(half-1)), j < half, computed exactly as util.py:86-88 (fp32 torch.exp)."""
    assert diffusion_step_embed_dim_in % 2 == 0
    half_dim = diffusion_step_embed_dim_in // 2
    _embed = np.log(10000) / (half_dim - 1)
    return torch.exp(torch.arange(half_dim) * -_embed)


def calc_diffusion_step_embedding(diffusion_steps, diffusion_step_embed_dim_in):
    """[B,1] steps -> [B, dim] sin/cos embedding (util.py:68-93); host/torch helper kept for API parity —
    the network itself computes this on device inside ap_embed."""
    _embed = embedding_frequencies(diffusion_step_embed_dim_in).to(diffusion_steps.device)
    _embed = diffusion_steps * _embed
    return torch.cat((torch.sin(_embed), torch.cos(_embed)), 1)


def std_normal(size, device=None):
    return torch.normal(0, 1, size=size, device=device)


def sampling(net, size, diffusion_hyperparams, noise_source=None):
    """Unconditional generation, the complete reverse process p(x_0 | x_T) (util.py:126-158; inference.py:73): the same
    chain of links the purifier runs, started from x_T ~ N(0, 1) with all T steps -- one native chain call per
    workspace-sized batch.  ``noise_source``: as ``DiffWave.set_noise_source`` (draw 0 is x_T)."""
    from ..diffwave_ddpm import DiffWave
    _dh = diffusion_hyperparams
    T = int(_dh["T"])
    assert len(_dh["Alpha"]) == T and len(_dh["Alpha_bar"]) == T and len(_dh["Sigma"]) == T and len(size) == 3
    print('begin sampling, total number of reverse steps = %s' % T)
    dw = DiffWave(model=net, diffusion_hyperparams=_dh, reverse_timestep=T)
    dw.set_noise_source(noise_source)
    dev = next(net.parameters()).device
    zeros = torch.zeros(tuple(size), device=dev)
    with torch.no_grad():                                   # x_T = 0 * x + 1 * z_0, then the T links
        return dw._chain(zeros, dw._ddpm_steps(T), 0.0, 1.0, n_draws=T)

