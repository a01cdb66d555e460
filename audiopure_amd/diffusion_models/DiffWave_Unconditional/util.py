"""Host-side schedule / embedding helpers with the reference's names and semantics
(diffusion_models/DiffWave_Unconditional/util.py:68-123), minus its hard-coded ``.cuda()``.

These run once at construction on the host; the per-step device work is in the HIP library.
"""
import numpy as np
import torch


def calc_diffusion_hyperparams(T, beta_0, beta_T):
    """The DDPM schedule tables (reference: util.py:96-123) -> dict of CPU tensors with keys T, Beta, Alpha, Alpha_bar, Sigma.
    alpha_bar is a running product taken one fp32 multiplication at a time (NOT cumprod: the two differ in the last bit from
    t ~ 20 on, and the chain's coefficients are read from these tables), beta~_t = beta_t (1 - abar_{t-1}) / (1 - abar_t) with the
    quotient formed first, beta~_0 = beta_0.  IEEE fp32 scalars throughout: bit-equal to the reference's tables
    (tests/test_host_logic_cpu.py against tests/golden sched/*)."""
    one = np.float32(1.0)
    beta = torch.linspace(beta_0, beta_T, T)
    b = beta.numpy()
    alpha = (one - b).astype(np.float32)
    alpha_bar, beta_tilde = np.empty(T, np.float32), np.empty(T, np.float32)
    running = one
    for t in range(T):
        before = running
        running = alpha[t] if t == 0 else np.float32(alpha[t] * before)
        alpha_bar[t] = running
        beta_tilde[t] = b[t] if t == 0 else np.float32(b[t] * np.float32(np.float32(one - before) / np.float32(one - running)))
    return {"T": T, "Beta": beta, "Alpha": torch.from_numpy(alpha), "Alpha_bar": torch.from_numpy(alpha_bar),
            "Sigma": torch.sqrt(torch.from_numpy(beta_tilde))}


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

