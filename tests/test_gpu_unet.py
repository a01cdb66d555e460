"""GPU parity of the Improved-Diffusion UNet and the spectrogram purifier against the oracle and
the reference-generated golden vectors."""
import os
import types

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from audiopure_amd.audio_models.convnets import synth_init
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
from audiopure_amd.diffusion_models.improved_diffusion_sde import RevImprovedDiffusion
from conftest import rel_err
from test_unet_oracle_golden import mini_unet, _x

pytestmark = pytest.mark.gpu
TOL = 3e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_unet_v1.npz"))


def test_primitives_match_torch(dev):
    import torch.nn.functional as F
    from audiopure_amd import _native as N
    lib = N.lib()
    x = torch.from_numpy(synth.uniform("gnx", (3, 64, 8, 8), 1, -2, 2))
    g, b = torch.from_numpy(synth.uniform("gng", (64,), 1, 0.5, 1.5)), torch.from_numpy(synth.uniform("gnb", (64,), 1))
    ss = torch.from_numpy(synth.uniform("gns", (3, 128), 1))
    ref = F.group_norm(x, 32, g, b, 1e-5) * (1 + ss[:, :64, None, None]) + ss[:, 64:, None, None]
    ref = ref * torch.sigmoid(ref)
    y = torch.empty_like(x, device=dev)
    xd, gd, bd, sd_ = x.to(dev), g.to(dev), b.to(dev), ss.to(dev)      # keep the device copies alive across the launch
    N.check(lib.ap_groupnorm_nchw(N.ptr(xd), N.ptr(gd), N.ptr(bd), N.ptr(sd_), N.ptr(y), 3, 64, 64, 32, 1e-5, 2, N.stream()))
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 3e-6
    for ch, T, heads in ((16, 256, 2), (64, 64, 4), (32, 100, 1), (64, 256, 3)):     # the ch = 64 cases run on the MFMA
        qkv = torch.from_numpy(synth.uniform(f"qkv{ch}", (2, heads * 3 * ch, T), 1, -1.5, 1.5))
        q, k, v = torch.split(qkv.reshape(2 * heads, 3 * ch, T), ch, dim=1)
        w = torch.softmax(torch.einsum("bct,bcs->bts", q * ch ** -0.25, k * ch ** -0.25), dim=-1)
        ref = torch.einsum("bts,bcs->bct", w, v).reshape(2, heads * ch, T)
        out = torch.empty((2, heads * ch, T), device=dev)
        qd = qkv.to(dev)
        N.check(lib.ap_attention_qkv(N.ptr(qd), N.ptr(out), 2, heads * ch, T, heads, N.stream()))
        assert rel_err(out.cpu().numpy(), ref.numpy()) < 3e-6


def test_mini_unet_matches_reference_golden(dev, gold):
    m = mini_unet().to(dev)
    for t in (0, 37, 999):
        eps = m(_x().to(dev), torch.tensor([float(t)] * 2))
        assert rel_err(eps.cpu().numpy(), gold[f"mini/eps_t{t}"]) < TOL, t
    eps = m(_x().to(dev), torch.tensor([5.0, 600.0]))
    assert rel_err(eps.cpu().numpy(), gold["mini/eps_tmixed"]) < TOL


def test_full_unet_matches_reference_golden(dev, gold):
    full = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
    eps = full(_x().to(dev), torch.tensor([37.0, 37.0]))
    assert rel_err(eps.cpu().numpy(), gold["full/eps_t37"]) < TOL


def test_spec_purifier_matches_oracle(dev):
    from oracle import unet_oracle as U
    m = mini_unet()
    args = types.SimpleNamespace(t=4, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(5)]
    ref = U.spec_sde_purify(m, img, 4, z)
    rev = RevImprovedDiffusion.from_model(mini_unet().to(dev), args)
    rev.set_noise_source(z)
    got = rev(img.to(dev))
    assert got.shape == img.shape
    assert rel_err(got.cpu().numpy(), ref.numpy()) < 1e-4


def test_ddpm_spec_purifier_matches_reference_golden(dev, gold):
    from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(5)]
    dd = ImprovedDiffusionDDPM(mini_unet().to(dev), reverse_timestep=4)
    dd.set_noise_source(z)
    got = dd(img.to(dev))
    assert rel_err(got.cpu().numpy(), gold["mini/ddpm_t4"]) < 1e-4
