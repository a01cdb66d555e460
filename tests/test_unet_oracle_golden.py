"""Pin the UNet / spectrogram-SDE oracle (oracle/unet_oracle.py) against outputs of the reference's own UNetModel and
continuous-beta RevVPSDE (tests/golden/golden_unet_v1.npz, made by tests/golden/make_golden_unet.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from synth_convnets import synth_init
from audiopure_amd.diffusion_models.improved_diffusion_unet import UNetModel, create_model, model_and_diffusion_defaults
from audiopure_amd.diffusion_models.improved_diffusion_sde import sde_step_table
from oracle import unet_oracle as U
from conftest import rel_err


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_unet_v1.npz"))


def _x():
    return torch.from_numpy(synth.uniform("specx", (2, 1, 32, 32), 3, -1.0, 1.0))


def mini_unet():
    return synth_init(UNetModel(in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(2, 4),
                                dropout=0.0, channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=True), 1)


def test_state_dict_keys_are_the_references(gold):
    assert list(mini_unet().state_dict().keys()) == list(gold["mini/keys"])
    full = create_model(**model_and_diffusion_defaults())
    assert list(full.state_dict().keys()) == list(gold["full/keys"])
    assert sum(p.numel() for p in full.parameters()) == 52538369


def test_mini_unet_oracle_matches_reference(gold):
    m = mini_unet()
    for t in (0, 37, 999):
        got = U.unet_forward(m, _x(), torch.tensor([float(t)] * 2))
        assert rel_err(got.numpy(), gold[f"mini/eps_t{t}"]) < 2e-6, t
    got = U.unet_forward(m, _x(), torch.tensor([5.0, 600.0]))
    assert rel_err(got.numpy(), gold["mini/eps_tmixed"]) < 2e-6


def test_full_unet_oracle_matches_reference(gold):
    full = synth_init(create_model(**model_and_diffusion_defaults()), 0)
    got = U.unet_forward(full, _x(), torch.tensor([37.0, 37.0]))
    assert rel_err(got.numpy(), gold["full/eps_t37"]) < 5e-6


def test_sde_drift_diffusion_match_reference(gold):
    m = mini_unet()
    xs = _x() * 0.8
    for tau in (0.0045, 0.02):
        tau_t = (1 - torch.tensor([1.0 - tau]))[0]          # the reference forms tau = 1 - s in float32
        f, g = U.sde_f_g(m, xs, tau_t)
        assert rel_err(f.reshape(2, -1).numpy(), gold[f"mini/sde_f_tau{tau}"]) < 5e-6
        np.testing.assert_allclose(np.full((2, 4), float(g), np.float32), gold[f"mini/sde_g_tau{tau}"], rtol=1e-6)


def test_step_table_equals_oracle_euler_coefficients():
    """The product's (ca, cb, cs) table is the Euler step x + f h + g sqrt(h) z of the oracle, term by term."""
    import math
    steps, grid = sde_step_table(5), U.sde_step_times(5)
    assert len(steps) == len(grid) == 5 and abs(float(grid[-1][1]) - (1e-3 - 1e-5)) < 2e-7
    # float32 time grid of the reference: the model timestep is floor((1 - s) * 1000) in float32, e.g. [4, 4, 3, 2, 1]
    assert [s_[0] for s_ in steps] == [float(int(g_[0] * np.float32(1000))) for g_ in grid]
    assert len(sde_step_table(30)) == 30 and len(sde_step_table(1)) == 1
    for (disc, ca, cb, cs), (tau, h) in zip(steps, grid):
        tau, h = float(tau), float(h)
        beta = 0.1 + tau * 19.9
        abar = math.exp(-0.5 * 19.9 * tau ** 2 - 0.1 * tau)
        assert disc == float(int(np.float32(tau) * np.float32(1000)))
        assert math.isclose(ca, 1 + 0.5 * beta * h, rel_tol=1e-6) and math.isclose(cs, math.sqrt(beta * h), rel_tol=1e-6)
        assert math.isclose(cb, -beta * h / math.sqrt(1 - abar), rel_tol=1e-5)


def test_ddpm_chain_oracle_matches_reference_gaussian_diffusion(gold):
    m = mini_unet()
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(5)]
    got = U.ddpm_spec_purify(m, img, 4, z)
    assert rel_err(got.numpy(), gold["mini/ddpm_t4"]) < 5e-6


def test_minimal_filtering_form_of_the_3x3_conv_is_the_conv():
    """The F(2,3)-along-W form the HIP library runs the UNet's 3 x 3 convolutions in (ap_conv_w3.hip), restated on the CPU with the
    kernel's operation order, equals nn.functional.conv2d at fp32 rounding for every map size the UNet has and for odd heights."""
    import torch
    import torch.nn.functional as F
    from oracle import unet_oracle as U
    g = torch.Generator().manual_seed(3)
    for (B, Cin, H, W, Cout) in [(2, 32, 32, 32, 16), (3, 16, 16, 16, 24), (2, 8, 8, 8, 8), (4, 8, 4, 4, 8), (1, 4, 5, 2, 3), (2, 4, 1, 6, 5)]:
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        got = U.conv3x3_minimal_filtering(x, w, b)
        plain = F.conv2d(x, w, b, padding=1)
        e_w = float((got.double() - ref).abs().max() / ref.abs().max())
        e_p = float((plain.double() - ref).abs().max() / ref.abs().max())
        assert e_w < 2e-6 and e_w < 4 * e_p + 2e-7, (B, Cin, H, W, Cout, e_w, e_p)
