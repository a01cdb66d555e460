"""Pin the CPU oracle (oracle/diffwave_oracle.py) against golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only.

Tolerances: the oracle performs the same fp32 ops in the same order through the same
PyTorch CPU kernels, so schedule/embedding/fold are bit-exact and the network outputs
agree to a few fp32 ulps of the largest activation (oneDNN may pick a different conv
blocking for the functional call than for the nn.Module, hence not asserted bit-exact).
"""
import numpy as np
import pytest
import torch

from audiopure_amd import synth
from oracle import diffwave_oracle as O
from conftest import rel_err

TOL_NET = 2e-6  # relative to max |reference|


def slices(h):
    """Same windows as tests/golden/make_golden.py::slices."""
    L = h.shape[-1]
    w = min(64, L)
    c = max(0, L // 2 - w // 2)
    return torch.cat([h[:, :4, :w], h[:, :4, c:c + w], h[:, :4, L - w:]], dim=-1).numpy()


@pytest.fixture(scope="module")
def dh():
    return O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG)


@pytest.fixture(scope="module")
def mini():
    cfg = synth.mini_wavenet_config(64, 12, 12)
    return cfg, O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))


@pytest.fixture(scope="module")
def full():
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    return cfg, O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))


def test_schedule_bit_exact(golden, dh):
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        assert np.array_equal(dh[k].numpy(), golden[f"sched/{k}"]), k
    # SURVEY.md section 8 a2 quotes these from the reference
    np.testing.assert_allclose(dh["Alpha_bar"][:5].numpy(),
                               [0.99989998, 0.99970001, 0.99940014, 0.99900037, 0.99850082], rtol=0, atol=1e-8)
    np.testing.assert_allclose(dh["Sigma"][:5].numpy(),
                               [0.01, 0.00816578, 0.01224867, 0.01549301, 0.01825905], rtol=0, atol=1e-8)


def test_step_embedding_bit_exact(golden):
    e = O.step_embedding(torch.from_numpy(golden["embed/steps"]), 128)
    assert np.array_equal(e.numpy(), golden["embed/out"])


def test_weight_norm_fold(golden, mini):
    _, w = mini
    p = "residual_layer.residual_blocks.0"
    assert rel_err(w[p + ".dilated_conv_layer.conv.weight"].numpy(), golden["mini/fold/dil0"]) < 2e-7
    assert rel_err(w[p + ".res_conv.weight"].numpy(), golden["mini/fold/res0"]) < 2e-7


@pytest.mark.parametrize("L", [16000, 4133, 1000])
def test_mini_eps(golden, mini, L):
    cfg, w = mini
    x = torch.from_numpy(synth.waveforms(2, L, seed=7)) * 2.0
    taps = {}
    with torch.no_grad():
        eps = O.eps_net(w, cfg, x, 3.0 * torch.ones(2, 1), taps)
    assert rel_err(eps.numpy(), golden[f"mini/L{L}/eps"]) < TOL_NET
    if L == 16000:
        for n in range(cfg["num_res_layers"]):
            assert rel_err(slices(taps[f"h{n}"]), golden[f"mini/L{L}/h{n}_slices"]) < TOL_NET, n
            s = taps[f"h{n}"].double()
            ref = golden[f"mini/L{L}/h{n}_sum"]
            assert abs(s.abs().sum().item() - ref[1]) / ref[1] < 1e-6


def test_mini_ddpm_chain(golden, mini, dh):
    cfg, w = mini
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=7))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=7)) for d in range(3)]
    x = O.ddpm_purify(w, cfg, dh, x0, 3, z)
    assert rel_err(x.numpy(), golden["mini/ddpm_n3"]) < TOL_NET


def test_sde_drift_diffusion_match_reference(golden, mini):
    cfg, w = mini
    tb = O.sde_tables()
    assert np.array_equal(tb["discrete_betas"].numpy(), golden["mini/sde/discrete_betas"])
    assert np.array_equal(tb["alphas_cumprod"].numpy(), golden["mini/sde/alphas_cumprod"])
    xs = (torch.from_numpy(synth.waveforms(2, 16000, seed=7)) * 1.5).view(2, -1)
    for k in (0, 4):
        with torch.no_grad():
            f, g = O.sde_f_g(w, cfg, tb, xs.clone(), k)
        assert rel_err(f.numpy(), golden[f"mini/sde/f_k{k}"]) < TOL_NET
        gg = float(g) if not torch.is_tensor(g) else float(g)
        np.testing.assert_allclose(np.full((2, 4), gg, np.float32), golden[f"mini/sde/g_k{k}"], rtol=1e-6, atol=0)


def test_sde_euler_is_first_order_ddpm(mini, dh):
    """the Euler step is the first-order expansion of the DDPM step with the same sigma."""
    cfg, w = mini
    tb = O.sde_tables()
    x0 = torch.from_numpy(synth.waveforms(2, 2000, seed=3))
    z = [torch.from_numpy(synth.noise(d, 2, 2000, seed=3)) for d in range(4)]
    a = O.ddpm_purify(w, cfg, dh, x0, 3, z)
    b = O.sde_purify(w, cfg, tb, x0, 3, z)
    assert rel_err(b.numpy(), a.numpy()) < 1e-4


@pytest.mark.parametrize("n", [1, 2, 5])
def test_full_config_ddpm_and_m5(golden, full, dh, n):
    cfg, w = full
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(n)]
    x, lp = O.purify_and_classify(w, cfg, dh, synth.m5_state_dict(10), x0, n, z)
    assert rel_err(x.numpy(), golden[f"full/ddpm_n{n}/x"]) < TOL_NET * 5
    np.testing.assert_allclose(lp.numpy(), golden[f"full/ddpm_n{n}/m5_logprobs"], rtol=0, atol=2e-5)
    if n == 1:
        np.testing.assert_allclose(lp.numpy(), golden["full/acoustic_system_n1"], rtol=0, atol=2e-5)


def test_full_config_helpers(golden, full, dh):
    cfg, w = full
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    with torch.no_grad():
        eps = O.eps_net(w, cfg, x0, 4.0 * torch.ones(2, 1))
    assert rel_err(eps.numpy(), golden["full/eps_t4"]) < TOL_NET * 5
    for t in (1, 25):
        assert rel_err(O.one_shot_denoise(w, cfg, dh, x0, t).numpy(), golden[f"full/one_shot_t{t}"]) < TOL_NET * 5
    assert rel_err(O.two_shot_denoise(w, cfg, dh, x0, 25).numpy(), golden["full/two_shot_t25"]) < TOL_NET * 5
    lp = O.m5_forward(synth.m5_state_dict(10), x0)
    np.testing.assert_allclose(lp.numpy(), golden["full/acoustic_system_nodefense"], rtol=0, atol=2e-5)


def test_mel_filterbank_closed_form():
    """parity unpinned for torchaudio: check the restated filterbank against its closed-form properties."""
    fb = O.mel_filterbank()
    assert fb.shape == (1025, 32) and (fb >= 0).all()
    # slaney area normalisation: each triangle integrates to ~1 over Hz (bin width 8000/1024)
    area = fb.sum(0) * (8000.0 / 1024)
    np.testing.assert_allclose(area, 1.0, atol=0.08)
    # centre frequencies are increasing and the first is at 3 mel-steps of 200/3 Hz spacing region
    peaks = fb.argmax(0)
    assert (np.diff(peaks) > 0).all()


def test_melspec_matches_direct_dft():
    """torch.stft-based restatement vs an explicit float64 DFT of the zero-padded, hann-windowed frames."""
    x = torch.from_numpy(synth.waveforms(1, 16000, seed=5))
    db = O.melspec_db(x)
    assert db.shape == (1, 1, 32, 32)
    xp = np.zeros(16000 + 2048, np.float64)
    xp[1024:1024 + 16000] = x.numpy()[0, 0]
    n = np.arange(2048)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / 2048)
    fb = O.mel_filterbank().astype(np.float64)
    for fr in (0, 7, 31):
        seg = xp[fr * 512: fr * 512 + 2048] * win
        P = np.abs(np.fft.rfft(seg)) ** 2
        ref = 10 * np.log10(np.maximum(fb.T @ P, 1e-10))
        np.testing.assert_allclose(db[0, 0, :, fr].numpy(), ref, atol=2e-3)


# ---- round-2 vectors (tests/golden/make_golden_v2.py): _diffusion, _reverse, fast_reverse of the reference's DiffWave ----
@pytest.fixture(scope="module")
def golden2():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v2.npz"))


def _nl(n, seed):
    return [torch.from_numpy(synth.noise(d, 2, 16000, seed=seed)) for d in range(n)]


def test_oracle_diffusion_and_reverse_match_reference(golden2, dh, mini):
    cfg, w = mini
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=21))
    xt = O.q_sample(dh, x0, 20, _nl(1, 21)[0])
    assert rel_err(xt.numpy(), golden2["mini/diffusion_t20"]) < 1e-6
    xr = O.ddpm_reverse(w, cfg, dh, x0 * 1.2, 4, _nl(3, 22))
    assert rel_err(xr.numpy(), golden2["mini/reverse_n4"]) < 1e-5


@pytest.mark.parametrize("ts", [20, 7])
def test_oracle_fast_reverse_matches_reference(golden2, dh, mini, ts):
    cfg, w = mini
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=21))
    xr = O.fast_reverse(w, cfg, dh, x0 * 1.1, ts, _nl(3, 23))
    assert rel_err(xr.numpy(), golden2[f"mini/fast_reverse_t{ts}"]) < 1e-5


# ---- round 5: the F(2,3) minimal-filtering form of the dilated conv (what the AP_PREC_F32 HIP block computes) ----
def test_winograd_form_of_the_dilated_conv_is_the_conv(mini):
    """Pair structure, padding and the four transformed products restate WaveNet.py:87 for every dilation, d >= L included."""
    cfg, w = mini
    p = "residual_layer.residual_blocks.0.dilated_conv_layer.conv"
    W, b = w[p + ".weight"], w[p + ".bias"]
    g = torch.Generator().manual_seed(5)
    for L, d in [(1000, 1), (1000, 2), (257, 8), (1000, 64), (1000, 512), (130, 256), (77, 2048), (1, 1), (2, 1), (3, 2)]:
        u = torch.randn(2, W.shape[1], L, generator=g)
        ref = torch.nn.functional.conv1d(u, W, b, dilation=d, padding=d)
        got = O.winograd_dilated_conv(u, W, b, d)
        assert rel_err(got.numpy(), ref.numpy()) < 2e-6, (L, d)


def test_winograd_form_holds_the_reference_golden_at_the_fp32_tolerances(golden, full, mini, dh):
    """with the dilated conv in F(2,3) form the reference's vectors are met at the tolerances the GPU
    tests use for the exact-fp32 block (tests/test_gpu_parity.py TOL_EVAL 2e-5, TOL_CHAIN 1e-4) -- measured 2.8e-6 / 2.8e-7."""
    cfg, w = full
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    with torch.no_grad():
        eps = O.eps_net(w, cfg, x0, 4.0 * torch.ones(2, 1), winograd=True)
    assert rel_err(eps.numpy(), golden["full/eps_t4"]) < 2e-5 / 4
    z = [torch.from_numpy(synth.noise(0, 2, 16000, seed=1234))]
    x = O.ddpm_purify(w, cfg, dh, x0, 1, z, winograd=True)
    assert rel_err(x.numpy(), golden["full/ddpm_n1/x"]) < 1e-4 / 4
    cfgm, wm = mini
    xm = torch.from_numpy(synth.waveforms(2, 4133, seed=7)) * 2.0
    with torch.no_grad():
        epsm = O.eps_net(wm, cfgm, xm, 3.0 * torch.ones(2, 1), winograd=True)
    assert rel_err(epsm.numpy(), golden["mini/L4133/eps"]) < TOL_NET * 2


def test_oracle_one_shot_votes_match_the_reference_smooth_predict(dh):
    """The oracle's one-shot denoise + M5 on the certification loop's noisy copies (certified_robust.py:45-55: x + sigma z, scaled by
    sqrt(alpha_bar*), one_shot_denoise at t*, classifier) against the scores the REFERENCE's RobustCertificate.smooth_predict
    produced on the same Philox draws (tests/golden/make_golden_f2.py) -- pins the oracle, oracle/philox.py's keying and the
    t* rule together.  First 8 of the 300 samples (CPU suite budget)."""
    import os
    from oracle.philox import philox_normal
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_f2_v1.npz"))
    cfg = synth.mini_wavenet_config(64, 12, 12)
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 1))
    sigma, n = 0.25, 8
    ab_star = 1 / (1 + sigma ** 2)
    t_star = int(torch.abs(dh["Alpha_bar"] - ab_star).min(0, keepdim=True)[1].item()) + 1      # certified_robust.py:99-107
    assert t_star == int(g["cert/t_star"][0])
    x = torch.from_numpy(synth.waveforms(1, 16000, seed=5))[0:1]
    z = torch.from_numpy(philox_normal(77, 0, 0, n, 16000)).reshape(n, 1, 16000)
    x_in = ab_star ** 0.5 * (x.repeat(n, 1, 1) + sigma * z)
    lp = O.m5_forward(synth.m5_state_dict(10, seed=11), O.one_shot_denoise(w, cfg, dh, x_in, t_star)).numpy()
    assert np.abs(lp - g["cert/scores"][:n]).max() < 1e-4
    assert (lp.argmax(1) == g["cert/pred"][:n]).all()
    assert int(g["cert/counts"].sum()) == 300 and (np.bincount(g["cert/pred"], minlength=10) == g["cert/counts"]).all()
