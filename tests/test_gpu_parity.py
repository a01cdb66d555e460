"""GPU parity tests: the HIP path (through the C-ABI, via the reference-shaped Python classes) against
the CPU oracle on the same seeded inputs and against the committed reference-generated golden vectors.

Stated fp32 tolerance: the kernels compute exact fp32 products (v_mfma_f32_32x32x2_f32) with a
different summation order than oneDNN, so one epsilon-evaluation agrees to ~1e-6 of max|eps|;
asserted 2e-5 per evaluation and 1e-4 after a 5-step chain (1e-4 * max|x|).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_EVAL = 2e-5
TOL_CHAIN = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch.device("cuda:0")


def _oracle():
    from oracle import diffwave_oracle as O
    return O


def _net(cfg, dev, seed=0):
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    sd = synth.wavenet_state_dict(cfg, seed)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.to(dev), sd


@pytest.fixture(scope="module")
def mini(dev):
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net, sd = _net(cfg, dev)
    return cfg, net, _oracle().fold_state_dict(sd)


@pytest.fixture(scope="module")
def full(dev):
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net, sd = _net(cfg, dev)
    return cfg, net, sd


@pytest.fixture(scope="module")
def dh():
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    return calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)


def test_native_library_loaded():
    from audiopure_amd import _native
    lib = _native.lib()
    assert lib.ap_version() >= 100


def test_weight_norm_fold_matches_oracle(mini, dev):
    from audiopure_amd import _native as N
    cfg, net, w = mini
    eng = net.engine()
    Cc = cfg["res_channels"]
    for which, key, n in ((0, "dilated_conv_layer.conv.weight", 2 * Cc * Cc * 3), (1, "res_conv.weight", Cc * Cc),
                          (2, "skip_conv.weight", Cc * Cc)):
        out = torch.empty(n, device=dev)
        N.check(eng.lib.ap_ctx_get_folded(eng.ctx, which, 5, N.ptr(out), n, N.stream()))
        ref = w[f"residual_layer.residual_blocks.5.{key}"].reshape(-1)
        assert rel_err(out.cpu().numpy(), ref.numpy()) < 5e-7


@pytest.mark.parametrize("C_,L,layer", [(64, 16000, 0), (64, 16000, 11), (64, 1000, 11), (64, 4133, 6), (64, 130, 3),
                                        (128, 2500, 9), (256, 1500, 2), (256, 2048, 10)])
def test_resblock_matches_oracle(dev, C_, L, layer):
    """One fused Residual_block.forward (WaveNet.py:75-97) vs the oracle, incl. ragged tiles and d >= L."""
    from audiopure_amd import _native as N
    O = _oracle()
    cfg = synth.mini_wavenet_config(C_, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    w = O.fold_state_dict(sd)
    eng = net.engine()
    B = 2
    h = torch.from_numpy(synth.uniform(f"h/{C_}/{L}", (B, C_, L), 1, -1.5, 1.5))
    skip0 = torch.from_numpy(synth.uniform(f"s/{C_}/{L}", (B, C_, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (B, 512), 1, -1.0, 1.0))
    emb[1] = emb[0]
    with torch.no_grad():
        p = f"residual_layer.residual_blocks.{layer}"
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        h_ref, s_ref = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
    hd, sk = h.to(dev), skip0.to(dev).clone()
    hout = torch.empty_like(hd)
    pt = part_t.to(dev).contiguous()
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
    assert rel_err(hout.cpu().numpy(), h_ref.numpy()) < 5e-6
    assert rel_err(sk.cpu().numpy(), (skip0 + s_ref).numpy()) < 5e-6
    sk2 = torch.full_like(sk, 7.0)
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk2), 0, B, L, N.stream()))
    assert rel_err(sk2.cpu().numpy(), s_ref.numpy()) < 5e-6


@pytest.mark.parametrize("L,layer", [(16000, 0), (16000, 1), (16000, 4), (16000, 5), (16000, 6), (16000, 7), (16000, 11), (4133, 8),
                                     (4133, 11), (1000, 11), (130, 3), (130, 9), (77, 0), (5, 1), (1, 0), (2, 0), (3, 1), (63, 5),
                                     (64, 5), (65, 5), (16001, 9), (333, 6)])
def test_minimal_filtering_resblock_matches_oracle(dev, L, layer):
    """The shipped-shape fp32 block (C = S = 256) in its F(2,3) form (ap_resblock_f32w.hip) vs the oracle's direct-form
    Residual_block.forward (WaveNet.py:75-97) at the direct kernel's tolerance: every dilation class (d < 32, = 32, > 32), d >= L,
    ragged and tiny clips, both skip modes -- and the direct form of the same context (ap_ctx_set_f32_form) beside it."""
    from audiopure_amd import _native as N
    O = _oracle()
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    w = O.fold_state_dict(sd)
    eng = net.engine()
    lib = eng.lib
    assert lib.ap_ctx_get_f32_form(eng.ctx) == 1
    B = 3
    h = torch.from_numpy(synth.uniform(f"h/256/{L}", (B, 256, L), 1, -1.5, 1.5))
    skip0 = torch.from_numpy(synth.uniform(f"s/256/{L}", (B, 256, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    with torch.no_grad():
        p = f"residual_layer.residual_blocks.{layer}"
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        h_ref, s_ref = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
    hd, pt = h.to(dev), part_t.to(dev).contiguous()
    outs = {}
    try:
        for form in (1, 0):
            N.check(lib.ap_ctx_set_f32_form(eng.ctx, form))
            sk = skip0.to(dev).clone()
            hout = torch.full_like(hd, 3.0)
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
            sk2 = torch.full_like(sk, 7.0)
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk2), 0, B, L, N.stream()))
            assert rel_err(hout.cpu().numpy(), h_ref.numpy()) < 5e-6, form
            assert rel_err(sk.cpu().numpy(), (skip0 + s_ref).numpy()) < 5e-6, form
            assert rel_err(sk2.cpu().numpy(), s_ref.numpy()) < 5e-6, form
            outs[form] = hout.cpu().numpy()
    finally:
        N.check(lib.ap_ctx_set_f32_form(eng.ctx, 1))
    assert rel_err(outs[1], outs[0]) < 3e-6


def test_embed_matches_oracle(mini, dev):
    from audiopure_amd import _native as N
    O = _oracle()
    cfg, net, w = mini
    eng = net.engine()
    NL, Cc = cfg["num_res_layers"], cfg["res_channels"]
    for step in (0.0, 4.0, 199.0):
        buf = torch.empty(NL * Cc + 512, device=dev)
        N.check(eng.lib.ap_embed(eng.ctx, step, N.ptr(buf), N.stream()))
        with torch.no_grad():
            e = O.step_embedding(torch.tensor([[step]]), 128)
            e = O._swish(torch.nn.functional.linear(e, w["residual_layer.fc_t1.weight"], w["residual_layer.fc_t1.bias"]))
            e = O._swish(torch.nn.functional.linear(e, w["residual_layer.fc_t2.weight"], w["residual_layer.fc_t2.bias"]))
            ref = torch.cat([torch.nn.functional.linear(e, w[f"residual_layer.residual_blocks.{n}.fc_t.weight"],
                                                        w[f"residual_layer.residual_blocks.{n}.fc_t.bias"]).reshape(-1)
                             for n in range(NL)])
        got = buf[:NL * Cc].cpu()
        assert (got - ref).abs().max().item() < 2e-5, step


@pytest.mark.parametrize("L", [16000, 4133, 1000])
def test_mini_eps_matches_reference_golden(golden, mini, dev, L):
    cfg, net, _ = mini
    x = torch.from_numpy(synth.waveforms(2, L, seed=7)) * 2.0
    with torch.no_grad():
        eps = net((x.to(dev), 3.0 * torch.ones(2, 1, device=dev)))
    assert rel_err(eps.cpu().numpy(), golden[f"mini/L{L}/eps"]) < TOL_EVAL


def test_mini_ddpm_chain_matches_reference_golden(golden, mini, dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=3)
    dw.set_noise_source([torch.from_numpy(synth.noise(d, 2, 16000, seed=7)) for d in range(3)])
    x = dw(torch.from_numpy(synth.waveforms(2, 16000, seed=7)).to(dev))
    assert rel_err(x.cpu().numpy(), golden["mini/ddpm_n3"]) < TOL_CHAIN


@pytest.mark.parametrize("mode", ["f32", "f32s", "bf16"])
def test_full_size_batch_is_the_small_batch_on_shared_clips(golden, dev, dh, mode):
    """BASELINE configs[1] at its full size (512 clips, shipped config, DDPM n = 5): utterances are independent and the
    Philox noise is keyed on the global utterance index, so clips 0-1 and 510-511 of the 512-clip run must equal, bit
    for bit, the same clips purified in batches of two; and in fp32 mode with the golden noise tensors the first two
    clips must meet the reference's golden vectors."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net, _ = _net(cfg, dev)
    net.set_precision(mode)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=5)
    B = 512
    x = torch.from_numpy(synth.waveforms(B, 16000, seed=77)).to(dev)
    x[:2] = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    dw.set_noise_source(("philox", 31, 0))
    big = dw(x)
    assert big.shape == x.shape and torch.isfinite(big).all()
    dw.set_noise_source(("philox", 31, 0))
    assert torch.equal(dw(x[:2]), big[:2])
    dw.set_noise_source(("philox", 31, 510))
    assert torch.equal(dw(x[510:]), big[510:])
    if mode == "f32":
        z = [torch.from_numpy(np.concatenate([synth.noise(d, 2, 16000, seed=1234), synth.noise(d, 2, 16000, seed=9)]))
             for d in range(5)]
        dw.set_noise_source(z)
        xp = dw(x[:4])
        assert rel_err(xp[:2].cpu().numpy(), golden["full/ddpm_n5/x"]) < TOL_CHAIN


def test_full_eps_matches_reference_golden(golden, full, dev):
    cfg, net, _ = full
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    with torch.no_grad():
        eps = net((x0, 4.0 * torch.ones(2, 1, device=dev)))
    assert rel_err(eps.cpu().numpy(), golden["full/eps_t4"]) < TOL_EVAL


@pytest.mark.parametrize("n", [1, 2, 5])
def test_full_ddpm_m5_acoustic_system_match_reference_golden(golden, full, dh, dev, n):
    """BASELINE config 1 inputs (B=2, L=16000, seed 1234) through AcousticSystem(M5, None, DiffWave)."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.audio_models.M5.M5Net import M5
    from audiopure_amd.acoustic_system import AcousticSystem
    cfg, net, _ = full
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=n)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.m5_state_dict(10).items()})
    m5 = m5.to(dev).eval()
    assert m5._get_name() == "M5"
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(n)]
    dw.set_noise_source(list(z))
    xp = dw(x0)
    assert rel_err(xp.cpu().numpy(), golden[f"full/ddpm_n{n}/x"]) < TOL_CHAIN
    system = AcousticSystem(classifier=m5, transform=None, defender=dw, defense_type="wave")
    dw.set_noise_source(list(z))
    lp = system(x0, True)
    np.testing.assert_allclose(lp.cpu().numpy(), golden[f"full/ddpm_n{n}/m5_logprobs"], rtol=0, atol=1e-3)
    lp0 = system(x0, False)
    np.testing.assert_allclose(lp0.cpu().numpy(), golden["full/acoustic_system_nodefense"], rtol=0, atol=2e-5)


def test_full_denoise_helpers_match_reference_golden(golden, full, dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = full
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    for t in (1, 25):
        dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=t)
        assert rel_err(dw.one_shot_denoise(x0).cpu().numpy(), golden[f"full/one_shot_t{t}"]) < TOL_EVAL
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=25)
    assert rel_err(dw.two_shot_denoise(x0).cpu().numpy(), golden["full/two_shot_t25"]) < 2 * TOL_EVAL
    eps, mu, sigma = dw.compute_coefficients(x0, 4)
    assert rel_err(eps.cpu().numpy(), golden["full/eps_t4"]) < TOL_EVAL
    a, ab = dh["Alpha"][4], dh["Alpha_bar"][4]
    mu_ref = (x0.cpu() - (1 - a) / torch.sqrt(1 - ab) * torch.from_numpy(golden["full/eps_t4"])) / torch.sqrt(a)
    assert rel_err(mu.cpu().numpy(), mu_ref.numpy()) < TOL_EVAL
    assert float(sigma) == float(dh["Sigma"][4])


def test_rev_vpsde_helper_functions_are_f_and_g_without_the_time_flip(mini, dh, dev):
    """diffwave_sde.RevVPSDE.vpsde_fn / rvpsde_fn (:73-116) against f / g (:118-134)."""
    import types
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    cfg, net, _ = mini
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=3)
    args = types.SimpleNamespace(t=3, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False, sample_step=1)
    sde = RevDiffWave.from_model(dw, args).rev_vpsde
    sde.audio_shape = (1, 1200)
    x = torch.from_numpy(synth.waveforms(2, 1200, seed=4)).to(dev).reshape(2, -1)
    t = torch.tensor([0.015])
    drift, diff = sde.vpsde_fn(t, x)
    beta = float(sde.discrete_betas[2]) * sde.N                      # _scale_timesteps(0.015) - 1 = 2
    assert torch.allclose(drift, -0.5 * beta * x) and abs(float(diff[0]) - beta ** 0.5) < 1e-7
    assert torch.equal(sde.rvpsde_fn(t, x, "drift"), -sde.f(1 - t, x))
    assert torch.equal(sde.rvpsde_fn(t, x, "diffusion"), sde.g(1 - t, x)[:, 0])


def test_unconditional_sampling_is_the_full_reverse_chain(mini, dh, dev):
    """util.sampling (util.py:126-158) with a short schedule: x_T = z_0, then T reverse steps, vs the oracle's
    ddpm_coefficients loop on the same noise tensors."""
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import sampling, calc_diffusion_hyperparams
    O = _oracle()
    cfg, net, sd = mini
    w = O.fold_state_dict(sd)
    T, B, L = 6, 2, 1800
    dh6 = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    odh = O.diffusion_hyperparams(T, 1e-4, 0.05)
    z = [torch.from_numpy(synth.noise(d, B, L, seed=33)) for d in range(T)]
    x = z[0].clone()
    with torch.no_grad():
        for i, t in enumerate(range(T - 1, -1, -1)):
            eps, mu, sigma = O.ddpm_coefficients(w, cfg, odh, x, t)
            x = mu + sigma * z[1 + i] if t > 0 else mu
    out = sampling(net, (B, 1, L), dh6, noise_source=list(z))
    assert out.shape == (B, 1, L)
    assert rel_err(out.cpu().numpy(), x.numpy()) < TOL_CHAIN


def test_reffwave_rounds_match_oracle_composition(mini, dh, dev):
    """ReffWave (diffwave_ddpm.py:251-313): num_re rounds of q-sample + one-shot denoise, each round one native chain
    call, against the same composition of the oracle's q_sample / one_shot_denoise on the same noise tensors."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import ReffWave
    O = _oracle()
    cfg, net, sd = mini
    w = O.fold_state_dict(sd)
    B, L, t_star, rounds = 2, 2500, 7, 3
    x0 = torch.from_numpy(synth.waveforms(B, L, seed=21))
    z = [torch.from_numpy(synth.noise(d, B, L, seed=21)) for d in range(rounds)]
    ref = x0
    with torch.no_grad():
        for i in range(rounds):
            ref = O.one_shot_denoise(w, cfg, dh, O.q_sample(dh, ref, t_star, z[i]), t_star)
    rw = ReffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=t_star, num_re=rounds)
    rw.set_noise_source(list(z))
    out = rw(x0.to(dev))
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 3 * TOL_EVAL
    rw.set_noise_source(("philox", 5, 0))                   # counter-based stream: a different key every round
    a = rw(x0.to(dev))
    rw.num_re = 1
    b = rw(x0.to(dev))
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_sde_matches_oracle_and_reference_drift(golden, mini, dh, dev):
    import types
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    O = _oracle()
    cfg, net, w = mini
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=4)
    args = types.SimpleNamespace(t=4, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False,
                                 sample_step=1, ddpm_path=None, ddpm_config=None)
    rev = RevDiffWave.from_model(dw, args)
    # per-step arithmetic pinned to the reference's RevVPSDE.f / .g
    xs = (torch.from_numpy(synth.waveforms(2, 16000, seed=7)) * 1.5).view(2, -1).to(dev)
    for k in (0, 4):
        s = torch.tensor([1.0 - (k + 1.5) / 200.0])
        assert rel_err(rev.rev_vpsde.f(s, xs).cpu().numpy(), golden[f"mini/sde/f_k{k}"]) < TOL_EVAL
        np.testing.assert_allclose(rev.rev_vpsde.g(s, xs)[:, :4].cpu().numpy(), golden[f"mini/sde/g_k{k}"], rtol=1e-6)
    assert np.array_equal(rev.rev_vpsde.discrete_betas.numpy(), golden["mini/sde/discrete_betas"])
    assert np.array_equal(rev.rev_vpsde.alphas_cumprod.numpy(), golden["mini/sde/alphas_cumprod"])
    # whole Euler-Maruyama chain vs the oracle restatement
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=7))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=7)) for d in range(5)]
    dw.set_noise_source(list(z))
    got = rev(x0.to(dev))
    ref = O.sde_purify(w, cfg, O.sde_tables(), x0, 4, z)
    assert rel_err(got.cpu().numpy(), ref.numpy()) < TOL_CHAIN


def test_philox_stream_is_shard_invariant_and_matches_fused_path(mini, dh, dev):
    """In-kernel Philox == ap_philox_normal fill fed back as explicit z; rows depend only on the global index."""
    from audiopure_amd import _native as N
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    lib = N.lib()
    B, L, n, seed = 4, 3000, 2, 99
    x0 = torch.from_numpy(synth.waveforms(B, L, seed=5)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=n)
    dw.set_noise_source(("philox", seed, 10))
    full = dw(x0)
    z = []
    for d in range(n):
        t = torch.empty(B, L, device=dev)
        N.check(lib.ap_philox_normal(N.ptr(t), seed, d, 10, B, L, N.stream()))
        z.append(t.view(B, 1, L))
    dw.set_noise_source(z)
    assert torch.equal(dw(x0), full)
    dw.set_noise_source(("philox", seed, 12))
    assert torch.equal(dw(x0[2:]), full[2:])
    zz = torch.cat([t.reshape(-1) for t in z]).cpu().double()
    assert abs(zz.mean().item()) < 0.02 and abs(zz.std().item() - 1.0) < 0.02


def test_philox_bits_match_numpy_restatement(dev):
    from audiopure_amd import _native as N
    from oracle.philox import philox_normal
    t = torch.empty(3, 1000, device=dev)
    N.check(N.lib().ap_philox_normal(N.ptr(t), 0x1234567890ABCDEF, 3, (1 << 33) + 5, 3, 1000, N.stream()))
    ref = philox_normal(0x1234567890ABCDEF, 3, (1 << 33) + 5, 3, 1000)
    np.testing.assert_allclose(t.cpu().numpy(), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("L", [16000, 8000, 700])
def test_m5_matches_oracle(dev, L):
    from audiopure_amd.audio_models.M5.M5Net import M5
    O = _oracle()
    sd = synth.m5_state_dict(35, seed=2)
    m5 = M5(n_input=1, n_output=35)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m5 = m5.to(dev).eval()
    x = torch.from_numpy(synth.waveforms(5, L, seed=8))
    if L < 1500:
        with pytest.raises(Exception):
            m5(x.to(dev))
        return
    np.testing.assert_allclose(m5(x.to(dev)).cpu().numpy(), O.m5_forward(sd, x).numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("mode", [0, 1])
def test_melspec_matches_oracle(dev, mode):
    from audiopure_amd.transforms import MelSpecDB, ToMelSpectrogramDB
    O = _oracle()
    x = torch.from_numpy(synth.waveforms(3, 16000, seed=5))
    tr = (MelSpecDB if mode == 0 else ToMelSpectrogramDB)(32)
    got = tr(x.to(dev)).cpu()
    ref = O.melspec_db(x, ref_max=bool(mode), top_db=80.0 if mode else None)
    assert got.shape == (3, 1, 32, 32)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=5e-3)


def test_errors_are_loud(mini, dh, dev):
    from audiopure_amd import _native as N
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    x = torch.from_numpy(synth.waveforms(1, 1000, seed=1)).to(dev)
    xg = x.clone().requires_grad_(True)
    with torch.enable_grad():
        assert net.eps(xg, 1.0).requires_grad          # an autograd node since round 2 (tests/test_gpu_dropin.py, test_gpu_grad.py)
    with pytest.raises(ValueError):
        net.eps(x[0], 1.0)                             # [1, L] instead of [B, 1, L]
    with pytest.raises(AssertionError):
        dw(x[0])
    with pytest.raises(TypeError):
        DiffWave(model=torch.nn.Linear(2, 2), diffusion_hyperparams=dh)
    eng = net.engine()
    with pytest.raises(N.NativeError):
        N.check(eng.lib.ap_eps_fwd(eng.ctx, N.ptr(x), 0.0, N.ptr(x), 1, 1000, None, 0, N.stream()), "ap_eps_fwd")
    # numpy input is accepted like the reference (diffwave_ddpm.py:38-39)
    out = dw(x.cpu().numpy())
    assert out.shape == x.shape and out.is_cuda


def test_empty_batch_flows_through_like_the_torch_modules(mini, dh, dev):
    """The reference's modules map an empty batch to empty outputs; so does the boundary (no launch, no error)."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.audio_models.M5.M5Net import M5
    from audiopure_amd.acoustic_system import AcousticSystem
    cfg, net, _ = mini
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.m5_state_dict(10).items()})
    m5 = m5.to(dev).eval()
    x = torch.empty((0, 1, 16000), device=dev)
    assert dw(x).shape == (0, 1, 16000)
    out = AcousticSystem(classifier=m5, transform=None, defender=dw, defense_type="wave")(x, True)
    assert out.shape[0] == 0


# ---- AP_PREC_BF16 (BASELINE configs[3]): bf16 MFMA operands, fp32 accumulate / storage -------------------------
# Every staging form of the bf16 block kernel meets the oracle: window staging with d = 1 (WS 1), d = 2 (WS 2), d = 4 .. 32 (WS 0),
# three-tap staging with the strided tile walk (d = 64 .. 2048, incl. d >= L), single-tile clips, ragged last tiles, and clip
# lengths that are not a multiple of four (masked tail of the last column quad).
@pytest.mark.parametrize("L,layer", [(2048, 0), (1500, 5), (4133, 11), (130, 3),
                                     (2048, 1), (2048, 2), (1536, 3), (2048, 4), (1920, 5), (2048, 6), (2048, 7), (4096, 9),
                                     (4096, 10), (1024, 11), (16000, 11), (128, 0), (64, 1), (132, 5), (4, 2), (1001, 0),
                                     (1002, 1), (1003, 4), (2049, 8), (16001, 10)])
def test_bf16_resblock_matches_bf16_emulating_oracle(dev, L, layer):
    """Tight check of the bf16 kernel's logic: the oracle rounds the same GEMM operands to bf16 (RNE) and keeps
    fp32 products/accumulation, so only summation order and bf16 rounding-boundary flips of g differ.
    Zero padding: WaveNet.py:26-27; gate :90; 1x1 convs :93-95."""
    from audiopure_amd import _native as N
    O = _oracle()
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    net.set_precision("bf16")
    w = O.fold_state_dict(sd)
    eng = net.engine()
    B, C_ = 2, 256
    h = torch.from_numpy(synth.uniform(f"hb/{L}", (B, C_, L), 1, -1.5, 1.5))
    skip0 = torch.from_numpy(synth.uniform(f"sb/{L}", (B, C_, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    with torch.no_grad():
        p = f"residual_layer.residual_blocks.{layer}"
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        h_q, s_q = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb, bf16_operands=True)
        h_f, s_f = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
    hd, sk = h.to(dev), skip0.to(dev).clone()
    hout = torch.empty_like(hd)
    pt = part_t.to(dev).contiguous()
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
    assert rel_err(hout.cpu().numpy(), h_q.numpy()) < 2e-3
    assert rel_err((sk.cpu() - skip0).numpy(), s_q.numpy()) < 4e-3
    # and the distance to the exact fp32 block is the bf16 operand rounding, reported tolerance 3e-2 of max
    assert rel_err(hout.cpu().numpy(), h_f.numpy()) < 3e-2
    assert rel_err((sk.cpu() - skip0).numpy(), s_f.numpy()) < 3e-2
    # deferred-skip form (round 4; include/audiopure.h): the block without skip_conv + the skip GEMM over its bf16 g image.
    # h' is the same pass of the same kernel: bit for bit.  A one-layer group accumulates in the fused form's order: bit for bit.
    hout2 = torch.empty_like(hd)
    gimg = torch.empty((B, L, C_), dtype=torch.bfloat16, device=dev)
    sk2 = skip0.to(dev).clone()
    N.check(eng.lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout2), gimg.data_ptr(), B, L, N.stream()))
    N.check(eng.lib.ap_skip_gemm(eng.ctx, layer, 1, gimg.data_ptr(), N.ptr(sk2), 1, B, L, N.stream()))
    assert torch.equal(hout2, hout)
    assert torch.equal(sk2, sk)
    gimg_noh = torch.zeros_like(gimg)                            # h_out = NULL (the net's last layer): the same g image, no h' written
    N.check(eng.lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(hd), N.ptr(pt), None, gimg_noh.data_ptr(), B, L, N.stream()))
    assert torch.equal(gimg_noh.view(torch.int16), gimg.view(torch.int16))
    sk3 = torch.full_like(sk2, 3.0)                              # accumulate_skip = 0 overwrites
    N.check(eng.lib.ap_skip_gemm(eng.ctx, layer, 1, gimg.data_ptr(), N.ptr(sk3), 0, B, L, N.stream()))
    assert rel_err(sk3.cpu().numpy(), s_q.numpy()) < 4e-3


@pytest.mark.parametrize("L,layer0,nl", [(16000, 0, 12), (4001, 3, 5), (1002, 9, 3), (130, 10, 2)])
def test_bf16_skip_gemm_over_a_group_of_layers_matches_the_oracle_sum(dev, L, layer0, nl):
    """WaveNet.py:131-133 `skip += skip_n` taken inside one K-concatenated GEMM: a group of consecutive layers run through
    ap_resblock_fwd_gate (each writing its bf16 g image) and ONE ap_skip_gemm, against the bf16-emulating oracle's sum of
    the per-layer skip outputs and against the fused per-layer kernel (fp32 summation order only: 5e-6)."""
    from audiopure_amd import _native as N
    O = _oracle()
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    net.set_precision("bf16")
    w = O.fold_state_dict(sd)
    eng = net.engine()
    B, C_ = 2, 256
    h = torch.from_numpy(synth.uniform(f"hg/{L}", (B, C_, L), 1, -1.5, 1.5))
    base = torch.from_numpy(synth.uniform(f"sg/{L}", (B, C_, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    s_sum, hq, pts = torch.zeros_like(h), h.clone(), []
    with torch.no_grad():
        for n in range(layer0, layer0 + nl):
            p = f"residual_layer.residual_blocks.{n}"
            pts.append(torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1))
            hq, s_n = O.residual_block(w, n, 2 ** (n % 12), hq, emb, bf16_operands=True)
            s_sum += s_n
    gimg = torch.empty((nl, B, L, C_), dtype=torch.bfloat16, device=dev)
    hin, sk_fused = h.to(dev), base.to(dev).clone()
    for i, n in enumerate(range(layer0, layer0 + nl)):
        ho, ho2 = torch.empty_like(hin), torch.empty_like(hin)
        pt = pts[i].to(dev).contiguous()
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, n, N.ptr(hin), N.ptr(pt), N.ptr(ho), N.ptr(sk_fused), 1, B, L, N.stream()))
        N.check(eng.lib.ap_resblock_fwd_gate(eng.ctx, n, N.ptr(hin), N.ptr(pt), N.ptr(ho2), gimg[i].data_ptr(), B, L, N.stream()))
        assert torch.equal(ho, ho2)
        hin = ho
    sk = base.to(dev).clone()
    N.check(eng.lib.ap_skip_gemm(eng.ctx, layer0, nl, gimg.data_ptr(), N.ptr(sk), 1, B, L, N.stream()))
    got = (sk.cpu() - base).numpy()
    # (the oracle's h runs ahead through nl layers of bf16 rounding-boundary flips: a looser bar than one layer's 4e-3)
    assert rel_err(got, s_sum.numpy()) < 4e-3 * max(1.0, nl ** 0.5)
    assert rel_err(got, (sk_fused.cpu() - base).numpy()) < 5e-6


@pytest.mark.parametrize("B,L,layer0,nl,acc", [(1, 4, 0, 1, 0), (3, 130, 2, 2, 1), (2, 1001, 5, 5, 1), (7, 257, 0, 12, 0), (2, 4096, 0, 36, 1),
                                               (5, 127, 30, 6, 0), (1, 16000, 35, 1, 1), (300, 128, 3, 3, 1)])
def test_bf16_skip_gemm_alone_matches_a_plain_gemm(dev, B, L, layer0, nl, acc):
    """ap_skip_gemm on RANDOM bf16 images (not produced by the block kernel), against the same sum written with torch matmuls on
    the device's folded skip_conv weights rounded to bf16: isolates the kernel's tiling -- partial tiles, single-column-quad clips,
    more tiles than CUs and fewer, groups of 1 .. 36 layers (odd and even chunk counts), write and accumulate -- from everything
    around it.  Products of bf16 values are exact in fp32, so only the fp32 summation order differs: 2e-5 of max."""
    from audiopure_amd import _native as N
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net, _ = _net(cfg, dev)
    net.set_precision("bf16")
    eng = net.engine()
    C_ = 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(B * 1000 + L + nl)
    g = (torch.rand((nl, B, L, C_), device=dev, generator=gen) * 2 - 1).to(torch.bfloat16).contiguous()
    base = torch.rand((B, C_, L), device=dev, generator=gen) - 0.5
    ref = base.clone() if acc else torch.zeros_like(base)
    for i in range(nl):
        w = torch.empty((C_, C_), device=dev)
        N.check(eng.lib.ap_ctx_get_folded(eng.ctx, 2, layer0 + i, N.ptr(w), w.numel(), N.stream()))
        bias = net.residual_layer.residual_blocks[layer0 + i].skip_conv.bias.detach().float()
        ref += torch.einsum("sc,blc->bsl", w.to(torch.bfloat16).float(), g[i].float()) + bias.view(1, C_, 1)
    out = base.clone() if acc else torch.full_like(base, float("nan"))
    N.check(eng.lib.ap_skip_gemm(eng.ctx, layer0, nl, g.data_ptr(), N.ptr(out), acc, B, L, N.stream()))
    assert torch.isfinite(out).all()
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("L", [16000, 1001])
def test_bf16_eps_is_the_same_for_every_skip_group_size(dev, L):
    """ap_ctx_set_skip_group: groups of G layers per skip GEMM (G = 1 accumulates in the fused form's order: bit-identical eps;
    larger groups change the fp32 summation order of skip only, which final_conv's bf16 operand rounding turns into isolated
    rounding-boundary flips: 3e-3 of max|eps|, inside the mode's 1e-2 per evaluation), uneven last group included."""
    cfg = synth.mini_wavenet_config(256, 14, 12)
    net, _ = _net(cfg, dev, seed=8)
    net.set_precision("bf16")
    eng = net.engine()
    x = torch.from_numpy(synth.waveforms(3, L, seed=L)).to(dev)
    try:
        eng.skip_group = 0
        ref = net.eps(x, 7.0)
        eng.skip_group = 1
        assert torch.equal(net.eps(x, 7.0), ref)
        for G in (4, 7, 14):
            eng.skip_group = G
            got = net.eps(x, 7.0)
            assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 3e-3, G
    finally:
        eng.skip_group = min(eng.SKIP_GROUP, 14)                 # the engine's fixed default (never a function of batch or memory)


TOL_BF16_CHAIN = 5e-3          # bf16-mode chain vs the bf16-emulating chain oracle (same operand roundings, fp32 elsewhere)
TOL_BF16_VS_FP32 = 1e-3        # bf16-mode chain vs the reference's fp32 result: 3 x the measured 3.3e-4


def test_bf16_full_chain_matches_bf16_emulating_chain_oracle_and_fp32_reference(golden, dev, dh):
    """Whole shipped-config DDPM n = 5 in bf16 mode (diffwave_ddpm.py:49-104): against the oracle chain whose every
    eps-evaluation rounds the same GEMM operands to bf16 (5e-3 of max|x|), and against the reference's fp32 golden
    vector (1e-3: the cost of the bf16 operands, measured 3.3e-4)."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    O = _oracle()
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net, sd = _net(cfg, dev)
    net.set_precision("bf16")
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=5)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(5)]
    dw.set_noise_source(list(z))
    xp = dw(x0.to(dev)).cpu().numpy()
    ref_q = O.ddpm_purify(O.fold_state_dict(sd), cfg, O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG), x0, 5, z,
                          bf16_operands=True)
    err_q = rel_err(xp, ref_q.numpy())
    err_f = rel_err(xp, golden["full/ddpm_n5/x"])
    print("bf16 chain rel err vs bf16-emulating oracle:", err_q, " vs fp32 reference:", err_f)
    assert err_q < TOL_BF16_CHAIN
    assert err_f < TOL_BF16_VS_FP32


@pytest.mark.parametrize("L", [23457, 5003, 1001])
def test_bf16_eps_on_variable_length_clips_meets_the_bf16_oracle(dev, L):
    """Clips that are not 1 s and not a multiple of four samples (kws_adaptive_attack_eval.py:178 sets audio_shape per
    utterance) in bf16 mode: all twelve dilations of a cycle through the persistent kernel's ragged instantiations, against
    the oracle network with the same operand roundings.  Stated tolerance for one eps-evaluation in this mode: 1e-2 of
    max|eps| (bf16 rounding-boundary flips of u and g, 2^-9 each, through twelve layers and final_conv); the ragged length
    must do no worse than the next multiple of four, which runs the aligned instantiations, by more than a factor 1.5."""
    O = _oracle()
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=5)
    net.set_precision("bf16")
    w = O.fold_state_dict(sd)
    errs = []
    for Lx in (L, L + 4 - L % 4):
        x = torch.from_numpy(synth.waveforms(2, Lx, seed=L))
        got = net.eps(x.to(dev), 3.0).cpu()
        with torch.no_grad():
            ref = O.eps_net(w, cfg, x, 3.0 * torch.ones(2, 1), bf16_operands=True)
        assert got.shape == ref.shape
        errs.append(rel_err(got.numpy(), ref.numpy()))
    print("bf16 eps rel err ragged / aligned:", errs)
    assert errs[0] < 1e-2 and errs[1] < 1e-2
    assert errs[0] < 1.5 * errs[1] + 1e-3


# ---- direct C-ABI entry points, chunking, graph capture ---------------------------------------------------------
# ---- AP_PREC_F32_SPLIT: fp32 operands as three bf16 parts, six partial products on the bf16 MFMA ------------------
@pytest.mark.parametrize("L,layer", [(16000, 0), (16000, 1), (16000, 4), (16000, 5), (16000, 6), (16000, 7), (16000, 11), (4133, 8),
                                     (4133, 11), (1000, 11), (130, 3), (130, 9), (77, 0), (5, 1), (1, 0), (2, 0), (3, 1), (63, 5),
                                     (64, 5), (65, 5), (127, 6), (129, 6), (16001, 9), (333, 6)])
def test_split_minimal_filtering_resblock_matches_oracle(dev, L, layer):
    """AP_PREC_F32_SPLIT with the dilated conv in F(2,3) form (ap_resblock_f32s2.hip: two launches, g handed on through h_out) vs the
    oracle's direct-form Residual_block.forward (WaveNet.py:75-97) at the exact-fp32 kernel's tolerance (5e-6 of max): every
    dilation class, d >= L, ragged and tiny clips, both skip modes, guard values around nothing (h_out is fully overwritten) -- and
    the direct-form split kernel of the same context (ap_ctx_set_f32_form 0) beside it."""
    from audiopure_amd import _native as N
    O = _oracle()
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    net.set_precision("f32sw")
    w = O.fold_state_dict(sd)
    eng = net.engine()
    lib = eng.lib
    assert lib.ap_ctx_get_f32_form(eng.ctx) == 1
    B = 3
    h = torch.from_numpy(synth.uniform(f"h/256/{L}", (B, 256, L), 1, -1.5, 1.5))
    skip0 = torch.from_numpy(synth.uniform(f"s/256/{L}", (B, 256, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    with torch.no_grad():
        p = f"residual_layer.residual_blocks.{layer}"
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        h_ref, s_ref = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
    hd, pt = h.to(dev), part_t.to(dev).contiguous()
    outs = {}
    try:
        for form in (1, 0):
            N.check(lib.ap_ctx_set_f32_form(eng.ctx, form))
            sk = skip0.to(dev).clone()
            hout = torch.full_like(hd, float("nan"))
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
            sk2 = torch.full_like(sk, 7.0)
            hout2 = torch.full_like(hd, float("nan"))
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout2), N.ptr(sk2), 0, B, L, N.stream()))
            assert torch.equal(hout, hout2)                                       # run-to-run, and independent of the skip mode
            assert rel_err(hout.cpu().numpy(), h_ref.numpy()) < 5e-6, form
            assert rel_err(sk.cpu().numpy(), (skip0 + s_ref).numpy()) < 5e-6, form
            assert rel_err(sk2.cpu().numpy(), s_ref.numpy()) < 5e-6, form
            outs[form] = hout.cpu().numpy()
    finally:
        N.check(lib.ap_ctx_set_f32_form(eng.ctx, 1))
    assert rel_err(outs[1], outs[0]) < 3e-6



@pytest.mark.parametrize("L,layer", [(1500, 2), (2048, 10), (4133, 11), (130, 3), (16000, 0)])
def test_split_resblock_matches_oracle_at_the_fp32_tolerance(dev, L, layer):
    """Same inputs and the SAME tolerance (5e-6 of max) as test_resblock_matches_oracle holds the exact fp32 MFMA
    kernel to; and the two kernels agree with each other to fp32 rounding noise."""
    from audiopure_amd import _native as N
    O = _oracle()
    C_ = 256
    cfg = synth.mini_wavenet_config(C_, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    w = O.fold_state_dict(sd)
    B = 2
    h = torch.from_numpy(synth.uniform(f"h/{C_}/{L}", (B, C_, L), 1, -1.5, 1.5))
    skip0 = torch.from_numpy(synth.uniform(f"s/{C_}/{L}", (B, C_, L), 1, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    with torch.no_grad():
        p = f"residual_layer.residual_blocks.{layer}"
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        h_ref, s_ref = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
    hd = h.to(dev)
    pt = part_t.to(dev).contiguous()
    outs = {}
    for mode in ("f32", "f32s"):
        net.set_precision(mode)
        eng = net.engine()
        sk = skip0.to(dev).clone()
        hout = torch.empty_like(hd)
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
        assert rel_err(hout.cpu().numpy(), h_ref.numpy()) < 5e-6, mode
        assert rel_err(sk.cpu().numpy(), (skip0 + s_ref).numpy()) < 5e-6, mode
        sk2 = torch.full_like(sk, 7.0)
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk2), 0, B, L, N.stream()))
        assert rel_err(sk2.cpu().numpy(), s_ref.numpy()) < 5e-6, mode
        outs[mode] = (hout.cpu().numpy(), sk2.cpu().numpy())
    for mode in ("f32s",):
        assert rel_err(outs[mode][0], outs["f32"][0]) < 2e-6, mode
        assert rel_err(outs[mode][1], outs["f32"][1]) < 2e-6, mode


def _adversarial_cases(B, C, L):
    g = torch.Generator().manual_seed(7)
    base = torch.randn(B, C, L, generator=g)
    out = {"unit": base}
    x = base.clone()                                                # cancellation: channel 2k+1 = -(channel 2k)(1 + 2^-12 noise), scale 64
    x[:, 1::2] = -x[:, 0::2] * (1 + 2.0 ** -12 * torch.randn(B, C // 2, L, generator=g))
    out["cancel"] = 64.0 * x
    for name, span in (("range10", 10), ("range20", 20)):             # dynamic range: channel c scaled by 2^e(c), e = -span .. span
        e = torch.linspace(-span, span, C).view(1, C, 1).round()
        out[name] = base * torch.pow(2.0, e)
    return out


@pytest.mark.parametrize("layer", [2, 7])
def test_fp32_class_modes_bound_their_error_on_adversarial_operands(dev, layer):
    """VERDICT r4 item 7 (promote or delete the split-operand kernels): AP_PREC_F32_SPLIT is promoted to a documented fp32-class
    mode on this evidence, the fp16 two-part mode was deleted on it (its error under a 2^20 dynamic range is 2e-1 of max:
    profiles/r5_fp32_class_adversarial_error.txt).  Per-dot-product error of the block against an fp64 evaluation of
    Residual_block.forward (WaveNet.py:75-97) on adversarial operands -- channel pairs that cancel to 2^-12, a 2^20 and a 2^40
    dynamic range across the contraction, plain unit-scale data: the split mode and the F(2,3) form stay within twice the
    direct-form fp32 kernel's own error (+ 3 x on the 2^40 case for F(2,3), whose input differences round once more;
    measured <= 1.2 x and <= 1.9 x)."""
    from audiopure_amd import _native as N
    O = _oracle()
    C_, L, B = 256, 2048, 2
    net, sd = _net(synth.mini_wavenet_config(C_, 12, 12), dev, seed=3)
    w = O.fold_state_dict(sd)
    w64 = {k: v.double() for k, v in w.items()}
    p = f"residual_layer.residual_blocks.{layer}"
    part = w[p + ".fc_t.bias"].clone()                               # emb = 0: part_t is fc_t's bias
    for name, h in _adversarial_cases(B, C_, L).items():
        with torch.no_grad():
            h64, s64 = O.residual_block(w64, layer, 2 ** layer, h.double(), torch.zeros(B, 512, dtype=torch.float64))
        err = {}
        for mode in ("f32d", "f32", "f32s", "f32sw"):
            net.set_precision(mode)
            eng = net.engine()
            hd, ptd = h.to(dev), part.to(dev).contiguous()
            ho, sk = torch.empty_like(hd), torch.zeros_like(hd)
            N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ho), N.ptr(sk), 0, B, L, N.stream()))
            err[mode] = (float((ho.cpu().double() - h64).abs().max() / h64.abs().max()),
                         float((sk.cpu().double() - s64).abs().max() / s64.abs().max()))
        print(f"layer {layer} {name}: " + "  ".join(f"{m} {a:.2e}/{b:.2e}" for m, (a, b) in err.items()))
        for k in (0, 1):
            floor = 2e-8
            assert err["f32s"][k] <= 2.0 * err["f32d"][k] + floor, (name, k, err)
            assert err["f32"][k] <= (3.0 if name == "range20" else 2.0) * err["f32d"][k] + floor, (name, k, err)
            # the split mode in F(2,3) form (opt-in): the form's allowance, not the split mode's (measured 2.7 x on range20, <= 1.3 x elsewhere)
            assert err["f32sw"][k] <= (3.0 if name == "range20" else 2.0) * err["f32d"][k] + floor, (name, k, err)
    net.set_precision("f32")


@pytest.mark.parametrize("mode", ["f32s"])
def test_split_full_chain_matches_reference_golden_at_the_fp32_tolerance(golden, dev, dh, mode):
    """Whole shipped-config DDPM n=5 + one-shot denoise in the split modes vs the reference's fp32 golden vectors, at the
    tolerances the exact-fp32 tests use (TOL_CHAIN / TOL_EVAL)."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net, _ = _net(cfg, dev)
    net.set_precision(mode)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    with torch.no_grad():
        eps = net((x0, 4.0 * torch.ones(2, 1, device=dev)))
    assert rel_err(eps.cpu().numpy(), golden["full/eps_t4"]) < TOL_EVAL
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=5)
    dw.set_noise_source([torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(5)])
    xp = dw(x0)
    assert rel_err(xp.cpu().numpy(), golden["full/ddpm_n5/x"]) < TOL_CHAIN
    # one-shot denoise against the reference's own vector (t* = 25 is what make_golden.py holds; a missing key is a KeyError,
    # never a HIP-vs-HIP comparison)
    dw25 = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=25)
    assert rel_err(dw25.one_shot_denoise(x0).cpu().numpy(), golden["full/one_shot_t25"]) < TOL_EVAL


@pytest.mark.parametrize("L", [16000, 2048, 1500, 132])
def test_bf16_chain_staging_from_the_previous_epilogue_is_bit_identical(dev, L):
    """The fused eps-evaluation (ap_eps_fwd: one workspace, ping-ponged h) against the same layers launched one by one
    through ap_resblock_fwd: bit for bit (partial tiles, d >= L).  (Round 1 also tried handing each layer a ready-made
    bf16 input image from the previous epilogue -- bit-identical by this test, 0 % faster, removed.)"""
    from audiopure_amd.diffusion_models._grad import EpsGrad
    cfg = synth.mini_wavenet_config(256, 14, 12)
    net, _ = _net(cfg, dev, seed=8)
    net.set_precision("bf16")
    net.engine().skip_group = 0                                  # the fused block per layer on both sides (the deferred-skip form
    x = torch.from_numpy(synth.waveforms(3, L, seed=L)).to(dev)  # sums skip in another order: its own tests above)
    eps_chain = net.eps(x, 7.0)
    eps_layers, _ = EpsGrad(net).forward_save(x, 7.0)
    assert torch.equal(eps_chain, eps_layers)


def test_all_three_block_kernels_agree_over_a_shape_sweep(dev):
    """Hazard / tiling sweep: the three residual-block kernels (fp32 MFMA, split-operand, bf16) against each other over
    lengths that hit every code path (L % 4 != 0 -> dword paths, partial last tiles, single tile, d >= L, d in {1,2} vs
    d % 4 == 0 staging), two launches back to back, batch sizes that are not powers of two.  fp32 vs split: 2e-6;
    bf16 vs fp32: the bf16 operand rounding (3e-2 of max)."""
    from audiopure_amd import _native as N
    C_ = 256
    cfg = synth.mini_wavenet_config(C_, 12, 12)
    net, _ = _net(cfg, dev, seed=11)
    rng = np.random.default_rng(0)
    cases = [(3, 128, 0), (1, 129, 1), (5, 1024, 2), (2, 1028, 7), (3, 3999, 4), (2, 4000, 11), (7, 640, 9), (1, 16000, 5)]
    cases += [(int(rng.integers(1, 6)), int(rng.integers(130, 5000)), int(rng.integers(0, 12))) for _ in range(6)]
    for B, L, layer in cases:
        h = torch.from_numpy(synth.uniform(f"sw/{L}/{layer}", (B, C_, L), 1, -1.5, 1.5)).to(dev)
        sk0 = torch.from_numpy(synth.uniform(f"sws/{L}/{layer}", (B, C_, L), 1, -1.0, 1.0)).to(dev)
        pt = torch.from_numpy(synth.uniform(f"swp/{layer}", (C_,), 1, -0.5, 0.5)).to(dev)
        outs = {}
        for mode in ("f32", "f32s", "bf16"):
            net.set_precision(mode)
            eng = net.engine()
            sk, ho = sk0.clone(), torch.empty_like(h)
            for _ in range(2):                                   # second launch accumulates onto the first's skip
                N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
            outs[mode] = (ho.cpu().numpy(), sk.cpu().numpy())
        for k in (0, 1):
            assert rel_err(outs["f32s"][k], outs["f32"][k]) < 2e-6, (B, L, layer, k)
            assert rel_err(outs["bf16"][k], outs["f32"][k]) < 3e-2, (B, L, layer, k)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_small_batches_replay_a_captured_graph_with_identical_results(mini, dh, dev, mode):
    """DiffWave._chain at the callers' batch sizes (adaptive_attack_eval.py:47: --batch_size 10) replays one captured
    ap_purify_chain call per (shape, coefficients, noise key): same kernels and arguments, so the result equals the eager call's
    bit for bit -- Philox noise, torch-generator noise under the same seed, a changed input, a changed reverse_timestep (new key);
    explicit noise tensors, large batches and a disabled switch stay eager."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    if mode == "f32":
        cfg, net, _ = mini
    else:
        net, _ = _net(synth.mini_wavenet_config(256, 12, 12), dev, seed=4)
        net.set_precision("bf16")
    x = torch.from_numpy(synth.waveforms(3, 2000, seed=5)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=3)
    eager = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=3)
    eager.graph_replay = False
    for src in (("philox", 21, 4), None):
        dw.set_noise_source(src)
        eager.set_noise_source(src)
        for k, xin in enumerate((x, x * 0.5, x)):
            torch.manual_seed(100 + k)
            ref = eager(xin)
            torch.manual_seed(100 + k)
            got = dw(xin)
            assert torch.equal(got, ref), (mode, src, k)
    assert len(dw._graphs) == 2 and not eager._graphs
    dw.reverse_timestep = eager.reverse_timestep = 2             # certification changes it per call (certified_robust.py:53)
    dw.set_noise_source(("philox", 21, 4)); eager.set_noise_source(("philox", 21, 4))
    assert torch.equal(dw(x), eager(x)) and len(dw._graphs) == 3
    assert torch.equal(dw.one_shot_denoise(x), eager.one_shot_denoise(x))
    n0 = len(dw._graphs)
    z = [torch.from_numpy(synth.noise(d, 3, 2000, seed=7)) for d in range(2)]
    dw.set_noise_source(list(z)); eager.set_noise_source(list(z))
    assert torch.equal(dw(x), eager(x)) and len(dw._graphs) == n0              # explicit noise tensors: eager
    big = torch.from_numpy(synth.waveforms(20, 16000, seed=6)).to(dev)          # 20 clips of 1 s: past the replay bound
    dw.set_noise_source(("philox", 3, 0))
    dw(big)
    assert len(dw._graphs) == n0


def test_graph_replay_cache_pins_one_workspace_and_falls_back_to_eager(mini, dh, dev, monkeypatch):
    """VERDICT r4 weak 10 / ADVICE r4: (1) after a larger batch made the engine replace its workspace, the next small call drops
    every graph captured on the old one -- the cache never pins more than the engine's current workspace, never replays into a
    freed one; (2) graphs of another engine (set_precision rebuilds it) are dropped the same way; (3) a capture that fails leaves
    a plain dw(x) working: the object turns its replay off and runs eager, same result."""
    import warnings
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    eng = net.engine()
    x = torch.from_numpy(synth.waveforms(2, 2000, seed=5)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    dw.set_noise_source(("philox", 5, 0))
    ref = dw(x).clone()
    dw.reverse_timestep = 3
    dw(x)
    assert len(dw._graphs) == 2 and all(e[5] is eng.ws for e in dw._graphs.values())
    old_ws = eng.ws
    big = torch.from_numpy(synth.waveforms(24, 16000, seed=6)).to(dev)          # past the replay bound AND past the workspace
    dw(big)
    assert eng.ws is not old_ws
    dw.reverse_timestep = 2
    assert torch.equal(dw(x), ref)
    assert len(dw._graphs) == 1 and all(e[5] is eng.ws for e in dw._graphs.values())
    del old_ws
    # (3) capture failure -> eager, once, with a warning
    dw2 = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    dw2.set_noise_source(("philox", 5, 0))

    class Boom:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused")
    monkeypatch.setattr(torch.cuda, "graph", Boom)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = dw2(x)
    assert torch.equal(got, ref) and dw2.graph_replay is False and not dw2._graphs
    assert any("capture" in str(r.message) for r in rec)
    assert torch.equal(dw2(x), ref)


def test_bf16_deferred_skip_chain_is_hip_graph_capturable(dh, dev):
    """The bf16 mode's two-kernel form (36 block launches + the skip GEMM per evaluation, ap_ctx_set_skip_group on the host side
    only) captures into a HIP graph like the fp32 chain does, and a replay reproduces the eager result bit for bit."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, _ = _net(cfg, dev, seed=4)
    net.set_precision("bf16")
    x0 = torch.from_numpy(synth.waveforms(2, 2000, seed=4)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    dw.set_noise_source(("philox", 11, 0))
    ref = dw(x0)                               # warm: engine, workspace and output allocations exist
    assert net.engine().skip_group > 0
    static_in = x0.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dw(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = dw(static_in)
    static_in.copy_(x0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("L,Bs,Bl", [(16000, 1, 3), (16000, 2, 5), (4133, 3, 9), (130, 7, 140), (64, 2, 300), (1, 5, 300)])
def test_bf16_small_batch_block_is_the_persistent_block_bit_for_bit(dev, L, Bs, Bl):
    """ap_resblock_fwd_gate on a launch of at most one 128-sample tile per CU runs ap_resblock_bf16s.hip (64-sample tiles, one per
    workgroup); on a larger batch the persistent kernel.  A clip's h' and gate image must not depend on the batch it travels in:
    the first Bs clips of a Bl-clip launch equal the Bs-clip launch bit for bit, for every dilation, with and without h' (the
    net's last layer), outputs inside untouched guard bands."""
    from audiopure_amd import _native as N
    net, _ = _net(synth.mini_wavenet_config(256, 12, 12), dev, seed=4)
    net.set_precision("bf16")
    eng = net.engine()
    lib = eng.lib
    tiles = (L + 127) // 128
    assert Bs * tiles <= 256 < Bl * tiles                        # the two launches are on different kernels
    h = (torch.rand(Bl, 256, L, device=dev) * 3 - 1.5)
    pt = torch.rand(256, device=dev) - 0.5

    def run(B, layer, with_h):
        G = 1024
        hb = torch.full((B * 256 * L + 2 * G,), 7.25, device=dev)
        gb = torch.full((B * L * 256 + 2 * G,), 7.25, device=dev, dtype=torch.bfloat16)
        ho, gi = hb[G:G + B * 256 * L], gb[G:G + B * L * 256]
        N.check(lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(h[:B].contiguous()), N.ptr(pt), N.ptr(ho) if with_h else None, gi.data_ptr(),
                                         B, L, N.stream()))
        for buf, n in ((hb, B * 256 * L), (gb, B * L * 256)):
            assert bool((buf[:G] == 7.25).all()) and bool((buf[G + n:] == 7.25).all()), "write outside the output"
        if not with_h:
            assert bool((ho == 7.25).all())                      # h' not wanted: not written
        return ho.view(B, 256, L), gi.view(B, L, 256)

    for layer in range(12):
        for with_h in (True, False):
            hs, gs = run(Bs, layer, with_h)
            hl, gl = run(Bl, layer, with_h)
            assert torch.equal(gs.view(torch.int16), gl[:Bs].view(torch.int16)), (layer, with_h)
            assert torch.equal(hs, hl[:Bs]), (layer, with_h)


def test_c_entry_points_match_python_chains(mini, dh, dev):
    """ap_purify_ddpm / ap_purify_sde / ap_one_shot_denoise (coefficients computed inside the library from the installed
    tables) give the same result as the chains the Python classes build with ap_purify_chain."""
    import types
    from audiopure_amd import _native as N
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    cfg, net, _ = mini
    B, L, n, seed = 3, 2500, 3, 7
    x0 = torch.from_numpy(synth.waveforms(B, L, seed=9)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=n)
    eng = dw._tables()
    ws = eng.workspace(B, L, dev)
    out = torch.empty_like(x0)

    dw.set_noise_source(("philox", seed, 5))
    ref = dw(x0)
    N.check(eng.lib.ap_purify_ddpm(eng.ctx, N.ptr(x0), n, 1, None, seed, 5, N.ptr(out), B, L, ws.data_ptr(), ws.numel(),
                                   N.stream()), "ap_purify_ddpm")
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-6

    ref1 = dw.one_shot_denoise(x0)
    N.check(eng.lib.ap_one_shot_denoise(eng.ctx, N.ptr(x0), n, N.ptr(out), B, L, ws.data_ptr(), ws.numel(), N.stream()))
    assert rel_err(out.cpu().numpy(), ref1.cpu().numpy()) < 2e-6

    args = types.SimpleNamespace(t=n, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False, sample_step=1)
    rev = RevDiffWave.from_model(dw, args)
    eng.set_sde_schedule(rev.rev_vpsde.discrete_betas, rev.rev_vpsde.alphas_cumprod)
    dw.set_noise_source(("philox", seed, 5))
    ref2 = rev(x0)
    N.check(eng.lib.ap_purify_sde(eng.ctx, N.ptr(x0), n, None, seed, 5, N.ptr(out), B, L, ws.data_ptr(), ws.numel(),
                                  N.stream()), "ap_purify_sde")
    assert rel_err(out.cpu().numpy(), ref2.cpu().numpy()) < 2e-6
    # sample_step = 2 concatenates along the batch axis like the reference (diffwave_sde.py:212)
    args.sample_step = 2
    assert RevDiffWave.from_model(dw, args)(x0).shape[0] == 2 * B


def test_c_entry_points_meet_reference_golden_directly(golden, mini, dh, dev):
    """ap_purify_ddpm and ap_one_shot_denoise called through ctypes -- no Python chain class in between -- against vectors
    of the REFERENCE's DiffWave (tests/golden/make_golden.py): mini net DDPM n = 3 with the injected noise list
    (diffwave_ddpm.py:36-104,143-164) and the shipped net's one_shot_denoise at t* = 25 (diffwave_ddpm.py:174-205)."""
    from audiopure_amd import _native as N
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    B, L, n = 2, 16000, 3
    eng = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=n)._tables()      # installs the schedule tables
    ws = eng.workspace(B, L, dev)
    x0 = torch.from_numpy(synth.waveforms(B, L, seed=7)).to(dev)
    z = torch.stack([torch.from_numpy(synth.noise(d, B, L, seed=7)) for d in range(n)]).to(dev).contiguous()
    out = torch.empty_like(x0)
    N.check(eng.lib.ap_purify_ddpm(eng.ctx, N.ptr(x0), n, 1, N.ptr(z), 0, 0, N.ptr(out), B, L, ws.data_ptr(), ws.numel(),
                                   N.stream()), "ap_purify_ddpm")
    assert rel_err(out.cpu().numpy(), golden["mini/ddpm_n3"]) < TOL_CHAIN

    fnet, _ = _net(dict(synth.FULL_WAVENET_CONFIG), dev)
    feng = DiffWave(model=fnet, diffusion_hyperparams=dh, reverse_timestep=25)._tables()
    fws = feng.workspace(B, L, dev)
    xf = torch.from_numpy(synth.waveforms(B, L, seed=1234)).to(dev)
    N.check(feng.lib.ap_one_shot_denoise(feng.ctx, N.ptr(xf), 25, N.ptr(out), B, L, fws.data_ptr(), fws.numel(), N.stream()),
            "ap_one_shot_denoise")
    assert rel_err(out.cpu().numpy(), golden["full/one_shot_t25"]) < TOL_EVAL


def test_chunked_batches_equal_one_call(mini, dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    x0 = torch.from_numpy(synth.waveforms(5, 1000, seed=2)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    dw.set_noise_source(("philox", 3, 0))
    eng = net.engine()
    old = eng.max_chunk
    try:
        eng.max_chunk = 512
        a = dw(x0)
        eng.max_chunk = 2                      # 3 native calls: utt_offset carries the global index
        b = dw(x0)
    finally:
        eng.max_chunk = old
    assert torch.equal(a, b)


def test_purify_is_hip_graph_capturable(mini, dh, dev):
    """The launch functions never allocate or synchronise (include/audiopure.h): capture a whole purification and replay it."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    cfg, net, _ = mini
    x0 = torch.from_numpy(synth.waveforms(2, 2000, seed=4)).to(dev)
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=2)
    dw.set_noise_source(("philox", 11, 0))
    ref = dw(x0)                               # warm: engine, workspace and output allocations exist
    static_in = x0.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dw(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        static_out = dw(static_in)
    static_in.copy_(x0 * 0.5)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, dw(x0 * 0.5))
    static_in.copy_(x0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref)


def test_bf16_final_conv_close_to_the_fp32_one(dev):
    """AP_PREC_BF16 runs final_conv's 1x1 convs on the bf16 pipe too (fp32 accumulate): eps and the fused update agree with
    the fp32 kernel to bf16 operand rounding, ragged last tile and L % 4 != 0 included."""
    from audiopure_amd import _native as N
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    nf, _ = _net(cfg, dev)
    nb, _ = _net(cfg, dev)
    nb.set_precision("bf16")
    ef, eb = nf.engine(), nb.engine()
    for B, L in ((2, 16000), (3, 1500), (2, 130), (1, 1001)):
        skip = torch.from_numpy(synth.uniform(f"fsk/{L}", (B, 256, L), 2, -6.0, 6.0)).to(dev)
        x = torch.from_numpy(synth.uniform(f"fx/{L}", (B, 1, L), 2, -1.0, 1.0)).to(dev)
        zt = torch.from_numpy(synth.uniform(f"fz/{L}", (B, 1, L), 2, -1.0, 1.0)).to(dev)
        outs = []
        for eng in (ef, eb):
            eps, out = torch.empty_like(x), torch.empty_like(x)
            N.check(eng.lib.ap_final_affine(eng.ctx, N.ptr(skip), N.ptr(x), N.ptr(eps), N.ptr(out), 1.01, -0.2, 0.05, N.ptr(zt),
                                            0, 0, 0, B, L, N.stream()))
            outs.append((eps.cpu().numpy(), out.cpu().numpy()))
        assert rel_err(outs[1][0], outs[0][0]) < 1e-2, (B, L)
        assert rel_err(outs[1][1], outs[0][1]) < 1e-2, (B, L)
