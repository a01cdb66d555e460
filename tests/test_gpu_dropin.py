"""GPU tests of round 2's boundary work: the drop-in path exactly as the *_eval.py scripts reach it (``create_model`` on a
whole-module pickle of the REFERENCE's M5 class, script-built torchaudio ``Compose``, a plain un-lowered ConvNet), the
previously untested ``DiffWave`` surface against reference-generated vectors (tests/golden/make_golden_v2.py), the
reference's ``rand_t`` semantics, and BASELINE configs[3] / configs[4] as composed workloads."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_EVAL, TOL_CHAIN = 2e-5, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def golden2():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))


@pytest.fixture(scope="module")
def dh():
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    return calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)


def _net(cfg, dev, seed=0):
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, seed).items()}, strict=True)
    return net.to(dev)


@pytest.fixture(scope="module")
def mini(dev):
    return _net(synth.mini_wavenet_config(64, 12, 12), dev)


@pytest.fixture(scope="module")
def full(dev):
    return _net(dict(synth.FULL_WAVENET_CONFIG), dev)


@pytest.fixture()
def dropin_path():
    """sys.path as `PYTHONPATH=<repo>/dropin:<repo> python adaptive_attack_eval.py` sees it (hook installed)."""
    d = os.path.join(ROOT, "dropin")
    sys.path.insert(0, d)
    import _audiopure_hook
    _audiopure_hook.install()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in ("acoustic_system", "audio_models", "diffusion_models", "robustness_eval", "M5Net")}
    yield d
    for k in list(sys.modules):
        if k.split(".")[0] in ("acoustic_system", "audio_models", "diffusion_models", "robustness_eval", "M5Net"):
            sys.modules.pop(k)
    sys.modules.update(saved)
    sys.path.remove(d)
    sys.meta_path[:] = [f for f in sys.meta_path if type(f).__module__ != "_audiopure_hook"]


# ---- 1. the scripts' own loading path ------------------------------------------------------------------------------------
def test_reference_pickled_m5_through_create_model_and_acoustic_system(golden, full, dh, dev, dropin_path):
    """adaptive_attack_eval.py:64-67,88-93,129-137 on a pickle of the reference's own M5 class: create_model ->
    .cuda() -> AcousticSystem(M5 => transform None) -> the reference's golden log-probabilities."""
    create_model = importlib.import_module("audio_models.create_model").create_model
    AcousticSystem = importlib.import_module("acoustic_system").AcousticSystem
    DiffWave = importlib.import_module("diffusion_models.diffwave_ddpm").DiffWave
    Classifier = create_model(os.path.join(ROOT, "tests", "golden", "ref_m5_module.pt"))
    Classifier.cuda()
    assert Classifier._get_name() == "M5" and type(Classifier).__module__.startswith("audiopure_amd.")
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    AS_MODEL = AcousticSystem(classifier=Classifier, transform=None, defender=None)
    AS_MODEL.eval()
    np.testing.assert_allclose(AS_MODEL(x0, False).cpu().numpy(), golden["full/acoustic_system_nodefense"], rtol=0, atol=2e-5)
    dw = DiffWave(model=full, diffusion_hyperparams=dh, reverse_timestep=1)
    dw.set_noise_source([torch.from_numpy(synth.noise(0, 2, 16000, seed=1234))])
    AS_MODEL = AcousticSystem(classifier=Classifier, transform=None, defender=dw, defense_type="wave").eval()
    np.testing.assert_allclose(AS_MODEL(x0, True).cpu().numpy(), golden["full/acoustic_system_n1"], rtol=0, atol=1e-3)
    # white-box route: the un-pickled classifier has its native input gradient too
    xg = x0.clone().requires_grad_(True)
    AcousticSystem(classifier=Classifier, transform=None, defender=None)(xg, False).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0


def test_script_built_mel_compose_and_plain_convnet_run_natively(dev):
    """adaptive_attack_eval.py:83-93 with the mel32 route: a torchaudio-shaped Compose and an un-lowered 2-D ConvNet go in,
    the native mel kernel and NativeConvNet run (the stand-ins raise if their forward is ever called)."""
    from fake_torchaudio import script_wave2spect
    from synth_convnets import CifarResNeXt, synth_init
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.convnet import NativeConvNet
    from audiopure_amd.transforms import MelSpecDB
    clf = synth_init(CifarResNeXt(10), 0).to(dev)
    AS_MODEL = AcousticSystem(classifier=clf, transform=script_wave2spect(32), defender=None)
    AS_MODEL.eval()
    assert isinstance(AS_MODEL.classifier, NativeConvNet) and type(AS_MODEL.transform) is MelSpecDB
    assert AS_MODEL.classifier._get_name() == "CifarResNeXt"
    x = torch.from_numpy(synth.waveforms(3, 16000, seed=5)).to(dev)
    got = AS_MODEL(x, False)
    mel = MelSpecDB(32)(x)
    assert mel.shape == (3, 1, 32, 32)
    with torch.no_grad():
        ref = clf(mel)                                              # torch module on the SAME native mel: classifier parity
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 1e-3
    assert torch.equal(got, NativeConvNet(clf, (1, 32, 32)).eval()(mel))   # lazily traced plan == explicitly traced plan


# ---- 2. surface that had no test: _diffusion, _reverse (noise list), fast_reverse -----------------------------------------
def _nl(n, seed):
    return [torch.from_numpy(synth.noise(d, 2, 16000, seed=seed)) for d in range(n)]


def test_diffusion_and_reverse_match_reference_golden(golden2, mini, dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=21)).to(dev)
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=20)
    dw.set_noise_source(_nl(1, 21))
    assert rel_err(dw._diffusion(x0).cpu().numpy(), golden2["mini/diffusion_t20"]) < 2e-6
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=4)
    dw.set_noise_source(_nl(3, 22))
    assert rel_err(dw._reverse(x0 * 1.2).cpu().numpy(), golden2["mini/reverse_n4"]) < TOL_CHAIN
    assert dw._noise == []                                          # exactly the reference's three draws were consumed


@pytest.mark.parametrize("ts", [20, 7])
def test_fast_reverse_matches_reference_golden(golden2, mini, dh, dev, ts):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=21)).to(dev)
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=ts)
    dw.set_noise_source(_nl(3, 23))
    assert rel_err(dw.fast_reverse(x0 * 1.1).cpu().numpy(), golden2[f"mini/fast_reverse_t{ts}"]) < TOL_CHAIN
    assert dw._noise == []


def test_fast_reverse_full_config_matches_reference_golden(golden2, full, dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    dw = DiffWave(model=full, diffusion_hyperparams=dh, reverse_timestep=25)
    dw.set_noise_source(_nl(3, 24))
    assert rel_err(dw.fast_reverse(x0).cpu().numpy(), golden2["full/fast_reverse_t25"]) < TOL_CHAIN


def test_forward_detaches_and_the_other_entry_points_differentiate(mini, dh, dev):
    """diffwave_ddpm.py:41-43: forward is no_grad (detached output even for a requires_grad input); _reverse /
    one_shot_denoise are differentiable torch code in the reference -- here the recompute-based autograd node."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from oracle import diffwave_oracle as O
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=2)
    x = torch.from_numpy(synth.waveforms(2, 2000, seed=3)).to(dev).requires_grad_(True)
    dw.set_noise_source(("philox", 5, 0))
    out = dw(x)
    assert not out.requires_grad
    y = dw.one_shot_denoise(x)
    assert y.requires_grad
    gsel = torch.from_numpy(synth.uniform("gsel", (2, 1, 2000), 2)).to(dev)
    (y * gsel).sum().backward()
    cfg = synth.mini_wavenet_config(64, 12, 12)
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    xc = x.detach().cpu().clone().requires_grad_(True)
    with torch.enable_grad():
        t = 1
        eps = O.eps_net(w, cfg, xc, float(t) * torch.ones((2, 1)))
        Ab = dh["Alpha_bar"]
        yr = (1 / Ab).sqrt()[t] * xc - (1 / Ab - 1).sqrt()[t] * eps
        (yr * gsel.cpu()).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), xc.grad.numpy()) < 1e-3
    # eps(x) of the network itself is an autograd node as well
    x2 = x.detach().clone().requires_grad_(True)
    e = mini.eps(x2, 1.0)
    assert e.requires_grad
    (e * gsel).sum().backward()
    assert torch.isfinite(x2.grad).all() and float(x2.grad.abs().max()) > 0


# ---- 3. rand_t (diffwave_sde.py:186-194) and a fresh NES key per instance --------------------------------------------------
def test_rand_t_diffuses_to_the_drawn_level_but_integrates_args_t_steps(mini, dh, dev, monkeypatch):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=4)
    args = types.SimpleNamespace(t=4, score_type="guided_diffusion", rand_t=True, t_delta=3, use_bm=True, sample_step=1,
                                 ddpm_path=None, ddpm_config=None)
    rev = RevDiffWave.from_model(dw, args)
    monkeypatch.setattr(np.random, "randint", lambda lo, hi: 2)            # total_noise_levels = 6
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=7)).to(dev)
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=7)) for d in range(5)]
    dw.set_noise_source(list(z))
    got = rev(x0)
    assert dw._noise == []                                                  # 1 + args.t draws, not 1 + 6
    a = float(rev.rev_vpsde.alphas_cumprod[6 - 1].double())                 # q-sample at the DRAWN level ...
    dw.set_noise_source(list(z))
    want = dw._chain(x0, rev.rev_vpsde.euler_steps(4), a ** 0.5, (1.0 - a) ** 0.5, n_draws=5)   # ... then args.t Euler steps
    assert torch.equal(got, want)
    from oracle import diffwave_oracle as O
    cfg = synth.mini_wavenet_config(64, 12, 12)
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    tb = O.sde_tables()
    ref = O.sde_purify(w, cfg, tb, x0.cpu(), 4, z, q_level=6)
    assert rel_err(got.cpu().numpy(), ref.numpy()) < TOL_CHAIN


def test_fresh_nes_instances_draw_different_directions(dev):
    """black_box_attack.py:181 builds a new NES every attack iteration; each must perturb along fresh directions."""
    from audiopure_amd.robustness_eval._NES import NES
    seen = []

    class _EOT:
        EOT_size, EOT_batch_size = 1, 1

        def __call__(self, xin, y):
            seen.append(xin.clone())
            n = xin.shape[0]
            return torch.zeros(n, 10, device=xin.device), torch.zeros(n, device=xin.device), None, [[0]] * n

    x = torch.from_numpy(synth.waveforms(1, 4000, seed=2)).to(dev)
    torch.manual_seed(123)
    a, b = NES(8, 8, 0.01, _EOT()), NES(8, 8, 0.01, _EOT())
    assert a.seed != b.seed
    a(x, [0]); b(x, [0])
    assert not torch.equal(seen[0], seen[1])
    torch.manual_seed(123)                                                  # "seed torch, get reproducible output"
    c = NES(8, 8, 0.01, _EOT())
    c(x, [0])
    assert torch.equal(seen[0], seen[2])


def test_non_finite_scores_are_refused_not_counted(dev):
    from audiopure_amd.robustness_eval.certified_robust import RobustCertificate

    class _Bad(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            out = torch.zeros(x.shape[0], 10, device=x.device)
            out[::3, 4] = float("nan")
            return out

    rc = RobustCertificate(classifier=_Bad().to(dev), transform=None, denoiser=None)
    with pytest.raises(FloatingPointError):
        rc.smooth_predict(torch.zeros(1, 1, 2000, device=dev), num_sampling=12, sigma=0.25, batch_size=4)


# ---- 4. BASELINE configs[3]: VP-SDE reverse, n = 10, bf16 --------------------------------------------------------------------
def test_config3_sde_n10_bf16_small_batch_vs_oracle_and_full_batch_identity(dh, dev):
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from oracle import diffwave_oracle as O
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net = _net(cfg, dev).set_precision("bf16")
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=10)
    args = types.SimpleNamespace(t=10, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False, sample_step=1,
                                 ddpm_path=None, ddpm_config=None)
    rev = RevDiffWave.from_model(dw, args)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(11)]
    dw.set_noise_source(list(z))
    got = rev(x0.to(dev))
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    # the chain oracle with the same operand roundings (bf16 GEMM operands, fp32 elsewhere): 5e-3; and the distance to the
    # fp32 oracle chain -- what the bf16 operands cost over ten steps -- stays under 2e-3
    ref_q = O.sde_purify(w, cfg, O.sde_tables(), x0, 10, z, bf16_operands=True)
    ref = O.sde_purify(w, cfg, O.sde_tables(), x0, 10, z)
    err_q, err_f = rel_err(got.cpu().numpy(), ref_q.numpy()), rel_err(got.cpu().numpy(), ref.numpy())
    print("configs[3] bf16 SDE n=10: vs bf16-emulating oracle", err_q, " vs fp32 oracle", err_f)
    assert err_q < 5e-3
    assert err_f < 2e-3
    # full size (B = 512): clips are independent and the Philox stream is keyed on the global utterance index
    B = 512
    x = torch.from_numpy(synth.waveforms(B, 16000, seed=78)).to(dev)
    dw.set_noise_source(("philox", 17, 0))
    big = rev(x)
    assert big.shape == x.shape and torch.isfinite(big).all()
    dw.set_noise_source(("philox", 17, 0))
    assert torch.equal(rev(x[:2]), big[:2])
    dw.set_noise_source(("philox", 17, 510))
    assert torch.equal(rev(x[510:]), big[510:])


# ---- 5. BASELINE configs[4]: shipped UNet, DDPM n = 5, + ResNeXt-29 --------------------------------------------------------
def test_config4_unet_ddpm_n5_resnext_matches_reference_golden_and_scales_to_b256(golden2, dev):
    from synth_convnets import CifarResNeXt, synth_init
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM
    from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
    unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
    dd = ImprovedDiffusionDDPM(unet, reverse_timestep=5)
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0)).to(dev)
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(6)]
    dd.set_noise_source(z[:5])                                              # q-sample draw + the four t > 0 draws
    got = dd(img)
    assert rel_err(got.cpu().numpy(), golden2["unetfull/ddpm_n5"]) < 2e-4
    clf = synth_init(CifarResNeXt(10), 0).to(dev)
    system = AcousticSystem(classifier=clf, transform=None, defender=dd, defense_type="spec").eval()
    dd.set_noise_source(z[:5])
    logits = system(img, True)
    assert rel_err(logits.cpu().numpy(), golden2["unetfull/ddpm_n5_logits"]) < 2e-3
    # B = 256 (the configuration's batch): rows 0-1 of the big run equal the 2-row run on the same noise, bit for bit
    B = 256
    big_img = torch.from_numpy(synth.uniform("meldbB", (B, 1, 32, 32), 6, -90.0, 30.0)).to(dev)
    big_img[:2] = img
    zb = [torch.from_numpy(synth.normal(f"szB{i}", (B, 1, 32, 32), 6)) for i in range(5)]
    for i in range(5):
        zb[i][:2] = z[i]
    dd.set_noise_source(zb)
    big = system(big_img, True)
    assert big.shape == (B, 10) and torch.isfinite(big).all()
    assert rel_err(big[:2].cpu().numpy(), golden2["unetfull/ddpm_n5_logits"]) < 2e-3


def test_rev_diffwave_sample_step_two_chains_the_purifier_like_the_reference(mini, dh, dev):
    """diffwave_sde.py:182-212: with sample_step = 2 the second pass purifies the FIRST pass's output and both are returned,
    concatenated along the batch axis; each pass consumes its own t + 1 draws."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from oracle import diffwave_oracle as O
    dw = DiffWave(model=mini, diffusion_hyperparams=dh, reverse_timestep=3)
    args = types.SimpleNamespace(t=3, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False, sample_step=2,
                                 ddpm_path=None, ddpm_config=None)
    rev = RevDiffWave.from_model(dw, args)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=7))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=7)) for d in range(8)]
    dw.set_noise_source(list(z))
    got = rev(x0.to(dev))
    assert got.shape == (4, 1, 16000) and dw._noise == []
    cfg = synth.mini_wavenet_config(64, 12, 12)
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    x1 = O.sde_purify(w, cfg, O.sde_tables(), x0, 3, z[:4])
    x2 = O.sde_purify(w, cfg, O.sde_tables(), x1, 3, z[4:])
    assert rel_err(got[:2].cpu().numpy(), x1.numpy()) < TOL_CHAIN
    assert rel_err(got[2:].cpu().numpy(), x2.numpy()) < 2 * TOL_CHAIN


# ---- 6. the constructors the eval scripts actually call: JSON config + {iter}.pkl checkpoint on disk ---------------------
def _write_checkpoint_and_config(tmp_path, cfg, seed=0):
    """What the scripts hand to the constructors: a ``.pkl`` holding ``model_state_dict`` (un-folded weight_g / weight_v
    tensors, the reference's state-dict key names) and a JSON with ``wavenet_config`` / ``diffusion_config``
    (diffwave_ddpm.py:395-411).  Both are generated from ``synth``; nothing of the reference's files is read."""
    import json
    ckpt = tmp_path / "1000000.pkl"
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, seed).items()},
                "optimizer_state_dict": {}, "iter": 1000000}, str(ckpt))
    conf = tmp_path / "config.json"
    conf.write_text(json.dumps({"wavenet_config": cfg, "diffusion_config": dict(synth.DIFFUSION_CONFIG),
                                "train_config": {"note": "ignored by the constructors"}}))
    return str(ckpt), str(conf)


def test_create_diffwave_model_from_files_one_shot_denoise_matches_reference_golden(golden, dev, tmp_path):
    """certified_robustness_eval.py:71-73 route: create_diffwave_model(model_path, config_path, reverse_timestep) ->
    one_shot_denoise, against the reference's own output on the same weights and input (full/one_shot_t25)."""
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave, create_diffwave_model
    ckpt, conf = _write_checkpoint_and_config(tmp_path, dict(synth.FULL_WAVENET_CONFIG))
    dw = create_diffwave_model(model_path=ckpt, config_path=conf, reverse_timestep=25)
    assert isinstance(dw, DiffWave) and dw.reverse_timestep == 25
    assert sum(p.numel() for p in dw.model.parameters()) == 24071681          # the shipped network
    assert next(dw.model.parameters()).device.type == "cuda"
    assert set(dw.diffusion_hyperparams) >= {"T", "Beta", "Alpha", "Alpha_bar", "Sigma"}
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234)).to(dev)
    assert rel_err(dw.one_shot_denoise(x0).cpu().numpy(), golden["full/one_shot_t25"]) < TOL_EVAL
    dw.reverse_timestep = 1                                                     # certification sets it per call
    assert rel_err(dw.one_shot_denoise(x0).cpu().numpy(), golden["full/one_shot_t1"]) < TOL_EVAL


def test_rev_diffwave_args_constructor_from_files_matches_sde_oracle(dev, tmp_path):
    """adaptive_attack_eval.py:99-100 route: RevDiffWave(args) with args.ddpm_path / args.ddpm_config on disk
    (diffwave_sde.py:138-161), then forward == audio_editing_sample, against the Euler oracle chain."""
    import argparse
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from oracle import diffwave_oracle as O
    cfg = synth.mini_wavenet_config(64, 12, 12)
    ckpt, conf = _write_checkpoint_and_config(tmp_path, cfg, seed=0)
    args = argparse.Namespace(ddpm_path=ckpt, ddpm_config=conf, t=4, score_type="guided_diffusion", rand_t=False,
                              t_delta=0, use_bm=False, sample_step=1)
    rev = RevDiffWave(args)
    assert rev._get_name() == "RevDiffWave" and rev.model.reverse_timestep == 4
    assert rev.rev_vpsde.noise_type == "diagonal" and rev.rev_vpsde.sde_type == "ito"
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=5))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=5)) for d in range(5)]
    rev.model.set_noise_source(list(z))
    got = rev(x0.to(dev))
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    ref = O.sde_purify(w, cfg, O.sde_tables(), x0, 4, z)
    assert got.shape == x0.shape
    assert rel_err(got.cpu().numpy(), ref.numpy()) < TOL_CHAIN


# ---- multi-GPU readiness: exercised by GPUTEST the day the box has two devices ---------------------------------
def test_two_rank_bench_over_rccl_reproduces_the_single_process_scores():
    """`python bench.py --gpus 2` as the driver runs it: the launcher starts two ranks as child processes (before anything touches
    a GPU, no re-exec), RCCL forms a 2-rank group, each rank binds its own device, and the gathered [16, 10] scores equal the
    single-process run's on the same global batch (inputs and Philox noise are keyed on the global utterance index).  Skipped
    on a one-GPU box -- which is every box this suite has seen so far."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on the node (torch.cuda.device_count() < 2)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-modes", "--no-other-configs", "--no-caller-shapes"]

    def run(gpus, batch):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--batch", str(batch)] + common,
                           capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    two, one = run(2, 8), run(1, 16)
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 16 and two["scaling"] == "weak"
    assert two["ranks"]["backend"] == "nccl" and two["ranks"]["rccl_ranks"] == 2 and two["ranks"]["distinct_devices"] == 2
    assert len({d.get("pci_bus_id") for d in two["ranks"]["devices"]}) == 2
    assert two["scores"]["shape"] == [16, 10] and two["scores"]["sha256"] == one["scores"]["sha256"]
