"""gradient of a loss through the purifier with respect to the audio, HIP path vs torch autograd
through the CPU oracle (the oracle's eps-network is plain differentiable torch ops)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiopure_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _net(cfg, dev, seed=0):
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    sd = synth.wavenet_state_dict(cfg, seed)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.to(dev), sd


@pytest.mark.parametrize("C_,NL,L,step", [(64, 12, 1000, 3.0), (128, 5, 777, 0.0), (256, 3, 640, 17.0)])
def test_eps_vjp_matches_oracle_autograd(dev, C_, NL, L, step):
    """J_eps(x, t)^T v for a random cotangent v: every backward piece (final conv, 3 GEMMs per block incl. dilations
    1..2048 >= L, gate derivative, init conv) against autograd through the oracle."""
    from oracle import diffwave_oracle as O
    from audiopure_amd.diffusion_models._grad import EpsGrad
    cfg = synth.mini_wavenet_config(C_, NL, 12)
    net, sd = _net(cfg, dev, seed=5)
    w = O.fold_state_dict(sd)
    B = 2
    x = torch.from_numpy(synth.waveforms(B, L, seed=11))
    v = torch.from_numpy(synth.uniform(f"v{L}", (B, 1, L), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    eps_ref = O.eps_net(w, cfg, xr, torch.full((B, 1), step))
    (g_ref,) = torch.autograd.grad(eps_ref, xr, v)
    eg = EpsGrad(net)
    eps, saved = eg.forward_save(x.to(dev), step)
    assert rel_err(eps.cpu().numpy(), eps_ref.detach().numpy()) < 2e-5
    g = eg.backward(saved, v.to(dev))
    assert rel_err(g.cpu().numpy(), g_ref.numpy()) < 1e-4


def test_eps_vjp_is_the_same_with_kept_and_with_recomputed_pre_gate_activations(dev):
    """ap_resblock_fwd_save keeps y = DilConv(u) + b per layer; without it the backward recomputes y with ap_conv2d_fwd.
    Same gradient either way (the fused block and the conv kernel sum K = 3C in different orders: 1e-5), and the chain
    falls back level by level when SAVE_BUDGET_BYTES is short (full -> layer inputs only -> recompute the link)."""
    from audiopure_amd.diffusion_models import _grad as G
    cfg = synth.mini_wavenet_config(256, 3, 12)
    net, _ = _net(cfg, dev, seed=7)
    B, L, step = 2, 1100, 4.0
    x = torch.from_numpy(synth.waveforms(B, L, seed=13)).to(dev)
    v = torch.from_numpy(synth.uniform("vk", (B, 1, L), 1, -1.0, 1.0)).to(dev)
    eg = G.EpsGrad(net)
    eps_a, saved_a = eg.forward_save(x, step)
    eps_b, saved_b = eg.forward_save(x, step, acts=False)
    assert saved_a[3] is not None and saved_a[3].shape == (3, B, 512, L) and saved_b[3] is None
    # (kept activations: the direct-form block with the pre-gate store; lean: the F(2,3) block -- the same eps to fp32 rounding)
    assert rel_err(eps_a.cpu().numpy(), eps_b.cpu().numpy()) < 2e-6
    ga, gb = eg.backward(saved_a, v), eg.backward(saved_b, v)
    assert rel_err(ga.cpu().numpy(), gb.cpu().numpy()) < 1e-5
    # chain level: three links under a budget that holds (a) everything, (b) one full link + lean ones, (c) nothing
    steps = [(2.0, 1.01, -0.02, 0.01, 1), (1.0, 1.01, -0.02, 0.01, 2), (0.0, 1.0, -0.01, 0.0, 0)]
    zs = [torch.from_numpy(synth.noise(dr, B, L, seed=14)).to(dev) for dr in range(3)]
    full = G._saved_bytes(saved_a)
    grads = []
    old = G.SAVE_BUDGET_BYTES
    try:
        for budget in (old, full + 2 * G._saved_bytes(saved_a[:3]) + 1, 1):
            G.SAVE_BUDGET_BYTES = budget
            xg = x.clone().requires_grad_(True)
            out = G.differentiable_chain(net, xg, steps, 0.9, 0.1, zs)
            (out * v).sum().backward()
            grads.append(xg.grad.cpu().numpy())
    finally:
        G.SAVE_BUDGET_BYTES = old
    assert rel_err(grads[1], grads[0]) < 1e-5 and rel_err(grads[2], grads[0]) < 1e-5


@pytest.mark.parametrize("mode", ["f32s"])
def test_eps_vjp_in_the_split_modes_matches_oracle_autograd(dev, mode):
    """The same check with the network in a split-operand mode (fused block forward in that mode, the backward GEMMs on
    ap_conv2d_fwd with the matching AP_CONV_SPLIT / AP_CONV_SPLIT_F16 flag): same tolerances as fp32."""
    from oracle import diffwave_oracle as O
    from audiopure_amd.diffusion_models._grad import EpsGrad
    cfg = synth.mini_wavenet_config(256, 4, 12)
    net, sd = _net(cfg, dev, seed=6)
    net.set_precision(mode)
    w = O.fold_state_dict(sd)
    B, L, step = 2, 900, 9.0
    x = torch.from_numpy(synth.waveforms(B, L, seed=12))
    v = torch.from_numpy(synth.uniform(f"vs{L}", (B, 1, L), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    eps_ref = O.eps_net(w, cfg, xr, torch.full((B, 1), step))
    (g_ref,) = torch.autograd.grad(eps_ref, xr, v)
    eg = EpsGrad(net)
    eps, saved = eg.forward_save(x.to(dev), step)
    assert rel_err(eps.cpu().numpy(), eps_ref.detach().numpy()) < 2e-5
    g = eg.backward(saved, v.to(dev))
    assert rel_err(g.cpu().numpy(), g_ref.numpy()) < 1e-4


def test_white_box_gradient_through_rev_diffwave_matches_oracle(dev):
    """loss(classifier-free surrogate: sum of w * purified) differentiated w.r.t. the audio through RevDiffWave's Euler
    chain (t* = 3), as white_box_attack.py:392,437-439 does; reference = autograd through the oracle's SDE chain."""
    from oracle import diffwave_oracle as O
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net, sd = _net(cfg, dev, seed=2)
    w = O.fold_state_dict(sd)
    dh = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    t_star, B, L = 3, 2, 1200
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=t_star)
    args = types.SimpleNamespace(t=t_star, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    runner = RevDiffWave.from_model(dw, args)
    zs = [torch.from_numpy(synth.noise(d, B, L, seed=9)) for d in range(t_star + 1)]
    wgt = torch.from_numpy(synth.uniform("lossw", (B, 1, L), 1, -1.0, 1.0))
    x = torch.from_numpy(synth.waveforms(B, L, seed=9))
    # HIP path
    dw.set_noise_source([z.clone() for z in zs])
    xd = x.to(dev).requires_grad_(True)
    out = runner(xd)
    loss = (out * wgt.to(dev)).sum()
    loss.backward()
    # oracle: the same Euler links with autograd
    xr = x.clone().requires_grad_(True)
    steps = runner.rev_vpsde.euler_steps(t_star)
    a = float(runner.rev_vpsde.alphas_cumprod[t_star - 1].double())
    cur = np.sqrt(a) * xr + np.sqrt(1.0 - a) * zs[0].reshape(B, 1, L)
    for (t, ca, cb, cs, draw) in steps:
        eps = O.eps_net(w, cfg, cur, torch.full((B, 1), float(t)))
        cur = ca * cur + cb * eps + (cs * zs[draw].reshape(B, 1, L) if cs != 0.0 else 0.0)
    loss_ref = (cur * wgt).sum()
    loss_ref.backward()
    assert rel_err(out.detach().cpu().numpy(), cur.detach().numpy()) < 1e-4
    # the gradient passes through two ReLU masks (init conv, final conv): forward differences of 1e-6 flip a mask at
    # isolated samples, so the max-norm tolerance is looser than the typical agreement
    ga, gr = xd.grad.cpu().numpy(), xr.grad.numpy()
    assert rel_err(ga, gr) < 1e-3
    assert float(np.median(np.abs(ga - gr))) < 2e-6 * float(np.abs(gr).max())
    # DiffWave.forward is no_grad in the reference (diffwave_ddpm.py:41-43): a detached output, as there
    dw.set_noise_source([z.clone() for z in zs])
    assert not dw(x.to(dev).requires_grad_(True)).requires_grad


def _m5_torch(sd, x, stride=16, eps=1e-5):
    """M5.forward (M5Net.py:20-38) in differentiable torch ops (the oracle's m5_forward without its no_grad)."""
    import torch.nn.functional as F
    t = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    h = x
    for i, s in ((1, stride), (2, 1), (3, 1), (4, 1)):
        h = F.conv1d(h, t[f"conv{i}.weight"], t[f"conv{i}.bias"], stride=s)
        h = F.batch_norm(h, t[f"bn{i}.running_mean"], t[f"bn{i}.running_var"], t[f"bn{i}.weight"], t[f"bn{i}.bias"],
                         training=False, eps=eps)
        h = F.max_pool1d(F.relu(h), 4)
    h = F.avg_pool1d(h, h.shape[-1]).view(h.size(0), -1)
    return F.log_softmax(F.linear(h, t["fc1.weight"], t["fc1.bias"]), dim=1)


@pytest.mark.parametrize("L", [16000, 8000])
def test_m5_input_gradient_matches_torch_autograd(dev, L):
    from audiopure_amd.audio_models.M5.M5Net import M5
    sd = synth.m5_state_dict(10)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m5 = m5.to(dev).eval()
    B = 3
    x = torch.from_numpy(synth.waveforms(B, L, seed=4))
    v = torch.from_numpy(synth.uniform("m5v", (B, 10), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    (g_ref,) = torch.autograd.grad(_m5_torch(sd, xr), xr, v)
    xd = x.to(dev).requires_grad_(True)
    lp = m5(xd)
    (g,) = torch.autograd.grad(lp, xd, v.to(dev))
    ga, gr = g.cpu().numpy(), g_ref.numpy()
    assert rel_err(ga, gr) < 1e-3                      # max-pool / ReLU selections can flip at isolated samples
    assert float(np.median(np.abs(ga - gr))) < 2e-6 * float(np.abs(gr).max())


def test_white_box_loss_gradient_end_to_end(dev):
    """nll_loss(AcousticSystem(M5, RevDiffWave)(x), y).backward() reaches the audio entirely on the HIP path: the call
    the PGD attack makes (white_box_attack.py:392,437-439)."""
    import torch.nn.functional as F
    from oracle import diffwave_oracle as O
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from audiopure_amd.audio_models.M5.M5Net import M5
    from audiopure_amd.acoustic_system import AcousticSystem
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net, sd = _net(cfg, dev, seed=2)
    w = O.fold_state_dict(sd)
    dh = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    t_star, B, L = 2, 2, 16000
    dw = DiffWave(model=net, diffusion_hyperparams=dh, reverse_timestep=t_star)
    args = types.SimpleNamespace(t=t_star, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    runner = RevDiffWave.from_model(dw, args)
    m5sd = synth.m5_state_dict(10)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in m5sd.items()})
    m5 = m5.to(dev).eval()
    system = AcousticSystem(classifier=m5, transform=None, defender=runner, defense_type="wave")
    zs = [torch.from_numpy(synth.noise(d, B, L, seed=3)) for d in range(t_star + 1)]
    x = torch.from_numpy(synth.waveforms(B, L, seed=3))
    y = torch.tensor([3, 7])
    dw.set_noise_source([z.clone() for z in zs])
    xd = x.to(dev).requires_grad_(True)
    loss = F.nll_loss(system(xd, True), y.to(dev))
    loss.backward()
    xr = x.clone().requires_grad_(True)
    a = float(runner.rev_vpsde.alphas_cumprod[t_star - 1].double())
    cur = np.sqrt(a) * xr + np.sqrt(1.0 - a) * zs[0].reshape(B, 1, L)
    for (t, ca, cb, cs, draw) in runner.rev_vpsde.euler_steps(t_star):
        eps = O.eps_net(w, cfg, cur, torch.full((B, 1), float(t)))
        cur = ca * cur + cb * eps + (cs * zs[draw].reshape(B, 1, L) if cs != 0.0 else 0.0)
    loss_ref = F.nll_loss(_m5_torch(m5sd, cur), y)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    ga, gr = xd.grad.cpu().numpy(), xr.grad.numpy()
    assert rel_err(ga, gr) < 2e-3
    assert float(np.median(np.abs(ga - gr))) < 1e-5 * float(np.abs(gr).max())


def test_pgd_step_and_eot_calls_of_the_reference_attack_work_unchanged(dev):
    """The exact call pattern of white_box_attack.py:380-440 (delta leaf, x_pert = x + delta, model(x_pert),
    criterion(...).backward(), delta.grad) and of _EOT.py:27-52 (repeat, retain_grad, backward(ones), .grad of the
    repeated batch) against the native AcousticSystem."""
    import torch.nn.functional as F
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from audiopure_amd.audio_models.M5.M5Net import M5
    from audiopure_amd.acoustic_system import AcousticSystem
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net, _ = _net(cfg, dev, seed=2)
    dw = DiffWave(model=net, diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=2)
    runner = RevDiffWave.from_model(dw, types.SimpleNamespace(t=2, rand_t=False, t_delta=0, use_bm=False, sample_step=1,
                                                              score_type="guided_diffusion"))
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.m5_state_dict(10).items()})
    model = AcousticSystem(classifier=m5.to(dev).eval(), transform=None, defender=runner, defense_type="wave")
    B, L = 2, 16000
    x = torch.from_numpy(synth.waveforms(B, L, seed=12)).to(dev)
    y = torch.tensor([2, 5], device=dev)
    dw.set_noise_source(("philox", 3, 0))
    # PGD iteration
    delta = torch.zeros_like(x, requires_grad=True)
    y_pert = model(x + delta)
    loss = F.cross_entropy(y_pert, y)
    loss.backward()
    g1 = delta.grad.clone()
    assert g1.shape == x.shape and torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    delta.data = delta.data + 0.001 * g1.sign()
    # EOT with use_grad=True, EOT_batch_size = 3
    x_pert = x + delta
    rep = x_pert.repeat(3, 1, 1)
    rep.retain_grad()
    scores = model(rep)
    loss_eot = F.cross_entropy(scores, y.repeat(3), reduction="none")
    loss_eot.backward(torch.ones_like(loss_eot))
    g_eot = rep.grad.view(3, -1, 1, L).mean(0)
    assert g_eot.shape == x.shape and torch.isfinite(g_eot).all() and float(g_eot.abs().max()) > 0
    # same Philox key and the same utterance indices for the first B rows -> the first replica reproduces a plain call
    delta2 = delta.detach().clone().requires_grad_(True)
    F.cross_entropy(model(x + delta2), y, reduction="sum").backward()
    assert rel_err(rep.grad[:B].cpu().numpy(), delta2.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("L,n_mels", [(16000, 32), (5000, 40)])
def test_mel_front_end_input_gradient_matches_torch_autograd(dev, L, n_mels):
    """d/dx of MelSpecDB (mode 0) against autograd through the oracle's torch.stft restatement."""
    from oracle import diffwave_oracle as O
    from audiopure_amd.transforms import MelSpecDB
    B = 2
    x = torch.from_numpy(synth.waveforms(B, L, seed=31))
    frames = 1 + L // 512
    v = torch.from_numpy(synth.uniform(f"melv{L}", (B, 1, n_mels, frames), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    (g_ref,) = torch.autograd.grad(O.melspec_db(xr, n_mels=n_mels), xr, v)
    xd = x.to(dev).requires_grad_(True)
    out = MelSpecDB(n_mels)(xd)
    (g,) = torch.autograd.grad(out, xd, v.to(dev))
    assert g.shape == x.shape
    assert rel_err(g.cpu().numpy(), g_ref.numpy()) < 2e-4


@pytest.mark.parametrize("family", ["vgg19_bn", "resnext29"])
def test_convnet_input_gradient_matches_torch_autograd(dev, family):
    """NativeConvNet backward (transposed convs incl. groups and stride 2, BN folded, ReLU / residual / max- and
    avg-pool) against torch autograd through the same eval-mode module on the CPU."""
    from synth_convnets import CifarResNeXt, vgg19_bn, synth_init
    from audiopure_amd.convnet import NativeConvNet
    torch.manual_seed(0)
    mod = synth_init(vgg19_bn(10) if family == "vgg19_bn" else CifarResNeXt(10), 3).eval()
    net = NativeConvNet(mod).eval()
    B = 2
    x = torch.from_numpy(synth.uniform(f"cg/{family}", (B, 1, 32, 32), 1, -2.0, 2.0))
    v = torch.from_numpy(synth.uniform("cgv", (B, 10), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    (g_ref,) = torch.autograd.grad(mod(xr), xr, v)
    xd = x.to(dev).requires_grad_(True)
    out = net(xd)
    (g,) = torch.autograd.grad(out, xd, v.to(dev))
    ga, gr = g.cpu().numpy(), g_ref.numpy()
    # the bulk agrees to fp32 rounding (VGG median 2.5e-7 of max); what differs is where a ReLU / max-pool selection of
    # the two forward passes differs by an ulp and flips -- isolated in VGG, spread by ResNeXt's 29 layers of 3x3 convs
    d = np.abs(ga - gr) / float(np.abs(gr).max())
    assert d.max() < 2e-2 and np.percentile(d, 99) < 5e-3 and np.median(d) < 1e-4, (d.max(), np.percentile(d, 99), np.median(d))
    # and the native gradient is the derivative of the native forward: central difference along sign(g)
    with torch.no_grad():
        dirn = g.sign()
        eps = 1e-4
        fd = float(((net((x.to(dev) + eps * dirn)) - net((x.to(dev) - eps * dirn))) * v.to(dev)).sum()) / (2 * eps)
    an = float((g * dirn).sum())
    assert abs(fd - an) < 2e-2 * abs(an), (fd, an)


def test_white_box_gradient_through_mel_and_spectrogram_classifier(dev):
    """The default route of adaptive_attack_eval.py (classifier_input mel32, --attack PGD, --defense Diffusion):
    cross_entropy(AcousticSystem(ResNeXt, mel32, RevDiffWave)(x), y).backward() entirely on the HIP path."""
    import torch.nn.functional as F
    from synth_convnets import CifarResNeXt, synth_init
    from audiopure_amd.convnet import NativeConvNet
    from audiopure_amd.transforms import MelSpecDB
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    cfg = synth.mini_wavenet_config(64, 12, 12)
    net, _ = _net(cfg, dev, seed=2)
    dw = DiffWave(model=net, diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=2)
    runner = RevDiffWave.from_model(dw, types.SimpleNamespace(t=2, rand_t=False, t_delta=0, use_bm=False, sample_step=1,
                                                              score_type="guided_diffusion"))
    clf = NativeConvNet(synth_init(CifarResNeXt(10), 1).eval()).eval()
    system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=runner, defense_type="wave")
    x = torch.from_numpy(synth.waveforms(2, 16000, seed=14)).to(dev)
    y = torch.tensor([1, 8], device=dev)
    dw.set_noise_source(("philox", 4, 0))
    delta = torch.zeros_like(x, requires_grad=True)
    loss = F.cross_entropy(system(x + delta, True), y)
    loss.backward()
    g = delta.grad
    assert g.shape == x.shape and torch.isfinite(g).all() and float(g.abs().max()) > 0
    # directional derivative check of the whole native chain (finite difference along the gradient's sign)
    eps = 2e-4
    with torch.no_grad():
        d = g.sign()
        dw.set_noise_source(("philox", 4, 0)); lp = F.cross_entropy(system(x + eps * d, True), y)
        dw.set_noise_source(("philox", 4, 0)); lm = F.cross_entropy(system(x - eps * d, True), y)
    fd = (lp - lm).item() / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd - an) < 0.15 * abs(an) + 1e-3, (fd, an)


class _CatSliceNet(torch.nn.Module):
    """DenseNet / DPN-style plumbing in miniature: torch.cat, channel slices feeding different consumers, a value used
    twice, grouped stride-2 conv, 1x1 convs, avg- and max-pool with padding, BatchNorm without a preceding conv."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        self.c1 = nn.Conv2d(1, 24, 3, padding=1, bias=False)
        self.b1 = nn.BatchNorm2d(24)
        self.c2 = nn.Conv2d(24, 16, 3, padding=1, bias=True)
        self.b2 = nn.BatchNorm2d(40)
        self.c3 = nn.Conv2d(40, 32, 3, stride=2, padding=1, groups=4, bias=False)
        self.b3 = nn.BatchNorm2d(32)
        self.c4 = nn.Conv2d(16, 16, 1, bias=False)
        self.c5 = nn.Conv2d(32, 48, 3, stride=2, padding=1, bias=False)
        self.fc = nn.Linear(48, 10)

    def forward(self, x):
        import torch.nn.functional as F
        a = F.relu(self.b1(self.c1(x)))
        b = self.c2(a)
        cat = torch.cat([a, b], 1)                                   # 40 channels
        d = F.relu(self.b3(self.c3(F.relu(self.b2(cat)))))           # 32 @ 16x16
        e = torch.cat([d[:, :16] + self.c4(d[:, 16:]), d[:, 16:]], 1)
        f = F.max_pool2d(e, 3, stride=2, padding=1)                  # 32 @ 8x8
        g = F.relu(self.c5(f))                                       # 48 @ 4x4
        h = F.avg_pool2d(g, 4)
        return self.fc(h.view(h.size(0), -1))


def test_convnet_backward_covers_cat_slices_and_pools(dev):
    from synth_convnets import synth_init
    from audiopure_amd.convnet import NativeConvNet
    mod = synth_init(_CatSliceNet(), 9).eval()
    net = NativeConvNet(mod).eval()
    x = torch.from_numpy(synth.uniform("csn", (3, 1, 32, 32), 1, -2.0, 2.0))
    v = torch.from_numpy(synth.uniform("csnv", (3, 10), 1, -1.0, 1.0))
    xr = x.clone().requires_grad_(True)
    out_ref = mod(xr)
    (g_ref,) = torch.autograd.grad(out_ref, xr, v)
    xd = x.to(dev).requires_grad_(True)
    out = net(xd)
    assert rel_err(out.detach().cpu().numpy(), out_ref.detach().numpy()) < 1e-5
    (g,) = torch.autograd.grad(out, xd, v.to(dev))
    d = np.abs(g.cpu().numpy() - g_ref.numpy()) / float(np.abs(g_ref.numpy()).max())
    assert d.max() < 2e-2 and np.median(d) < 1e-5, (d.max(), np.median(d))


def _guarded(shape, dev, dtype=torch.float32, fill=5.0, guard=2048):
    """A tensor inside guard bands of a sentinel: (buffer, view, check) -- check() asserts nothing was written outside the view."""
    n = 1
    for v in shape:
        n *= v
    buf = torch.full((n + 2 * guard,), 7.25, device=dev, dtype=dtype)
    view = buf[guard:guard + n].view(*shape)
    view.fill_(fill)

    def check():
        assert bool((buf[:guard] == 7.25).all()) and bool((buf[guard + n:] == 7.25).all()), "write outside the output buffer"
    return buf, view, check


@pytest.mark.parametrize("L,layer", [(1100, 0), (2048, 5), (1000, 11), (16000, 6), (130, 3), (77, 9), (5, 1), (129, 6), (4133, 7), (16000, 1),
                                     (1, 0), (64, 4), (4100, 10)])
def test_fused_block_backward_matches_autograd_through_the_oracle(dev, L, layer):
    """ap_resblock_fwd_save + ap_resblock_bwd (one block's backward as two fused launches) at the shipped
    width: the kept pre-gate activations equal the oracle's y = DilConv(u) + b, the save-forward equals the plain forward bit
    for bit, and dh_in equals torch autograd through the oracle's Residual_block.forward (WaveNet.py:75-97) for random
    cotangents on h' and skip_n -- every dilation class, d >= L, ragged and tiny clips."""
    import torch.nn.functional as F
    from oracle import diffwave_oracle as O
    from audiopure_amd import _native as N
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    w = O.fold_state_dict(sd)
    eng = net.engine()
    lib = eng.lib
    B, C_, d = 2, 256, 2 ** (layer % 12)
    assert lib.ap_resblock_bwd_available(eng.ctx, B, L) == 1
    N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))      # the backward's own weight images: built once per load, never by a launch
    h = torch.from_numpy(synth.uniform(f"gh/{L}", (B, C_, L), 1, -1.5, 1.5))
    gh = torch.from_numpy(synth.uniform(f"gg/{L}", (B, C_, L), 2, -1.0, 1.0))
    gs = torch.from_numpy(synth.uniform(f"gs/{L}", (B, C_, L), 3, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    p = f"residual_layer.residual_blocks.{layer}"
    hr = h.clone().requires_grad_(True)
    h_ref, s_ref = O.residual_block(w, layer, d, hr, emb)
    (g_ref,) = torch.autograd.grad([h_ref, s_ref], hr, [gh, gs])
    with torch.no_grad():
        part_t = F.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
        y_ref = F.conv1d(h + part_t.view(1, -1, 1), w[p + ".dilated_conv_layer.conv.weight"], w[p + ".dilated_conv_layer.conv.bias"],
                         dilation=d, padding=d)
    hd, pt = h.to(dev), part_t.to(dev).contiguous()
    hout, hout2 = torch.empty_like(hd), torch.empty_like(hd)
    sk, sk2 = torch.zeros_like(hd), torch.zeros_like(hd)
    pre = torch.full((B, 2 * C_, L), 9.0, device=dev)
    N.check(lib.ap_resblock_fwd_save(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), N.ptr(pre), 0, B, L, N.stream()))
    N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout2), N.ptr(sk2), 0, B, L, N.stream()))
    assert torch.equal(hout, hout2) and torch.equal(sk, sk2)
    assert rel_err(hout.cpu().numpy(), h_ref.detach().numpy()) < 5e-6
    assert rel_err(pre.cpu().numpy(), y_ref.numpy()) < 5e-6
    _, dy, dy_ok = _guarded(pre.shape, dev)                      # both outputs inside guard bands that must stay untouched
    _, dh_in, dh_ok = _guarded(hd.shape, dev)
    ghd, gsd = gh.to(dev), gs.to(dev)                            # (held: a temporary's block would be reused by the next .to())
    N.check(lib.ap_resblock_bwd(eng.ctx, layer, N.ptr(ghd), N.ptr(gsd), N.ptr(pre), N.ptr(dy), N.ptr(dh_in), B, L, N.stream()))
    assert rel_err(dh_in.cpu().numpy(), g_ref.numpy()) < 1e-5
    dy_ok(); dh_ok()
    first = dh_in.clone()
    N.check(lib.ap_resblock_bwd(eng.ctx, layer, N.ptr(ghd), N.ptr(gsd), N.ptr(pre), N.ptr(dy), N.ptr(dh_in), B, L, N.stream()))
    assert torch.equal(first, dh_in)                             # run to run bit-identical


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("L,layer", [(1100, 0), (2048, 5), (1000, 11), (16000, 6), (130, 3), (77, 9), (64, 4), (4100, 10), (16000, 1),
                                     (5, 1), (1, 0), (129, 6), (4133, 7), (127, 2), (128, 8)])
def test_bf16_block_backward_matches_autograd_through_the_bf16_oracle(dev, L, layer):
    """ap_resblock_bwd_bf16 (bf16 MFMA operands, the dilated conv recomputed from the layer input) against
    torch autograd through the bf16-emulating oracle block (both forward GEMMs see bf16-rounded operands; its backward multiplies
    the same rounded weights with fp32 cotangents): cosine >= 0.999 and max deviation <= 2e-2 of the largest gradient entry."""
    from oracle import diffwave_oracle as O
    from audiopure_amd import _native as N
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net, sd = _net(cfg, dev, seed=3)
    net.set_precision("bf16")
    w = O.fold_state_dict(sd)
    eng = net.engine()
    lib = eng.lib
    B, C_, d = 2, 256, 2 ** (layer % 12)
    assert lib.ap_resblock_bwd_bf16_available(eng.ctx, B, L) == 1
    N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
    h = torch.from_numpy(synth.uniform(f"gh/{L}", (B, C_, L), 1, -1.5, 1.5))
    gh = torch.from_numpy(synth.uniform(f"gg/{L}", (B, C_, L), 2, -1.0, 1.0))
    gs = torch.from_numpy(synth.uniform(f"gs/{L}", (B, C_, L), 3, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    p = f"residual_layer.residual_blocks.{layer}"
    hr = h.clone().requires_grad_(True)
    h_ref, s_ref = O.residual_block(w, layer, d, hr, emb, bf16_operands=True)
    (g_ref,) = torch.autograd.grad([h_ref, s_ref], hr, [gh, gs])
    with torch.no_grad():
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
    hd, ptd, ghd, gsd = h.to(dev), part_t.to(dev).contiguous(), gh.to(dev), gs.to(dev)
    _, dy, dy_ok = _guarded((B, L, 2 * C_), dev, torch.bfloat16, 0.0)   # both outputs inside guard bands that must stay untouched
    _, dh_in, dh_ok = _guarded(hd.shape, dev)
    N.check(lib.ap_resblock_bwd_bf16(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ghd), N.ptr(gsd), dy.data_ptr(), N.ptr(dh_in), B, L, N.stream()))
    got = dh_in.cpu()
    assert torch.isfinite(got).all()
    assert _cos(got, g_ref) >= 0.999, (_cos(got, g_ref),)
    assert rel_err(got.numpy(), g_ref.numpy()) <= 2e-2
    dy_ok(); dh_ok()
    N.check(lib.ap_resblock_bwd_bf16(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ghd), N.ptr(gsd), dy.data_ptr(), N.ptr(dh_in), B, L, N.stream()))
    assert torch.equal(got, dh_in.cpu())                         # run to run bit-identical
    # the same gradient from KEPT gate factors (round 6): ap_resblock_fwd_gate_save writes them beside a bit-identical h' / g image,
    # ap_resblock_bwd_bf16_saved reads them instead of recomputing the dilated conv -- held to the same bars, and close to the
    # recomputing form (the factors are the forward's fast-exp gate quantities at fp16 instead of a compensated exp at fp32)
    ho, ho2 = torch.empty_like(hd), torch.empty_like(hd)
    gi, gi2 = (torch.empty((B, L, C_), dtype=torch.bfloat16, device=dev) for _ in range(2))
    fac = torch.zeros(lib.ap_gate_factor_bytes(B, L), dtype=torch.uint8, device=dev)
    N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ho), gi.data_ptr(), fac.data_ptr(), B, L, N.stream()))
    N.check(lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ho2), gi2.data_ptr(), B, L, N.stream()))
    assert torch.equal(ho, ho2) and torch.equal(gi.view(torch.int16), gi2.view(torch.int16))
    _, dy2, dy2_ok = _guarded((B, L, 2 * C_), dev, torch.bfloat16, 0.0)
    _, dh2, dh2_ok = _guarded(hd.shape, dev)
    N.check(lib.ap_resblock_bwd_bf16_saved(eng.ctx, layer, fac.data_ptr(), N.ptr(ghd), N.ptr(gsd), 0, dy2.data_ptr(), N.ptr(dh2), B, L, N.stream()))
    got2 = dh2.cpu()
    assert torch.isfinite(got2).all()
    assert _cos(got2, g_ref) >= 0.999, (_cos(got2, g_ref),)
    assert rel_err(got2.numpy(), g_ref.numpy()) <= 2e-2
    assert rel_err(got2.numpy(), got.numpy()) <= 5e-3
    dy2_ok(); dh2_ok()
    # dskip handed over once as the bf16 image [B][L][S] (ap_bwd_bf16_rows_image): what the staging rounds it to anyway -> bit-identical
    dsk = torch.empty((B, L, C_), dtype=torch.bfloat16, device=dev)
    N.check(lib.ap_bwd_bf16_rows_image(N.ptr(gsd), dsk.data_ptr(), B, C_, L, N.stream()))
    assert torch.equal(dsk.float().cpu(), gs.to(torch.bfloat16).float().permute(0, 2, 1))
    dh3 = torch.empty_like(hd)
    N.check(lib.ap_resblock_bwd_bf16_saved(eng.ctx, layer, fac.data_ptr(), N.ptr(ghd), dsk.data_ptr(), 1, dy2.data_ptr(), N.ptr(dh3), B, L, N.stream()))
    assert torch.equal(dh3.cpu(), got2)
    gi3 = torch.empty_like(gi)                                   # h_out = NULL (the net's last layer): same g image, same factors
    fac3 = torch.zeros_like(fac)
    N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), None, gi3.data_ptr(), fac3.data_ptr(), B, L, N.stream()))
    assert torch.equal(gi3.view(torch.int16), gi.view(torch.int16)) and torch.equal(fac3, fac)


def test_bf16_eps_vjp_runs_on_the_bf16_backward_and_matches_the_bf16_oracle(dev):
    """The whole eps VJP in bf16 mode, both forms: the forward keeping the gate's derivative factors + ap_resblock_bwd_bf16_saved per
    layer (default since round 6), and the forward keeping layer inputs + ap_resblock_bwd_bf16 (the dilated conv recomputed).
    (1) Against the composed fp32 backward on the SAME stored layer inputs -- the backward kernels' own error through six layers:
    cosine >= 0.9999, max deviation <= 2e-2 of the largest entry (the per-block bar).
    (2) Against autograd through the bf16-emulating oracle network: cosine >= 0.999.  The largest deviation is NOT held to 2e-2
    here: two implementations of the same bf16 arithmetic differ in accumulation order, operands on a rounding boundary flip,
    and this net's input gradient amplifies that (tools/check_bwd_bf16.py: the composed fp32 backward sits at the same 3.8e-2
    relative L2 from the oracle as the fused one); the bound that holds by construction is the distance between the oracle's own
    fp32 and bf16 gradients, and that is what is asserted."""
    from oracle import diffwave_oracle as O
    from audiopure_amd.diffusion_models._grad import EpsGrad
    cfg = synth.mini_wavenet_config(256, 6, 12)
    net, sd = _net(cfg, dev, seed=6)
    net.set_precision("bf16")
    w = O.fold_state_dict(sd)
    B, L, step = 2, 1500, 3.0
    x = torch.from_numpy(synth.waveforms(B, L, seed=11))
    v = torch.from_numpy(synth.uniform(f"v{L}", (B, 1, L), 1, -1.0, 1.0))
    refs = {}
    for bf in (True, False):
        xr = x.clone().requires_grad_(True)
        eps_ref = O.eps_net(w, cfg, xr, torch.full((B, 1), step), bf16_operands=bf)
        (refs[bf],) = torch.autograd.grad(eps_ref, xr, v)
    eg = EpsGrad(net)
    xd, vd = x.to(dev), v.to(dev)
    # round 6: by default the forward pass keeps the gate's derivative factors (an opaque uint8 image per layer) and only the
    # first layer's input; the backward reads them instead of recomputing the dilated conv
    eps, saved = eg.forward_save(xd, step)
    assert saved[3] is not None and saved[3].dtype == torch.uint8 and saved[0].shape[0] == 3
    assert torch.equal(eps, eg.eps_only(xd, step))               # the saving forward runs the chain's own deferred-skip form: same eps, bit for bit
    g_kept = eg.backward(saved, vd).cpu()
    eg.keep_gate_factors = False                                 # the round-5 form: every layer's input kept, the dilated conv recomputed
    eps2, saved2 = eg.forward_save(xd, step)
    assert saved2[3] is None and torch.equal(eps2, eps)
    g = eg.backward(saved2, vd).cpu()
    eg.fused_bf16 = False
    g_fp32 = eg.backward(saved2, vd).cpu()                       # same layer inputs, fp32 GEMMs
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    for name, gg in (("kept factors", g_kept), ("recomputing", g)):
        assert _cos(gg, g_fp32) >= 0.9999 and rel_err(gg.numpy(), g_fp32.numpy()) <= 2e-2, (name, _cos(gg, g_fp32), rel_err(gg.numpy(), g_fp32.numpy()))
        assert _cos(gg, refs[True]) >= 0.999, (name, _cos(gg, refs[True]))
        assert l2(gg, refs[True]) <= l2(refs[False], refs[True]), (name, l2(gg, refs[True]), l2(refs[False], refs[True]))


def test_backward_launches_refuse_to_run_before_the_images_are_built(dev):
    """the backward launch functions allocate nothing (include/audiopure.h: device memory is allocated in
    ap_ctx_load_wavenet / ap_ctx_prepare_backward / ap_m5_create only; launch functions are hipGraph-capturable).  Before
    ap_ctx_prepare_backward -- and again after the weights are re-loaded -- they return -22 and say what to call."""
    from audiopure_amd import _native as N
    cfg = synth.mini_wavenet_config(256, 12, 12)
    for mode, fn in (("f32", "ap_resblock_bwd"), ("bf16", "ap_resblock_bwd_bf16")):
        net, _ = _net(cfg, dev, seed=3)
        net.set_precision(mode)
        eng = net.engine()
        lib = eng.lib
        B, L = 1, 256
        t = torch.zeros((B, 512, L), device=dev)
        h = torch.zeros((B, 256, L), device=dev)
        pt = torch.zeros(256, device=dev)
        o = torch.empty_like(h)

        def call():
            if mode == "f32":
                return lib.ap_resblock_bwd(eng.ctx, 0, N.ptr(h), N.ptr(h), N.ptr(t), N.ptr(t), N.ptr(o), B, L, N.stream())
            return lib.ap_resblock_bwd_bf16(eng.ctx, 0, N.ptr(h), N.ptr(pt), N.ptr(h), N.ptr(h), t.data_ptr(), N.ptr(o), B, L, N.stream())
        assert call() == -22 and b"ap_ctx_prepare_backward" in lib.ap_last_error()
        N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
        N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))      # a second call is a no-op
        assert call() == 0
        with torch.no_grad():
            net.init_conv[0].conv.bias.add_(1.0)                       # parameters changed: engine() re-loads, the images are stale
        eng2 = net.engine()
        assert eng2 is eng
        assert call() == -22
        N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
        assert call() == 0
