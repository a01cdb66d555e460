"""GPU parity of the Improved-Diffusion UNet and the spectrogram purifier against the oracle and
the reference-generated golden vectors."""
import os
import types

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from synth_convnets import synth_init
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
    # every form of the one-pass kernel (a wave or the block per (sample, group) slab, 1 .. 16 vectors per thread), the slab
    # count not a multiple of the slabs per block, and the three-pass fallback (H W % 4 != 0, slab > 16384 floats)
    for (B, C_, H, W, G, act, use_ss) in ((3, 64, 2, 2, 32, 2, True), (5, 64, 4, 4, 32, 2, False), (3, 96, 8, 8, 32, 0, True),
                                          (2, 128, 16, 8, 32, 1, False), (2, 64, 32, 32, 32, 2, True), (2, 128, 32, 32, 32, 2, True),
                                          (2, 256, 32, 32, 32, 0, False), (1, 384, 32, 32, 32, 2, True), (1, 96, 64, 64, 32, 2, True),
                                          (1, 64, 64, 48, 8, 2, False), (2, 64, 5, 5, 32, 2, True), (1, 32, 96, 96, 4, 2, True)):
        x = torch.from_numpy(synth.uniform(f"gnx{C_}{H}{W}", (B, C_, H, W), 1, -2, 2)) + 0.7
        g, b = torch.from_numpy(synth.uniform("gng", (C_,), 1, 0.5, 1.5)), torch.from_numpy(synth.uniform("gnb", (C_,), 1))
        ss = torch.from_numpy(synth.uniform("gns", (B, 2 * C_), 1))
        ref = F.group_norm(x, G, g, b, 1e-5)
        if use_ss:
            ref = ref * (1 + ss[:, :C_, None, None]) + ss[:, C_:, None, None]
        ref = ref * torch.sigmoid(ref) if act == 2 else ref.relu() if act == 1 else ref
        xd, gd, bd, sd_ = x.to(dev), g.to(dev), b.to(dev), ss.to(dev)
        y = torch.full_like(xd, float("nan"))
        N.check(lib.ap_groupnorm_nchw(N.ptr(xd), N.ptr(gd), N.ptr(bd), N.ptr(sd_) if use_ss else None, N.ptr(y), B, C_, H * W, G, 1e-5,
                                      act, N.stream()))
        assert rel_err(y.cpu().numpy(), ref.numpy()) < 3e-6, (B, C_, H, W, G, act, use_ss)
    # a 4-byte-aligned (not 16-byte-aligned) input takes the three-pass kernel: same result
    flat = torch.zeros(2 * 64 * 8 * 8 + 1, device=dev)
    xm = flat[1:].view(2, 64, 8, 8)
    xm.copy_(torch.from_numpy(synth.uniform("gnmis", (2, 64, 8, 8), 1, -2, 2)))
    g64, b64 = torch.from_numpy(synth.uniform("gng", (64,), 1, 0.5, 1.5)).to(dev), torch.from_numpy(synth.uniform("gnb", (64,), 1)).to(dev)
    ym = torch.empty(2, 64, 8, 8, device=dev)
    N.check(lib.ap_groupnorm_nchw(N.ptr(xm), N.ptr(g64), N.ptr(b64), None, N.ptr(ym), 2, 64, 64, 32, 1e-5, 2, N.stream()))
    refm = F.group_norm(xm.cpu(), 32, g64.cpu(), b64.cpu(), 1e-5)
    assert rel_err(ym.cpu().numpy(), (refm * torch.sigmoid(refm)).numpy()) < 3e-6
    # channel-slice copy (torch.cat / slices): the 16-byte 2-D form and the element form (H W = 25, an unaligned offset)
    for (B, C_, HW, scs, sco, dcs, dco) in ((3, 8, 64, 8, 0, 20, 12), (2, 5, 1024, 9, 3, 5, 0), (2, 6, 25, 7, 1, 9, 2), (2, 3, 6, 4, 1, 7, 3),
                                            (4, 10, 1, 16, 4, 12, 1), (2, 8, 1, 16, 4, 12, 4)):
        src = torch.from_numpy(synth.uniform(f"cp{C_}{HW}", (B, scs, HW), 1)).to(dev)
        dst = torch.zeros((B, dcs, HW), device=dev)
        N.check(lib.ap_copy_channels(N.ptr(src), N.ptr(dst), B, C_, HW, scs, sco, dcs, dco, N.stream()))
        ref = torch.zeros_like(dst)
        ref[:, dco:dco + C_] = src[:, sco:sco + C_]
        assert torch.equal(dst, ref), (B, C_, HW, scs, sco, dcs, dco)
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


def test_full_unet_with_in_place_concatenation_is_bit_identical_to_the_copying_form(dev):
    """At inference the block that produces h writes it straight into the buffer of `th.cat([h, hs.pop()], dim=1)`
    (improved_diffusion/unet.py:490-491; ap_conv2d_fwd_slice); with a tape (forward_save, the gradient's forward pass) both halves
    are copied.  Same kernels, same arithmetic: the two evaluations must agree bit for bit (batch 1: the slice view is
    "contiguous" there; batch 3: it is not)."""
    full = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
    for B in (1, 3):
        x = torch.from_numpy(synth.uniform(f"ipc{B}", (B, 1, 32, 32), 1, -1, 1)).to(dev)
        t = torch.tensor([37.0] * B)
        with torch.no_grad():
            direct = full(x, t)
            taped, _ = full.forward_save(x, t)
        assert torch.equal(direct, taped), B


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


# ---- input gradient (white-box attack through the DiffSpec purifier) ------------------------------------------------------
def test_backward_primitives_match_autograd(dev):
    import torch.nn.functional as F
    from audiopure_amd import _native as N
    lib = N.lib()
    for act, use_ss in ((2, True), (0, False), (2, False)):
        x = torch.from_numpy(synth.uniform("gnx", (3, 64, 8, 8), 1, -2, 2)).requires_grad_(True)
        g, b = torch.from_numpy(synth.uniform("gng", (64,), 1, 0.5, 1.5)), torch.from_numpy(synth.uniform("gnb", (64,), 1))
        ss = torch.from_numpy(synth.uniform("gns", (3, 128), 1))
        dy = torch.from_numpy(synth.uniform("gndy", (3, 64, 8, 8), 1))
        ref = F.group_norm(x, 32, g, b, 1e-5)
        if use_ss:
            ref = ref * (1 + ss[:, :64, None, None]) + ss[:, 64:, None, None]
        if act == 2:
            ref = ref * torch.sigmoid(ref)
        (ref * dy).sum().backward()
        xd, gd, bd, sd_, dyd = x.detach().to(dev), g.to(dev), b.to(dev), ss.to(dev), dy.to(dev)
        dx = torch.empty_like(xd)
        N.check(lib.ap_groupnorm_bwd(N.ptr(xd), N.ptr(gd), N.ptr(bd), N.ptr(sd_) if use_ss else None, N.ptr(dyd), N.ptr(dx), 3, 64, 64,
                                     32, 1e-5, act, N.stream()))
        assert rel_err(dx.cpu().numpy(), x.grad.numpy()) < 1e-5, (act, use_ss)
    for ch, T, heads in ((16, 256, 2), (64, 64, 4), (32, 100, 1), (64, 256, 3), (8, 17, 2)):
        qkv = torch.from_numpy(synth.uniform(f"qkv{ch}", (2, heads * 3 * ch, T), 1, -1.5, 1.5)).requires_grad_(True)
        do = torch.from_numpy(synth.uniform(f"do{ch}", (2, heads * ch, T), 1))
        q, k, v = torch.split(qkv.reshape(2 * heads, 3 * ch, T), ch, dim=1)
        w = torch.softmax(torch.einsum("bct,bcs->bts", q * ch ** -0.25, k * ch ** -0.25), dim=-1)
        ref = torch.einsum("bts,bcs->bct", w, v).reshape(2, heads * ch, T)
        (ref * do).sum().backward()
        qd, od, dod = qkv.detach().to(dev), ref.detach().to(dev), do.to(dev)
        dq = torch.empty_like(qd)
        stats = torch.empty(2 * heads * T * 3, device=dev)
        N.check(lib.ap_attention_qkv_bwd(N.ptr(qd), N.ptr(od), N.ptr(dod), N.ptr(dq), N.ptr(stats), 2, heads * ch, T, heads, N.stream()))
        assert rel_err(dq.cpu().numpy(), qkv.grad.numpy()) < 1e-5, (ch, T, heads)
    dy = torch.from_numpy(synth.uniform("updy", (6, 10, 14), 1)).to(dev)
    dx = torch.empty((6, 5, 7), device=dev)
    N.check(lib.ap_upsample_nearest2x_bwd(N.ptr(dy), N.ptr(dx), 6, 5, 7, N.stream()))
    assert torch.allclose(dx, dy.reshape(6, 5, 2, 7, 2).sum(dim=(2, 4)), atol=1e-6)


def _unet_vjp_case(dev, model_cpu, t, mode="f32", tol=2e-4):
    from oracle import unet_oracle as U
    x = _x()
    v = torch.from_numpy(synth.uniform("unetv", tuple(x.shape), 3))
    xr = x.clone().requires_grad_(True)
    (U.unet_forward(model_cpu, xr, torch.tensor([float(t)] * x.shape[0]), grad=True) * v).sum().backward()
    import copy
    m = copy.deepcopy(model_cpu).to(dev).set_precision(mode)
    xg = x.to(dev).requires_grad_(True)
    out = m(xg, torch.tensor([float(t)] * x.shape[0]))
    (out * v.to(dev)).sum().backward()
    with torch.no_grad():
        assert torch.equal(out.detach(), m(x.to(dev), torch.tensor([float(t)] * x.shape[0])))
    err = rel_err(xg.grad.cpu().numpy(), xr.grad.numpy())
    assert err < tol, (t, mode, err)


@pytest.mark.parametrize("mode", ["f32", "f32s", "f32h"])
def test_mini_unet_input_gradient_matches_autograd_of_the_oracle(dev, mode):
    for t in (0, 37, 999):
        _unet_vjp_case(dev, mini_unet(), t, mode)


def test_full_unet_input_gradient_matches_autograd_of_the_oracle(dev):
    _unet_vjp_case(dev, synth_init(create_model(**model_and_diffusion_defaults()), 0), 37, tol=5e-4)


def test_white_box_gradient_through_the_spectrogram_purifier(dev):
    """d loss / d mel through RevImprovedDiffusion (chain of Euler links, each J^T on the HIP path) vs autograd through
    the oracle's restatement of the same chain."""
    from oracle import unet_oracle as U
    args = types.SimpleNamespace(t=3, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(4)]
    v = torch.from_numpy(synth.uniform("specv", (2, 1, 32, 32), 3))
    ir = img.clone().requires_grad_(True)
    (U.spec_sde_purify(mini_unet(), ir, 3, z, grad=True) * v).sum().backward()
    rev = RevImprovedDiffusion.from_model(mini_unet().to(dev), args)
    rev.set_noise_source(z)
    ig = img.to(dev).requires_grad_(True)
    out = rev(ig)
    (out * v.to(dev)).sum().backward()
    rev.set_noise_source(z)
    with torch.no_grad():
        assert rel_err(out.detach().cpu().numpy(), rev(img.to(dev)).cpu().numpy()) < 1e-6
    assert rel_err(ig.grad.cpu().numpy(), ir.grad.numpy()) < 5e-4


def test_white_box_loss_gradient_end_to_end_with_the_diffspec_defense(dev):
    """adaptive_attack_eval.py --defense DiffSpec --attack PGD: cross_entropy(AcousticSystem(ResNeXt, mel32,
    RevImprovedDiffusion, 'spec')(x), y).backward() on the HIP path, checked by a central difference of the native loss."""
    import torch.nn.functional as F
    from synth_convnets import CifarResNeXt
    from audiopure_amd.convnet import NativeConvNet
    from audiopure_amd.transforms import MelSpecDB
    from audiopure_amd.acoustic_system import AcousticSystem
    args = types.SimpleNamespace(t=2, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    rev = RevImprovedDiffusion.from_model(mini_unet().to(dev), args)
    clf = NativeConvNet(synth_init(CifarResNeXt(10), 1).eval()).eval()
    system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=rev, defense_type="spec")
    x = torch.from_numpy(synth.waveforms(2, 16000, seed=14)).to(dev)
    y = torch.tensor([1, 8], device=dev)
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(3)]
    rev.set_noise_source(z)
    delta = torch.zeros_like(x, requires_grad=True)
    F.cross_entropy(system(x + delta, True), y).backward()
    g = delta.grad
    assert g.shape == x.shape and torch.isfinite(g).all() and float(g.abs().max()) > 0
    eps = 2e-4
    with torch.no_grad():
        d = g.sign()
        rev.set_noise_source(z); lp = F.cross_entropy(system(x + eps * d, True), y)
        rev.set_noise_source(z); lm = F.cross_entropy(system(x - eps * d, True), y)
    fd = (lp - lm).item() / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd - an) < 0.15 * abs(an) + 1e-3, (fd, an)


def test_spec_rev_vpsde_drift_and_diffusion_match_oracle(dev):
    """improved_diffusion_sde.RevVPSDE.f / g (torchsde's callbacks, :118-136) with the score from one native UNet
    evaluation, against the oracle's sde_f_g at a few reference times; RevImprovedDiffusion exposes it as .rev_vpsde."""
    from oracle import unet_oracle as U
    from audiopure_amd.diffusion_models.improved_diffusion_sde import RevVPSDE
    m_cpu = mini_unet()
    args = types.SimpleNamespace(t=3, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
    rev = RevImprovedDiffusion.from_model(mini_unet().to(dev), args)
    sde = rev.rev_vpsde
    assert isinstance(sde, RevVPSDE) and sde.noise_type == "diagonal" and sde.sde_type == "ito"
    x = torch.from_numpy(synth.uniform("sdex", (2, 1, 32, 32), 4, -1.0, 1.0))
    for tau in (0.004, 0.0125, 0.25):
        f_ref, g_ref = U.sde_f_g(m_cpu, x, torch.tensor(tau, dtype=torch.float32))
        s = torch.tensor([1.0 - tau], dtype=torch.float32)
        f = sde.f(s, x.to(dev).reshape(2, -1))
        g = sde.g(s, x.to(dev).reshape(2, -1))
        assert f.shape == (2, 1024) and g.shape == (2, 1024)
        assert rel_err(f.cpu().numpy().reshape(x.shape), f_ref.numpy()) < 1e-4, tau
        assert abs(float(g[0, 0]) - float(g_ref)) < 1e-6 * max(1.0, float(g_ref))
