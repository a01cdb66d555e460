"""GPU parity of the lowered 2-D ConvNet classifiers: HIP conv-as-GEMM executor vs the
PyTorch-CPU module (the oracle for an arbitrary classifier is the module itself) and vs logits produced by the
REFERENCE's own model classes (tests/golden/golden_convnets_v1.npz)."""
import os

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from synth_convnets import FAMILIES, CifarResNeXt, synth_init, vgg19_bn
from audiopure_amd.convnet import NativeConvNet
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5          # exact fp32 products, different summation order, through up to ~50 layers


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_convnets_v1.npz"))


def _x(B=2):
    return torch.from_numpy(synth.uniform("mel", (B, 1, 32, 32), 3, -2.0, 2.0))


@pytest.mark.parametrize("name", sorted(FAMILIES))
def test_full_models_match_reference_golden(dev, gold, name):
    """All six families of models/__init__.py:8-45 on the HIP executor against logits of the REFERENCE's own classes."""
    m = synth_init(FAMILIES[name](), 0)
    assert set(m.state_dict()) == set(str(k) for k in gold[f"{name}/keys"])
    net = NativeConvNet(m).eval()
    y = net(_x().to(dev)).cpu().numpy()
    assert y.shape == (2, 10)
    assert rel_err(y, gold[f"{name}/logits"]) < TOL
    assert net._get_name() == type(m).__name__
    # batch-size independence of the plan + row independence (FAKEBOB reshapes scores per sample, _utils.py:118-119)
    y5 = net(_x(5).to(dev)).cpu().numpy()
    with torch.no_grad():
        ref5 = m(_x(5)).numpy()
    assert rel_err(y5, ref5) < TOL


@pytest.mark.parametrize("name", sorted(FAMILIES))
def test_lazily_wrapped_families_run_on_the_hip_kernels(dev, gold, name):
    """the route the eval scripts take -- a module wrapped on sight by lower_classifier (input shape unknown
    until the first call) -- ran NATIVELY for every family: conv launches counted by the library's own hook, the same logits
    as the reference's classes, no fallback warning (an error in this suite) and no fallback under AUDIOPURE_STRICT."""
    import ctypes as C
    from audiopure_amd import _native as N
    from audiopure_amd.lowering import lower_classifier
    lib = N.lib()
    m = synth_init(FAMILIES[name](), 0).to(dev).eval()
    net = lower_classifier(m)
    assert isinstance(net, NativeConvNet) and net.plan is None
    N.check(lib.ap_conv_profile_enable(1))
    try:
        y = net(_x().to(dev))
        torch.cuda.synchronize()
        ms, fl, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int64 * 8)()
        N.check(lib.ap_conv_profile_read(ms, fl, n, 8))
    finally:
        N.check(lib.ap_conv_profile_enable(0))
    n_conv = sum(1 for st in net.plan.steps if st.kind == "conv")
    assert net.plan is not None and not net._foreign and net.native_calls == 1
    assert sum(n) >= n_conv > 0, (list(n), n_conv)
    assert rel_err(y.cpu().numpy(), gold[f"{name}/logits"]) < TOL


def test_family_structures_match_module(dev):
    from test_convnet_lowering_cpu import DenseDPN
    m = synth_init(DenseDPN(), 2)
    x = _x(7)
    with torch.no_grad():
        ref = m(x).numpy()
    assert rel_err(NativeConvNet(m).eval()(x.to(dev)).cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("flags", [0, 0x100, 0x400])
def test_conv2d_primitive_edge_shapes(dev, flags):
    """ap_conv2d_fwd alone: grouped, strided, 1x1, Cout not a multiple of the tile, N not a multiple of the tile; with
    flags = AP_CONV_SPLIT / AP_CONV_SPLIT_F16 the eligible layers run on the bf16 / fp16 MFMA with split operands -- same
    tolerance."""
    import torch.nn.functional as F
    from audiopure_amd import _native as N
    lib = N.lib()
    # the last three reach the 128 x 128 kernels (>= 512 tiles): streamed-weight path incl. ragged M / N, groups,
    # stride 2, 1x1, and (Cin/g = 24) the LDS-staged one
    for (B, Cin, H, Cout, k, s, p, g) in [(3, 8, 9, 12, 3, 1, 1, 1), (2, 16, 16, 40, 3, 2, 1, 4), (5, 33, 7, 70, 1, 1, 0, 1),
                                          (1, 64, 4, 64, 3, 1, 1, 8), (4, 1, 32, 64, 3, 1, 1, 1),
                                          (33, 32, 32, 200, 3, 1, 1, 1), (70, 64, 31, 272, 3, 2, 1, 2),
                                          (40, 48, 30, 160, 1, 1, 0, 1), (36, 24, 32, 136, 3, 1, 1, 1),
                                          (7, 32, 5, 200, 3, 1, 1, 1),           # few tiles: the 128 x 64 variant
                                          (9, 128, 16, 176, 3, 1, 1, 2)]:        # Cout/g = 88: the 64 x 128 variant
        x = torch.from_numpy(synth.uniform(f"cx{Cin}{H}", (B, Cin, H, H), 1))
        w = torch.from_numpy(synth.uniform(f"cw{Cin}{Cout}", (Cout, Cin // g, k, k), 1))
        b = torch.from_numpy(synth.uniform(f"cb{Cout}", (Cout,), 1))
        ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p, groups=g))
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin // g, k, k, g), device=dev)
        N.check(lib.ap_conv2d_pack(N.ptr(wd), None, N.ptr(wT), Cout, Cin // g, k, k, g, N.stream()))
        out = torch.empty(ref.shape, device=dev)
        N.check(lib.ap_conv2d_fwd(N.ptr(xd), N.ptr(wT), N.ptr(bd), None, N.ptr(out), B, Cin, H, H, Cout, k, k, s, p, g,
                                  1 | flags, Cin, 0, N.stream()))
        assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-6, (B, Cin, H, Cout, k, s, p, g)


def test_mel_classifier_pipeline_end_to_end(dev):
    """AcousticSystem(classifier=ResNeXt, transform=mel-dB, defender=DiffWave) fully on the HIP path
    (the eval scripts' default wave-defense + spectrogram-classifier setup, adaptive_attack_eval.py:83-137)."""
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.transforms import MelSpecDB
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from oracle import diffwave_oracle as O
    cfg = synth.mini_wavenet_config(64, 12, 12)
    sd = synth.wavenet_state_dict(cfg, 0)
    wn = WaveNet_Speech_Commands(**cfg)
    wn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    dw = DiffWave(wn.to(dev), calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=2)
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=11)) for d in range(2)]
    dw.set_noise_source(list(z))
    clf = synth_init(CifarResNeXt(10, cardinality=4, base_width=8), 1)
    system = AcousticSystem(classifier=NativeConvNet(clf).eval(), transform=MelSpecDB(32), defender=dw, defense_type="wave")
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=11))
    logits = system(x0.to(dev), True).cpu()
    xp = O.ddpm_purify(O.fold_state_dict(sd), cfg, O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG), x0, 2, z)
    with torch.no_grad():
        ref = clf(O.melspec_db(xp))
    assert rel_err(logits.numpy(), ref.numpy()) < 5e-3        # mel dB amplifies tiny spectral differences


def test_split_k_and_few_output_paths_match_torch_and_are_deterministic(dev):
    """Round-3 paths of ap_conv2d_fwd: split-K for layers with too few output tiles (the UNet's 4 x 4 maps,
    improved_diffusion/unet.py:60-104 levels; partial sums in the caller's workspace, slices summed in order) with and
    without the workspace, with bias + residual; and the one-thread-per-pixel kernel for Cout <= 4 (the UNet's output
    convolution, unet.py:432-436)."""
    from audiopure_amd import _native as N
    lib = N.lib()
    torch.manual_seed(3)

    def conv(x, w, b, res, pad, use_ws):
        Cout, Cin, kh, kw = w.shape
        wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, kh, kw, 1), device=dev)
        N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, kh, kw, 1, N.stream()))
        if use_ws:
            N.use_conv_workspace(dev)
        else:
            N.check(lib.ap_conv2d_set_workspace(None, 0))
        B, _, H, W = x.shape
        out = torch.empty(B, Cout, H + 2 * pad - kh + 1, W + 2 * pad - kw + 1, device=dev)
        N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), N.ptr(res), N.ptr(out), B, Cin, H, W, Cout, kh, kw, 1, pad, 1, 0,
                                  Cin, 0, N.stream()))
        return out

    # 4 x 4 map, K = 256 * 9 = 2304 (144 chunks), 256 output channels: the split-K shape of BASELINE configs[4]
    x = torch.randn(64, 256, 4, 4, device=dev)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.03
    b = torch.randn(256, device=dev)
    res = torch.randn(64, 256, 4, 4, device=dev)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1) + res
    split_a, split_b, plain = conv(x, w, b, res, 1, True), conv(x, w, b, res, 1, True), conv(x, w, b, res, 1, False)
    assert torch.equal(split_a, split_b)                                     # slices summed in a fixed order
    assert rel_err(split_a.cpu().numpy(), ref.cpu().numpy()) < 5e-6
    assert rel_err(plain.cpu().numpy(), ref.cpu().numpy()) < 5e-6
    # Cout = 1 and 3: the few-output kernel, with padding and a ragged pixel count
    for cout in (1, 3):
        x = torch.randn(5, 128, 9, 7, device=dev)
        w = torch.randn(cout, 128, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        got = conv(x, w, b, None, 1, True)
        assert rel_err(got.cpu().numpy(), torch.nn.functional.conv2d(x, w, b, padding=1).cpu().numpy()) < 5e-6
    N.use_conv_workspace(dev)


def test_conv_output_written_as_a_channel_slice_is_bit_identical_and_touches_nothing_else(dev):
    """ap_conv2d_fwd_slice (the producer of h in `h = th.cat([h, hs.pop()], dim=1)`, improved_diffusion/unet.py:490-491, writes
    its half of the concatenation in place): the slice equals ap_conv2d_fwd's output bit for bit on the main 128 x 128 kernel, the
    pointwise 128 x 64 form and the split-K path (reduce kernel), the rest of the wider tensor is untouched; layers the
    streamed-weight kernel does not serve are refused."""
    from audiopure_amd import _native as N
    lib = N.lib()
    torch.manual_seed(5)
    N.use_conv_workspace(dev)
    for (B, Cin, H, W, Cout, k, extra, coff) in ((16, 128, 32, 32, 128, 3, 64, 0), (16, 256, 16, 16, 256, 1, 128, 0), (64, 256, 4, 4, 256, 3, 256, 0),
                                                (8, 64, 16, 16, 192, 3, 32, 16), (3, 32, 9, 7, 64, 1, 8, 8)):
        x = torch.randn(B, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        res = torch.randn(B, Cout, H, W, device=dev)
        wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, k, k, 1), device=dev)
        N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, k, k, 1, N.stream()))
        plain = torch.empty(B, Cout, H, W, device=dev)
        N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), N.ptr(res), N.ptr(plain), B, Cin, H, W, Cout, k, k, 1, k // 2, 1, 0, Cin, 0,
                                  N.stream()))
        wide = torch.full((B, Cout + extra, H, W), 7.0, device=dev)
        N.check(lib.ap_conv2d_fwd_slice(N.ptr(x), N.ptr(wT), N.ptr(b), N.ptr(res), N.ptr(wide), B, Cin, H, W, Cout, k, k, 1, k // 2, 1, 0, Cin, 0,
                                        Cout + extra, coff, N.stream()))
        assert torch.equal(wide[:, coff:coff + Cout], plain), (B, Cin, H, W, Cout, k)
        rest = torch.cat([wide[:, :coff], wide[:, coff + Cout:]], 1)
        assert bool((rest == 7.0).all())
    # Cin % 16 != 0: not a layer of the streamed-weight kernel -> refused, nothing written
    x = torch.randn(2, 24, 8, 8, device=dev)
    w = torch.randn(64, 24, 3, 3, device=dev)
    wT = torch.empty(lib.ap_conv2d_packed_elems(64, 24, 3, 3, 1), device=dev)
    N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), 64, 24, 3, 3, 1, N.stream()))
    wide = torch.zeros(2, 96, 8, 8, device=dev)
    assert lib.ap_conv2d_fwd_slice(N.ptr(x), N.ptr(wT), None, None, N.ptr(wide), 2, 24, 8, 8, 64, 3, 3, 1, 1, 1, 0, 24, 0, 96, 0, N.stream()) == -22
    assert float(wide.abs().max()) == 0.0


def test_conv3x3_minimal_filtering_kernel_matches_conv2d(dev):
    """ap_conv2d_fwd on the layers its F(2,3)-along-W kernel serves (3 x 3, stride 1, pad 1, ungrouped, Cin % 32 == 0, Cout % 128
    == 0, even W: the UNet's ResBlock convolutions, improved_diffusion/unet.py:150-197) vs torch's conv2d: both workgroup shapes
    (Cout % 256 == 0 and not), maps from 2 x 2 to 32 x 32, ragged pair-column tiles, bias / residual / ReLU, a channel-sliced input
    and a channel-sliced output."""
    import torch.nn.functional as F
    from audiopure_amd import _native as N
    lib = N.lib()
    # (the launcher takes this kernel from 512 tiles on: B sized so that every case has them -- small maps need many images)
    cases = [(2050, 32, 8, 8, 128, True, True, 1), (4100, 64, 4, 4, 256, True, False, 0), (260, 128, 32, 32, 128, False, True, 1), (258, 32, 16, 16, 256, True, True, 0),
             (33000, 32, 2, 2, 128, True, False, 0), (1101, 96, 6, 10, 384, True, True, 1), (2731, 32, 3, 4, 512, False, False, 1),
             # 32 .. 511 tiles: K slices into the split-K workspace + the reduce kernel (the UNet's 8 x 8 / 4 x 4 maps at B = 256)
             (256, 256, 8, 8, 256, True, True, 1), (256, 256, 4, 4, 256, True, False, 0), (300, 96, 8, 8, 128, False, True, 1), (301, 512, 4, 4, 256, True, True, 0)]
    N.use_conv_workspace(dev)
    for (B, Cin, H, W, Cout, has_b, has_r, relu) in cases:
        x = torch.from_numpy(synth.uniform(f"w3x{Cin}{H}{W}", (B, Cin + 5, H, W), 1)).to(dev)       # the conv reads channels [3, 3 + Cin)
        w = torch.from_numpy(synth.uniform(f"w3w{Cin}{Cout}", (Cout, Cin, 3, 3), 1)).to(dev) * 0.1
        b = torch.from_numpy(synth.uniform(f"w3b{Cout}", (Cout,), 1)).to(dev) if has_b else None
        r = torch.from_numpy(synth.uniform(f"w3r{Cout}{H}", (B, Cout, H, W), 1)).to(dev) if has_r else None
        ref = F.conv2d(x[:, 3:3 + Cin].double(), w.double(), None if b is None else b.double(), padding=1)
        if r is not None:
            ref = ref + r.double()
        if relu:
            ref = F.relu(ref)
        wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, 3, 3, 1), device=dev)
        N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, 3, 3, 1, N.stream()))
        out = torch.full((B, Cout, H, W), 9.0, device=dev)
        N.check(lib.ap_conv_profile_enable(1))
        N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), N.ptr(r), N.ptr(out), B, Cin, H, W, Cout, 3, 3, 1, 1, 1, relu, Cin + 5, 3, N.stream()))
        import ctypes as C
        ms, fl, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int64 * 8)()
        N.check(lib.ap_conv_profile_read(ms, fl, n, 8))
        N.check(lib.ap_conv_profile_enable(0))
        assert n[6] == 1, (list(n), "the F(2,3) kernel did not take this layer")
        assert rel_err(out.cpu().numpy(), ref.float().cpu().numpy()) < 3e-6, (B, Cin, H, W, Cout)
        wide = torch.full((B, Cout + 7, H, W), 4.0, device=dev)                                       # out = channels [2, 2 + Cout) of a wider tensor
        N.check(lib.ap_conv2d_fwd_slice(N.ptr(x), N.ptr(wT), N.ptr(b), N.ptr(r), N.ptr(wide), B, Cin, H, W, Cout, 3, 3, 1, 1, 1, relu, Cin + 5, 3,
                                        Cout + 7, 2, N.stream()))
        assert torch.equal(wide[:, 2:2 + Cout], out) and bool((wide[:, :2] == 4.0).all()) and bool((wide[:, 2 + Cout:] == 4.0).all())

