"""Host-side logic of the reference-shaped classes that needs no GPU: coefficient tables of the sampling chains,
state-dict compatibility with the reference's key names, loud failure without a device."""
import math
import types

import numpy as np
import pytest
import torch

from audiopure_amd import synth
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import (calc_diffusion_hyperparams,
                                                                      calc_diffusion_step_embedding)
from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
from audiopure_amd.diffusion_models.diffwave_sde import RevVPSDE
from audiopure_amd.acoustic_system import AcousticSystem
from audiopure_amd.audio_models.M5.M5Net import M5
from audiopure_amd import _native as N


def test_schedule_and_embedding_match_reference_golden(golden):
    dh = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        assert np.array_equal(dh[k].numpy(), golden[f"sched/{k}"])
    e = calc_diffusion_step_embedding(torch.from_numpy(golden["embed/steps"]), 128)
    assert np.array_equal(e.numpy(), golden["embed/out"])


def test_state_dict_keys_are_the_references():
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net = WaveNet_Speech_Commands(**cfg)
    ref_sd = synth.wavenet_state_dict(cfg, 0)          # reference key names / shapes (probe)
    assert list(net.state_dict().keys()) == list(ref_sd.keys()) and len(ref_sd) == 408
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == ref_sd[k].shape, k
    assert sum(p.numel() for p in net.parameters()) == 24071681
    m5 = M5(n_input=1, n_output=10)
    assert m5._get_name() == "M5" and sum(p.numel() for p in m5.parameters()) == 25290


def test_ddpm_chain_coefficients():
    dh = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    net = WaveNet_Speech_Commands(**synth.mini_wavenet_config(64, 2, 2))
    dw = DiffWave(net, dh, reverse_timestep=5)
    steps = dw._ddpm_steps(5)
    assert [s[0] for s in steps] == [4.0, 3.0, 2.0, 1.0, 0.0] and [s[4] for s in steps] == [1, 2, 3, 4, 0]
    for (t, ca, cb, cs, _), tt in zip(steps, range(4, -1, -1)):
        a, ab = float(dh["Alpha"][tt]), float(dh["Alpha_bar"][tt])
        assert math.isclose(ca, 1 / math.sqrt(a), rel_tol=1e-12)
        assert math.isclose(cb, -(1 - a) / math.sqrt(1 - ab) / math.sqrt(a), rel_tol=1e-12)
        assert cs == (float(dh["Sigma"][tt]) if tt > 0 else 0.0)


def test_sde_euler_coefficients_are_first_order_ddpm():
    dh = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    sde = RevVPSDE(model=None, score_type="guided_diffusion", beta_min=0.0001 * 200, beta_max=0.02 * 200, N=200)
    for (k, ca, cb, cs, draw), i in zip(sde.euler_steps(10), range(10)):
        kk = int(k)
        assert kk == 9 - i and draw == 1 + i
        beta = float(sde.discrete_betas[kk])
        assert math.isclose(ca, 1 + beta / 2) and math.isclose(cb, -beta / math.sqrt(1 - float(sde.alphas_cumprod[kk])))
        # same sigma as DDPM up to the cumprod-vs-loop table difference
        assert abs(cs - (float(dh["Sigma"][kk]) if kk > 0 else 0.0)) < 1e-6
        a = float(dh["Alpha"][kk])
        assert abs(ca - 1 / math.sqrt(a)) < 2e-4 * beta * 100


def test_no_device_no_fallback():
    net = WaveNet_Speech_Commands(**synth.mini_wavenet_config(64, 2, 2))
    with pytest.raises(N.NativeError):
        net((torch.zeros(1, 1, 256), torch.zeros(1, 1)))
    with pytest.raises(N.NativeError):
        M5(n_output=10).eval()(torch.zeros(1, 1, 16000))
    with pytest.raises(NotImplementedError):
        AcousticSystem(classifier=torch.nn.Identity(), transform=None, defender=None, defense_type="nope")
    with pytest.raises(NotImplementedError):
        M5(n_output=10)(torch.zeros(1, 1, 16000))       # training mode
    with pytest.raises(TypeError):
        DiffWave(torch.nn.Identity(), {}, 5)


def test_acoustic_system_dispatch_order():
    calls = []

    class Rec(torch.nn.Module):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def forward(self, x):
            calls.append(self.name)
            return x

    AcousticSystem(Rec("cls"), Rec("tr"), Rec("def"), "wave")(torch.zeros(1), True)
    AcousticSystem(Rec("cls"), Rec("tr"), Rec("def"), "spec")(torch.zeros(1), True)
    AcousticSystem(Rec("cls"), None, Rec("def"), "wave")(torch.zeros(1), False)
    assert calls == ["def", "tr", "cls", "tr", "def", "cls", "cls"]      # acoustic_system.py:35-51


def test_script_constructors_read_json_config_and_pkl_checkpoint(tmp_path):
    """The constructors the eval scripts call (diffwave_ddpm.py:395-411, diffwave_sde.py:138-161) on files written in the
    reference's formats -- JSON with wavenet_config / diffusion_config, .pkl with model_state_dict -- generated from synth
    (nothing of the reference's is read).  CPU: construction and parsing only; the forward needs the GPU
    (tests/test_gpu_dropin.py runs both constructors end to end)."""
    import argparse
    import json

    import torch

    from audiopure_amd import synth
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave, create_diffwave_model
    from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    sd = synth.wavenet_state_dict(cfg, 0)
    ckpt, conf = tmp_path / "1000000.pkl", tmp_path / "config.json"
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer_state_dict": {}}, str(ckpt))
    conf.write_text(json.dumps({"wavenet_config": cfg, "diffusion_config": dict(synth.DIFFUSION_CONFIG)}))
    dw = create_diffwave_model(str(ckpt), str(conf), reverse_timestep=25, device=torch.device("cpu"))
    assert isinstance(dw, DiffWave) and dw.reverse_timestep == 25
    assert sum(p.numel() for p in dw.model.parameters()) == 24071681
    got = dw.model.state_dict()
    assert set(got) == set(sd)
    for k in ("init_conv.0.conv.weight_g", "residual_layer.residual_blocks.35.skip_conv.weight_v", "final_conv.2.conv.weight"):
        assert torch.equal(got[k], torch.from_numpy(sd[k])), k
    assert float(dw.diffusion_hyperparams["Sigma"][0]) == pytest.approx(0.01)
    args = argparse.Namespace(ddpm_path=str(ckpt), ddpm_config=str(conf), t=4, score_type="guided_diffusion", rand_t=False,
                              t_delta=0, use_bm=False, sample_step=1)
    rev = RevDiffWave(args, device=torch.device("cpu"))
    assert rev._get_name() == "RevDiffWave" and rev.model.reverse_timestep == 4 and rev.T == 200
    assert rev.rev_vpsde.noise_type == "diagonal" and rev.rev_vpsde.sde_type == "ito"
    rev.rev_vpsde.audio_shape = (1, 8000)                                      # writable, as the scripts do
    with pytest.raises(KeyError):
        bad = tmp_path / "bad.pkl"
        torch.save({"state_dict": {}}, str(bad))
        create_diffwave_model(str(bad), str(conf), device=torch.device("cpu"))


def test_engine_walks_a_batch_in_calls_that_fit_the_free_memory(monkeypatch):
    """bf16 mode keeps one [B][L][C] bf16 gate image per layer of the skip group (ap_ctx_set_skip_group): 344 MB per clip-second
    at the shipped config.  Where a batch's workspace exceeds the free device memory the engine splits the batch (results do
    not depend on the split: noise is keyed on the global utterance index) -- it never changes the group size, which would
    change the fp32 summation order of skip."""
    import torch
    from audiopure_amd import _native as N, synth
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import NativeEngine
    eng = NativeEngine(dict(synth.FULL_WAVENET_CONFIG), N.AP_PREC_BF16)
    assert eng.skip_group == 36
    dev = torch.device("cpu")
    eng._sync_group()                                            # (the context learns the group size with the first workspace request)
    per_clip = eng.lib.ap_workspace_bytes(eng.ctx, 2, 16000) - eng.lib.ap_workspace_bytes(eng.ctx, 1, 16000)
    assert 340e6 < per_clip < 350e6                              # 49 MB of activations + 36 x 8.2 MB of gate images
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(100e9), int(288e9)))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda d=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d=None: 0)
    fit = eng.clips_that_fit(16000, dev)
    assert fit == int(0.7 * 100e9) // per_clip and 190 < fit < 210
    spans = list(eng.chunks(512, 16000, dev))
    assert spans[0] == (0, fit) and spans[-1][1] == 512 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert eng.skip_group == 36                                  # never traded for memory
    eng.skip_group = 0                                           # the fused form needs 49 MB per clip: one call
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(280e9), int(288e9)))
    assert list(eng.chunks(512, 16000, dev)) == [(0, 512)]
    f32 = NativeEngine(dict(synth.FULL_WAVENET_CONFIG), N.AP_PREC_F32)
    assert list(f32.chunks(700, 16000, dev)) == [(0, 512), (512, 700)]          # max_chunk still applies
