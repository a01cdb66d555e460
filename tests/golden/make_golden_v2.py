#!/usr/bin/env python3
"""Round-2 golden vectors, produced by running the REFERENCE's own classes (imported from /root/reference; build
container only, CPU).  Nothing of the reference is stored: inputs are seed recipes of ``audiopure_amd.synth``, outputs
are the reference's numbers, and ``ref_m5_module.pt`` is ``torch.save`` of a reference-class ``M5Net.M5`` INSTANCE
(a pickle records the class by name -- ``M5Net.M5`` -- and the tensors; no source text).

    python tests/golden/make_golden_v2.py           # ~3 min on 8 cores

Contents of golden_v2.npz
  mini/diffusion_t20        DiffWave._diffusion           (diffwave_ddpm.py:49-73), reverse_timestep 20, injected noise
  mini/reverse_n4           DiffWave._reverse             (:75-104), 4 steps, 3 injected draws
  mini/fast_reverse_t{20,7} DiffWave.fast_reverse         (:106-141), K = 3 respaced steps, 3 injected draws
  full/fast_reverse_t25     the same on the shipped WaveNet configuration
  unetfull/ddpm_n5          GaussianDiffusion.q_sample + 5 x p_sample (gaussian_diffusion.py:188-206,356-387) on the
                            shipped 52.5 M-parameter UNet, mel-dB in / out (BASELINE configs[4])
  unetfull/ddpm_n5_logits   ... followed by the reference's ResNeXt-29 8x64d (models/resnext.py) on the result
ref_m5_module.pt            whole-module pickle of the reference's M5 with synth weights (what create_model() loads)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up sys.path, the third-party mocks, the no-op .cuda() and the noise injector)

from audiopure_amd import synth  # noqa: E402

torch.set_grad_enabled(False)


def noise_list(n, B, L, seed):
    return [torch.from_numpy(synth.noise(d, B, L, seed=seed)) for d in range(n)]


def main():
    out = {}
    dh = G.calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    mnet = G.build_ref_net(synth.mini_wavenet_config(64, 12, 12), seed=0)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=21))

    dw = G.DiffWave(model=mnet, diffusion_hyperparams=dh, reverse_timestep=20)
    G.INJ.queue = noise_list(1, 2, 16000, 21)
    out["mini/diffusion_t20"] = dw._diffusion(x0.clone()).numpy().copy()
    assert not G.INJ.queue

    dw = G.DiffWave(model=mnet, diffusion_hyperparams=dh, reverse_timestep=4)
    G.INJ.queue = noise_list(3, 2, 16000, 22)
    out["mini/reverse_n4"] = dw._reverse((x0 * 1.2).clone()).numpy().copy()
    assert not G.INJ.queue

    for ts in (20, 7):
        dw = G.DiffWave(model=mnet, diffusion_hyperparams=dh, reverse_timestep=ts)
        G.INJ.queue = noise_list(3, 2, 16000, 23)
        out[f"mini/fast_reverse_t{ts}"] = dw.fast_reverse((x0 * 1.1).clone()).numpy().copy()
        assert not G.INJ.queue

    fnet = G.build_ref_net(dict(synth.FULL_WAVENET_CONFIG), seed=0)
    xf = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    dw = G.DiffWave(model=fnet, diffusion_hyperparams=dh, reverse_timestep=25)
    G.INJ.queue = noise_list(3, 2, 16000, 24)
    out["full/fast_reverse_t25"] = dw.fast_reverse(xf.clone()).numpy().copy()
    assert not G.INJ.queue

    # --- whole-module pickle of the reference's M5 (audio_models/create_model.py:10 loads exactly this kind of file)
    m5 = G.build_ref_m5(10, seed=0)
    assert type(m5).__module__ == "M5Net" and type(m5).__name__ == "M5"
    torch.save(m5, os.path.join(HERE, "ref_m5_module.pt"))

    # --- shipped UNet, DDPM n = 5, then ResNeXt-29 (BASELINE configs[4])
    REF = G.REF
    sys.path.insert(0, os.path.join(REF, "audio_models/ConvNets_SpeechCommands"))
    sys.path.insert(1, os.path.join(G.ROOT, "tools"))
    from synth_convnets import synth_init
    from diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion import script_util as su
    import diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion.gaussian_diffusion as gd
    import models as ref_models
    d = su.model_and_diffusion_defaults()
    unet = synth_init(su.create_model(d["image_size"], d["num_channels"], d["num_res_blocks"], learn_sigma=d["learn_sigma"],
                                      class_cond=d["class_cond"], use_checkpoint=False,
                                      attention_resolutions=d["attention_resolutions"], num_heads=d["num_heads"],
                                      num_heads_upsample=d["num_heads_upsample"],
                                      use_scale_shift_norm=d["use_scale_shift_norm"], dropout=d["dropout"]), 0)
    diff = su.create_gaussian_diffusion(steps=d["diffusion_steps"], learn_sigma=d["learn_sigma"], sigma_small=d["sigma_small"],
                                        noise_schedule=d["noise_schedule"], use_kl=d["use_kl"],
                                        predict_xstart=d["predict_xstart"], rescale_timesteps=d["rescale_timesteps"],
                                        rescale_learned_sigmas=d["rescale_learned_sigmas"],
                                        timestep_respacing=d["timestep_respacing"])
    img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
    n = 5
    z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(n + 1)]   # p_sample draws at t = 0 too (masked)
    xq = diff.q_sample(2 * (img + 100.0) / 138.22 - 1, torch.tensor([n - 1, n - 1]), noise=z[0])
    queue = list(z[1:])
    gd.th.randn_like = lambda t_: queue.pop(0)
    for i in range(n - 1, -1, -1):
        xq = diff.p_sample(unet, xq, torch.tensor([i, i]))["sample"]
    assert not queue
    purified = (xq + 1) * 138.22 / 2 - 100.0
    out["unetfull/ddpm_n5"] = purified.numpy().copy()
    rx = synth_init(ref_models.create_model("resnext29_8_64", 10, 1), seed=0)
    out["unetfull/ddpm_n5_logits"] = rx(purified).numpy().copy()

    path = os.path.join(HERE, "golden_v2.npz")
    np.savez(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
