#!/usr/bin/env python3
"""Golden outputs of the REFERENCE's Improved-Diffusion UNetModel and of its continuous-beta RevVPSDE drift/diffusion
(imported from /root/reference; build container only) on seeded synthetic weights / inputs.

    python tests/golden/make_golden_unet.py
"""
import os
import sys
import warnings
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AUDIOPURE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
for m in ["torchvision", "torchvision.utils", "torchvision.datasets", "torchvision.transforms", "torchaudio", "torchsde",
          "librosa", "mpi4py", "blobfile", "torchaudio.transforms"]:
    sys.modules[m] = MagicMock()

from diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion.unet import UNetModel as RefUNet   # noqa: E402
from diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion.script_util import (                 # noqa: E402
    create_model, model_and_diffusion_defaults)
from audiopure_amd import synth                                                                                # noqa: E402
sys.path.insert(1, os.path.join(ROOT, "tools"))
from synth_convnets import synth_init                                                     # noqa: E402

torch.set_grad_enabled(False)
out = {}
x = torch.from_numpy(synth.uniform("specx", (2, 1, 32, 32), 3, -1.0, 1.0))

# mini UNet: 32 base channels, 1 res block per level, attention at 16x16 and 8x8 with 2 heads (ch/head = 16, 32)
mini = synth_init(RefUNet(in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(2, 4),
                          dropout=0.0, channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=True), 1)
out["mini/keys"] = np.array(list(mini.state_dict().keys()))
for t in (0.0, 37.0, 999.0):
    out[f"mini/eps_t{int(t)}"] = mini(x, torch.tensor([t, t])).numpy().copy()
out["mini/eps_tmixed"] = mini(x, torch.tensor([5.0, 600.0])).numpy().copy()

# shipped configuration (script_util.py:15-35): 128 ch, 3 res blocks, (1,2,2,2), attention 16,8, 4 heads; 52.5 M params
d = model_and_diffusion_defaults()
full = synth_init(create_model(d["image_size"], d["num_channels"], d["num_res_blocks"], learn_sigma=d["learn_sigma"],
                               class_cond=d["class_cond"], use_checkpoint=False, attention_resolutions=d["attention_resolutions"],
                               num_heads=d["num_heads"], num_heads_upsample=d["num_heads_upsample"],
                               use_scale_shift_norm=d["use_scale_shift_norm"], dropout=d["dropout"]), 0)
print("full params", sum(p.numel() for p in full.parameters()))
out["full/keys"] = np.array(list(full.state_dict().keys()))
out["full/eps_t37"] = full(x, torch.tensor([37.0, 37.0])).numpy().copy()

# the reference's continuous-beta RevVPSDE f / g on the mini net (torchsde stubbed)
from diffusion_models.improved_diffusion_sde import RevVPSDE                                                     # noqa: E402
sde = RevVPSDE(model=mini, score_type="guided_diffusion", img_shape=(1, 32, 32), model_kwargs=None)
xs = (x * 0.8).view(2, -1)
for tau in (0.0045, 0.02):
    s = torch.tensor([1.0 - tau])
    out[f"mini/sde_f_tau{tau}"] = sde.f(s, xs.clone()).numpy().copy()
    out[f"mini/sde_g_tau{tau}"] = sde.g(s, xs.clone()).numpy()[:, :4].copy()
# GaussianDiffusion DDPM chain of the reference (script_util.py:231-269 defaults: T=200, linear, eps-prediction, FIXED_LARGE):
# q_sample at t*-1 then p_sample for t = t*-1..0 on the mini UNet, noise injected by patching th.randn_like
from diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion import script_util as su              # noqa: E402
import diffusion_models.Improved_Diffusion_Unconditional.improved_diffusion.gaussian_diffusion as gd             # noqa: E402
dd = su.model_and_diffusion_defaults()
diff = su.create_gaussian_diffusion(steps=dd["diffusion_steps"], learn_sigma=dd["learn_sigma"], sigma_small=dd["sigma_small"],
                                    noise_schedule=dd["noise_schedule"], use_kl=dd["use_kl"], predict_xstart=dd["predict_xstart"],
                                    rescale_timesteps=dd["rescale_timesteps"], rescale_learned_sigmas=dd["rescale_learned_sigmas"],
                                    timestep_respacing=dd["timestep_respacing"])
img = torch.from_numpy(synth.uniform("meldb", (2, 1, 32, 32), 5, -90.0, 30.0))
z = [torch.from_numpy(synth.normal(f"sz{i}", (2, 1, 32, 32), 5)) for i in range(5)]
xq = diff.q_sample(2 * (img + 100.0) / 138.22 - 1, torch.tensor([3, 3]), noise=z[0])
queue = list(z[1:])
gd.th.randn_like = lambda t_: queue.pop(0)
for i in range(3, -1, -1):
    xq = diff.p_sample(mini, xq, torch.tensor([i, i]))["sample"]
out["mini/ddpm_t4"] = ((xq + 1) * 138.22 / 2 - 100.0).numpy().copy()

np.savez(os.path.join(HERE, "golden_unet_v1.npz"), **out)
print("wrote golden_unet_v1.npz", {k: v.shape for k, v in out.items() if "keys" not in k})
