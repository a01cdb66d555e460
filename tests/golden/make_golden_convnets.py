#!/usr/bin/env python3
"""Golden logits of the REFERENCE's own 2-D ConvNet classifiers (imported from /root/reference; build container
only) on seeded synthetic weights / inputs, plus a check that every family lowers (audiopure_amd.convnet.lower) to a
plan that reproduces the reference module.  Stores inputs' recipes + outputs only.

    python tests/golden/make_golden_convnets.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AUDIOPURE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "audio_models/ConvNets_SpeechCommands"))
warnings.filterwarnings("ignore")

import models as ref_models                                            # noqa: E402  (reference package)
from audiopure_amd import synth                                        # noqa: E402
sys.path.insert(1, os.path.join(ROOT, "tools"))
from synth_convnets import synth_init             # noqa: E402
from audiopure_amd.convnet import lower                                # noqa: E402
from oracle.convnet_plan_oracle import run_plan_torch                  # noqa: E402

torch.set_grad_enabled(False)
out = {}
x = torch.from_numpy(synth.uniform("mel", (2, 1, 32, 32), 3, -2.0, 2.0))
for name in ["vgg19_bn", "resnet50", "wideresnet28_10", "resnext29_8_64", "dpn92", "densenet_bc_100_12"]:
    m = synth_init(ref_models.create_model(name, 10, 1), seed=0)
    y = m(x)
    plan = lower(m)
    yp = run_plan_torch(plan, x)
    err = float((y - yp).abs().max() / y.abs().max())
    kinds = {}
    for s in plan.steps:
        kinds[s.kind] = kinds.get(s.kind, 0) + 1
    print(f"{name:22s} params {sum(p.numel() for p in m.parameters()):>9d}  |logit|max {float(y.abs().max()):.3f}  "
          f"plan-vs-module rel err {err:.2e}  steps {kinds}")
    assert err < 1e-4, name
    out[f"{name}/logits"] = y.numpy().copy()
    out[f"{name}/keys"] = np.array(list(m.state_dict().keys()))
np.savez(os.path.join(HERE, "golden_convnets_v1.npz"), **out)
print("wrote golden_convnets_v1.npz")
