"""Golden vectors of the KWS classifier from the REFERENCE's own model class (audio_models/RCNN_KWS/model.py, loaded by
file path because the package __init__ imports librosa).  Run in the build container only; commits
tests/golden/golden_kws_v1.npz (inputs are regenerated from audiopure_amd.synth, weights are stored: 60k floats)."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiopure_amd import synth  # noqa: E402

spec = importlib.util.spec_from_file_location("kws_model_ref", "/root/reference/audio_models/RCNN_KWS/model.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

out = {}
for n_mels in (40, 32):
    torch.manual_seed(100 + n_mels)
    m = mod.KWSModel(in_size=n_mels).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        out[f"m{n_mels}/sd/{k}"] = v.numpy()
    for T in (81, 161, 47):                               # 1 s, 2 s and a short clip of mel frames
        x = torch.from_numpy(synth.uniform(f"kwsx/{n_mels}/{T}", (3, 1, n_mels, T), 1, -80.0, 20.0))
        with torch.no_grad():
            out[f"m{n_mels}/logp_T{T}"] = m(x).numpy()
    x1 = torch.from_numpy(synth.uniform(f"kwsx/{n_mels}/81", (3, 1, n_mels, 81), 1, -80.0, 20.0))[:1]
    with torch.no_grad():
        out[f"m{n_mels}/logp_T81_b1"] = m(x1).numpy()      # the B = 1 squeeze path (model.py:58,111-112)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_kws_v1.npz"), **out)
print("wrote", len(out), "arrays")
