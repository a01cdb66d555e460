"""Drop-in for ``audio_models/create_model.py`` (the scripts do ``from audio_models.create_model import *`` and call
``create_model(args.classifier_path)``, adaptive_attack_eval.py:64-67).

Same contract -- un-pickle a whole classifier module, unwrap ``DataParallel``, ``float().eval()`` -- with two additions:
the pickle's ``M5Net.M5`` resolves to the native class (``dropin/M5Net.py`` is bound in ``sys.modules`` before the
reference's own directory can shadow it), and whatever comes out is lowered onto the HIP path
(``audiopure_amd.lowering.lower_classifier``: M5 / KWSModel -> native modules, 2-D ConvNets -> ``NativeConvNet``).
The reference's model directories still go on ``sys.path`` (create_model.py:4-6): the ConvNet pickles need the reference's
own ``models`` package to un-pickle; only their execution is replaced.
"""
import importlib
import os
import sys

import torch

from audiopure_amd.lowering import lower_classifier

__all__ = ["create_model", "torch", "sys"]

sys.modules.setdefault("M5Net", importlib.import_module("audiopure_amd.audio_models.M5.M5Net"))
for _d in ("./audio_models/RCNN_KWS", "./audio_models/M5", "./audio_models/ConvNets_SpeechCommands"):
    if os.path.isdir(_d) and _d not in sys.path:
        sys.path.insert(0, _d)


def create_model(path):
    model = torch.load(path, map_location="cpu", weights_only=False)      # whole-module pickle (trusted checkpoint)
    model = getattr(model, "module", model) if isinstance(model, torch.nn.DataParallel) else model
    assert isinstance(model, torch.nn.Module)
    return lower_classifier(model.float().eval())
