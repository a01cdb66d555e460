"""``AcousticSystem`` — the reference's dispatch (acoustic_system.py:5-53): a defender that acts either on the waveform
or on the spectrogram, an optional waveform -> spectrogram transform, then the classifier.  The members are the native
modules of this package (DiffWave / RevDiffWave / RevImprovedDiffusion, MelSpecDB, M5 / NativeConvNet / KWSModel), so
the whole defended forward runs in the HIP library; this class only orders the calls.

The eval scripts build the system from a classifier they un-pickled (audio_models/create_model.py:8-17) and a torchaudio
``Compose`` they assembled themselves (adaptive_attack_eval.py:83-93,129-137); ``__init__`` lowers both onto the native
modules (``audiopure_amd.lowering``), so the scripts run on the HIP path without edits."""
import torch

from . import _native as N
from .lowering import lower_classifier, lower_transform

_STAGE_OF = {"wave": 0, "spec": 1}        # where in the pipeline the defender sits


class AcousticSystem(torch.nn.Module):
    def __init__(self, classifier: torch.nn.Module, transform, defender: torch.nn.Module = None,
                 defense_type: str = "wave"):
        super().__init__()
        if defense_type not in _STAGE_OF:                                   # same refusal as acoustic_system.py:26-27
            raise NotImplementedError("argument defense_type should be 'wave' or 'spec'!")
        self.classifier, self.transform = lower_classifier(classifier), lower_transform(transform)
        self.defender, self.defense_type = defender, defense_type

    def _defends_at(self, stage: int, defend) -> bool:
        # `defend == True` on purpose: the scripts pass booleans, and anything else means "no defense" there too (:35,:45)
        return defend == True and self.defender is not None and _STAGE_OF.get(self.defense_type, -1) == stage  # noqa: E712

    @N.on_device
    def forward(self, x, defend=True):
        signal = self.defender(x) if self._defends_at(0, defend) else x                       # :35-38
        feats = signal if self.transform is None else self.transform(signal)                  # :41-42
        if self._defends_at(1, defend):                                                        # :45-48
            feats = self.defender(feats)
        return self.classifier(feats)                                                          # :51
