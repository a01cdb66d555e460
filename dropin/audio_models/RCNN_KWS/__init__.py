"""Drop-in for the reference's ``audio_models.RCNN_KWS`` (kws_adaptive_attack_eval.py:70-75): the native ``KWSModel``
plus whatever else the reference's package exports (``config``, the Qualcomm dataset classes), taken from the reference's
own files when they are importable (they need librosa)."""
import os
import sys
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
for _d in __path__[1:]:                       # the reference's package __init__ puts its directory on sys.path (:5-6)
    if os.path.isdir(_d) and _d not in sys.path:
        sys.path.insert(0, _d)
try:
    from config import *  # noqa: F401,F403
except Exception:  # pragma: no cover - reference checkout absent
    pass
try:
    from qualcomm_kws_dataset import *  # noqa: F401,F403
except Exception:  # pragma: no cover - librosa / reference checkout absent
    pass
from audiopure_amd.audio_models.RCNN_KWS.model import KWSModel  # noqa: F401,E402
