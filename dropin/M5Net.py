"""Drop-in for ``audio_models/M5/M5Net.py`` (the scripts put that directory on sys.path and unpickle ``M5Net.M5``,
audio_models/create_model.py:4-10)."""
from audiopure_amd.audio_models.M5.M5Net import M5  # noqa: F401
