"""Device-side classifier front-end transforms."""
from .melspec import MelSpecDB, MelSpecDBHTK, ToMelSpectrogramDB  # noqa: F401
