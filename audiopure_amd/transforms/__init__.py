"""Device-side classifier front-end transforms."""
from .melspec import MelSpecDB, ToMelSpectrogramDB  # noqa: F401
