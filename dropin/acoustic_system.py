"""Drop-in for the reference's top-level ``acoustic_system`` module (acoustic_system.py:5-53)."""
from audiopure_amd.acoustic_system import AcousticSystem  # noqa: F401
