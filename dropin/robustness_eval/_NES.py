"""Drop-in for the reference's ``robustness_eval._NES`` (black_box_attack.py:5,181)."""
from audiopure_amd.robustness_eval._NES import NES, resolve_prediction  # noqa: F401
