"""Drop-in for the reference's ``robustness_eval.certified_robust`` (certified_robustness_eval.py:87)."""
from audiopure_amd.robustness_eval.certified_robust import *  # noqa: F401,F403
from audiopure_amd.robustness_eval.certified_robust import RobustCertificate  # noqa: F401
