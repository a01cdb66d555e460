"""Drop-in for ``diffusion_models/diffwave_sde.py`` (RevVPSDE, RevDiffWave); the scripts do ``from ... import *``."""
from audiopure_amd.diffusion_models.diffwave_sde import RevVPSDE, RevDiffWave, DiffWave, create_diffwave_model  # noqa: F401
