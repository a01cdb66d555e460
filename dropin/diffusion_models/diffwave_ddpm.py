"""Drop-in for ``diffusion_models/diffwave_ddpm.py`` (DiffWave, ReffWave, create_diffwave_model)."""
from audiopure_amd.diffusion_models.diffwave_ddpm import *  # noqa: F401,F403
from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave, ReffWave, create_diffwave_model  # noqa: F401
