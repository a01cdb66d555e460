"""Drop-in for ``diffusion_models/improved_diffusion_sde.py`` (RevVPSDE, RevImprovedDiffusion); the scripts do ``from ... import *``."""
from audiopure_amd.diffusion_models.improved_diffusion_sde import RevVPSDE, RevImprovedDiffusion  # noqa: F401
