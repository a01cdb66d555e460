from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import *  # noqa: F401,F403
