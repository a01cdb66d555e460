"""audiopure_amd — MI355X-native diffusion-purification hot path of AudioPure.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
the per-step work is hand-written gfx950 HIP behind the C-ABI of
``include/audiopure.h`` (``audiopure_amd/lib/libaudiopure_hip.so``).
There is no CPU fallback: every op raises if the native library is missing.
"""
__version__ = "0.1.0"
