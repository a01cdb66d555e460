"""Deterministic synthetic weights / inputs / noise, independent of the torch RNG.

The reference ships no checkpoints (``.gitignore`` lists ``*.pkl``) and its
``ZeroConv1d`` makes the freshly constructed net output exactly zero
(``diffusion_models/DiffWave_Unconditional/WaveNet.py:39-48``), so parity and
throughput runs need weights that (a) are identical on both sides of a
comparison without being stored (24 M parameters = 96 MB) and (b) keep the
activations O(1) through 36 layers.  Everything here is a pure function of
``(name, index, seed)`` through a SplitMix64 counter hash, so the fixture
generator (which loads these tensors into the reference's own modules), the
oracle, the tests and ``bench.py`` all see the same numbers.

The state-dict KEY NAMES and SHAPES are the reference's
(``WaveNet.py:138-172``; probe in SURVEY.md section 8b).
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _bits(name: str, n: int, seed: int, stream: int = 0) -> np.ndarray:
    base = (_fnv1a64(name) ^ (seed * 0xD1342543DE82EF95) ^ (stream * 0xA0761D6478BD642F)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base)) & _MASK
    return _splitmix64(idx)


def uniform(name: str, shape, seed: int = 0, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 uniform in [lo, hi) determined only by (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_bits(name, n, seed) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(name: str, shape, seed: int = 0) -> np.ndarray:
    """float32 standard normal (Box-Muller in float64, rounded once)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = ((_bits(name, n, seed, 1) >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))
    u2 = (_bits(name, n, seed, 2) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# DiffWave epsilon-network (reference: WaveNet.py:138-172; configs/config.json)
# ---------------------------------------------------------------------------
FULL_WAVENET_CONFIG = dict(
    in_channels=1, res_channels=256, skip_channels=256, out_channels=1,
    num_res_layers=36, dilation_cycle=12,
    diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
    diffusion_step_embed_dim_out=512,
)
DIFFUSION_CONFIG = dict(T=200, beta_0=0.0001, beta_T=0.02)


def mini_wavenet_config(channels: int = 64, layers: int = 12, cycle: int = 12) -> dict:
    cfg = dict(FULL_WAVENET_CONFIG)
    cfg.update(res_channels=channels, skip_channels=channels, num_res_layers=layers, dilation_cycle=cycle)
    return cfg


def wavenet_state_dict(cfg: dict, seed: int = 0) -> "dict[str, np.ndarray]":
    """Reference-named state dict (``weight_g``/``weight_v`` un-folded)."""
    C, S = cfg["res_channels"], cfg["skip_channels"]
    Ein, Emid, Eout = (cfg["diffusion_step_embed_dim_in"], cfg["diffusion_step_embed_dim_mid"],
                       cfg["diffusion_step_embed_dim_out"])
    sd: dict[str, np.ndarray] = {}

    def wn(prefix: str, O: int, I: int, k: int, g_lo: float, g_hi: float, b: float):
        sd[prefix + ".bias"] = uniform(prefix + ".bias", (O,), seed, -b, b)
        sd[prefix + ".weight_g"] = uniform(prefix + ".weight_g", (O, 1, 1), seed, g_lo, g_hi)
        sd[prefix + ".weight_v"] = uniform(prefix + ".weight_v", (O, I, k), seed, -1.0, 1.0)

    def lin(prefix: str, O: int, I: int, scale: float, b: float):
        sd[prefix + ".weight"] = uniform(prefix + ".weight", (O, I), seed, -scale, scale)
        sd[prefix + ".bias"] = uniform(prefix + ".bias", (O,), seed, -b, b)

    wn("init_conv.0.conv", C, cfg["in_channels"], 1, 0.5, 1.5, 0.3)
    lin("residual_layer.fc_t1", Emid, Ein, (3.0 / Ein) ** 0.5, 0.1)
    lin("residual_layer.fc_t2", Eout, Emid, (3.0 / Emid) ** 0.5, 0.1)
    for n in range(cfg["num_res_layers"]):
        p = f"residual_layer.residual_blocks.{n}"
        lin(p + ".fc_t", C, Eout, (3.0 / Eout) ** 0.5, 0.1)
        # row norm of the folded weight is |g|: pre-gate activations ~ N(0, g^2 * mean(u^2))
        wn(p + ".dilated_conv_layer.conv", 2 * C, C, 3, 1.0, 1.6, 0.1)
        wn(p + ".res_conv", C, C, 1, 0.8, 1.2, 0.05)
        wn(p + ".skip_conv", S, C, 1, 0.8, 1.2, 0.05)
    wn("final_conv.0.conv", S, S, 1, 0.8, 1.2, 0.05)
    # ZeroConv1d in the reference (WaveNet.py:39-48): a bare Conv1d, zero at init; non-zero here
    sd["final_conv.2.conv.weight"] = uniform("final_conv.2.conv.weight", (cfg["out_channels"], S, 1), seed,
                                             -1.0, 1.0) * np.float32(4.0 / S ** 0.5)
    sd["final_conv.2.conv.bias"] = uniform("final_conv.2.conv.bias", (cfg["out_channels"],), seed, -0.05, 0.05)
    return sd


# ---------------------------------------------------------------------------
# M5 classifier (reference: audio_models/M5/M5Net.py:4-38)
# ---------------------------------------------------------------------------
def m5_state_dict(n_output: int = 10, n_channel: int = 32, first_kernel_size: int = 80,
                  seed: int = 0) -> "dict[str, np.ndarray]":
    sd: dict[str, np.ndarray] = {}
    chans = [(1, n_channel, first_kernel_size), (n_channel, n_channel, 3),
             (n_channel, 2 * n_channel, 3), (2 * n_channel, 2 * n_channel, 3)]
    for i, (ci, co, k) in enumerate(chans, start=1):
        s = (6.0 / (ci * k)) ** 0.5
        sd[f"conv{i}.weight"] = uniform(f"m5.conv{i}.weight", (co, ci, k), seed, -s, s)
        sd[f"conv{i}.bias"] = uniform(f"m5.conv{i}.bias", (co,), seed, -0.1, 0.1)
        sd[f"bn{i}.weight"] = uniform(f"m5.bn{i}.weight", (co,), seed, 0.8, 1.2)
        sd[f"bn{i}.bias"] = uniform(f"m5.bn{i}.bias", (co,), seed, -0.1, 0.1)
        sd[f"bn{i}.running_mean"] = uniform(f"m5.bn{i}.running_mean", (co,), seed, -0.1, 0.1)
        sd[f"bn{i}.running_var"] = uniform(f"m5.bn{i}.running_var", (co,), seed, 0.5, 1.5)
        sd[f"bn{i}.num_batches_tracked"] = np.array(100, dtype=np.int64)
    s = (3.0 / (2 * n_channel)) ** 0.5
    sd["fc1.weight"] = uniform("m5.fc1.weight", (n_output, 2 * n_channel), seed, -s, s)
    sd["fc1.bias"] = uniform("m5.fc1.bias", (n_output,), seed, -0.1, 0.1)
    return sd


# ---------------------------------------------------------------------------
# inputs / injected noise
# ---------------------------------------------------------------------------
def waveforms(B: int, L: int = 16000, seed: int = 1234, utt_offset: int = 0) -> np.ndarray:
    """``0.5*U(-1,1)`` clips ``[B,1,L]`` (attack code asserts ``-1 <= x < 1``,
    ``robustness_eval/black_box_attack.py:197``).  Row ``i`` depends only on the
    GLOBAL utterance index ``utt_offset + i`` so shards reproduce the full batch."""
    out = np.empty((B, 1, L), dtype=np.float32)
    for i in range(B):
        out[i, 0] = uniform(f"x0/{utt_offset + i}", (L,), seed, -0.5, 0.5)
    return out


def noise(draw: int, B: int, L: int = 16000, seed: int = 1234, utt_offset: int = 0) -> np.ndarray:
    """Injected N(0,1) tensor number ``draw`` of a purification call
    (draw 0 = q-sample noise, draw k>=1 = k-th reverse-step noise)."""
    out = np.empty((B, 1, L), dtype=np.float32)
    for i in range(B):
        out[i, 0] = normal(f"z/{draw}/{utt_offset + i}", (L,), seed)
    return out


def synth_init(model, seed: int = 0):
    """Deterministic, torch-RNG-independent weights keyed on the state-dict names (He-scaled convs, BN statistics away
    from the identity so the fold is exercised)."""
    import torch
    sd = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(100)
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(uniform("cn/" + k, shape, seed, 0.5, 1.5))
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(uniform("cn/" + k, shape, seed, -0.2, 0.2))
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            a = (6.0 / fan_in) ** 0.5
            sd[k] = torch.from_numpy(uniform("cn/" + k, shape, seed, -a, a))
        elif k.endswith("weight"):                       # BN gamma
            sd[k] = torch.from_numpy(uniform("cn/" + k, shape, seed, 0.7, 1.3))
        else:                                            # biases / BN beta
            sd[k] = torch.from_numpy(uniform("cn/" + k, shape, seed, -0.1, 0.1))
    model.load_state_dict(sd)
    return model.eval()
