"""ORACLE — test infrastructure only.  numpy restatement of the library's counter-based noise:
Philox4x32-10 (Salmon et al., SC'11; the same round function as Random123/cuRAND) keyed on
(seed; quad index, draw, global utterance index) + Box-Muller in float32, as documented in
include/audiopure.h.  No reference counterpart (the reference draws from torch's global RNG,
diffwave_ddpm.py:66,100); this pins the HIP generator's integer stream bit-for-bit."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_normal(seed: int, draw: int, utt_offset: int, B: int, L: int) -> np.ndarray:
    out = np.empty((B, L), dtype=np.float32)
    nq = (L + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    for b in range(B):
        u = utt_offset + b
        r = philox4x32_10(q, np.full(nq, draw), np.full(nq, u & 0xFFFFFFFF), np.full(nq, u >> 32),
                          seed & 0xFFFFFFFF, seed >> 32)
        z = np.empty((nq, 4), dtype=np.float32)
        for p in range(2):
            u1 = ((r[2 * p] >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
            u2 = (r[2 * p + 1] >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
            rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
            ang = (np.float32(6.283185307179586) * u2).astype(np.float32)
            z[:, 2 * p] = rad * np.cos(ang.astype(np.float64)).astype(np.float32)
            z[:, 2 * p + 1] = rad * np.sin(ang.astype(np.float64)).astype(np.float32)
        out[b] = z.reshape(-1)[:L]
    return out
