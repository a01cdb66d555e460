"""CPU pin of the arithmetic AP_PREC_F32_SPLIT / AP_CONV_SPLIT rely on (csrc/ap_resblock_f32s.hip, ap_convnet.hip):
an fp32 value is the exact sum of three bf16 parts, and the six partial products kept reproduce an fp32 GEMM to fp32
rounding noise.  torch's float32 -> bfloat16 conversion is round-to-nearest-even, the rounding of v_cvt_pk_bf16_f32."""
import numpy as np
import torch


def split3(a):
    h = a.bfloat16().float()
    r = a - h
    m = r.bfloat16().float()
    lo = (r - m).bfloat16().float()
    return h, m, lo


def test_three_bf16_parts_reconstruct_fp32_exactly():
    g = torch.Generator().manual_seed(0)
    vals = [torch.randn(1 << 16, generator=g) * s for s in (1e-6, 1e-3, 1.0, 37.5, 1e4, 1e12)]
    vals.append(torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 3.0 * 2 ** -20, 2.0 ** -60, -2.0 ** 60]))
    x = torch.cat(vals)
    h, m, lo = split3(x)
    assert torch.equal(h + m + lo, x)                 # exact in fp32, in this order
    assert torch.equal((h.double() + m.double() + lo.double()).float(), x)
    nz = x != 0
    assert (m.abs()[nz] <= h.abs()[nz] * 2.0 ** -8).all() and (lo.abs()[nz] <= h.abs()[nz] * 2.0 ** -16).all()


def test_six_partial_products_match_an_fp32_gemm():
    """K = 768 like GEMM1 of the residual block, accumulated in fp32 in blocks of 16 (one MFMA k-step)."""
    g = torch.Generator().manual_seed(1)
    K, M, N = 768, 128, 256
    w = torch.randn(M, K, generator=g) * 0.05
    x = torch.randn(K, N, generator=g) * 1.2
    ref = w.double() @ x.double()
    ws, xs = split3(w), split3(x)

    def emulate(terms):
        acc = torch.zeros(M, N)
        for k0 in range(0, K, 16):
            for i, j in terms:
                acc += ws[i][:, k0:k0 + 16] @ xs[j][k0:k0 + 16, :]      # every product exact: bf16 x bf16 fits fp32
        return acc

    six = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    nine = six + [(1, 2), (2, 1), (2, 2)]
    plain = torch.zeros(M, N)
    for k0 in range(0, K, 16):
        plain += w[:, k0:k0 + 16] @ x[k0:k0 + 16, :]
    rms = lambda v: float((v.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e_plain, e6, e9, e1 = rms(plain), rms(emulate(six)), rms(emulate(nine)), rms(emulate(six[:1]))
    assert e6 < 4 * e_plain and e6 < 1e-6              # fp32-class: the longer accumulation chain, not the dropped terms,
    assert abs(e6 - e9) < 0.2 * e9                     # is what separates it from the plain fp32 GEMM
    assert e1 > 1000 * e6                              # (plain bf16 operands are 3-4 orders of magnitude away)
