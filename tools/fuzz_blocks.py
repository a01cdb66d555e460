"""Randomised hazard hunt over the residual-block kernels (direct fp32, F(2,3) fp32, 3-way split -- direct and F(2,3) two-launch form --, bf16), the bf16
deferred-skip pair and the bf16-storage block (AP_PREC_BF16_STORE): python tools/fuzz_blocks.py [cases] [seed]
Random batch / length (multiples of 4 and not) / layer (dilation) / accumulate flag; every mode is run twice (bit-identical
results required, outputs inside guard bands that must stay untouched) and compared with the exact fp32 kernel (split modes 5e-6 of max, bf16 3e-2)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
nets = {}
for mode in ("f32d", "f32", "f32s", "f32sw", "bf16", "bf16s"):
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
    nets[mode] = net.to(dev).set_precision(mode)
tol = {"f32": 5e-6, "f32s": 5e-6, "f32sw": 5e-6, "bf16": 3e-2}      # rounding-noise level at input amplitudes up to 3; a hazard shows as 1e-2 .. 1
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
worst = {k: 0.0 for k in tol}
bad = 0
for i in range(cases):
    B = int(rng.integers(1, 7))
    L = int(rng.choice([rng.integers(1, 260), rng.integers(260, 5000), 4 * rng.integers(32, 4100), 16000]))
    layer = int(rng.integers(0, 12))
    acc = int(rng.integers(0, 2))
    h = torch.randn(B, 256, L, device=dev) * float(rng.choice([0.3, 1.0, 3.0]))
    sk0 = torch.randn(B, 256, L, device=dev)
    pt = torch.randn(256, device=dev) * 0.5
    res = {}
    for mode, net in nets.items():
        if mode == "bf16s":                                      # (its block interface is the u image: below)
            continue
        eng = net.engine()
        outs = []
        for rep in range(2):
            # outputs live inside guard bands (4 KB of a sentinel either side): nothing may be written outside them
            G, n = 1024, h.numel()
            hb, sb = torch.full((n + 2 * G,), 7.25, device=dev), torch.full((n + 2 * G,), 7.25, device=dev)
            ho, sk = hb[G:G + n].view_as(h), sb[G:G + n].view_as(h)
            ho.fill_(float("nan")); sk.copy_(sk0)
            N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), acc, B, L, N.stream()))
            for nm, buf in (("h'", hb), ("skip", sb)):
                if not (bool((buf[:G] == 7.25).all()) and bool((buf[G + n:] == 7.25).all())):
                    print(f"OUT-OF-BOUNDS WRITE {nm} {mode} B={B} L={L} layer={layer} acc={acc}"); bad += 1
            outs.append((ho.clone(), sk.clone()))
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            # (the fp32 kernel adds into skip with memory-side float atomics, but one add per element per launch: deterministic too)
            print(f"NONDETERMINISTIC {mode} B={B} L={L} layer={layer} acc={acc}"); bad += 1
        if not (torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()):
            print(f"NONFINITE {mode} B={B} L={L} layer={layer} acc={acc}"); bad += 1
        res[mode] = outs[0]
    # bf16 mode's deferred-skip pair (ap_resblock_fwd_gate + ap_skip_gemm, one-layer group): h' and skip must equal the fused bf16
    # kernel's bit for bit, twice, with every output (h', the bf16 g image, skip) inside untouched guard bands
    eng = nets["bf16"].engine()
    for rep in range(2):
        G, n = 1024, h.numel()
        hb, sb = torch.full((n + 2 * G,), 7.25, device=dev), torch.full((n + 2 * G,), 7.25, device=dev)
        gb = torch.full((n + 2 * G,), 7.25, device=dev, dtype=torch.bfloat16)
        ho, sk, gi = hb[G:G + n].view_as(h), sb[G:G + n].view_as(h), gb[G:G + n]
        ho.fill_(float("nan")); sk.copy_(sk0)
        N.check(eng.lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), gi.data_ptr(), B, L, N.stream()))
        N.check(eng.lib.ap_skip_gemm(eng.ctx, layer, 1, gi.data_ptr(), N.ptr(sk), acc, B, L, N.stream()))
        for nm, buf in (("h'", hb), ("skip", sb), ("g image", gb)):
            if not (bool((buf[:G] == 7.25).all()) and bool((buf[G + n:] == 7.25).all())):
                print(f"OUT-OF-BOUNDS WRITE {nm} deferred-skip B={B} L={L} layer={layer} acc={acc}"); bad += 1
        if not (torch.equal(ho, res["bf16"][0]) and torch.equal(sk, res["bf16"][1])):
            print(f"DEFERRED-SKIP != FUSED bf16 B={B} L={L} layer={layer} acc={acc}: h' {rel(ho, res['bf16'][0]):.2e} skip {rel(sk, res['bf16'][1]):.2e}"); bad += 1
    # AP_PREC_BF16_STORE: the block on u images.  Fed u = bf16-exact values, the bf16 block with part_t = 0 sees the same GEMM operands and
    # the same residual: rounding ITS h' + part_t(next) to bf16 must give the stored image up to isolated rounding-boundary flips; twice,
    # bit-identical, every output (u', g image) inside untouched guard bands; the last-layer form (u' = NULL) writes the same g image
    eu, eb = nets["bf16s"].engine(), nets["bf16"].engine()
    PERM = torch.tensor([(p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1) for p in range(32)], device=dev)
    to_img = lambda u: u.reshape(B, 8, 32, L)[:, :, PERM, :].permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
    from_img = lambda im: im.float().permute(0, 1, 3, 2)[:, :, PERM, :].reshape(B, 256, L)
    ub = h.to(torch.bfloat16).float()
    uin, zero = to_img(ub), torch.zeros(256, device=dev)
    # both kernels of the mode (tools library: ap_debug_no_bf16us 1 = the persistent 128-sample-tile kernel, 2 = the 64-sample-tile kernel
    # of one- and two-clip launches): each twice inside guard bands, the last-layer form, and the two against each other bit for bit
    try:
        toggle = eu.lib.ap_debug_no_bf16us
        flags = (1, 2)
    except AttributeError:
        toggle, flags = None, (0,)
    per_kernel = {}
    for flag in flags:
        if toggle is not None:
            toggle(flag)
        prev = None
        for rep in range(2):
            G, n = 1024, h.numel()
            ubuf = torch.full((n + 2 * G,), 7.25, device=dev, dtype=torch.bfloat16)
            gb = torch.full((n + 2 * G,), 7.25, device=dev, dtype=torch.bfloat16)
            uo, gi = ubuf[G:G + n].view(B, 8, L, 32), gb[G:G + n]
            N.check(eu.lib.ap_resblock_fwd_u(eu.ctx, layer, uin.data_ptr(), N.ptr(pt), uo.data_ptr(), gi.data_ptr(), B, L, N.stream()))
            for nm, buf in (("u'", ubuf), ("g image", gb)):
                if not (bool((buf[:G] == 7.25).all()) and bool((buf[G + n:] == 7.25).all())):
                    print(f"OUT-OF-BOUNDS WRITE {nm} bf16s kernel {flag} B={B} L={L} layer={layer}"); bad += 1
            if prev is not None and not (torch.equal(prev[0].view(torch.int16), uo.view(torch.int16)) and torch.equal(prev[1].view(torch.int16), gi.view(torch.int16))):
                print(f"NONDETERMINISTIC bf16s kernel {flag} B={B} L={L} layer={layer}"); bad += 1
            prev = (uo.clone(), gi.clone())
        g2 = torch.empty_like(prev[1])
        N.check(eu.lib.ap_resblock_fwd_u(eu.ctx, layer, uin.data_ptr(), None, None, g2.data_ptr(), B, L, N.stream()))
        if not torch.equal(g2.view(torch.int16), prev[1].view(torch.int16)):
            print(f"LAST-LAYER FORM: OTHER g IMAGE bf16s kernel {flag} B={B} L={L} layer={layer}"); bad += 1
        per_kernel[flag] = prev
    if toggle is not None:
        toggle(0)
        if not (torch.equal(per_kernel[1][0].view(torch.int16), per_kernel[2][0].view(torch.int16)) and
                torch.equal(per_kernel[1][1].view(torch.int16), per_kernel[2][1].view(torch.int16))):
            d_u = float((per_kernel[1][0].float() - per_kernel[2][0].float()).abs().max())
            print(f"SMALL-TILE KERNEL != PERSISTENT KERNEL bf16s B={B} L={L} layer={layer}: max |du'| {d_u:.3e}"); bad += 1
    hb_, gb_ = torch.empty_like(h), torch.empty_like(prev[1])
    N.check(eb.lib.ap_resblock_fwd_gate(eb.ctx, layer, N.ptr(ub), N.ptr(zero), N.ptr(hb_), gb_.data_ptr(), B, L, N.stream()))
    ref_u = (hb_ + pt.view(1, -1, 1)).to(torch.bfloat16).float()
    got_u = from_img(prev[0])
    same = float((got_u == ref_u).float().mean())
    far = float(((got_u - ref_u).abs() > torch.exp2(torch.floor(torch.log2(ref_u.abs().clamp_min(1e-30))) - 7)).float().mean())
    if not (torch.isfinite(got_u).all() and same > 0.99 and far < 5e-3):
        print(f"MISMATCH bf16s vs the bf16 block: bit-equal {same:.4f}, beyond one ulp {far:.2e} B={B} L={L} layer={layer}"); bad += 1
    for mode, t in tol.items():
        e = max(rel(res[mode][0], res["f32d"][0]), rel(res[mode][1], res["f32d"][1]))   # reference: the direct-form fp32 kernel
        worst[mode] = max(worst[mode], e)
        if e > t:
            print(f"MISMATCH {mode} {e:.3e} B={B} L={L} layer={layer} acc={acc}"); bad += 1
print(f"{cases} cases, {bad} failures; worst vs fp32 kernel: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
sys.exit(1 if bad else 0)
