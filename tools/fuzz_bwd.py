"""Randomised hazard hunt over the fused backward block kernels (ap_resblock_bwd fp32, ap_resblock_bwd_bf16): python tools/fuzz_bwd.py [cases] [seed]
Random batch / clip length (multiples of 4, 64, 128 and not) / layer (dilation): both kernels twice (bit-identical results required),
every output (dy scratch, dh_in) inside guard bands that must stay untouched, results finite, and the bf16 gradient within 3e-2 of the
largest entry of the fp32 one (the two arithmetics differ by bf16 rounding: a hazard shows as 1e-1 .. 1)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
nets = {}
for mode in ("f32", "bf16"):
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
    nets[mode] = net.to(dev).set_precision(mode)
G = 2048
def guarded(shape, dtype=torch.float32):
    n = int(np.prod(shape))
    buf = torch.full((n + 2 * G,), 7.25, device=dev, dtype=dtype)
    return buf, buf[G:G + n].view(*shape), n
def intact(buf, n):
    return bool((buf[:G] == 7.25).all()) and bool((buf[G + n:] == 7.25).all())
bad, worst = 0, 0.0
for i in range(cases):
    B = int(rng.integers(1, 6))
    L = int(rng.choice([rng.integers(1, 300), rng.integers(300, 6000), 4 * rng.integers(16, 2100), 128 * rng.integers(1, 60), 16000]))
    layer = int(rng.integers(0, 12))
    h = torch.randn(B, 256, L, device=dev) * float(rng.choice([0.5, 1.0, 2.0]))
    gh, gs = torch.randn_like(h), torch.randn_like(h)
    pt = torch.randn(256, device=dev) * 0.5
    tag = f"B={B} L={L} layer={layer}"
    res = {}
    # fp32: the saving forward keeps the pre-gate activations, then ap_resblock_bwd
    e = nets["f32"].engine()
    N.check(e.lib.ap_ctx_prepare_backward(e.ctx, N.stream()))
    ho, sk, pre = torch.empty_like(h), torch.zeros_like(h), torch.empty(B, 512, L, device=dev)
    N.check(e.lib.ap_resblock_fwd_save(e.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), N.ptr(pre), 0, B, L, N.stream()))
    outs = []
    for rep in range(2):
        yb, dy, ny = guarded((B, 512, L)); db, dh, nd = guarded((B, 256, L))
        N.check(e.lib.ap_resblock_bwd(e.ctx, layer, N.ptr(gh), N.ptr(gs), N.ptr(pre), N.ptr(dy), N.ptr(dh), B, L, N.stream()))
        if not (intact(yb, ny) and intact(db, nd)): print("OUT-OF-BOUNDS WRITE fp32", tag); bad += 1
        outs.append(dh.clone())
    if not torch.equal(outs[0], outs[1]): print("NONDETERMINISTIC fp32", tag); bad += 1
    res["f32"] = outs[0]
    e = nets["bf16"].engine()
    N.check(e.lib.ap_ctx_prepare_backward(e.ctx, N.stream()))
    outs = []
    for rep in range(2):
        yb, dy, ny = guarded((B, L, 512), torch.bfloat16); db, dh, nd = guarded((B, 256, L))
        N.check(e.lib.ap_resblock_bwd_bf16(e.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(gh), N.ptr(gs), dy.data_ptr(), N.ptr(dh), B, L, N.stream()))
        if not (intact(yb, ny) and intact(db, nd)): print("OUT-OF-BOUNDS WRITE bf16", tag); bad += 1
        outs.append(dh.clone())
    if not torch.equal(outs[0], outs[1]): print("NONDETERMINISTIC bf16", tag); bad += 1
    res["bf16"] = outs[0]
    # bf16 from kept gate factors: ap_resblock_fwd_gate_save (h' / g image bit-identical to ap_resblock_fwd_gate) + ap_resblock_bwd_bf16_saved
    ho1, ho2 = torch.empty_like(h), torch.empty_like(h)
    g1, g2 = (torch.empty((B, L, 256), dtype=torch.bfloat16, device=dev) for _ in range(2))
    fac = torch.empty(e.lib.ap_gate_factor_bytes(B, L), dtype=torch.uint8, device=dev)
    N.check(e.lib.ap_resblock_fwd_gate_save(e.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho1), g1.data_ptr(), fac.data_ptr(), B, L, N.stream()))
    N.check(e.lib.ap_resblock_fwd_gate(e.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho2), g2.data_ptr(), B, L, N.stream()))
    if not (torch.equal(ho1, ho2) and torch.equal(g1.view(torch.int16), g2.view(torch.int16))): print("SAVE FORWARD DIFFERS", tag); bad += 1
    yb, dy, ny = guarded((B, L, 512), torch.bfloat16); db, dh, nd = guarded((B, 256, L))
    N.check(e.lib.ap_resblock_bwd_bf16_saved(e.ctx, layer, fac.data_ptr(), N.ptr(gh), N.ptr(gs), 0, dy.data_ptr(), N.ptr(dh), B, L, N.stream()))
    if not (intact(yb, ny) and intact(db, nd)): print("OUT-OF-BOUNDS WRITE bf16 saved", tag); bad += 1
    es = float((dh - res["bf16"]).abs().max() / res["bf16"].abs().max())
    if not es < 1e-2: print("SAVED-FACTOR FORM FAR FROM THE RECOMPUTING ONE", es, tag); bad += 1
    for m, t in res.items():
        if not bool(torch.isfinite(t).all()): print("NONFINITE", m, tag); bad += 1
    err = float((res["bf16"] - res["f32"]).abs().max() / res["f32"].abs().max())
    worst = max(worst, err)
    if err > 3e-2: print(f"MISMATCH bf16 vs fp32 {err:.3e}", tag); bad += 1
print(f"{cases} cases, {bad} failures; worst bf16 vs fp32 backward: {worst:.2e}")
sys.exit(1 if bad else 0)
