"""bf16 block, deferred-skip form (round 4; ap_skipgemm_bf16.hip, resblock_bf16p_kernel<DS>) against the fused block.

1. exactness: per layer and clip length, h' of ap_resblock_fwd_gate must equal ap_resblock_fwd's bit for bit; skip of
   ap_skip_gemm over a group against the per-layer accumulation (fp32 summation order only);
2. eps of the shipped net with groups of G layers against the fused form;
3. timing at B clips: one eps evaluation (36 layers + final conv) with G in {0, 6, 12, 18, 36}, interleaved, and the split of the
   layers' time into block and skip-GEMM launches (ap_profile_read_split).

python tools/ab_bf16_ds.py [B] [reps]      (uses the PRODUCT library: the form ships)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))  # noqa: E401,E702
import ctypes as C
import sys

import torch

from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
GROUPS = [int(v) for v in _os.environ.get("AP_DS_GROUPS", "0,6,12,18,36").split(",")]
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine()
lib = eng.lib
bad = 0


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


# ---- 1. single layers through the C entry points
print("== layers: h' bit identity, skip of a 3-layer group vs per-layer accumulation")
for L in (16000, 4001, 1002, 643, 130):
    b = 2
    g = torch.Generator(device=dev)
    g.manual_seed(L)
    h = torch.randn((b, 256, L), device=dev, generator=g)
    pt = torch.randn((36, 256), device=dev, generator=g) * 0.1
    for l0 in (0, 1, 5, 9, 11):                                   # d = 1, 2, 32, 512, 2048 first layers of a 3-layer group
        layers = [l0, l0 + 12, l0 + 24]
        skip_ref = torch.zeros_like(h)
        skip_ds = torch.full_like(h, 7.0)                           # (accumulate = 0 must overwrite)
        gimg = torch.zeros((3, b, L, 256), dtype=torch.bfloat16, device=dev)
        hin = h
        same = True
        for i, l in enumerate(layers):
            ho_ref, ho_ds = torch.empty_like(h), torch.empty_like(h)
            N.check(lib.ap_resblock_fwd(eng.ctx, l, N.ptr(hin), N.ptr(pt[l].contiguous()), N.ptr(ho_ref), N.ptr(skip_ref), int(i > 0), b, L, N.stream()))
            N.check(lib.ap_resblock_fwd_gate(eng.ctx, l, N.ptr(hin), N.ptr(pt[l].contiguous()), N.ptr(ho_ds), gimg[i].data_ptr(), b, L, N.stream()))
            same &= torch.equal(ho_ref, ho_ds)
            hin = ho_ref
        # the group's layers are not consecutive here, so one ap_skip_gemm per layer slot (n_layers = 1) accumulating, and -- for
        # consecutive layers -- the grouped call below
        for i, l in enumerate(layers):
            N.check(lib.ap_skip_gemm(eng.ctx, l, 1, gimg[i].data_ptr(), N.ptr(skip_ds), int(i > 0), b, L, N.stream()))
        torch.cuda.synchronize()
        e = rel(skip_ds, skip_ref)
        ok = same and e < 2e-6 and bool(torch.isfinite(skip_ds).all())
        bad += not ok
        print(f"L = {L:5d} layers {layers}: h' {'bit-identical' if same else 'DIFFERENT'}, skip rel err {e:.2e} {'ok' if ok else 'FAIL'}")
    # consecutive group of 5 layers in one call
    layers = list(range(7, 12))
    skip_ref = torch.zeros_like(h)
    gimg = torch.zeros((5, b, L, 256), dtype=torch.bfloat16, device=dev)
    hin = h
    for i, l in enumerate(layers):
        ho_ref, ho_ds = torch.empty_like(h), torch.empty_like(h)
        N.check(lib.ap_resblock_fwd(eng.ctx, l, N.ptr(hin), N.ptr(pt[l].contiguous()), N.ptr(ho_ref), N.ptr(skip_ref), int(i > 0), b, L, N.stream()))
        N.check(lib.ap_resblock_fwd_gate(eng.ctx, l, N.ptr(hin), N.ptr(pt[l].contiguous()), N.ptr(ho_ds), gimg[i].data_ptr(), b, L, N.stream()))
        hin = ho_ref
    base = torch.randn((b, 256, L), device=dev, generator=g)
    skip_ds = base.clone()
    N.check(lib.ap_skip_gemm(eng.ctx, 7, 5, gimg.data_ptr(), N.ptr(skip_ds), 1, b, L, N.stream()))
    torch.cuda.synchronize()
    e = rel(skip_ds - base, skip_ref)
    ok = e < 5e-6
    bad += not ok
    print(f"L = {L:5d} group 7..11 in one call, accumulating onto a random tensor: rel err {e:.2e} {'ok' if ok else 'FAIL'}")

# ---- 2. eps of the whole net
print("== eps of the shipped net (36 layers), groups of G against the fused form")
for L in (16000, 4001, 1002):
    x = torch.from_numpy(synth.waveforms(3, L, seed=5)).to(dev).reshape(3, 1, L)
    with torch.no_grad():
        eng.skip_group = 0
        ref = net.eps(x, 3.0)
        for G in (1, 5, 12, 36):
            eng.skip_group = G
            a = net.eps(x, 3.0)
            e = rel(a, ref)
            ok = e < (1e-9 if G == 1 else 4e-3) and bool(torch.isfinite(a).all())      # (final_conv rounds skip to bf16: order changes flip roundings)
            bad += not ok
            print(f"L = {L:5d} G = {G:2d}: eps rel err vs fused {e:.2e} {'ok' if ok else 'FAIL'}")

# ---- 3. timing
print(f"== timing, B = {B}, L = 16000, one eps evaluation")
L = 16000
x = torch.from_numpy(synth.waveforms(B, L, seed=6)).to(dev).reshape(B, 1, L)
res = {}
for rnd in range(2):
    for G in GROUPS:
        eng.skip_group = G
        try:
            with torch.no_grad():
                net.eps(x, 3.0)
                torch.cuda.synchronize()
                N.check(lib.ap_profile_enable(eng.ctx, 1))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    net.eps(x, 3.0)
                e1.record()
                torch.cuda.synchronize()
                ms, n = (C.c_double * 2)(), (C.c_int64 * 2)()
                N.check(lib.ap_profile_read_split(eng.ctx, ms, n))
                N.check(lib.ap_profile_enable(eng.ctx, 0))
        except (RuntimeError, N.NativeError) as ex:
            print(f"G = {G}: {ex}")
            eng.ws = None
            continue
        layer_ms = (ms[0] + ms[1]) / max(n[0], 1)
        frac = 65.536e6 * B / (layer_ms * 1e-3) / 8e12
        print(f"round {rnd} G = {G:2d}: eps {e0.elapsed_time(e1) / reps:8.2f} ms; per layer {layer_ms:7.3f} ms = block {ms[0] / max(n[0], 1):7.3f} + skip GEMM "
              f"{ms[1] / max(n[0], 1):6.3f} ({n[1]} launches of {ms[1] / max(n[1], 1):7.3f} ms); roofline frac {frac:.4f}; workspace {eng.ws.numel() / 2**30:.1f} GiB")
        res.setdefault(G, []).append(layer_ms)
        eng.ws = None
        torch.cuda.empty_cache()
if 0 in res:
    for G in GROUPS:
        if G and G in res:
            print(f"G = {G:2d}: {100 * (min(res[G]) / min(res[0]) - 1):+.1f} % time per layer against the fused form")
sys.exit(1 if bad else 0)
