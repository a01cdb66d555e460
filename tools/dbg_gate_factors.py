import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine(); lib = eng.lib
for L, layer in ((1100, 0), (2048, 5), (1024, 11)):
    B = 2
    h = torch.from_numpy(synth.uniform(f"gh/{L}", (B, 256, L), 1, -1.5, 1.5)).to(dev)
    pt = torch.zeros(256, device=dev)
    ho = torch.empty_like(h)
    g1 = torch.empty((B, L, 256), dtype=torch.bfloat16, device=dev); g2 = torch.empty_like(g1)
    n = lib.ap_gate_factor_bytes(B, L)
    f1 = torch.zeros(n, dtype=torch.uint8, device=dev); f2 = torch.zeros_like(f1); f3 = torch.zeros_like(f1)
    N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), g1.data_ptr(), f1.data_ptr(), B, L, N.stream()))
    N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), g1.data_ptr(), f3.data_ptr(), B, L, N.stream()))
    N.check(lib.ap_resblock_fwd_gate_save(eng.ctx, layer, N.ptr(h), N.ptr(pt), None, g2.data_ptr(), f2.data_ptr(), B, L, N.stream()))
    torch.cuda.synchronize()
    d = (f1 != f2).nonzero().flatten()
    print(L, layer, "bytes", n, "differ NOH vs not:", d.numel(), " run-to-run:", int((f1 != f3).sum()), " g equal:", torch.equal(g1.view(torch.int16), g2.view(torch.int16)))
    if d.numel():
        idx = d[:8].tolist()
        for i in idx:
            tile, r = divmod(i, 131072); w, r = divmod(r, 16384); ct, r = divmod(r, 4096); q, r = divmod(r, 1024); lane, byte = divmod(r, 16)
            print("   byte", i, "tile", tile, "wave", w, "ct", ct, "q", q, "lane", lane, "byte", byte, int(f1[i]), int(f2[i]))
        a = f1.view(torch.float16).float(); b = f2.view(torch.float16).float()
        print("   max |diff| as fp16 values:", float((a - b).abs().max()))
