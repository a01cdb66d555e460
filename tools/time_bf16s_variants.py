"""Times one eps evaluation (B clips; AP_PREC_BF16_STORE, or the mode in AP_TIME_MODE) per library given on the command line, each in its own
process, round-robin.   python tools/time_bf16s_variants.py B rounds lib1.so lib2.so ...     (child: --child)"""
import os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
if sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(here))
    import torch
    from audiopure_amd import synth
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    B = int(sys.argv[2])
    dev = torch.device("cuda:0")
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
    net = net.to(dev).set_precision(os.environ.get("AP_TIME_MODE", "bf16s"))
    x = torch.from_numpy(synth.waveforms(B, 16000, seed=6)).to(dev).reshape(B, 1, 16000)
    with torch.no_grad():
        for _ in range(2):
            net.eps(x, 3.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            net.eps(x, 3.0)
        torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 4 * 1e3:.2f}")
    sys.exit(0)
B, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for r in range(rounds):
    row = []
    for lib in libs:
        env = dict(os.environ, AUDIOPURE_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else "ERR " + out.stderr[-200:])
    print(f"round {r}: " + "  ".join(f"{os.path.basename(l)}: {v} ms" for l, v in zip(libs, row)), flush=True)
