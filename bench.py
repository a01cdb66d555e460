#!/usr/bin/env python3
"""Headline benchmark: purified 1 s @ 16 kHz utterances / s at 5 reverse DDPM steps (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clips per GPU:
q-sample -> 5 x (36 fused residual blocks + final conv + x_{t-1} update) -> M5 classify
(BASELINE.json configs[1]: DiffWave DDPM n=5, batch=512 per GPU, fp32).  Inputs, weights and the
workspace are resident in HBM before the timed region.  N > 1: launched by torch.distributed.run, one
rank per GPU, utterances sharded contiguously (weak scaling: 512 per GPU), counter-based noise keyed on
the global utterance index, one RCCL all_gather of the [B,10] log-probabilities at the end of each step.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself as CHILD processes
(`python -m torch.distributed.run ... bench.py --gpus N ...`, before anything touches a GPU) and relays rank 0's line;
`--dry-run` does the same on CPU with the gloo backend (launcher / sharding / gather plumbing only, no kernels).
At N = 1 the same run then times the other arithmetic modes of the path (`other_modes`: bf16, the fp32-class split mode f32s, the direct-form fp32 block),
BASELINE configs[3] / configs[4] as their own workloads and the callers' batch shapes (`other_configs`), each with its own roofline object.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = fused residual block, timed with HIP
events on its launch stream inside the timed region) and `cpu_baseline` (the CPU oracle on BASELINE
config 1, timed on this node's host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# a ConvNet whose trace meets an operator without a kernel must fail the leg, not run on PyTorch operators behind a warning and
# still report a throughput (audiopure_amd/convnet.py: NativeConvNet._off_native)
os.environ.setdefault("AUDIOPURE_STRICT", "1")

FLOP_PER_LAYER_UTT = 2.0 * 16000 * (512 * 768 + 512 * 256)      # 16.777 GFLOP: the layer as the reference states it
FLOP_EXEC_F32W_UTT = 2.0 * 16000 * (512 * 512 + 512 * 256)      # 12.583 GFLOP: what the F(2,3) block executes (4 of 6 dilated-conv products)
BYTES_PER_LAYER_UTT = (2 * 256 + 2 * 256) * 16000 * 4.0          # read h, write h', read+write skip (fp32)
BYTES_PER_LAYER_UTT_BF16STORE = BYTES_PER_LAYER_UTT / 2.0        # SURVEY 8(d), bf16 storage: 1.180 GB per clip and step / 36 layers = 32.8 MB
PEAK_BF16_MFMA_TFLOPS = 2500.0                                   # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table
PEAK_F32_MFMA_TFLOPS = 157.3                                     # MI355X_MICROARCH.md chip table
N_LAYERS = 36


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


_CPU_CHILD = r"""
import sys, time, torch
sys.path.insert(0, %r)
from audiopure_amd import synth
from oracle import diffwave_oracle as O
torch.set_num_threads(int(sys.argv[1]))
cfg = dict(synth.FULL_WAVENET_CONFIG)
w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
dh = O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
m5 = synth.m5_state_dict(10)
x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
z = [torch.from_numpy(synth.noise(0, 2, 16000, seed=1234))]
print("READY", flush=True)
t = time.time(); O.purify_and_classify(w, cfg, dh, m5, x0, 1, z); print("CALL", time.time() - t, flush=True)
t = time.time(); O.purify_and_classify(w, cfg, dh, m5, x0, 1, z); print("CALL", time.time() - t, flush=True)
"""


def _all_cores_attempt(threads: int, bound_s: float = 20.0):
    """SURVEY 8(d) asks for torch.set_num_threads(os.cpu_count()).  On the GPU hosts seen so far (2 x 64-core EPYC, 256 hardware
    threads) one 2-clip evaluation at that setting takes ~90 s (oneDNN's fork-join over 256 threads on a batch of two), three
    times the whole default bench -- so the attempt runs in a child process, is cut off after `bound_s` of compute, and the
    object reports what happened; `value` is then the thread count that performs (`cores` says which)."""
    import subprocess
    t0 = time.time()
    calls, ready = [], None
    try:
        p = subprocess.Popen([sys.executable, "-c", _CPU_CHILD % ROOT, str(threads)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except OSError as e:
        return {"threads": threads, "status": f"not started ({e})"}
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(p.stdout, selectors.EVENT_READ)
    deadline = None
    while True:
        now = time.time()
        if ready is None and now - t0 > 120:                      # imports + weight synthesis never finished
            break
        if deadline is not None and now > deadline:
            break
        if not sel.select(timeout=0.5):
            if p.poll() is not None:
                break
            continue
        line = p.stdout.readline()
        if not line:
            break
        if line.startswith("READY"):
            ready = time.time()
            deadline = ready + bound_s
        elif line.startswith("CALL"):
            calls.append(float(line.split()[1]))
            if len(calls) == 2:
                break
    if p.poll() is None:
        p.kill()
    p.wait()
    if len(calls) == 2:
        return {"threads": threads, "status": "completed", "s_per_call": round(calls[1], 3), "value": round(2.0 / calls[1], 4)}
    return {"threads": threads, "status": f"cut off after {bound_s:.0f} s of compute with {len(calls)} of 2 calls finished",
            "s_first_call": round(calls[0], 3) if calls else None,
            "note": "one 2-clip, 1-step evaluation at this thread count measured ~90 s on this host class (profiles/r5_cpu_baseline_threads.txt)"}


def cpu_baseline(budget_s: float = 25.0):
    """The oracle ("port") on BASELINE configs[0] as SURVEY.md section 8(d) fixes it -- 2 clips, DiffWave DDPM n = 1 + M5, fp32,
    CPU model and thread count in the object.  `all_cores`: the same call at torch.set_num_threads(os.cpu_count()), attempted in a
    bounded child process (see _all_cores_attempt); when it completes and beats the 32-thread figure it IS `value`.  The same 2
    clips at the metric's n = 5 ride along as `n5_value`.  Bounded: 10-30 s of CPU work here, at most 20 s more in the child."""
    import torch
    from audiopure_amd import synth
    from oracle import diffwave_oracle as O
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)
    allc = _all_cores_attempt(ncpu) if ncpu > cores else None
    torch.set_num_threads(cores)
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
    dh = O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
    m5 = synth.m5_state_dict(10)
    x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
    z = [torch.from_numpy(synth.noise(d, 2, 16000, seed=1234)) for d in range(5)]
    t0 = time.time()
    O.purify_and_classify(w, cfg, dh, m5, x0, 1, z[:1])        # warm-up (oneDNN primitive creation), one step
    best1, runs1 = None, 0
    while runs1 < 3 and (runs1 == 0 or (time.time() - t0) + best1 < budget_s * 0.4):
        t1 = time.time()
        O.purify_and_classify(w, cfg, dh, m5, x0, 1, z[:1])
        dt = time.time() - t1
        best1 = dt if best1 is None else min(best1, dt)
        runs1 += 1
    best5, runs5 = None, 0
    while runs5 < 2 and (runs5 == 0 or (time.time() - t0) + best5 < budget_s):
        t1 = time.time()
        O.purify_and_classify(w, cfg, dh, m5, x0, 5, z)
        dt = time.time() - t1
        best5 = dt if best5 is None else min(best5, dt)
        runs5 += 1
    value, used = 2.0 / best1, cores
    if allc and allc.get("status") == "completed" and allc["value"] > value:
        value, used = allc["value"], allc["threads"]
    return {"value": round(value, 4), "unit": "utterances/s", "cores": used, "kind": "port", "cpu_model": _cpu_model(),
            "host_threads": ncpu, "n_reverse_steps": 1, "all_cores": allc, "n5_value": round(2.0 / best5, 4), "n5_cores": cores,
            "sample": f"CPU oracle (PyTorch-CPU fp32 restatement of the reference path) on BASELINE configs[0]: 2 clips x 1 reverse "
                      f"step + M5 on {cores} threads, best of {runs1} calls ({best1:.2f} s per call) after a one-step warm-up; "
                      f"`all_cores`: the same call at os.cpu_count() = {ncpu} threads in a bounded child process; n5_value: the same "
                      f"2 clips x 5 reverse steps (the metric's step count), best of {runs5} ({best5:.2f} s per call)"}


PREC_NAME = {"f32": "fp32 (v_mfma_f32_32x32x2_f32; dilated conv in F(2,3) minimal-filtering form)",
             "f32d": "fp32 (v_mfma_f32_32x32x2_f32; direct-form dilated conv: the round 1-4 kernel)", "bf16": "bf16",
             "bf16s": "bf16 MFMA operands, fp32 accumulate, bf16 storage of the residual stream (SURVEY 8d's third precision row; fp32 skip)",
             "f32s": "fp32-class: exact 3-way bf16 operand split, 6 partial products on the bf16 MFMA, fp32 accumulate (direct-form "
                     "dilated conv; held to the fp32 tolerances and to 2 x the fp32 kernel's error on adversarial operands)"}


PMC_FILES = {"f32": ["r6_f32w_pmc_traffic.json", "r5_f32w_pmc_traffic.json"], "f32d": ["r3_pmc_traffic.json", "r2_pmc_traffic.json"], "f32s": ["r3_f32s_pmc_traffic.json", "r2_f32s_pmc_traffic.json"],
             "bf16": ["r6_bf16_pmc_traffic.json", "r4_bf16_pmc_traffic.json"], "bf16s": ["r6_bf16s_pmc_traffic.json"]}


class PowerSampler:
    """Board power and shader clock of GPU 0 while a leg runs (`rocm-smi` polled from a side thread every 0.5 s: a host process,
    nothing on the GPU's queues) -> the `power` object of that leg: the bf16 / f32s block kernels sit at the package's
    power cap and the firmware lowers the clock under them, and the line should say so itself.  None when
    rocm-smi is missing or prints nothing usable."""

    def __init__(self, device_index=0, enabled=True, pci_bus_id=None):
        """`device_index` is torch's LOGICAL index; rocm-smi numbers PHYSICAL boards.  With `pci_bus_id` (device_descriptor) the
        board is looked up in `rocm-smi --showbus`, so under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES the sampled board is the
        one the kernels run on; the result says which index was polled and how it was found."""
        import shutil
        self.exe = (shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)) if enabled else None
        self.dev, self.dev_source = device_index, "logical index (no PCI match)"
        if self.exe and pci_bus_id:
            self._resolve(pci_bus_id)
        self.samples, self.stop, self.thread = [], False, None

    def _resolve(self, pci_bus_id):
        import re
        import subprocess
        try:
            out = subprocess.run([self.exe, "--showbus"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return
        want = pci_bus_id.lower()
        for m in re.finditer(r"GPU\[(\d+)\]\s*:\s*PCI Bus:\s*([0-9a-fA-F:.]+)", out):
            if m.group(2).lower().startswith(want):
                self.dev, self.dev_source = int(m.group(1)), f"rocm-smi --showbus match of {pci_bus_id}"
                return

    def _read(self, extra=()):
        import re
        import subprocess
        try:
            out = subprocess.run([self.exe, "-d", str(self.dev), "--showpower", "--showclocks", *extra], capture_output=True, text=True,
                                 timeout=10).stdout
        except Exception:
            return None, None, None
        p = re.search(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)", out)
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", out)
        return (float(p.group(1)) if p else None, int(c.group(1)) if c else None, float(m.group(1)) if m else None)

    def __enter__(self):
        if self.exe:
            import threading

            def loop():
                time.sleep(0.4)                                     # (the first poll would land before the leg's first kernel is running)
                while not self.stop:
                    w, c, _ = self._read()
                    if w is not None and c is not None:
                        self.samples.append((w, c))
                    time.sleep(0.5)
            self.thread = threading.Thread(target=loop, daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        if self.thread:
            self.thread.join(timeout=15)

    def result(self):
        if not self.samples:
            return None
        ws, cs = [w for w, _ in self.samples], [c for _, c in self.samples]
        cap = self._read(("--showmaxpower",))[2]
        return {"board_W_mean": round(sum(ws) / len(ws), 1), "board_W_max": round(max(ws), 1), "cap_W": cap,
                "sclk_MHz_mean": round(sum(cs) / len(cs)), "sclk_MHz_min": min(cs), "samples": len(ws),
                "smi_index": self.dev, "smi_index_source": self.dev_source,
                "source": "rocm-smi --showpower --showclocks polled every 0.5 s over the timed region"}


def pmc_traffic(B, precision="f32"):
    """HBM-side bytes per residual-block launch from the newest committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 calibration, + WRITE_SIZE), scaled from the 512-clip launch it was measured on -> (bytes or None, source).
    The counters cannot be read from inside this process; the line says where the number comes from."""
    for name in PMC_FILES.get(precision, []):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return round(d["traffic_bytes_per_launch"] * B / 512.0), f"profiles/{name} (separate rocprofv3 --pmc passes of this command, not measured in this run)"
        except Exception:
            continue
    return None, None


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv) -> int:
    """--gpus N > 1 outside torch.distributed.run: start the ranks as children of this (GPU-untouched) process."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def rank_evidence(use_dist, elapsed_local, device_desc, backend):
    """What makes an N > 1 line self-evidencing: the size of the process group the collective backend actually formed
    (`rccl_ranks` on the GPU path, after the `nccl` init), every rank's own elapsed time for the K timed steps (min / max:
    `ms_per_step` is the max), and the device each rank bound -- N ranks on N distinct GPUs is visible in the JSON."""
    import torch.distributed as dist
    if not use_dist:
        return {"backend": None, "rccl_ranks" if backend == "nccl" else "ranks": 0, "rank_elapsed_s": [round(elapsed_local, 4)],
                "rank_elapsed_min_s": round(elapsed_local, 4), "rank_elapsed_max_s": round(elapsed_local, 4),
                "devices": [device_desc], "distinct_devices": 1}
    world = dist.get_world_size()
    objs = [None] * world
    dist.all_gather_object(objs, {"rank": dist.get_rank(), "elapsed_s": elapsed_local, "device": device_desc})
    objs.sort(key=lambda o: o["rank"])
    el = [o["elapsed_s"] for o in objs]
    devs = [o["device"] for o in objs]
    return {"backend": dist.get_backend(), "rccl_ranks" if backend == "nccl" else "ranks": world,
            "rank_elapsed_s": [round(e, 4) for e in el], "rank_elapsed_min_s": round(min(el), 4),
            "rank_elapsed_max_s": round(max(el), 4), "devices": devs,
            "distinct_devices": len({d.get("pci_bus_id") or d.get("uuid") or d.get("index") for d in devs})}


def device_descriptor(torch, local):
    """Name + PCI address (+ uuid where torch exposes it) of the HIP device this rank bound."""
    p = torch.cuda.get_device_properties(local)
    d = {"index": local, "name": torch.cuda.get_device_name(local)}
    dom, bus, devid = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if bus is not None:
        d["pci_bus_id"] = f"{(dom or 0):04x}:{bus:02x}:{(devid or 0):02x}"
    if getattr(p, "uuid", None) is not None:
        d["uuid"] = str(p.uuid)
    return d


def global_batch_shard(torch, world, rank, B, L, device, seed=1234):
    """The GLOBAL batch of world * B synthetic clips (0.5 U(-1, 1)) from ONE seed, and this rank's contiguous shard of it: clip i of the
    job is the same waveform however many ranks share the work (with the Philox noise keyed on the global index too, an N-rank run
    reproduces the 1-rank run's scores bit for bit)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (torch.rand((world * B, 1, L), device=device, generator=g) - 0.5)[rank * B:(rank + 1) * B].contiguous()


def make_step(score_fn, x0, use_dist, n_total):
    """One pass of the hot path over this rank's shard: `score_fn(x0)` -> [B, K] scores, then the path's only collective (an
    all_gather of the score rows into global utterance order).  The timed loop of every leg and the CPU rehearsal call this."""
    from audiopure_amd.sharding import all_gather_scores

    def step():
        lp = score_fn(x0)
        return all_gather_scores(lp, n_total) if use_dist else lp
    return step


def scores_digest(scores):
    """sha256 of the gathered [global_batch, K] scores: equal for every split of the same global batch over ranks."""
    import hashlib
    s = scores.detach().float().cpu().contiguous()
    return {"shape": list(s.shape), "sha256": hashlib.sha256(s.numpy().tobytes()).hexdigest(), "argmax_head": s[:8].argmax(1).tolist()}


def dry_run(args) -> None:
    """CPU rehearsal of the multi-rank protocol (gloo): shard bounds, barrier, the scores all_gather, max-over-ranks
    timing, one JSON line from rank 0.  No kernel runs and `value` is not a throughput."""
    import torch
    import torch.distributed as dist
    from audiopure_amd.sharding import all_gather_scores, shard_bounds
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    use_dist = "RANK" in os.environ
    if use_dist:
        dist.init_process_group("gloo")
    B = args.batch
    lo, hi = shard_bounds(world * B, rank, world)
    assert (lo, hi) == (rank * B, (rank + 1) * B)
    # the real step function (global batch from one seed, contiguous shard, scores all_gather) around a CPU stand-in scorer whose
    # row i depends on clip i only -- like the HIP path's -- so the gathered scores' digest must not depend on the split
    Ld = 64
    x0 = global_batch_shard(torch, world, rank, B, Ld, torch.device("cpu"))
    wgt = torch.sin(torch.arange(10 * Ld, dtype=torch.float64).reshape(10, Ld) * 0.37).float()
    step = make_step(lambda x: torch.log_softmax(8.0 * x.reshape(x.shape[0], -1) @ wgt.t(), dim=1), x0, use_dist, world * B)

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lp = step()
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    ranks = rank_evidence(use_dist, el, {"index": rank, "name": "cpu (dry run)", "pci_bus_id": f"cpu:{rank}"}, "gloo")
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    assert lp.shape == (world * B, 10)
    if rank == 0:
        print(json.dumps({"metric": f"purified 1s@16kHz utterances/sec at {args.reverse_steps} reverse steps", "dry_run": True,
                          "value": None, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el * 1e3 / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.precision, "data": "none (protocol rehearsal on CPU, gloo)",
                          "ranks": ranks, "scores": scores_digest(lp),
                          "config": {"workload": "dry run: launcher + sharding + scores all_gather only",
                                     "global_batch": world * B, "parallelism": f"utterance-sharded x{world}, logits all_gather"}}),
              flush=True)
    if use_dist:
        dist.destroy_process_group()


UNET_GFLOP_PER_EVAL = 16.76        # shipped Improved-Diffusion UNet (52.5 M parameters) at 1 x 32 x 32, per sample
RESNEXT29_GFLOP = 10.77            # ResNeXt-29 8x64d at 1 x 32 x 32, per sample


def bench_config4(dev, steps, B=256, n=5):
    """BASELINE configs[4]: Improved-Diffusion UNet DDPM n = 5 (GaussianDiffusion q_sample + p_sample chain on mel-dB
    spectrograms) + ResNeXt-29 classifier, batch 256, fp32 MFMA conv-as-GEMM.  Timed with HIP events on the launch
    stream (torch's current stream is the stream every ap_* call of this path is issued on)."""
    import torch
    sys.path.insert(1, os.path.join(ROOT, "tools"))
    from synth_convnets import CifarResNeXt, synth_init
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM
    from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
    from audiopure_amd.transforms import MelSpecDB
    unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
    clf = synth_init(CifarResNeXt(10), 0).to(dev)                       # a plain module: AcousticSystem lowers it
    system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=ImprovedDiffusionDDPM(unet, reverse_timestep=n),
                            defense_type="spec").eval()
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    x = (torch.rand((B, 1, 16000), device=dev, generator=g) - 0.5).contiguous()
    with torch.no_grad():
        y = system(x, True)                                             # warm-up: lowering, weight packing
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with PowerSampler(dev.index or 0, pci_bus_id=device_descriptor(torch, dev.index or 0).get("pci_bus_id")) as ps:
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                y = system(x, True)
            e1.record()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    assert y.shape == (B, 10) and torch.isfinite(y).all()
    ev_ms = e0.elapsed_time(e1) / steps
    gflop = n * UNET_GFLOP_PER_EVAL + RESNEXT29_GFLOP
    tf = gflop * B / (ev_ms * 1e-3) / 1e3
    # per-kernel roofline of the conv-as-GEMM family: one more pass of the same step (outside the timed region) with every
    # ap_conv2d_fwd launch bracketed by HIP events on its launch stream; algorithmic flops 2 N M K per launch, by kernel class
    import ctypes as C
    from audiopure_amd import _native as N
    lib = N.lib()
    NC = 8
    names = ["conv2d_f32_big2_kernel<128,128>", "conv2d_f32_big2_kernel<64,128>", "conv2d_f32_big2_kernel<128,64>",
             "conv2d_split_kernel (split operands)", "conv2d_f32_big_kernel", "conv2d_f32_kernel (generic)",
             "conv2d_w3_kernel (3x3 in F(2,3) form along W: executes 2/3 of the direct form's flops)",
             "conv1x1_stream_kernel (tools builds only)"]
    N.check(lib.ap_conv_profile_enable(1))
    with torch.no_grad():
        system(x, True)
    torch.cuda.synchronize()
    ms, fl, ln = (C.c_double * NC)(), (C.c_double * NC)(), (C.c_int64 * NC)()
    N.check(lib.ap_conv_profile_read(ms, fl, ln, NC))
    N.check(lib.ap_conv_profile_enable(0))
    # executed flops: the F(2,3) class performs 12 of the direct form's 18 multiplications per output pair and input channel
    ex = [fl[c] * (2.0 / 3.0 if c == 6 else 1.0) for c in range(NC)]
    by_kernel = {names[c]: {"launches": int(ln[c]), "ms": round(ms[c], 3), "GFLOP": round(fl[c] / 1e9, 1),
                            "executed_GFLOP": round(ex[c] / 1e9, 1),
                            "TFLOPs": round(ex[c] / (ms[c] * 1e-3) / 1e12, 2),
                            "frac": round(ex[c] / (ms[c] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                            "algorithmic_TFLOPs": round(fl[c] / (ms[c] * 1e-3) / 1e12, 2)}
                 for c in range(NC) if ln[c]}
    tot_ms, tot_fl, tot_ex = sum(ms), sum(fl), sum(ex)
    ktf = tot_ex / (tot_ms * 1e-3) / 1e12
    ktf_alg = tot_fl / (tot_ms * 1e-3) / 1e12
    dom = max(range(NC), key=lambda c: ms[c])
    return {"workload": f"mel-dB front-end -> Improved-Diffusion UNet DDPM n={n} (ImprovedDiffusionDDPM) -> ResNeXt-29 8x64d, "
                        f"batch={B}, fp32 MFMA conv-as-GEMM, 1 s @ 16 kHz clips",
            "value": round(B * steps / el, 3), "unit": "utterances/s", "steps": steps, "warmup": 1,
            "ms_per_step": round(el * 1e3 / steps, 3), "dtype": "f32", "power": ps.result(),
            "roofline": {"bound": "mfma", "kernel": f"conv-as-GEMM family (every conv / linear layer of the step); dominant: {names[dom]}",
                         "achieved": round(ktf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ktf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                         "attainable": "the same matrix instruction fed from LDS reaches 0.84-0.94 of this peak in a bare loop "
                                       "(profiles/r3_mfma_f32_operand_delivery.txt)",
                         "note": "per-kernel: algorithmic flops (2 N M K) of every ap_conv2d_fwd launch of one step / the HIP-event "
                                 "time of those launches on their launch stream (a separate pass of the same step, outside the "
                                 "timed region); by_kernel gives each kernel class its own rate",
                         "algorithmic": {"achieved": round(ktf_alg, 2), "frac": round(ktf_alg / PEAK_F32_MFMA_TFLOPS, 4),
                                         "note": "the layers' direct-form flops (2 N M K) / their kernel time; `achieved` / `frac` above price "
                                                 "the flops the kernels EXECUTE (the 3 x 3 layers run in F(2,3) form: 2/3 of 2 N M K)"},
                         "conv_launches": int(sum(ln)), "conv_ms_per_step": round(tot_ms, 3),
                         "conv_GFLOP_per_step": round(tot_fl / 1e9, 1), "conv_executed_GFLOP_per_step": round(tot_ex / 1e9, 1),
                         "by_kernel": by_kernel,
                         "whole_step": {"achieved": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                                        "event_ms_per_step": round(ev_ms, 3),
                                        "note": f"whole-step algorithmic flops ({n} x {UNET_GFLOP_PER_EVAL} + {RESNEXT29_GFLOP} GFLOP "
                                                "per sample) / HIP-event time of the step (GroupNorm, attention, mel and the sampler "
                                                "updates inside the interval)"}}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="clips per GPU per step")
    ap.add_argument("--reverse-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="time only --precision (skip the bf16 / f32s / f32d legs)")
    ap.add_argument("--precision", choices=["f32", "f32s", "bf16", "bf16s"], default="f32",
                    help="f32 = exact fp32 MFMA (headline, BASELINE configs[1]); bf16 = bf16 MFMA operands, fp32 accumulate/storage; "
                         "bf16s = the same with the residual stream stored as bf16")
    ap.add_argument("--sampler", choices=["ddpm", "sde"], default="ddpm")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs[3] / configs[4] legs")
    ap.add_argument("--no-caller-shapes", action="store_true", help="skip the callers' batch shapes leg (B = 1, 2, 10, 50, ...)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo rehearsal of the launch + gather protocol; no kernels")
    ap.add_argument("--chunk", type=int, default=0,
                    help="clips per chain call (0 = the engine's default for the mode): the batch of a step is walked in chunks "
                         "whose activations (3 x 65.5 MB per clip) stay resident in the 256 MB Infinity Cache")
    args = ap.parse_args()

    if args.chunk > 0 and args.batch % args.chunk:
        raise SystemExit("--chunk must divide --batch (the roofline object prices every launch at `chunk` clips)")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # Children, not exec: nothing in this process has touched a GPU, and it only relays the ranks' exit code.
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if world_env != args.gpus and "RANK" in os.environ:
        raise SystemExit(f"WORLD_SIZE={world_env} but --gpus {args.gpus}")
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    from audiopure_amd import synth, _native as N
    from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
    from audiopure_amd.audio_models.M5.M5Net import M5
    from audiopure_amd.acoustic_system import AcousticSystem
    from audiopure_amd.sharding import all_gather_scores

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP devices visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run even N=1 exercises RCCL
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    B, L, n = args.batch, 16000, args.reverse_steps
    import numpy as np
    import types
    cfg = dict(synth.FULL_WAVENET_CONFIG)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
    net = net.to(dev)
    m5 = M5(n_input=1, n_output=10)
    m5.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.m5_state_dict(10).items()})
    m5 = m5.to(dev).eval()
    # synthetic 0.5*U(-1,1) clips, generated on device (resident in HBM before timing)
    x0_full = global_batch_shard(torch, world, rank, B, L, dev)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_mode(precision, steps, warmup, sampler=None, n=n, batch=None, profile=True):
        """W untimed + K timed passes of the hot path in one arithmetic mode -> (elapsed s, kernel ms, launches).
        `batch`: the first `batch` clips of the step's batch (the callers' shapes leg); default the whole batch."""
        sampler = sampler or args.sampler
        x0 = x0_full if batch is None else x0_full[:batch].contiguous()
        B = x0.shape[0]
        net.set_precision(precision)
        dw = DiffWave(model=net, diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG),
                      reverse_timestep=n)
        dw.set_noise_source(("philox", 1234, rank * B))          # global utterance index = rank*B + b
        defender = dw
        if sampler == "sde":                                     # BASELINE configs[3]: RevDiffWave VP-SDE Euler chain
            from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
            defender = RevDiffWave.from_model(dw, types.SimpleNamespace(
                t=n, score_type="guided_diffusion", rand_t=False, t_delta=0, use_bm=False, sample_step=1))
        system = AcousticSystem(classifier=m5, transform=None, defender=defender, defense_type="wave")
        eng = net.engine()
        eng.max_chunk = args.chunk if args.chunk > 0 else B
        run_mode.chunk = min(eng.max_chunk, B)

        # purify (n reverse steps) + classify, all in HIP; then the path's only collective: [B,10] scores / rank
        step = make_step(lambda x: system(x, True), x0, use_dist, world * B)

        with torch.no_grad():
            for _ in range(warmup):
                step()
            # (the per-launch event hook keeps small-batch chains out of their replayed graph: the callers' shapes leg times without it)
            N.check(eng.lib.ap_profile_enable(eng.ctx, 1 if profile else 0))
            fence()
            with PowerSampler(local, enabled=rank == 0, pci_bus_id=device_descriptor(torch, local).get("pci_bus_id")) as ps:    # (rank 0's GPU only)
                t0 = time.perf_counter()
                for _ in range(steps):
                    lp = step()
                fence()
                elapsed = time.perf_counter() - t0
            run_mode.power = ps.result()
        ms2, n2 = (C.c_double * 2)(), (C.c_int64 * 2)()            # [0] residual-block launches, [1] skip-GEMM launches (bf16 mode)
        N.check(eng.lib.ap_profile_read_split(eng.ctx, ms2, n2))
        N.check(eng.lib.ap_profile_enable(eng.ctx, 0))
        assert torch.isfinite(lp).all()
        run_mode.last_scores = lp                                 # [world * B, 10] after the gather (every rank holds all rows)
        run_mode.local_elapsed = elapsed                          # this rank's own clock, before the max over ranks
        run_mode.split = {"block_ms": ms2[0] / max(n2[0], 1), "block_launches": int(n2[0]),
                          "skip_gemm_ms_per_block_launch": ms2[1] / max(n2[0], 1), "skip_gemm_launches": int(n2[1]),
                          "skip_gemm_ms_per_launch": ms2[1] / max(n2[1], 1), "skip_group": int(getattr(eng, "skip_group", 0) or 0)}
        if use_dist:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # per residual LAYER: the block launch plus its share of the group's skip GEMM (zero in the fp32-class modes)
        return elapsed, (ms2[0] + ms2[1]) / max(n2[0], 1), int(n2[0])

    def roofline(precision, k_ms, launches):
        Bl = getattr(run_mode, "chunk", B)                       # clips per residual-block launch
        return roofline_b(precision, k_ms, launches, Bl)

    def roofline_b(precision, k_ms, launches, B):
        achieved = FLOP_PER_LAYER_UTT * B / (k_ms * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(B, precision)
        if precision == "f32" and N.lib().ap_ctx_get_f32_form(net.engine().ctx) == 1:
            # The block computes the reference's layer (16.777 GFLOP as stated) with 12.583 GFLOP of matrix work: `achieved` / `frac`
            # price the flops the kernel EXECUTES against the fp32 MFMA peak (the kernel-quality figure); `algorithmic` is SURVEY
            # 8(d)'s figure -- the layer's stated flops per launch / the launch time -- which may exceed the peak: that excess is
            # the arithmetic the minimal-filtering form removed, not matrix-pipe throughput.
            executed = FLOP_EXEC_F32W_UTT * B / (k_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "resblock_f32w_kernel (F(2,3) over the dilation pair; persistent, one 512-register wave per SIMD)",
                    "achieved": round(executed, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "executed_flop_per_launch": FLOP_EXEC_F32W_UTT * B,
                    "algorithmic": {"achieved": round(achieved, 2), "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                                    "flop_per_launch": FLOP_PER_LAYER_UTT * B,
                                    "note": "SURVEY 8(d) flops of the layer (direct form) / launch time"},
                    "note": "4 instead of 6 [2C x C] products per dilation pair: 12.583 of the layer's 16.777 GFLOP per clip are executed; "
                            "v_mfma_f32_32x32x2_f32 issues at 0.98 of this peak from registers and at 0.84-0.94 when its operands "
                            "arrive from LDS (profiles/r3_mfma_f32_operand_delivery.txt)"}
        elif precision in ("f32", "f32d"):
            roof = {"bound": "mfma", "kernel": "resblock_f32_kernel<256,64>", "achieved": round(achieved, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": traffic,
                    "note": "v_mfma_f32_32x32x2_f32 issues at 0.98 of this peak from registers and at 0.84-0.94 when its operands "
                            "arrive from LDS (4-byte / 16-byte reads): profiles/r3_mfma_f32_operand_delivery.txt, DESIGN.md 3.5"}
        elif precision == "f32s":
            # fp32 operands as three bf16 parts, 6 bf16 MFMAs per fp32 MFMA-equivalent: the ceiling for ALGORITHMIC
            # flops is the dense bf16 MFMA peak / 6
            peak = 2500.0 / 6.0
            roof = {"bound": "mfma", "kernel": "resblock_f32s_kernel<256>", "achieved": round(achieved, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "traffic": traffic,
                    "note": "algorithmic fp32 flops; each is 6 v_mfma_f32_32x32x16_bf16 partial products (exact 3-way "
                            "bf16 operand split, fp32 accumulate), so peak = 2500 TFLOP/s dense bf16 / 6",
                    "executed_bf16_TFLOPs": round(6 * achieved, 1)}
        elif precision == "bf16s":
            # SURVEY 8(d), third precision row: bf16 MFMA + bf16 storage is COMPUTE-bound (830 against 1 356 utt/s per GPU), so the
            # line is priced on the dense bf16 MFMA peak; the HBM side is reported beside it on bf16-storage algorithmic bytes
            # (32.8 MB per clip and layer: read u, write u', read + write skip at two bytes per element).
            sp = getattr(run_mode, "split", {})
            gbs = BYTES_PER_LAYER_UTT_BF16STORE * B / (k_ms * 1e-3) / 1e9
            roof = {"bound": "mfma",
                    "kernel": "resblock_bf16u_kernel (persistent; bf16 u image in, u' + bf16 gate image out) + skipgemm_bf16_kernel (one "
                              f"K-concatenated skip GEMM per {sp.get('skip_group', 0) or 36} layers)",
                    "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": traffic,
                    "hbm": {"algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4),
                            "algorithmic_bytes_per_launch": BYTES_PER_LAYER_UTT_BF16STORE * B,
                            "physical_GBps": round(traffic / (k_ms * 1e-3) / 1e9, 1) if traffic else None,
                            "note": "bf16-storage algorithmic bytes (SURVEY 8d: 32.8 MB per clip and layer); the kernels move u in, u' + a bf16 "
                                    "gate image out, the image in once more and fp32 skip out once per group"},
                    "per_layer_ms": {"block_launch": round(sp.get("block_ms", k_ms), 4),
                                     "skip_gemm_share": round(sp.get("skip_gemm_ms_per_block_launch", 0.0), 4),
                                     "skip_gemm_launches": sp.get("skip_gemm_launches", 0),
                                     "skip_gemm_ms_per_launch": round(sp.get("skip_gemm_ms_per_launch", 0.0), 4)},
                    "accounting": "achieved = the layer's algorithmic flops (16.777 GFLOP per clip: SURVEY 8d) / (block launch + its share of "
                                  "the group's skip GEMM), against the dense bf16 MFMA peak",
                    "note": "power-capped like AP_PREC_BF16 (see this leg's `power`): ~0.63 pJ per bf16 flop on random operands is 5.4 J "
                            "per 512-clip launch = 3.9 ms at 1400 W before any byte moves (profiles/r3_mfma_power_calibration.txt)"}
        else:
            gbs = BYTES_PER_LAYER_UTT * B / (k_ms * 1e-3) / 1e9
            sp = getattr(run_mode, "split", {})
            roof = {"bound": "hbm",
                    "kernel": "resblock_bf16p_kernel<DS> (persistent; h' + bf16 gate image) + skipgemm_bf16_kernel (one K-concatenated "
                              f"skip GEMM per {sp.get('skip_group', 0)} layers)" if sp.get("skip_group") else "resblock_bf16p_kernel (persistent, fused skip)",
                    "achieved": round(gbs, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": traffic,
                    # what the memory system actually moved (the committed PMC passes' bytes per layer / this run's time): the
                    # deferred-skip kernels move FEWER bytes than the algorithmic figure `achieved` prices
                    "physical_GBps": round(traffic / (k_ms * 1e-3) / 1e9, 1) if traffic else None,
                    "physical_frac": round(traffic / (k_ms * 1e-3) / 8e12, 4) if traffic else None,
                    "mfma_TFLOPs": round(achieved, 1), "mfma_frac_of_2500": round(achieved / 2500.0, 4),
                    "per_layer_ms": {"block_launch": round(sp.get("block_ms", k_ms), 4),
                                     "skip_gemm_share": round(sp.get("skip_gemm_ms_per_block_launch", 0.0), 4),
                                     "skip_gemm_launches": sp.get("skip_gemm_launches", 0),
                                     "skip_gemm_ms_per_launch": round(sp.get("skip_gemm_ms_per_launch", 0.0), 4)},
                    "accounting": "achieved = the layer's ALGORITHMIC bytes (read h, write h', read + write fp32 skip: 65.5 MB per clip, "
                                  "SURVEY 8d) / (block launch + its share of the group's skip GEMM); the deferred-skip form moves fewer "
                                  "bytes than that (h in, h' + a bf16 gate image out, the image in once more, skip once per group)",
                    "note": "the kernel runs at the board's 1400 W power cap (see this leg's `power`); calibrated with bare MFMA and "
                            "HBM-copy kernels (profiles/r3_mfma_power_calibration.txt: ~0.63 pJ per bf16 flop on random operands, ~120 pJ "
                            "per HBM byte) the ceiling of this algorithm under the cap is 0.48-0.54 of the 8 TB/s roofline; a synthetic kernel with the same "
                            "MFMA : HBM mix and nothing else measures 0.46-0.48 (profiles/r3_mfma_hbm_mix.txt, DESIGN.md 3.4)"}
        bpl = BYTES_PER_LAYER_UTT_BF16STORE if precision == "bf16s" else BYTES_PER_LAYER_UTT
        roof.update({"traffic_source": traffic_source, "launches": launches, "clips_per_launch": B, "avg_launch_ms": round(k_ms, 4),
                     "flop_per_launch": FLOP_PER_LAYER_UTT * B, "algorithmic_bytes_per_launch": bpl * B,
                     "hbm_algorithmic_GBps": round(bpl * B / (k_ms * 1e-3) / 1e9, 1),
                     "hbm_frac_of_8TBps": round(bpl * B / (k_ms * 1e-3) / 8e12, 4)})
        return roof

    elapsed, k_ms, launches = run_mode(args.precision, args.steps, args.warmup)
    head_digest = scores_digest(run_mode.last_scores)
    head_scores = run_mode.last_scores.detach().float().cpu()   # the same clips and the same Philox draws run in every mode below
    head_power = run_mode.power
    head_roof = roofline(args.precision, k_ms, launches)        # (now: run_mode.chunk / .split describe the run just made)
    ranks = rank_evidence(use_dist, run_mode.local_elapsed, device_descriptor(torch, local), "nccl")
    # the other arithmetic modes of the same path, measured in the same run (N = 1 only: they are extra evidence,
    # not the headline): same inputs, same chain, same timing brackets
    others = {}
    if world == 1 and not args.no_other_modes:
        for prec in ("bf16", "bf16s", "f32s", "f32d", "f32"):
            if prec == args.precision:
                continue
            if prec == "f32d" or (prec == "f32" and args.precision != "f32"):   # (one step: the direct-form A/B of the same run)
                e2, k2, l2 = run_mode(prec, 1, 0)
                st = 1
            else:                                  # extra evidence, not the headline: bounded so the default run stays short
                st = max(1, min(args.steps, 3))
                e2, k2, l2 = run_mode(prec, st, min(args.warmup, 1))
            others[prec] = {"arithmetic": PREC_NAME[prec], "value": round(B * st / e2, 3), "unit": "utterances/s",
                            "steps": st, "ms_per_step": round(e2 * 1e3 / st, 3), "roofline": roofline(prec, k2, l2),
                            "power": run_mode.power}
            sc = run_mode.last_scores.detach().float().cpu()
            # the reduced-precision paths' agreement with the headline arithmetic on the same clips and noise, reported
            others[prec]["scores_vs_headline"] = {
                "clips": int(sc.shape[0]), "classes_in_headline_decisions": int(head_scores.argmax(1).unique().numel()),
                "argmax_agreement": round(float((sc.argmax(1) == head_scores.argmax(1)).float().mean()), 5),
                "max_abs_dlogp": float(f"{float((sc - head_scores).abs().max()):.3e}")}

    other_configs = {}
    if world == 1 and not args.no_other_configs:
        # BASELINE configs[3]: DiffWave VP-SDE reverse (diffwave_sde.py), n = 10, batch 512, bf16
        st = max(1, min(args.steps, 2))
        e3, k3, l3 = run_mode("bf16", st, 1, sampler="sde", n=10)
        other_configs["configs[3]"] = {
            "workload": f"DiffWave VP-SDE (RevDiffWave Euler chain) n=10 + M5 classify, batch={B}, bf16 MFMA operands / fp32 "
                        "accumulate and storage, 1 s @ 16 kHz clips",
            "value": round(B * st / e3, 3), "unit": "utterances/s", "steps": st, "warmup": 1,
            "ms_per_step": round(e3 * 1e3 / st, 3), "dtype": "bf16", "roofline": roofline("bf16", k3, l3), "power": run_mode.power}
        # the same workload with the residual stream stored as bf16 (AP_PREC_BF16_STORE: SURVEY 8d's third precision row) -- its own
        # arithmetic (one more rounding per layer), reported beside configs[3], not in place of it
        e3s, k3s, l3s = run_mode("bf16s", 1, 1, sampler="sde", n=10)
        other_configs["configs[3]"]["bf16_storage"] = {
            "value": round(B / e3s, 3), "unit": "utterances/s", "steps": 1, "warmup": 1, "ms_per_step": round(e3s * 1e3, 3), "dtype": "bf16",
            "arithmetic": PREC_NAME["bf16s"], "roofline": roofline("bf16s", k3s, l3s), "power": run_mode.power}
        net.set_precision(args.precision)
        other_configs["configs[4]"] = bench_config4(dev, max(1, min(args.steps, 5)))

    # the batch shapes the reference's callers use (adaptive_attack_eval.py:47,156-160: --batch_size 10, FAKEBOB's 10 x 50 copies;
    # certified_robust.py:46-57: one-shot denoise of a batch of copies; white_box_attack.py:392,437-439: PGD through the purifier
    # at B = 10; BASELINE configs[0]: B = 2, n = 1): per-clip rate of the same path relative to the 512-clip batch
    caller_shapes = None
    if world == 1 and not args.no_other_configs and not args.no_caller_shapes:
        from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
        dh_ = calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
        ref_rate = {"f32": (world * B * args.steps / elapsed) if args.precision == "f32" else others.get("f32", {}).get("value"),
                    "bf16": (world * B * args.steps / elapsed) if args.precision == "bf16" else others.get("bf16", {}).get("value"),
                    "bf16s": (world * B * args.steps / elapsed) if args.precision == "bf16s" else others.get("bf16s", {}).get("value")}

        def timed(fn, reps=2):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps

        caller_shapes = {"note": "DDPM n=5 + M5 at the callers' batch sizes: utterances/s and the per-clip rate relative to this run's "
                                 f"{B}-clip batch in the same arithmetic mode; then one_shot_denoise + M5 at B=10 (certification), one "
                                 "white-box gradient step (RevDiffWave t=5 + M5 + nll_loss, forward + backward w.r.t. the audio) at B=10, and "
                                 "BASELINE configs[0]'s B=2, n=1 on the GPU", "ddpm_n5": {}}
        for prec in ("f32", "bf16", "bf16s"):
            rows = {}
            for b in (1, 2, 10, 50) + ((500,) if prec != "f32" else ()):
                if b > B:
                    continue
                st = 2
                st = 4 if b <= 16 else 2
                e_, _, _ = run_mode(prec, st, 2 if b <= 16 else 1, batch=b, profile=False)   # (B <= 16: the chain is a replayed HIP graph)
                v = b * st / e_
                rows[f"B={b}"] = {"value": round(v, 3), "unit": "utterances/s", "ms_per_step": round(e_ * 1e3 / st, 3),
                                  "graph_replay": bool(b * L <= DiffWave.GRAPH_MAX_SAMPLES and DiffWave.graph_replay),
                                  "per_clip_rate_vs_full_batch": round(v / ref_rate[prec], 4) if ref_rate.get(prec) else None}
            caller_shapes["ddpm_n5"][prec] = rows
        x10 = x0_full[:min(10, B)].contiguous()
        y10 = torch.zeros(x10.shape[0], dtype=torch.long, device=dev)
        one_shot, grad_step = {}, {}
        for prec in ("f32", "bf16", "bf16s"):
            net.set_precision(prec)
            dwc = DiffWave(model=net, diffusion_hyperparams=dh_, reverse_timestep=n)
            dwc.set_noise_source(("philox", 1234, 0))
            with torch.no_grad():
                t_os = timed(lambda: m5(dwc.one_shot_denoise(x10)))
            one_shot[prec] = {"B": x10.shape[0], "ms": round(t_os * 1e3, 3), "clips_per_s": round(x10.shape[0] / t_os, 2)}
            # fp32 keeps the pre-gate activations and runs ap_resblock_bwd; bf16 / bf16s keep the gate's derivative factors (an fp16 pair per
            # element) and run ap_resblock_bwd_bf16_saved on the bf16 matrix pipe
            runner = RevDiffWave.from_model(dwc, types.SimpleNamespace(t=n, score_type="guided_diffusion", rand_t=False, t_delta=0,
                                                                         use_bm=False, sample_step=1))
            sysg = AcousticSystem(classifier=m5, transform=None, defender=runner, defense_type="wave")

            def gstep():
                xg = x10.clone().requires_grad_(True)
                torch.nn.functional.nll_loss(sysg(xg, True), y10).backward()
                return xg.grad

            t_g = timed(gstep)
            with torch.no_grad():
                t_f = timed(lambda: sysg(x10, True), reps=2)
            grad_step[prec] = {"B": x10.shape[0], "ms": round(t_g * 1e3, 2), "clips_per_s": round(x10.shape[0] / t_g, 2),
                               "forward_only_ms": round(t_f * 1e3, 2), "step_over_forward": round(t_g / t_f, 2)}
        caller_shapes["one_shot_denoise_B10"] = one_shot
        caller_shapes["white_box_gradient_step_B10"] = grad_step
        e0_, k0_, _ = run_mode("f32", 3, 2, n=1, batch=min(2, B), profile=False)
        caller_shapes["configs[0]_on_gpu"] = {"workload": "DiffWave DDPM n=1 + M5, batch=2, fp32 (BASELINE configs[0], the CPU reference's case)",
                                              "value": round(min(2, B) * 3 / e0_, 3), "unit": "utterances/s", "ms_per_step": round(e0_ * 1e3 / 3, 3)}
        net.set_precision(args.precision)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B * args.steps / elapsed
        roof = head_roof
        out = {
            "metric": f"purified 1s@16kHz utterances/sec at {n} reverse steps",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32s" else ("bf16" if args.precision == "bf16s" else args.precision), "data": "synthetic",
            "config": {"workload": f"DiffWave {args.sampler.upper()} purify n={n} + M5 classify, batch={B}/GPU, 1 s @ 16 kHz "
                                   f"clips, {PREC_NAME[args.precision]} (BASELINE.json configs[{3 if args.precision == 'bf16' else 1}]); "
                                   "shipped config C=S=256, 36 layers",
                       "global_batch": world * B, "clip_samples": L, "reverse_steps": n,
                       "parallelism": f"utterance-sharded x{world}, logits all_gather"},
            "roofline": roof,
            "power": head_power,
            "ranks": ranks,
            # digest of the job's gathered [global_batch, 10] scores: equal for every split of the same global batch over ranks
            "scores": head_digest,
        }
        if others:
            out["other_modes"] = others
        if other_configs:
            out["other_configs"] = other_configs
        if caller_shapes:
            out.setdefault("other_configs", {})["caller_shapes"] = caller_shapes
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
