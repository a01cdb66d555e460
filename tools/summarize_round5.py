#!/usr/bin/env python3
"""Turn one tools/profile_round5.sh run (gpurun_out/<dir>) into the committed evidence under profiles/ (r5_*).
   python tools/summarize_round5.py gpurun_out/r5"""
import json
import os
import shutil
import sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
B = 512
ALG = (2 * 256 + 2 * 256) * 16000 * 4.0 * B                      # bytes per layer launch: read h, write h', read + write skip
FLOP_DIRECT = 2.0 * 16000 * (512 * 768 + 512 * 256) * B
FLOP_EXEC = 2.0 * 16000 * (512 * 512 + 512 * 256) * B


def cp(name, dst):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(P, dst))
        return True
    print("missing:", name)
    return False


for a, b in (("bench.json", "r5_bench.json"), ("bench_torchrun_n1.json", "r5_bench_torchrun_n1.json"), ("kernel_stats.csv", "r5_kernel_stats.csv"),
             ("f32w_ablation.txt", "r5_f32w_ablation.txt"), ("f32w_ab.txt", "r5_f32w_ab.txt"), ("conv_w3_ab.txt", "r5_conv_w3_ab.txt"),
             ("cfg4_conv_by_shape.txt", "r5_cfg4_conv_by_shape.txt"), ("cfg4_kernel_stats.csv", "r5_cfg4_kernel_stats.csv"),
             ("whitebox.txt", "r5_whitebox_gradient_step.txt"), ("whitebox_kernel_stats.csv", "r5_whitebox_kernel_stats.csv"),
             ("whitebox_bf16.txt", "r5_whitebox_bf16_gradient_step.txt"), ("whitebox_bf16_kernel_stats.csv", "r5_whitebox_bf16_kernel_stats.csv"),
             ("pmc_bwdb_gate/summary.json", "r5_whitebox_bf16_gate_kernel_pmc.json"), ("pmc_bwdb_conv/summary.json", "r5_whitebox_bf16_conv_kernel_pmc.json"),
             ("bwd_bf16_deviation.txt", "r5_whitebox_bf16_deviation.txt"),
             ("adversarial_error.txt", "r5_fp32_class_adversarial_error_rerun.txt")):
    cp(a, b)
for key, kern, flops in (("f32w", "resblock_f32w_kernel (F(2,3) form)", FLOP_EXEC), ("f32d", "resblock_f32_kernel<256,64> (direct form)", FLOP_DIRECT)):
    p = os.path.join(src, f"pmc_{key}", "summary.json")
    if not os.path.exists(p):
        print("missing:", p)
        continue
    d = json.load(open(p))
    rd, wr = d.get("fetch_bytes_corrected", 0.0), d.get("write_bytes", 0.0)
    o = {"kernel": kern, "batch": B, "launch": f"tools/run_resblock.py {B} {'f32' if key == 'f32w' else 'f32d'} 2 (layer 5, d = 32) under rocprofv3: one plain --kernel-trace "
                                               "pass, then --pmc SQ+GRBM / FETCH_SIZE / WRITE_SIZE passes of the same command; the LAST dispatch",
         "ms_per_launch_profiled": d.get("ms_last"), "FETCH_SIZE_KB_raw": d.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": d.get("WRITE_SIZE"),
         "fetch_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": ALG,
         "traffic_over_algorithmic": (rd + wr) / ALG, "executed_flop_per_launch": flops,
         "note": "FETCH_SIZE doubled per the gfx950 calibration (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
         "sq": {k: d[k] for k in d if k.startswith("SQ_") or k.startswith("GRBM")},
         "fractions_of_wave_cycles": {k: round(d[k], 4) for k in ("frac_wait_any", "frac_wait_inst", "frac_active") if k in d},
         "mfma_busy_of_cu_busy": round(d.get("mfma_busy_of_cu_busy", 0.0), 4),
         "clock_GHz": round(d["GRBM_GUI_ACTIVE"] / 8 / (d["ms_last"] * 1e-3) / 1e9, 3) if d.get("GRBM_GUI_ACTIVE") and d.get("ms_last") else None}
    json.dump(o, open(os.path.join(P, f"r5_{key}_pmc_traffic.json"), "w"), indent=1)
    print(key, "traffic x algorithmic", round(o["traffic_over_algorithmic"], 3), "mfma busy", o["mfma_busy_of_cu_busy"], "clock", o["clock_GHz"])
p = os.path.join(src, "pmc_w3", "summary.json")
if os.path.exists(p):
    shutil.copy(p, os.path.join(P, "r5_conv_w3_pmc.json"))
try:
    d = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    print("bench:", d["value"], "utt/s, roofline", d["roofline"]["frac"], "algorithmic", d["roofline"].get("algorithmic", {}).get("frac"))
    for k, v in d.get("other_modes", {}).items():
        print("  ", k, v["value"], v["roofline"]["frac"])
    oc = d.get("other_configs", {})
    print("   cfg3", oc.get("configs[3]", {}).get("value"), "cfg4", oc.get("configs[4]", {}).get("value"))
except Exception as e:
    print("bench.json unreadable:", e)
