#!/usr/bin/env python3
"""Turn one tools/profile_round6.sh run (gpurun_out/<dir>) into the committed evidence under profiles/ (r6_*).
   python tools/summarize_round6.py gpurun_out/r6"""
import json
import os
import shutil
import subprocess
import sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
B = 512
ALG = (2 * 256 + 2 * 256) * 16000 * 4.0 * B
FLOP_EXEC = 2.0 * 16000 * (512 * 512 + 512 * 256) * B


def cp(name, dst, json_line=False):
    p = os.path.join(src, name)
    if not (os.path.exists(p) and os.path.getsize(p) > 0):
        print("missing:", name)
        return False
    if json_line:                                        # the JSON line only (RCCL prints its banner on stdout too)
        lines = [l for l in open(p) if l.lstrip().startswith("{")]
        if not lines:
            print("no JSON line in", name)
            return False
        open(os.path.join(P, dst), "w").write(lines[-1])
    else:
        shutil.copy(p, os.path.join(P, dst))
    return True


cp("bench.json", "r6_bench.json", True)
cp("bench_torchrun_n1.json", "r6_bench_torchrun_n1.json", True)
for a, b in (("kernel_stats.csv", "r6_kernel_stats.csv"), ("f32s_forms.txt", "r6_f32s_forms_rerun.txt"), ("bf16u_flops_ab.txt", "r6_bf16s_fewer_products_upper_bound_rerun.txt"),   # (runs up to tree f081186 only: the hook was removed afterwards)
            
             ("whitebox.txt", "r6_whitebox_gradient_step.txt"), ("whitebox_bf16.txt", "r6_whitebox_bf16_gradient_step.txt"), ("whitebox_bf16s.txt", "r6_whitebox_bf16s_gradient_step.txt"),
             ("whitebox_kernel_stats.csv", "r6_whitebox_kernel_stats.csv"), ("whitebox_bf16_kernel_stats.csv", "r6_whitebox_bf16_kernel_stats.csv"),
             ("pmc_bwdb_gate/summary.json", "r6_whitebox_bf16_gate_kernel_pmc.json"), ("pmc_bwdb_conv/summary.json", "r6_whitebox_bf16_conv_kernel_pmc.json"),
             ("cfg4_conv_by_shape.txt", "r6_cfg4_conv_by_shape.txt"), ("cfg4_kernel_stats.csv", "r6_cfg4_kernel_stats.csv"),
             ("adversarial_error.txt", "r6_fp32_class_adversarial_error.txt")):
    cp(a, b)
p = os.path.join(src, "pmc_f32w", "summary.json")
if os.path.exists(p):
    d = json.load(open(p))
    rd, wr = d.get("fetch_bytes_corrected", 0.0), d.get("write_bytes", 0.0)
    o = {"kernel": "resblock_f32w_kernel (F(2,3) form)", "batch": B,
         "launch": f"tools/run_resblock.py {B} f32 2 (layer 5, d = 32) under rocprofv3: one plain --kernel-trace pass, then --pmc SQ+GRBM / FETCH_SIZE / "
                   "WRITE_SIZE passes of the same command; the LAST dispatch",
         "ms_per_launch_profiled": d.get("ms_last"), "FETCH_SIZE_KB_raw": d.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": d.get("WRITE_SIZE"),
         "fetch_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": ALG,
         "traffic_over_algorithmic": (rd + wr) / ALG, "executed_flop_per_launch": FLOP_EXEC,
         "note": "FETCH_SIZE doubled per the gfx950 calibration (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
         "sq": {k: d[k] for k in d if k.startswith("SQ_") or k.startswith("GRBM")},
         "fractions_of_wave_cycles": {k: round(d[k], 4) for k in ("frac_wait_any", "frac_wait_inst", "frac_active") if k in d},
         "mfma_busy_of_cu_busy": round(d.get("mfma_busy_of_cu_busy", 0.0), 4),
         "clock_GHz": round(d["GRBM_GUI_ACTIVE"] / 8 / (d["ms_last"] * 1e-3) / 1e9, 3) if d.get("GRBM_GUI_ACTIVE") and d.get("ms_last") else None}
    json.dump(o, open(os.path.join(P, "r6_f32w_pmc_traffic.json"), "w"), indent=1)
    print("f32w traffic x algorithmic", round(o["traffic_over_algorithmic"], 3), "mfma busy", o["mfma_busy_of_cu_busy"], "clock", o["clock_GHz"])
if os.path.isdir(os.path.join(src, "bf16_modes")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_bf16_modes.py"), os.path.join(src, "bf16_modes"), "r6"], env=dict(os.environ, B=str(B)),
                   stdout=subprocess.DEVNULL)
    for m in ("bf16", "bf16s"):
        import glob
        f = glob.glob(os.path.join(src, "bf16_modes", f"time_{m}", "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(P, f"r6_{m}_eps_kernel_stats.csv"))
try:
    d = json.loads(open(os.path.join(P, "r6_bench.json")).read())
    print("bench:", d["value"], "utt/s, roofline", d["roofline"]["frac"], "algorithmic", d["roofline"].get("algorithmic", {}).get("frac"))
    for k, v in d.get("other_modes", {}).items():
        print("  ", k, v["value"], v["roofline"]["frac"])
    oc = d.get("other_configs", {})
    print("   cfg3", oc.get("configs[3]", {}).get("value"), "cfg4", oc.get("configs[4]", {}).get("value"))
    wb = oc.get("caller_shapes", {}).get("white_box_gradient_step_B10", {})
    print("   white-box", {k: (v.get("ms"), v.get("step_over_forward")) for k, v in wb.items() if isinstance(v, dict)})
except Exception as e:
    print("bench.json unreadable:", e)
