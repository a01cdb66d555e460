#!/usr/bin/env python3
"""Turn one tools/profile_round3.sh run (gpurun_out/<dir>) into the committed evidence under profiles/:
   r3_bench.json, r3_bench_torchrun_n1.json, r3_kernel_stats.csv, r3_pmc_by_mode_and_layer.json (+ the per-mode traffic files
   bench.py reads its `traffic` from), and the text files of the tools-build passes.
   python tools/summarize_round3.py gpurun_out/r3"""
import collections, csv, glob, json, os, shutil, sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
B, LAYERS, REPS = 512, [0, 5, 9, 11], 3
ALG = (2 * 256 + 2 * 256) * 16000 * 4.0 * B
FLOP = 2.0 * 16000 * (512 * 768 + 512 * 256) * B
KERNEL = {"f32": "resblock_f32_kernel", "f32s": "resblock_f32s_kernel", "f32h": "resblock_f32h_kernel", "bf16": "resblock_bf16p_kernel"}


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def per_layer(d, prec, counter=None):
    """dispatches of the block kernel in launch order -> {layer: [values of the last REPS - 1 launches]} (first launch: warm-up)"""
    if counter is None:
        r = [x for x in rows(d, "kernel_trace.csv") if KERNEL[prec] in x["Kernel_Name"]]
        r.sort(key=lambda x: int(x["Start_Timestamp"]))
        vals = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in r]
    else:
        r = [x for x in rows(d, "counter_collection.csv") if KERNEL[prec] in x["Kernel_Name"] and x["Counter_Name"] == counter]
        r.sort(key=lambda x: int(x["Dispatch_Id"]))
        vals = [float(x["Counter_Value"]) for x in r]
    if len(vals) != REPS * len(LAYERS):
        return None
    return {l: vals[i * REPS + 1:(i + 1) * REPS] for i, l in enumerate(LAYERS)}


def mean(v):
    return sum(v) / len(v)


summary = {"command": f"tools/run_resblock_layers.py {B} <mode> {REPS} " + " ".join(map(str, LAYERS)) +
                      " under rocprofv3 (one plain --kernel-trace pass, then --pmc FETCH_SIZE / WRITE_SIZE / SQ+GRBM passes); "
                      "per layer: mean over the launches after the first",
           "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; fetch_bytes = FETCH_SIZE x 2 x 1024 (gfx950 calibration, "
                    "MI355X_MICROARCH.md HBM); clock = GRBM_GUI_ACTIVE / 8 / profiled kernel time",
           "algorithmic_bytes_per_launch": ALG, "flop_per_launch": FLOP, "modes": {}}
for prec in ("f32", "f32s", "f32h", "bf16"):
    t = per_layer(f"time_{prec}", prec)
    fe = per_layer(f"fetch_{prec}", prec, "FETCH_SIZE")
    wr = per_layer(f"write_{prec}", prec, "WRITE_SIZE")
    sqt = per_layer(f"sq_{prec}", prec)
    sq = {c: per_layer(f"sq_{prec}", prec, c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_LDS_BANK_CONFLICT",
                                                        "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")}
    m = {}
    for l in LAYERS:
        e = {"d": 1 << (l % 12)}
        if t:
            e["ms"] = round(mean(t[l]), 4)
            e["TFLOPs"] = round(FLOP / (e["ms"] * 1e-3) / 1e12, 1)
            e["algorithmic_GBps"] = round(ALG / (e["ms"] * 1e-3) / 1e9, 1)
        if fe and wr:
            rd, w = mean(fe[l]) * 2 * 1024, mean(wr[l]) * 1024
            e.update({"FETCH_SIZE_KB_raw": round(mean(fe[l]), 1), "WRITE_SIZE_KB_raw": round(mean(wr[l]), 1), "fetch_bytes": rd,
                      "write_bytes": w, "traffic_bytes": rd + w, "traffic_over_algorithmic": round((rd + w) / ALG, 4)})
        if all(sq.values()):
            s = {c: mean(v[l]) for c, v in sq.items()}
            wc = s["SQ_WAVE_CYCLES"]
            e["sq"] = {"wait_any": round(s["SQ_WAIT_ANY"] / wc, 4), "wait_inst_any": round(s["SQ_WAIT_INST_ANY"] / wc, 4),
                       "active_inst_any": round(s["SQ_ACTIVE_INST_ANY"] / wc, 4),
                       "mfma_busy_of_cu_busy": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_BUSY_CU_CYCLES"] / 4, 4),
                       "lds_bank_conflict_share": round(s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_LDS_IDX_ACTIVE"], 1), 4),
                       "raw": {k: round(v) for k, v in s.items()}}
            if sqt:
                e["sq"]["profiled_ms"] = round(mean(sqt[l]), 4)
                e["sq"]["clock_GHz"] = round(s["GRBM_GUI_ACTIVE"] / 8 / (mean(sqt[l]) * 1e-3) / 1e9, 3)
        m[str(l)] = e
    summary["modes"][prec] = m
    # the per-mode traffic file bench.py quotes: mean over the four layers
    tr = [e["traffic_bytes"] for e in m.values() if "traffic_bytes" in e]
    if tr:
        name = {"f32": "r3_pmc_traffic.json", "f32s": "r3_f32s_pmc_traffic.json", "f32h": "r3_f32h_pmc_traffic.json", "bf16": "r3_bf16_pmc_traffic.json"}[prec]
        json.dump({"kernel": KERNEL[prec], "batch": B, "layers": LAYERS, "launch": summary["command"],
                   "traffic_bytes_per_launch": mean(tr), "algorithmic_bytes_per_launch": ALG,
                   "traffic_over_algorithmic": round(mean(tr) / ALG, 4),
                   "per_layer_traffic_over_algorithmic": {k: e.get("traffic_over_algorithmic") for k, e in m.items()},
                   "note": "FETCH_SIZE doubled per the gfx950 calibration; mean over the layers d = 1, 32, 512, 2048; details in "
                           "r3_pmc_by_mode_and_layer.json"}, open(os.path.join(P, name), "w"), indent=1)
json.dump(summary, open(os.path.join(P, "r3_pmc_by_mode_and_layer.json"), "w"), indent=1)
for prec, m in summary["modes"].items():
    for l, e in m.items():
        print(prec, "layer", l, {k: v for k, v in e.items() if k in ("ms", "traffic_over_algorithmic", "algorithmic_GBps", "TFLOPs")},
              {k: v for k, v in e.get("sq", {}).items() if k != "raw"})


def text(name, dst):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(P, dst), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)


for n, d in (("bench.json", "r3_bench.json"), ("bench_torchrun_n1.json", "r3_bench_torchrun_n1.json"), ("kernel_stats.csv", "r3_kernel_stats.csv")):
    if not os.path.exists(os.path.join(src, n)):
        continue
    if n.endswith(".json"):                          # the JSON line only (RCCL prints its version banner on stdout too)
        lines = [l for l in open(os.path.join(src, n)) if l.lstrip().startswith("{")]
        if lines:
            open(os.path.join(P, d), "w").write(lines[-1])
    else:
        shutil.copy(os.path.join(src, n), os.path.join(P, d))
text("phase_trace.txt", "r3_bf16_phase_trace.txt")
text("ablation.txt", "r3_bf16_ablation_persistent.txt")
text("cmp_kernels.txt", "r3_bf16_persistent_vs_pertile.txt")
text("bf16w_experiment.txt", "r3_bf16w_one_wave_per_simd_experiment.txt")
text("cfg4_conv_by_shape.txt", "r3_cfg4_conv_by_shape.txt")
text("power_by_mode.txt", "r3_power_by_mode.txt")
text("bf16_energy_ablation.txt", "r3_bf16_energy_ablation.txt")
text("bf16_operand_images_experiment.txt", "r3_bf16_operand_images_experiment.txt")
if os.path.exists(os.path.join(src, "cfg4_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "cfg4_kernel_stats.csv"), os.path.join(P, "r3_cfg4_kernel_stats.csv"))
text("kstep_trace.txt", "r3_bf16_kstep_trace.txt")
text("mfma_power_calibration.txt", "r3_mfma_power_calibration.txt")
text("mfma_hbm_mix.txt", "r3_mfma_hbm_mix.txt")
