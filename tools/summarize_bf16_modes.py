#!/usr/bin/env python3
"""Turn one tools/profile_bf16_modes.sh run (gpurun_out/<dir>) into committed evidence under profiles/:
<round>_bf16_modes_pmc_by_kernel.json and, per mode, <round>_bf16_pmc_traffic.json / <round>_bf16s_pmc_traffic.json (the files
bench.py quotes its `traffic` from).      python tools/summarize_bf16_modes.py gpurun_out/<dir> r6 [--no-copy]"""
import csv, glob, json, os, sys

src, rp = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r6")
copy = "--no-copy" not in sys.argv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
B, NL = int(os.environ.get("B", 512)), 36
ALG = {"bf16": (2 * 256 + 2 * 256) * 16000 * 4.0 * B,            # per layer, fp32 storage: read h, write h', read + write skip
       "bf16s": (2 * 256 + 2 * 256) * 16000 * 2.0 * B}           # per layer, bf16 storage (1.180 GB per clip and step / 36)
FLOP = 2.0 * 16000 * (512 * 768 + 512 * 256) * B
KERNELS = {"bf16": {"block": "resblock_bf16p_kernel", "skip_gemm": "skipgemm_bf16_kernel"},
           "bf16s": {"block": "resblock_bf16u_kernel", "skip_gemm": "skipgemm_bf16_kernel", "init_conv_u": "init_conv_u_kernel"}}
PER = {"block": NL, "skip_gemm": 1, "init_conv_u": 1}
SQC = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE")


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def last_rep(vals, per_rep):
    return vals[-per_rep:] if per_rep and len(vals) >= per_rep else []


def times(d, kern, per_rep):
    r = [x for x in rows(d, "kernel_trace.csv") if kern in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return last_rep([(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in r], per_rep)


def counter(d, kern, name, per_rep):
    r = [x for x in rows(d, "counter_collection.csv") if kern in x["Kernel_Name"] and x["Counter_Name"] == name]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return last_rep([float(x["Counter_Value"]) for x in r], per_rep)


summary = {"command": f"tools/run_eps_bf16.py {B} 2 -1 <mode> under rocprofv3 (tools/profile_bf16_modes.sh: one plain --kernel-trace pass, then separate "
                      "--pmc passes FETCH_SIZE / WRITE_SIZE / SQ+GRBM / LDS); figures are sums over the dispatches of the LAST evaluation, per kernel",
           "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; fetch_bytes = FETCH_SIZE x 2 x 1024 (gfx950 calibration, "
                    "MI355X_MICROARCH.md HBM); clock = GRBM_GUI_ACTIVE / 8 / profiled kernel time",
           "flop_per_layer": FLOP, "modes": {}}
for mode, kerns in KERNELS.items():
    f = {"algorithmic_bytes_per_layer": ALG[mode]}
    tot_ms, tot_traffic = 0.0, 0.0
    for key, kern in kerns.items():
        per = PER[key]
        t = times(f"time_{mode}", kern, per)
        fe, wr = counter(f"fetch_{mode}", kern, "FETCH_SIZE", per), counter(f"write_{mode}", kern, "WRITE_SIZE", per)
        e = {"dispatches_per_evaluation": per}
        if t:
            e["ms_per_evaluation"] = round(sum(t), 3)
            e["ms_per_dispatch"] = round(sum(t) / per, 4)
            if key == "block":
                e["ms_by_layer"] = [round(v, 3) for v in t]
            tot_ms += sum(t)
        if fe and wr:
            rd, w = sum(fe) * 2 * 1024, sum(wr) * 1024
            e.update({"fetch_bytes": rd, "write_bytes": w, "traffic_bytes": rd + w})
            tot_traffic += rd + w
        sq = {c: counter(f"sq_{mode}", kern, c, per) for c in SQC}
        sqt = times(f"sq_{mode}", kern, per)
        if all(sq.values()):
            s = {c: sum(v) for c, v in sq.items()}
            e["sq"] = {"wait_any": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4), "wait_inst_any": round(s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
                       "active_inst_any": round(s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
                       "mfma_busy_of_cu_busy": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_BUSY_CU_CYCLES"] / 4, 4)}
            if sqt:
                e["sq"]["clock_GHz"] = round(s["GRBM_GUI_ACTIVE"] / 8 / (sum(sqt) * 1e-3) / 1e9, 3)
        lc, li = counter(f"lds_{mode}", kern, "SQ_LDS_BANK_CONFLICT", per), counter(f"lds_{mode}", kern, "SQ_LDS_IDX_ACTIVE", per)
        if lc and li:
            e["lds_bank_conflict_share"] = round(sum(lc) / max(sum(li), 1), 4)
        f[key] = e
    if tot_ms:
        ms = tot_ms / NL
        f["per_layer"] = {"ms": round(ms, 4), "algorithmic_GBps": round(ALG[mode] / (ms * 1e-3) / 1e9, 1),
                          "hbm_frac_of_8TBps": round(ALG[mode] / (ms * 1e-3) / 8e12, 4), "TFLOPs": round(FLOP / (ms * 1e-3) / 1e12, 1),
                          "mfma_frac_of_2500": round(FLOP / (ms * 1e-3) / 2.5e15, 4)}
        if tot_traffic:
            f["per_layer"].update({"traffic_bytes": tot_traffic / NL, "traffic_over_algorithmic": round(tot_traffic / NL / ALG[mode], 4),
                                   "physical_GBps": round(tot_traffic / NL / (ms * 1e-3) / 1e9, 1)})
    summary["modes"][mode] = f
print(json.dumps(summary["modes"], indent=1))
if copy:
    json.dump(summary, open(os.path.join(P, f"{rp}_bf16_modes_pmc_by_kernel.json"), "w"), indent=1)
    for mode, f in summary["modes"].items():
        pl = f.get("per_layer", {})
        if "traffic_bytes" in pl:
            json.dump({"kernel": " + ".join(f"{k} x {PER[n]}" for n, k in KERNELS[mode].items()) + " per evaluation, per layer", "batch": B,
                       "launch": summary["command"], "traffic_bytes_per_launch": pl["traffic_bytes"], "algorithmic_bytes_per_launch": ALG[mode],
                       "traffic_over_algorithmic": pl["traffic_over_algorithmic"],
                       "note": f"FETCH_SIZE doubled per the gfx950 calibration; (sum over the dispatches of one eps evaluation) / 36; details in "
                               f"{rp}_bf16_modes_pmc_by_kernel.json"}, open(os.path.join(P, f"{rp}_{mode}_pmc_traffic.json"), "w"), indent=1)
else:
    json.dump(summary, open(os.path.join(src, "summary.json"), "w"), indent=1)
