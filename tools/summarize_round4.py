#!/usr/bin/env python3
"""Turn one tools/profile_round4.sh run (gpurun_out/<dir>) into the committed evidence under profiles/: r4_bench.json,
r4_bench_torchrun_n1.json, r4_kernel_stats.csv, r4_bf16_pmc_by_kernel.json (+ r4_bf16_pmc_traffic.json, which bench.py quotes its
`traffic` from), and the text files.      python tools/summarize_round4.py gpurun_out/r4"""
import csv, glob, json, os, shutil, sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
B, REPS, NL = 512, 2, 36
ALG = (2 * 256 + 2 * 256) * 16000 * 4.0 * B                      # per layer
FLOP = 2.0 * 16000 * (512 * 768 + 512 * 256) * B
KERNELS = {"block": "resblock_bf16p_kernel", "skip_gemm": "skipgemm_bf16_kernel"}
SQC = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES",
       "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE")


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def last_rep(vals, per_rep):
    """values in dispatch order -> those of the LAST evaluation (the first is the warm-up)"""
    return vals[-per_rep:] if per_rep and len(vals) >= per_rep else []


def times(d, kern, per_rep):
    r = [x for x in rows(d, "kernel_trace.csv") if kern in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return last_rep([(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in r], per_rep)


def counter(d, kern, name, per_rep):
    r = [x for x in rows(d, "counter_collection.csv") if kern in x["Kernel_Name"] and x["Counter_Name"] == name]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return last_rep([float(x["Counter_Value"]) for x in r], per_rep)


summary = {"command": f"tools/run_eps_bf16.py {B} {REPS} [skip_group] under rocprofv3 (one plain --kernel-trace pass, then --pmc FETCH_SIZE / "
                      "WRITE_SIZE / SQ+GRBM passes); figures are sums over the dispatches of the LAST evaluation, per kernel",
           "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; fetch_bytes = FETCH_SIZE x 2 x 1024 (gfx950 calibration, "
                    "MI355X_MICROARCH.md HBM); clock = GRBM_GUI_ACTIVE / 8 / profiled kernel time",
           "algorithmic_bytes_per_layer": ALG, "flop_per_layer": FLOP, "forms": {}}
for form in ("ds", "fused"):
    f = {}
    tot_ms, tot_traffic = 0.0, 0.0
    for key, kern in KERNELS.items():
        per = NL if key == "block" else (1 if form == "ds" else 0)
        if not per:
            continue
        t = times(f"time_{form}", kern, per)
        fe, wr = counter(f"fetch_{form}", kern, "FETCH_SIZE", per), counter(f"write_{form}", kern, "WRITE_SIZE", per)
        e = {"dispatches_per_evaluation": per}
        if t:
            e["ms_per_evaluation"] = round(sum(t), 3)
            e["ms_per_dispatch"] = round(sum(t) / per, 4)
            tot_ms += sum(t)
        if fe and wr:
            rd, w = sum(fe) * 2 * 1024, sum(wr) * 1024
            e.update({"fetch_bytes": rd, "write_bytes": w, "traffic_bytes": rd + w})
            tot_traffic += rd + w
        sq = {c: counter(f"sq_{form}", kern, c, per) for c in SQC}
        sqt = times(f"sq_{form}", kern, per)
        if all(sq.values()):
            s = {c: sum(v) for c, v in sq.items()}
            e["sq"] = {"wait_any": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4), "wait_inst_any": round(s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
                       "active_inst_any": round(s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
                       "mfma_busy_of_cu_busy": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_BUSY_CU_CYCLES"] / 4, 4),
                       "lds_bank_conflict_share": round(s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_LDS_IDX_ACTIVE"], 1), 4)}
            if sqt:
                e["sq"]["clock_GHz"] = round(s["GRBM_GUI_ACTIVE"] / 8 / (sum(sqt) * 1e-3) / 1e9, 3)
        f[key] = e
    if tot_ms:
        f["per_layer"] = {"ms": round(tot_ms / NL, 4), "algorithmic_GBps": round(ALG / (tot_ms / NL * 1e-3) / 1e9, 1),
                          "roofline_frac_of_8TBps": round(ALG / (tot_ms / NL * 1e-3) / 8e12, 4), "TFLOPs": round(FLOP / (tot_ms / NL * 1e-3) / 1e12, 1)}
        if tot_traffic:
            f["per_layer"].update({"traffic_bytes": tot_traffic / NL, "traffic_over_algorithmic": round(tot_traffic / NL / ALG, 4)})
    summary["forms"][form] = f
json.dump(summary, open(os.path.join(P, "r4_bf16_pmc_by_kernel.json"), "w"), indent=1)
ds = summary["forms"].get("ds", {}).get("per_layer", {})
if "traffic_bytes" in ds:
    json.dump({"kernel": "resblock_bf16p_kernel<DS> x 36 + skipgemm_bf16_kernel x 1 per evaluation, per layer", "batch": B,
               "launch": summary["command"], "traffic_bytes_per_launch": ds["traffic_bytes"], "algorithmic_bytes_per_launch": ALG,
               "traffic_over_algorithmic": ds["traffic_over_algorithmic"],
               "note": "FETCH_SIZE doubled per the gfx950 calibration; (sum over the 36 block launches and the skip GEMM of one eps "
                       "evaluation) / 36; details in r4_bf16_pmc_by_kernel.json"}, open(os.path.join(P, "r4_bf16_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summary["forms"], indent=1))


def text(name, dst):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(P, dst), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)


for n, d in (("bench.json", "r4_bench.json"), ("bench_torchrun_n1.json", "r4_bench_torchrun_n1.json"), ("kernel_stats.csv", "r4_kernel_stats.csv"),
             ("cfg4_kernel_stats.csv", "r4_cfg4_kernel_stats.csv")):
    if not os.path.exists(os.path.join(src, n)):
        continue
    if n.endswith(".json"):                          # the JSON line only (RCCL prints its version banner on stdout too)
        lines = [l for l in open(os.path.join(src, n)) if l.lstrip().startswith("{")]
        if lines:
            open(os.path.join(P, d), "w").write(lines[-1])
    else:
        shutil.copy(os.path.join(src, n), os.path.join(P, d))
text("ab_bf16_ds.txt", "r4_bf16_deferred_skip_ab.txt")
text("cfg4_conv_by_shape.txt", "r4_cfg4_conv_by_shape.txt")
text("mfma_hbm_mix.txt", "r4_mfma_hbm_mix.txt")
