#!/usr/bin/env python3
"""Copy the evidence of one tools/profile_round2.sh run (gpurun_out/<dir>) into profiles/ under the round's names and refresh the
PMC summary files bench.py reads its `traffic` from.   python tools/update_profiles.py gpurun_out/r2m [round-prefix, default r2]"""
import json, os, shutil, sys

src = sys.argv[1]
rp = sys.argv[2] if len(sys.argv) > 2 else "r2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
ALG = 33554432000.0                                   # (2C + 2S) L 4 B x 512 clips


def text(name, dst):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f, open(os.path.join(P, dst), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)


shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, f"{rp}_kernel_stats.csv"))
text("phase_trace.txt", f"{rp}_bf16_phase_trace.txt")
text("ablation.txt", f"{rp}_bf16_ablation_persistent.txt")
text("cmp_kernels.txt", f"{rp}_bf16_persistent_vs_pertile.txt")
text("fetch_by_layer.txt", f"{rp}_bf16_fetch_by_layer.txt")
for n, d in (("bench.json", f"{rp}_bench.json"), ("bench_torchrun_n1.json", f"{rp}_bench_torchrun_n1.json")):
    if os.path.exists(os.path.join(src, n)):
        shutil.copy(os.path.join(src, n), os.path.join(P, d))
pm = json.load(open(os.path.join(src, "pmc_summary.json")))
for prec, fn in (("bf16", f"{rp}_bf16_pmc_traffic.json"), ("f32", f"{rp}_pmc_traffic.json")):
    o = json.load(open(os.path.join(P, fn)))
    v = pm[prec]
    rd, wr = v["FETCH_SIZE"] * 2 * 1024, v["WRITE_SIZE"] * 1024
    o.update({"FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB_raw": v["WRITE_SIZE"], "fetch_bytes_corrected": rd, "write_bytes": wr,
              "traffic_bytes_per_launch": rd + wr, "traffic_over_algorithmic": (rd + wr) / o["algorithmic_bytes_per_launch"],
              "TCC_HIT_sum": v["TCC_HIT_sum"], "TCC_MISS_sum": v["TCC_MISS_sum"], "measured": f"final build of the round ({src} via tools/profile_round2.sh)"})
    json.dump(o, open(os.path.join(P, fn), "w"), indent=1)
    print(fn, round(o["traffic_over_algorithmic"], 3))
s = pm["sq_bf16_B256"]
o = json.load(open(os.path.join(P, f"{rp}_pmc_sq_bf16.json")))
o["per_dispatch"] = s
o["fractions_of_wave_cycles"] = {k: round(s[k] / s["SQ_WAVE_CYCLES"], 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
o["lds_bank_conflict_share_of_lds_cycles"] = round(s["SQ_LDS_BANK_CONFLICT"] / s["SQ_LDS_IDX_ACTIVE"], 4)
o["mfma_busy_of_cu_busy"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / s["SQ_BUSY_CU_CYCLES"] / 4, 4)
json.dump(o, open(os.path.join(P, f"{rp}_pmc_sq_bf16.json"), "w"), indent=1)
print("sq:", o["fractions_of_wave_cycles"], o["lds_bank_conflict_share_of_lds_cycles"], o["mfma_busy_of_cu_busy"])
sp = os.path.join(os.path.dirname(src.rstrip("/")), "split_pmc.json")
if os.path.exists(sp):
    d = json.load(open(sp))
    for p_, k in (("f32s", "resblock_f32s_kernel<256>"), ("f32h", "resblock_f32h_kernel<256>")):
        if p_ not in d or "FETCH_SIZE" not in d[p_]:
            continue
        v = d[p_]
        rd, wr = v["FETCH_SIZE"] * 2 * 1024, v["WRITE_SIZE"] * 1024
        o = {"kernel": k, "batch": 512, "launch": f"tools/run_resblock.py 512 {p_} 2 (layer 5, d = 32), one rocprofv3 --pmc pass per counter set",
             "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB_raw": v["WRITE_SIZE"], "fetch_bytes_corrected": rd, "write_bytes": wr,
             "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": ALG, "traffic_over_algorithmic": (rd + wr) / ALG,
             "TCC_HIT_sum": v.get("TCC_HIT_sum"), "TCC_MISS_sum": v.get("TCC_MISS_sum"),
             "note": "FETCH_SIZE doubled per the gfx950 calibration; final build of the round (nt policy, dilation-strided walk)"}
        json.dump(o, open(os.path.join(P, f"{rp}_{p_}_pmc_traffic.json"), "w"), indent=1)
        print(p_, round(o["traffic_over_algorithmic"], 3))
