#!/bin/bash
# SQ counters of the AP_PREC_F32_SPLIT block kernels (one launch set at B = 512): MFMA busy, wait fractions, LDS bank conflicts, clock.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${MODES:-f32sw f32s}; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/s2pmc_${m}_a -o r -- python3 $R/tools/run_resblock.py 512 $m 3 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/s2pmc_${m}_b -o r -- python3 $R/tools/run_resblock.py 512 $m 3 5 > /dev/null 2>&1
python3 - $R/gpurun_out/s2pmc_${m}_a $R/gpurun_out/s2pmc_${m}_b <<'PY'
import csv,glob,sys,collections
v=collections.defaultdict(lambda: collections.defaultdict(list)); t=collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:40]
            if 'f32s' in k: v[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:40]
            if 'f32s' in k: t[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,c in v.items():
    m={n:sum(x)/len(x) for n,x in c.items()}
    ms=sorted(t[k])[len(t[k])//2]
    print(k, f"{ms:.2f} ms  clock {m['GRBM_GUI_ACTIVE']/8/(ms*1e-3)/1e9:.2f} GHz  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/m['SQ_BUSY_CU_CYCLES']/4:.3f}  wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}  wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}  active_inst {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}")
    print("    lds conflict share", round(m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1),3), " valu active / wave cycles", round(m.get('SQ_ACTIVE_INST_VALU',0)/m['SQ_WAVE_CYCLES'],3), " lds active", round(m.get('SQ_ACTIVE_INST_LDS',0)/m['SQ_WAVE_CYCLES'],3), " wait lds", round(m.get('SQ_WAIT_INST_LDS',0)/m['SQ_WAVE_CYCLES'],3), " vmem inst cycles", round(m.get('SQ_INST_CYCLES_VMEM',0)/m['SQ_WAVE_CYCLES'],3))
PY
done
