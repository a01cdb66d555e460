cd /tmp; export TMPDIR=/tmp
for prec in f32s bf16 f32; do
B=256; [ $prec = f32 ] && B=64
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_$prec -o r -- python3 $GRAFT_REPO_ROOT/tools/run_resblock.py $B $prec 3 > /tmp/clk_$prec.log 2>&1
python3 - $prec <<'PY'
import sys, csv, glob
prec = sys.argv[1]
cc = {}
for f in glob.glob(f"/tmp/clk_{prec}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resblock" in r["Kernel_Name"]: cc[r["Dispatch_Id"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(f"/tmp/clk_{prec}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resblock" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in cc:
    if k in dur: print(prec, "dispatch", k, "GUI_ACTIVE", cc[k], "ns", dur[k], "=> GHz", cc[k] / dur[k])
PY
done
