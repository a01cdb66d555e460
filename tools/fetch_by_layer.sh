# L2-miss read bytes (FETCH_SIZE x 2, the gfx950 calibration) of one bf16 residual-block launch at B = 512, per layer of a dilation
# cycle: bash tools/fetch_by_layer.sh   (on the GPU box; every profiler pass bounded by timeout)
cd /tmp; export TMPDIR=/tmp
for lay in 0 1 2 3 4 5 6 7 8 9 10 11; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ft_$lay
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ft_$lay -o r -- python3 $GRAFT_REPO_ROOT/tools/run_resblock.py 512 bf16 2 $lay > /dev/null 2>&1
done
python3 - <<PY
import glob,csv
for lay in range(12):
    v=[]
    for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ft_%d/**/*counter_collection.csv"%lay, recursive=True):
        for r in csv.DictReader(open(f)):
            if "resblock" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
    rd=sum(v)/len(v)*2048
    print("layer",lay,"read GB %.2f  traffic/algorithmic %.3f"%(rd/1e9,(rd+16.78e9)/33.554e9))
PY
