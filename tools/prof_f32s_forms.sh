cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in f32sw f32s; do
for layer in 5 10; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s2_${m}_$layer -o r -- python3 $R/tools/run_resblock.py 512 $m 6 $layer > /dev/null 2>&1
echo "== $m layer $layer"; python3 - $R/gpurun_out/s2_${m}_$layer <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'f32s' in r['Name'] or 'resblock' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6, 'ms min', float(r['MinNs'])/1e6)
PY
done; done
