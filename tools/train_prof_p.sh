cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_prof_p; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/rows -o t -- python3 $GRAFT_REPO_ROOT/tools/train_time.py p-nd-np > $OUT/rows.log 2>&1
python3 - <<'PY'
import csv, glob, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/train_prof_p'
f=glob.glob(root+'/rows/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total %.1f ms'%(tot/1e6))
for r in rows[:26]:
    print('  %-100s %5d %8.2f ms %5.1f%% avg %7.1f us'%(r['Name'][:100],int(r['Calls']),float(r['TotalDurationNs'])/1e6,100*float(r['TotalDurationNs'])/tot,float(r['AverageNs'])/1e3))
PY
