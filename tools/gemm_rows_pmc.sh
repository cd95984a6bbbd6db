set -u
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rows_pmc; rm -rf $OUT; mkdir -p $OUT
for MODE in rows tiled; do
  if [ $MODE = tiled ]; then export PDP_TRAIN_GEMM=tiled; else unset PDP_TRAIN_GEMM; fi
  for SH in "128 384" "129 100"; do
    TAG=${MODE}_$(echo $SH | tr ' ' '_')
    rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -d $OUT/$TAG/p1 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_rows_pmc.py $SH > $OUT/$TAG.p1.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $OUT/$TAG/p2 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_rows_pmc.py $SH > $OUT/$TAG.p2.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $OUT/$TAG/p3 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_rows_pmc.py $SH > $OUT/$TAG.p3.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/$TAG/p4 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_rows_pmc.py $SH > $OUT/$TAG.p4.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/rows_pmc'
for tag in sorted(os.listdir(root)):
    if not os.path.isdir(root+'/'+tag): continue
    agg=collections.defaultdict(float); n=0
    for f in glob.glob(root+'/'+tag+'/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm' in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(tag, {k: '%.3e'%v for k,v in sorted(agg.items())})
PY
