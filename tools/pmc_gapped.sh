# PMC passes for the gapped stage (one counter group per run, 1 M reads, never combined with tracing); prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gapped
mkdir -p $O
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $g --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --reads 1000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:60]
    if 'gapped' not in k and 'seed_extend' not in k and 'sort_consensus' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    print(k, {c:(v/cnt[(k,c)], cnt[(k,c)]) for c,v in agg[k].items()})
PY
  rm -rf $O/p$i
done
