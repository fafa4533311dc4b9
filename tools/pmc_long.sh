# PMC passes over tools/long_bench.py (one counter group per run, never combined with tracing); per-kernel means per dispatch
# usage: bash tools/pmc_long.sh <tag> "<long_bench args>" [kernel-name-filter]
TAG=${1:-q}
LB_ARGS=${2:-"600 300000"}
F=${3:-gapped}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
i=0
: > $O/summary.txt
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $g --output-format csv -d $O/p$i -o p -- python3 $R/tools/long_bench.py $LB_ARGS > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$F" >> $O/summary.txt <<'PY'
import csv,sys,collections
f=sys.argv[1]; flt=sys.argv[2].split(',')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:60]
    if not any(x in k for x in flt): continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    print(k, {c:("%.4g" % (v/cnt[(k,c)]), cnt[(k,c)]) for c,v in agg[k].items()})
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
