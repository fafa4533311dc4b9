# Round profile (rounds 3 and 4): the bench line, rocprofv3 kernel stats of the same command, PMC passes (one counter group per run,
# never combined with tracing) at the bench's own scale (10 M reads: the region order of the gapped stage depends on it).
# usage: bash tools/round4_profile.sh [tag]   -> gpurun_out/<tag>_profile/
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
: > $O/pmc.txt
i=0
for g in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $g --output-format csv -d $O/p$i -o p -- python3 $R/tools/quick_bench.py 10000000 2 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $O/pmc.txt <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:70]
    if not any(x in k for x in ('gapped','seed_extend','sort_consensus','k_reg_','k_seg_','k_dust')): continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in sorted(agg):
    print(k, {c:{"mean_per_dispatch":v/cnt[(k,c)],"dispatches":cnt[(k,c)]} for c,v in agg[k].items()})
PY
  rm -rf $O/p$i
done
head -30 $O/kernel_stats.csv
cat $O/pmc.txt
tail -c 3000 $O/bench.json
