# Round profile: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes (never combined with tracing).
# usage: bash tools/round_profile.sh r02   -> gpurun_out/<tag>_profile/
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
for g in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  n=$(echo $g | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -o p -- python3 $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$n.log 2>&1
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" > $O/pmc_$n.txt <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:70]
    if not any(x in k for x in ('gapped','seed_extend','sort_consensus')): continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in sorted(agg):
    print(k, {c:{"sum":v,"dispatches":cnt[(k,c)]} for c,v in agg[k].items()})
PY
  rm -rf $O/pmc_$n
done
head -15 $O/kernel_stats.csv
cat $O/pmc_*.txt
tail -c 2500 $O/bench.json
