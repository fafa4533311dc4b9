# kernel-level breakdown of tools/long_bench.py: bash tools/prof_long.sh <tag> <args of long_bench.py...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/long_bench.py "$@" > $O/run.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
tail -3 $O/run.log
python3 - $O/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-100s calls=%s avg_ms=%.3f total_ms=%.1f" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
