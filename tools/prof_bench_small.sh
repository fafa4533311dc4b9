# small traced bench run (1 M reads, 2 steps): kernel stats of the whole step
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/small
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/small/prof -- python3 bench.py --steps 2 --warmup 1 --reads 1000000 --no-cpu-baseline > gpurun_out/small/bench.log 2> gpurun_out/small/bench.err
find gpurun_out/small/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/small/kernel_stats.csv
rm -rf gpurun_out/small/prof
head -20 gpurun_out/small/kernel_stats.csv
