cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02d
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r02d/prof -- python3 bench.py --steps 2 --warmup 1 --reads 1000000 --no-cpu-baseline > gpurun_out/r02d/bench.log 2> gpurun_out/r02d/bench.err
find gpurun_out/r02d/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02d/kernel_stats.csv
rm -rf gpurun_out/r02d/prof
head -12 gpurun_out/r02d/kernel_stats.csv
tail -c 600 gpurun_out/r02d/bench.log
