# PMC passes for profiles/: one counter group per run, 2 M reads per launch, never combined with tracing
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for g in "$@"; do
  n=$(echo $g | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $g --output-format csv -d $R/gpurun_out/pmc_$n -o p -- python3 $R/bench.py --reads 2000000 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$n.log 2>&1
done
