export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5prof -- python $R/bench.py --legs c5_sampling --steps 10 --warmup 2 --no-cpu-baseline --no-parity --launch eager > /dev/null 2>&1
find $R/gpurun_out/c5prof -name "*_kernel_trace.csv" -delete
head -12 $R/gpurun_out/c5prof/*/*kernel_stats.csv | cut -c1-150
