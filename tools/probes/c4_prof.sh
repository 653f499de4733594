python -m pytest tests/test_gpu_teacher.py -x -q -k "training_step_on_kernels and 128 and bf16" 2>&1 | grep -E "assert|Error|cos" | head -10
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4prof -- python $GRAFT_REPO_ROOT/bench.py --legs c4_train --steps 5 --warmup 2 --no-cpu-baseline --no-parity > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/c4prof -name "*_kernel_trace.csv" -delete
head -14 $GRAFT_REPO_ROOT/gpurun_out/c4prof/*/*kernel_stats.csv | cut -c1-130
