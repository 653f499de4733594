#!/bin/bash
# End-of-round check on one MI355X: the whole GPU suite, smoke(), then the driver-shaped bench run.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/final_tests.log 2>&1
echo "gpu suite exit $?" | tee -a gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err
tail -4 gpurun_out/final_tests.log; tail -2 gpurun_out/final_smoke.log; tail -c 2600 gpurun_out/final_bench.log
