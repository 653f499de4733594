#!/bin/bash
# GPU check of the r04 normalisation work: layer norm (fused + training kernels), instance / layer norm on token tiles,
# the near-tie proof of the full-size fp32 flips; then a short bench of the two encoder-heavy legs (regression check).
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_norms_mlp_grad.py -q --maxfail=25 --tb=short -p no:cacheprovider > gpurun_out/norm_tests.log 2>&1
echo "layer_norm tests exit $?" | tee -a gpurun_out/norm_tests.log
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_decode.py -q --tb=short -k "leaves_the_packed or full_size_tsp100_b4096_vs or full_size_cvrp100_b4096_vs" > gpurun_out/norm_tests2.log 2>&1
echo "decode/graph tests exit $?" | tee -a gpurun_out/norm_tests2.log
timeout 600 python bench.py --legs c2_greedy,c5_sampling --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/norm_bench.log 2>&1
tail -c 1500 gpurun_out/norm_tests.log; tail -c 800 gpurun_out/norm_tests2.log; tail -c 600 gpurun_out/norm_bench.log
