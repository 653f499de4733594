#!/bin/bash
# GPU box, one call: the round's measurement set. Kernel stats + HBM counters per leg, counters of the training kernels
# and of the fp32 encoder, and the driver's own bench command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err
cp gpurun_out/bench_detail.json gpurun_out/r04_bench_detail.json
bash tools/profile_legs.sh r04 "c2_greedy c2_greedy_fp32 c3_greedy c5_sampling c2_sampling" pmc > gpurun_out/r04_profile_legs.log 2>&1
bash tools/train_pmc.sh r04pmc > gpurun_out/r04_train_pmc.log 2>&1
bash tools/enc_pmc.sh r04encf32 enc_f32_bench.py am_encoder_f32 > gpurun_out/r04_enc_f32_pmc.log 2>&1
tail -3 gpurun_out/r04_bench.err; wc -c gpurun_out/r04_bench_line.json; tail -5 gpurun_out/r04_enc_f32_pmc.log
