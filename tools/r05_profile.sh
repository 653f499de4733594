#!/bin/bash
# GPU box, one call: the round's measurement set (r05). The driver's own bench command, kernel stats + HBM counters per leg,
# counters of every training kernel >= 2 % of the step, of the fused 16-bit encoder, the fp32 encoder and the C5 attention.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
cp gpurun_out/bench_detail.json gpurun_out/r05_bench_detail.json
bash tools/profile_legs.sh r05 "c2_greedy c2_greedy_fp32 c3_greedy c5_sampling c2_sampling c4_train" pmc > gpurun_out/r05_profile_legs.log 2>&1
bash tools/train_pmc.sh r05pmc > gpurun_out/r05_train_pmc.log 2>&1
bash tools/enc_pmc.sh r05enc enc_bench.py am_encoder > gpurun_out/r05_enc_pmc.log 2>&1
bash tools/enc_pmc.sh r05encf32 enc_f32_bench.py am_encoder_f32 > gpurun_out/r05_enc_f32_pmc.log 2>&1
bash tools/enc_pmc.sh r05flash c5_encoder_bench.py attn_flash > gpurun_out/r05_flash_pmc.log 2>&1
python tools/enc_power_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_power_probe.txt
tail -3 gpurun_out/r05_bench.err; wc -c gpurun_out/r05_bench_line.json; tail -4 gpurun_out/r05_power_probe.txt
