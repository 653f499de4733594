timeout 600 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_teacher_mma.py -x -q 2>&1 | tail -2
timeout 200 python bench.py --legs c4_train --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3k_train.json 2>/dev/null
python - <<P
import json
d=json.loads(open("gpurun_out/r3k_train.json").read().strip().splitlines()[-1])
print("train", d.get("ms_per_step"), d["roofline"]["launch_ms_mean"], d["legs"] if False else "")
P
timeout 400 python tools/teacher_clock_probe.py run > gpurun_out/r3l_teacher_clk.json 2> gpurun_out/r3l_teacher_clk.err; echo rc=$?
