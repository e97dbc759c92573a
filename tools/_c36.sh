mkdir -p gpurun_out/r3/c36
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r3/c36/test.txt 2>&1; echo rc=$?; tail -3 gpurun_out/r3/c36/test.txt
timeout -k 10 120 python tools/gemm_bench.py --bwd-epi > gpurun_out/r3/c36/new.txt 2>&1; echo rc=$?
ADN_LIB=$PWD/audio-depth-estimation_amd/libadn_prev.so timeout -k 10 120 python tools/gemm_bench.py --bwd-epi > gpurun_out/r3/c36/old.txt 2>&1; echo rc=$?
timeout -k 10 120 python tools/gemm_bench.py --bwd-epi > gpurun_out/r3/c36/new2.txt 2>&1; echo rc=$?
paste gpurun_out/r3/c36/new.txt gpurun_out/r3/c36/old.txt gpurun_out/r3/c36/new2.txt | grep -v amdgpu | awk -F'\t' '{print substr($1,1,70) " | " substr($2,40,30) " | " substr($3,40,30)}'
timeout -k 10 200 python bench.py --no-f32 --no-cpu-baseline > gpurun_out/r3/c36/bench_new.json 2>gpurun_out/r3/c36/bench_new.err; echo rc=$?
ADN_LIB=$PWD/audio-depth-estimation_amd/libadn_prev.so timeout -k 10 200 python bench.py --no-f32 --no-cpu-baseline > gpurun_out/r3/c36/bench_old.json 2>gpurun_out/r3/c36/bench_old.err; echo rc=$?
python - <<'P'
import json
for n in ("new","old"):
    d=json.loads(open("gpurun_out/r3/c36/bench_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["sustained"]["median_ms_per_step"], d["roofline"]["frac"], d["final_loss"])
P
timeout -k 10 400 python -m pytest tests/test_gpu_unet.py -x -q -m gpu > gpurun_out/r3/c36/test2.txt 2>&1; echo rc=$?; tail -3 gpurun_out/r3/c36/test2.txt
