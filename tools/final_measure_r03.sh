set -x
mkdir -p gpurun_out/r3/final
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r3/final/bench_n1.json 2> gpurun_out/r3/final/bench_n1.err || exit 1
python bench.py --no-graph --no-f32 --no-cpu-baseline > gpurun_out/r3/final/bench_plan.json 2>/dev/null || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/final/prof_bench -o bench -- python3 bench.py --steps 50 --warmup 10 --no-f32 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r3/final/prof_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r3/final/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-f32 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r3/final/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r3/final/pmc_write -o w -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-f32 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r3/final/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/r3/final/pmc_mfma -o m -- python3 tools/gemm_bench.py --iters 3 > gpurun_out/r3/final/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/r3/final/pmc_lds -o l -- python3 tools/gemm_bench.py --iters 3 > gpurun_out/r3/final/pmc_lds.log 2>&1 || exit 1
python tools/gemm_bench.py > gpurun_out/r3/final/gemm_microbench.txt 2>&1
python tools/gemm_bench.py --s1 > gpurun_out/r3/final/s1_gemm_microbench.txt 2>&1
python tools/gemm_bench.py --mx8 > gpurun_out/r3/final/mx8_microbench.txt 2>&1
ls gpurun_out/r3/final
python tools/gemm_bench.py --deep > gpurun_out/r3/final/deep_microbench.txt 2>&1
python tools/gemm_bench.py --bwd-epi > gpurun_out/r3/final/gemm_bwd_hot.txt 2>&1
python tools/gemm_bench.py --bwd-epi --cold > gpurun_out/r3/final/gemm_bwd_cold.txt 2>&1
python tools/step_timeline.py > gpurun_out/r3/final/step_timeline.txt 2>&1
ls gpurun_out/r3/final
