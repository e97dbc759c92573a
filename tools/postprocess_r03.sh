# Turns gpurun_out/r3/final (written on the GPU box by tools/final_measure_r03.sh + tools/secondary_r03.sh) into the
# committed summaries profiles/r03_*.  Run in the build container after the gpurun call.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/r3/final
python tools/rocpd_export.py stats $F/prof_bench/bench_results.db > profiles/r03_bench_n1_kernel_stats.csv
python tools/rocpd_export.py traffic $F/pmc_fetch/f_results.db $F/pmc_write/w_results.db > /tmp/adn_traffic.json
python - <<'P'
import json
d = json.load(open('/tmp/adn_traffic.json'))
d = {k: v for k, v in d.items() if not (k.startswith('__amd') or k.startswith('at::'))}
import subprocess
d['_commit'] = open('gpurun_out/r3/final/commit.txt').read().strip()
d['_command'] = 'rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-f32 --no-cpu-baseline --sustain-seconds 0'
json.dump(d, open('profiles/r03_pmc_traffic_gemm.json', 'w'), indent=1)
fam = [v for k, v in d.items() if k.startswith(('igemm_ring_kernel', 'igemm_patch_kernel', 'igemm_mfma_kernel'))]
print('igemm family: mean HBM bytes per launch %.1f MB' % (sum(r['launches'] * r['hbm_bytes_per_launch'] for r in fam) / sum(r['launches'] for r in fam) / 1e6))
g = json.load(open('gpurun_out/r3/final/bench_n1.json'))
p = json.loads(open('gpurun_out/r3/final/bench_plan.json').read().strip().splitlines()[-1])
open('profiles/r03_bench_plan_vs_graph.txt', 'w').write(
    "# bench.py (hipGraph replay) vs bench.py --no-graph (launch-plan replay), same box, N = 1\n"
    f"hipGraph      {g['ms_per_step']:.4f} ms/step {g['value']:.1f} maps/s; sustained median {g['sustained']['median_ms_per_step']:.4f}\n"
    f"launch plan   {p['ms_per_step']:.4f} ms/step {p['value']:.1f} maps/s; sustained median {p['sustained']['median_ms_per_step']:.4f}\n")
P
PAT='^(igemm_ring_kernel|igemm_mfma_kernel<unsigned short, 128, 128, 2, 0, true>|igemm_patch_kernel|wgrad_k4_patch_kernel)'
{
  echo "# rocprofv3 --pmc (two passes) over tools/gemm_bench.py --iters 3 (unet_256 layer shapes, B = 32, bf16), mean per launch"
  echo "# pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  python tools/rocpd_export.py counters $F/pmc_mfma/m_results.db | grep -A4 -E "$PAT" | grep -v "^--"
  echo "# pass 2: SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS"
  python tools/rocpd_export.py counters $F/pmc_lds/l_results.db | grep -A4 -E "$PAT" | grep -v "^--"
} > profiles/r03_pmc_gemm_counters.txt
cp $F/bench_n1.json profiles/r03_bench_n1.json
for f in gemm_microbench s1_gemm_microbench mx8_microbench; do grep -v amdgpu.ids $F/$f.txt > profiles/r03_$f.txt; done
if [ -f $F/secondary_workloads.txt ]; then
  { echo "# tools/secondary_r03.sh: tools/bench_model.py per workload (B = 32, one MI355X): GEMM families (ms per step, launches, TFLOP/s) + the hipGraph step"; cat $F/secondary_workloads.txt; } > profiles/r03_secondary_workloads.txt
fi
ls -la profiles/r03_*
