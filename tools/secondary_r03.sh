# Secondary workloads of DESIGN section 5 (one box, B = 32): prints one JSON line per workload.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3/final
OUT=gpurun_out/r3/final/secondary_workloads.txt
: > $OUT
for m in rgb binaural adabins baseres; do
  python tools/bench_model.py --model $m 2>/dev/null | grep -E "^adn_igemm|^adn_wgrad|^adn_attn|^\{" | sed "s/^/$m  /" >> $OUT
done
python tools/bench_model.py --model rgb --size 512 2>/dev/null | grep -E "^adn_igemm|^adn_wgrad|^\{" | sed "s/^/rgb512-bf16  /" >> $OUT
python tools/bench_model.py --model rgb --size 512 --dtype mxfp8 2>/dev/null | grep -E "^adn_igemm|^adn_wgrad|^adn_conv3x3_mx8|^\{" | sed "s/^/rgb512-mxfp8  /" >> $OUT
cat $OUT
