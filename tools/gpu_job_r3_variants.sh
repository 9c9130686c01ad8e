# usage: bash tools/gpu_job_r3_variants.sh <tag> <config> name1 name2 ...   -- A/B of library variants (make variant NAME=x), product first and last
TAG=$1; CFG=$2; shift 2
mkdir -p gpurun_out/r3_$TAG
for v in product "$@" product; do
  LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip.so; [ $v != product ] && LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_$v.so
  OFDM_HIP_LIB=$LIB timeout -k 10 300 python bench.py --config $CFG --steps 8 --warmup 2 --cpu-packets 0 > gpurun_out/r3_$TAG/bench_$v.json 2> gpurun_out/r3_$TAG/bench_$v.err
  echo "== $v"; python tools/show_bench.py gpurun_out/r3_$TAG/bench_$v.json 2>/dev/null || tail -n 3 gpurun_out/r3_$TAG/bench_$v.err
done
