# usage: bash tools/gpu_job_r3_ablate.sh <tag> "<ablate values>"  -- stamps+diag build, one run per OFDM_ABLATE value
TAG=${1:-x}; VALS=${2:-"0 4 8 12"}
mkdir -p gpurun_out/r3_$TAG
for v in $VALS; do
  echo "== OFDM_ABLATE=$v"
  OFDM_ABLATE=$v OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_stamps.so timeout -k 10 300 python bench.py --steps 1 --warmup 1 --cpu-packets 0 --no-pipeline > gpurun_out/r3_$TAG/ab_$v.json 2> gpurun_out/r3_$TAG/ab_$v.err; tail -n 11 gpurun_out/r3_$TAG/ab_$v.err | cut -c1-70
  python tools/show_bench.py gpurun_out/r3_$TAG/ab_$v.json | tail -n 1
done
