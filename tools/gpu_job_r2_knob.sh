# usage: bash tools/gpu_job_r2_knob.sh <tag> <config> <ENVVAR> <value>... -- bench of one config with an env knob at each value (twice)
TAG=${1:-x}; CFG=$2; VAR=$3; shift 3
mkdir -p gpurun_out/r2_$TAG
for rep in 1 2; do
for V in "$@"; do
env $VAR=$V timeout -k 10 300 python bench.py --config $CFG --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_$V.json 2> gpurun_out/r2_$TAG/bench_$V.err; echo "$VAR=$V bench exit=$?"
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$V.json 2>/dev/null || tail -c 800 gpurun_out/r2_$TAG/bench_$V.err
done
done
