#!/bin/bash
# The GPU-box jobs of this repository, one sub-command each (run through gpurun from the repository root):
#   bash tools/gpu_job.sh test <tag> [pytest args...]     GPU tests (default: the whole -m gpu suite), then a short C2 bench
#   bash tools/gpu_job.sh variants <tag> <config> v1 v2.. A/B of library variants (make variant NAME=x), product first and last
#   bash tools/gpu_job.sh modes <tag> [config]            pipelined / sequential / four-round-trip bench of one build
#   bash tools/gpu_job.sh stamps <tag> [config]           product bench, then the stamps build's per-phase profile of k_sync
#   bash tools/gpu_job.sh ablate <tag> "<values>"         stamps+diag build, one run per OFDM_ABLATE value
#   bash tools/gpu_job.sh soak <tag> <seed> [par reuse stream sense]  every randomised soak (seconds each) + the edge inputs
#   bash tools/gpu_job.sh evidence <config> <packets>     rocprofv3 kernel trace + PMC passes (gpu_job_prof.sh) + bench lines
#   bash tools/gpu_job.sh front                           the same evidence for the opt-in fused front end (OFDM_FRONT=1) at C2
#   bash tools/gpu_job.sh cache                           Infinity-Cache microbenchmark + C2 bench against the batch size
CMD=$1; shift
show() { python tools/show_bench.py "$1" 2>/dev/null || tail -c 1500 "${1%.json}.err"; }
bench() { timeout -k 10 300 python bench.py --cpu-packets 0 "$@"; }
case $CMD in
test)
  TAG=${1:-x}; shift; ARGS=${@:-tests -m gpu -x -q}; O=gpurun_out/r3_$TAG; mkdir -p $O
  timeout -k 10 900 python -m pytest $ARGS > $O/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -n 25 $O/pytest.log
  [ $rc -eq 0 ] || exit $rc
  bench --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; echo bench exit=$?; show $O/bench_c2.json ;;
variants)
  TAG=$1; CFG=$2; shift 2; O=gpurun_out/r3_$TAG; mkdir -p $O
  for v in product "$@" product; do
    LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip.so; [ $v != product ] && LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_$v.so
    OFDM_HIP_LIB=$LIB bench --config $CFG --steps 8 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err
    echo "== $v"; show $O/bench_$v.json
  done ;;
modes)
  TAG=${1:-x}; CFG=${2:-c2}; O=gpurun_out/r3_$TAG; mkdir -p $O
  for mode in pipe seq seq4; do
    ARGS=""; [ $mode != pipe ] && ARGS="--no-pipeline"
    [ $mode = seq4 ] && export OFDM_RX_SYNCS=1
    bench --config $CFG --steps 8 --warmup 2 $ARGS > $O/bench_${CFG}_$mode.json 2> $O/bench_${CFG}_$mode.err; echo "$mode exit=$?"
    show $O/bench_${CFG}_$mode.json | head -n 1
    unset OFDM_RX_SYNCS
  done ;;
stamps)
  TAG=${1:-x}; CFG=${2:-c2}; O=gpurun_out/r3_$TAG; mkdir -p $O
  bench --config $CFG --steps 5 --warmup 2 > $O/bench_$CFG.json 2> $O/bench_$CFG.err; echo bench exit=$?; show $O/bench_$CFG.json
  OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_stamps.so bench --config $CFG --steps 1 --warmup 1 --no-pipeline > $O/stamps.json 2> $O/stamps.err
  tail -n 12 $O/stamps.err ;;
ablate)
  TAG=${1:-x}; VALS=${2:-"0 4 8 12"}; O=gpurun_out/r3_$TAG; mkdir -p $O
  for v in $VALS; do
    echo "== OFDM_ABLATE=$v"
    OFDM_ABLATE=$v OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_stamps.so bench --steps 1 --warmup 1 --no-pipeline > $O/ab_$v.json 2> $O/ab_$v.err
    tail -n 11 $O/ab_$v.err | cut -c1-70; show $O/ab_$v.json | tail -n 1
  done ;;
soak)
  TAG=${1:-x}; SEED=${2:-2}; SP=${3:-300}; SR=${4:-250}; SS=${5:-150}; SN=${6:-80}; O=gpurun_out/soak3_$TAG; mkdir -p $O
  run() { [ $2 -gt 0 ] || return 0; timeout -k 10 $(($2 + 240)) python tests/soak/$1.py $2 $SEED > $O/$1.log 2>&1; echo "$1 exit=$?"; tail -n 1 $O/$1.log; grep MISMATCH $O/$1.log | cut -c1-300 | sed -n 1,4p; }
  ( while sleep 60; do echo "[soak] $(date +%T) still running"; done ) & HB=$!   # (gpurun takes 7 silent minutes for a hang)
  run fuzz_parity $SP; run fuzz_reuse $SR; run fuzz_stream $SS; run fuzz_sense $SN
  timeout -k 10 300 python tests/soak/edge_inputs.py > $O/edge_inputs.log 2>&1; echo "edge exit=$?"; tail -n 2 $O/edge_inputs.log
  kill $HB
  true ;;
evidence)
  CFG=${1:-c2}; PK=${2:-65536}; O=gpurun_out/prof_r3f_$CFG
  bash tools/gpu_job_prof.sh r3f_$CFG $PK $CFG > $O.log 2>&1; tail -n 6 $O.log
  python tools/pmc_summary.py $O > $O/summary.txt 2>&1
  bench --config $CFG --no-pipeline > $O/bench_seq.json 2> /dev/null; echo seq exit=$?
  show $O/bench_$CFG.json ;;
front)
  export OFDM_FRONT=1; O=gpurun_out/prof_r3f_front
  bash tools/gpu_job_prof.sh r3f_front 65536 c2 > $O.log 2>&1; tail -n 6 $O.log
  python tools/pmc_summary.py $O > $O/summary.txt 2>&1
  show $O/bench_c2.json ;;
cache)
  O=gpurun_out/r3_mall; mkdir -p $O
  timeout -k 10 300 python tools/ubench/mallbw.py > $O/mallbw.txt 2>&1; cat $O/mallbw.txt
  for p in 256 512 1024 2048 4096 16384 65536; do
    bench --packets $p --steps 20 --warmup 3 --no-pipeline > $O/bench_$p.json 2> $O/bench_$p.err || exit 1
    echo "== $p packets"; show $O/bench_$p.json
  done ;;
*) sed -n 2,12p $0; exit 2 ;;
esac
