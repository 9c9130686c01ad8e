# usage: bash tools/gpu_job_prof.sh <tag> [packets] [config]
# rocprofv3 passes of `bench.py --no-pipeline` (steps strictly in sequence: clean per-kernel durations and counters):
# kernel trace + stats, two SQ counter passes, FETCH_SIZE, WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes)
TAG=${1:-x}; PK=${2:-16384}; CFG=${3:-c2}
mkdir -p gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--config $CFG --packets $PK --warmup 1 --cpu-packets 0 --no-pipeline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python3 $R/bench.py $ARGS --steps 3 > $R/gpurun_out/prof_$TAG/trace.log 2>&1; echo trace exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/prof_$TAG/pmc1 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc1.log 2>&1; echo pmc1 exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_$TAG/pmc2 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc2.log 2>&1; echo pmc2 exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc3 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc3.log 2>&1; echo pmc3 exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc4 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc4.log 2>&1; echo pmc4 exit=$?
cd $R && timeout -k 10 300 python3 bench.py --config $CFG --packets $PK --cpu-packets 0 > gpurun_out/prof_$TAG/bench_$CFG.json 2> gpurun_out/prof_$TAG/bench_$CFG.err; echo bench exit=$?
