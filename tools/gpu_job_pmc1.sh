# usage: bash tools/gpu_job_pmc1.sh <tag> [packets] [config] -- one SQ counter pass (VALU/LDS/wait) of bench.py --no-pipeline
TAG=${1:-x}; PK=${2:-65536}; CFG=${3:-c2}
mkdir -p gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--config $CFG --packets $PK --warmup 1 --cpu-packets 0 --no-pipeline"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/prof_$TAG/pmc1 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc1.log 2>&1; echo pmc1 exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_$TAG/pmc2 -- python3 $R/bench.py $ARGS --steps 2 > $R/gpurun_out/prof_$TAG/pmc2.log 2>&1; echo pmc2 exit=$?
