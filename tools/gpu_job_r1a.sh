mkdir -p gpurun_out/prof
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo exit=$? >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 500 python bench.py --cpu-packets 2048 > gpurun_out/bench_full.log 2>&1; echo exit=$? >> gpurun_out/bench_full.log; tail -2 gpurun_out/bench_full.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -- python3 $GRAFT_REPO_ROOT/bench.py --packets 16384 --steps 3 --warmup 1 --cpu-packets 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace.log 2>&1; echo trace exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --packets 16384 --steps 2 --warmup 1 --cpu-packets 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc1.log 2>&1; echo pmc1 exit=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --packets 16384 --steps 2 --warmup 1 --cpu-packets 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc2.log 2>&1; echo pmc2 exit=$?
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*.csv" | head -20; du -sh gpurun_out/prof
