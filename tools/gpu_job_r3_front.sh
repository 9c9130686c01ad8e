# usage: bash tools/gpu_job_r3_front.sh  -- the opt-in fused front end (OFDM_FRONT=1) at C2: kernel trace + HBM traffic counters
export OFDM_FRONT=1
bash tools/gpu_job_prof.sh r3f_front 65536 c2 > gpurun_out/prof_r3f_front.log 2>&1; tail -n 6 gpurun_out/prof_r3f_front.log
python tools/pmc_summary.py gpurun_out/prof_r3f_front > gpurun_out/prof_r3f_front/summary.txt 2>&1
python tools/show_bench.py gpurun_out/prof_r3f_front/bench_c2.json
