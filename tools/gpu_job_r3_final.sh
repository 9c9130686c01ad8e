# usage: bash tools/gpu_job_r3_final.sh <config> <packets>  -- the round's evidence for one configuration:
# rocprofv3 kernel trace + the four PMC passes of `bench.py --no-pipeline`, then the default (pipelined) bench line
CFG=${1:-c2}; PK=${2:-65536}
bash tools/gpu_job_prof.sh r3f_$CFG $PK $CFG > gpurun_out/prof_r3f_$CFG.log 2>&1; tail -n 6 gpurun_out/prof_r3f_$CFG.log
python tools/pmc_summary.py gpurun_out/prof_r3f_$CFG > gpurun_out/prof_r3f_$CFG/summary.txt 2>&1
timeout -k 10 300 python bench.py --config $CFG --no-pipeline --cpu-packets 0 > gpurun_out/prof_r3f_$CFG/bench_seq.json 2> /dev/null; echo seq exit=$?
python tools/show_bench.py gpurun_out/prof_r3f_$CFG/bench_$CFG.json
