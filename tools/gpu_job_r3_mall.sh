# usage: bash tools/gpu_job_r3_mall.sh  -- cache-residency microbenchmark + C2 bench against the batch size
mkdir -p gpurun_out/r3_mall
timeout -k 10 300 python tools/ubench/mallbw.py > gpurun_out/r3_mall/mallbw.txt 2>&1; cat gpurun_out/r3_mall/mallbw.txt
for p in 256 512 1024 2048 4096 16384 65536; do
  timeout -k 10 300 python bench.py --packets $p --steps 20 --warmup 3 --cpu-packets 0 --no-pipeline > gpurun_out/r3_mall/bench_$p.json 2> gpurun_out/r3_mall/bench_$p.err || exit 1
  echo "== $p packets"; python tools/show_bench.py gpurun_out/r3_mall/bench_$p.json
done
