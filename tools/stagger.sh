for a in 0 8000 16000 24000 32000; do
  OFDM_STAGGER=$a timeout -k 10 200 python bench.py --packets 16384 --steps 3 --warmup 1 --cpu-packets 0 > gpurun_out/sg_$a.log 2>&1
  echo -n "stagger=$a  "; python tools/show_bench.py gpurun_out/sg_$a.log | tail -1
done
