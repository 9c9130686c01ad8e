# TX ablation: k_tx_mod time of library variants built with -DTX_ABLATE=<bits> (1: no transform, 2: no mapper, 4: no channel)
mkdir -p gpurun_out/r3_txab
for v in product txa1 txa2 txa4 txa7 product; do
  LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip.so; [ $v != product ] && LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_$v.so
  echo "== $v"; OFDM_HIP_LIB=$LIB timeout -k 10 200 python tools/tx_chan_cost.py 2>&1 | grep channel
done > gpurun_out/r3_txab/out.txt 2>&1
cat gpurun_out/r3_txab/out.txt
