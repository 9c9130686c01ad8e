# needs the diagnostic library: make -C ofdm_uhd_amd/csrc diag (built before the gpurun call)
export OFDM_HIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/ofdm_uhd_amd/csrc/libofdm_hip_diag.so
# usage: bash tools/ablate_pmc.sh  -- VALU/SALU/LDS instruction counts of k_sync under the OFDM_ABLATE phases
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ablate
for AB in 0 1 2 3; do
  export OFDM_ABLATE=$AB
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/ablate/ab$AB -- python3 $R/bench.py --packets 16384 --steps 2 --warmup 1 --cpu-packets 0 > $R/gpurun_out/ablate/ab$AB.log 2>&1 || echo "ab$AB failed"
done
python3 - <<'P'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for ab in range(4):
    fs=glob.glob(R+"/gpurun_out/ablate/ab%d/*/*_counter_collection.csv"%ab)
    if not fs: continue
    agg=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(fs[0])):
        if "k_sync" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"]=="SQ_WAVES": n+=1
    print("ablate",ab,{k:"%.4g"%(v/max(n,1)) for k,v in agg.items()}, "launches",n)
P
