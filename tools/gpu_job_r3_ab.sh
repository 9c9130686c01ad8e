# usage: bash tools/gpu_job_r3_ab.sh <tag> <variant> [pytest file]  -- parity tests, then A/B product vs a library variant at C2 / C3 / C5
TAG=$1; VAR=$2; PT=${3:-tests/test_gpu_parity.py}
mkdir -p gpurun_out/r3_$TAG
timeout -k 10 600 python -m pytest $PT -x -q > gpurun_out/r3_$TAG/pytest.log 2>&1; rc=$?; tail -n 3 gpurun_out/r3_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_job_r3_variants.sh ${TAG}_c2 c2 $VAR && bash tools/gpu_job_r3_variants.sh ${TAG}_c3 c3 $VAR && bash tools/gpu_job_r3_variants.sh ${TAG}_c5 c5 $VAR
