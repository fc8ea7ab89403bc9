# GPU box: try single counters on the kernels matching a regex (unknown counters just fail their pass).  usage: gpurun -- "bash scripts/gpu_pmc_try.sh <tag> '<regex>' CNT1 CNT2 ..."
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; RE=$2; shift; shift
for C in "$@"; do
  mkdir -p gpurun_out/try_$TAG/$C
  timeout -k 10 200 rocprofv3 --pmc $C SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/try_$TAG/$C -o p -- python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/try_$TAG/$C/stdout.json 2> gpurun_out/try_$TAG/$C/stderr.txt && echo "$C ok" || echo "$C failed"
done
