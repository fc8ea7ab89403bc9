# GPU box: instruction-fetch / occupancy counters of the kernels matching a regex.  usage as gpu_pmc_probe.sh; output gpurun_out/probe2_<tag>/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
RE=$2
shift; shift
EXTRA="$@"
i=0
for CNTS in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_LEVEL_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  mkdir -p gpurun_out/probe2_$TAG/pass$i
  rocprofv3 --pmc $CNTS --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/probe2_$TAG/pass$i -o p -- python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/probe2_$TAG/pass$i/stdout.json 2> gpurun_out/probe2_$TAG/pass$i/stderr.txt || echo "pass $i failed"
done
echo probed
