# GPU box: kernel trace only (quick look at per-launch-shape times).  usage: gpurun -- "bash scripts/gpu_trace.sh <tag> [bench args]"; then
# python3 scripts/summarize_profiles.py <tag> here.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
shift
EXTRA="$@"
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 bench.py --serial --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > gpurun_out/prof_$TAG/bench_stdout.json 2> gpurun_out/prof_$TAG/bench_stderr.txt
echo traced
