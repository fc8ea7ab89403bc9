# GPU box: kernel trace of the OVERLAPPED default run (for scripts/timeline_slices.py).  usage: gpurun -- "bash scripts/gpu_trace_overlap.sh <tag>"
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ovl_$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl_$1 -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ovl_$1/stdout.json 2> gpurun_out/ovl_$1/stderr.txt
echo traced
