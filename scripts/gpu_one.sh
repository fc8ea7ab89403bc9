# GPU box: run the given test files (default: all GPU tests) into gpurun_out/one.txt, under a hard timeout.
# usage: gpurun -- 'bash scripts/gpu_one.sh tests/test_gpu_mest.py'
set -e
cd $GRAFT_REPO_ROOT
T=${@:-tests}
timeout -k 10 ${GPU_ONE_TIMEOUT:-600} python -m pytest $T -m gpu -x -q > gpurun_out/one.txt 2>&1 || (grep -n "^E " gpurun_out/one.txt | head -30; tail -5 gpurun_out/one.txt; exit 1)
tail -3 gpurun_out/one.txt
