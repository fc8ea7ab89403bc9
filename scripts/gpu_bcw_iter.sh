# GPU box: encoder-level tests by name with their output kept (iteration helper)
# usage: gpurun -- 'bash scripts/gpu_bcw_iter.sh <tag> <pytest -k expression> [more pytest args]'
cd $GRAFT_REPO_ROOT
TAG=$1; K=$2; shift; shift
timeout -k 10 1150 python -m pytest tests/test_gpu_encoder_dropin.py -m gpu -q -s -k "$K" "$@" > gpurun_out/iter_$TAG.log 2>&1
rc=$?
tail -c 2500 gpurun_out/iter_$TAG.log
exit $rc
