# GPU box: one encoder-level test by name with its output kept (iteration helper)
# usage: gpurun -- 'bash scripts/gpu_bcw_iter.sh <pytest -k expression> [more pytest args]'
cd $GRAFT_REPO_ROOT
K=$1; shift
timeout -k 10 1100 python -m pytest tests/test_gpu_encoder_dropin.py -m gpu -x -q -s -k "$K" "$@" > gpurun_out/iter_$K.log 2>&1
rc=$?
tail -c 3000 gpurun_out/iter_$K.log
exit $rc
