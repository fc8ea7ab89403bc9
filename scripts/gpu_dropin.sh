# GPU box: distortion parity + the encoder-level drop-in test, verbose.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_encoder_dropin.py -m gpu -x -q -s > gpurun_out/dropin.txt 2>&1 || (tail -40 gpurun_out/dropin.txt; exit 1)
tail -5 gpurun_out/dropin.txt
