# GPU box: kernel trace of the one-call-per-CU replay (tests/test_gpu_pis_golden.py, ra records): which launches one CU-level call makes, how long they run, how long the gaps are
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_pis_cu
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 -m pytest tests/test_gpu_pis_golden.py -m gpu -q -x -k "one_call and ra" > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 scripts/trace_pis_cu.py $OUT > gpurun_out/trace_pis_cu_summary.txt 2>&1
tail -40 gpurun_out/trace_pis_cu_summary.txt
