set -e
cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || (tail -20 gpurun_out/bench1.err; exit 1)
cat gpurun_out/bench1.json
