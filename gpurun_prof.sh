set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -3
mkdir -p gpurun_out/prof_r01
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r01/bench_stdout.json 2> gpurun_out/prof_r01/bench_stderr.txt || (tail -20 gpurun_out/prof_r01/bench_stderr.txt; exit 1)
ls -R gpurun_out/prof_r01 | head -30
