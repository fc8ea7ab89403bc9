set -e
cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 2 2>gpurun_out/bench_err.txt | tail -1 > gpurun_out/bench_r01_d.json || (tail -30 gpurun_out/bench_err.txt; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/bench_r01_d.json')); print({k:d[k] for k in ('value','ms_per_step','satd_gblocks_per_s')}); print(d['stages']); print(d['roofline']); print(d['cpu_baseline'])"
