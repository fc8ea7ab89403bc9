set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_r01_c.json
python -c "
import json; d=json.load(open('gpurun_out/bench_r01_c.json')); print({k:d[k] for k in ('value','ms_per_step','satd_gblocks_per_s','tz_candidates_per_picture')}, d['roofline'], d['cpu_baseline'])"
