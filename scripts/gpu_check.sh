# GPU box: parity tests + a short bench line.  usage: gpurun -- 'bash scripts/gpu_check.sh'
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/pl.txt 2>&1 || (grep -n "^E " gpurun_out/pl.txt | head -20; tail -5 gpurun_out/pl.txt; exit 1)
tail -2 gpurun_out/pl.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_err.txt | tail -1 > gpurun_out/bench_tmp.json || (tail -30 gpurun_out/bench_err.txt; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/bench_tmp.json')); print({k:d[k] for k in ('value','ms_per_step','satd_gblocks_per_s')}); print(d['stages']); print(d['roofline'])"
