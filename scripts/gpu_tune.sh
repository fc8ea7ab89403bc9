# GPU box: quick bench under several tuning settings (no tests).  usage: gpurun -- 'bash scripts/gpu_tune.sh "A=1 B=2" "A=3"'
set -e
cd $GRAFT_REPO_ROOT
i=0
for cfg in "" "$@"; do
  i=$((i+1))
  env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/tune_err_$i.txt | tail -1 > gpurun_out/tune_$i.json || (tail -20 gpurun_out/tune_err_$i.txt; exit 1)
  python -c "
import json,sys; d=json.load(open('gpurun_out/tune_$i.json')); print('[$cfg]', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['stages'].items()})"
done
