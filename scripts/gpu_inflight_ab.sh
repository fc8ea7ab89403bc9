cd $GRAFT_REPO_ROOT
for n in 1 2 3; do python bench.py --no-cpu-baseline --steps 30 --inflight $n > gpurun_out/inf.json 2> gpurun_out/inf.err || { tail -3 gpurun_out/inf.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/inf.json')); print('inflight $n', round(d['ms_per_step'],3))"; done
