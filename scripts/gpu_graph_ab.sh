cd $GRAFT_REPO_ROOT
for g in off on; do for n in 1 8; do
VTM_BENCH_SIMULATE_WORLD=$n timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --graph $g > gpurun_out/graph.json 2> gpurun_out/graph.err || { tail -5 gpurun_out/graph.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/graph.json')); print('graph $g world $n', round(d['ms_per_step'],3))"
done; done
