# GPU box: the bench lines of the operating points DESIGN.md section 5 lists (each with the reference CPU chain beside it).  usage: gpurun -- "bash scripts/gpu_bench_points.sh <tag>"
cd $GRAFT_REPO_ROOT
T=$1
run() { name=$1; shift; python bench.py "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err || { tail -5 gpurun_out/${T}_$name.err; exit 1; }; python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1]); c=d.get("cpu_baseline") or {}
print("$name", round(d["value"],2), "pictures/s", round(d["ms_per_step"],3), "ms | cpu s/picture", c.get("seconds_per_picture"), "mismatches", c.get("mismatches_vs_gpu"))
PY
}
run ra_4k_default
run ra_4k_dpoc8_16 --no-encoder-level --dpoc 8,16
run ra_4k_lite --no-encoder-level --lite
run ra_4k_qp27 --no-encoder-level --qp 27
run ra_1080p --no-encoder-level --width 1920 --height 1080
run ldp_1080p --no-encoder-level --width 1920 --height 1080 --config ldp
run ra_8k_qp22 --no-encoder-level --width 7680 --height 4320 --qp 22 --steps 10
run ra_4k_btt --no-encoder-level --partition btt
VTM_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/${T}_force_dist.json 2> gpurun_out/${T}_force_dist.err && python -c "
import json; d=json.loads(open('gpurun_out/${T}_force_dist.json').read().strip().splitlines()[-1]); print('force_dist', round(d['value'],2), round(d['ms_per_step'],3))"
for n in 2 4 8; do VTM_BENCH_SIMULATE_WORLD=$n python bench.py --no-cpu-baseline > gpurun_out/${T}_sim_world$n.json 2> gpurun_out/${T}_sim$n.err && python -c "
import json; d=json.loads(open('gpurun_out/${T}_sim_world$n.json').read().strip().splitlines()[-1]); print('simulated rank share of world $n', round(d['ms_per_step'],3), 'ms')"; done
