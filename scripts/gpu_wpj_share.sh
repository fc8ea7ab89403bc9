# GPU box: waves per search for a rank's share of the picture (few searches per launch).  usage: gpurun -- 'bash scripts/gpu_wpj_share.sh'
cd $GRAFT_REPO_ROOT
for n in 8 4 2; do for v in "" "128:16" "128:16,64:4" "128:16,64:4,32:2"; do
VTM_BENCH_SIMULATE_WORLD=$n VTM_AMD_TZ_WPJ="$v" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/ws.json 2> gpurun_out/ws.err || { tail -3 gpurun_out/ws.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/ws.json')); print('world $n [$v]', round(d['ms_per_step'],3))"
done; done
