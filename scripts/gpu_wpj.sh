# GPU box: the waves-per-search table of the integer search (VTM_AMD_TZ_WPJ overrides of pipeline.WAVES_PER_JOB), default bench without the CPU chain.
# usage: gpurun -- 'bash scripts/gpu_wpj.sh "128:16" "128:4" "64:4" ...'
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  VTM_AMD_TZ_WPJ="$v" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/wpj.json 2> gpurun_out/wpj.err || { tail -5 gpurun_out/wpj.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/wpj.json"))
print("[$v]", round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if "tz_" in k}, round(d["stages_ms"]["uni_me"], 3))
PY
done
