# GPU box: bench kernel times of prebuilt library variants vtm_amd/libvtmhip_<tag>.so WITHOUT parity tests (experiments whose results are wrong on purpose).
# usage: gpurun -- 'bash scripts/gpu_lib_time.sh <kernel substring> tag1 tag2 ...'
set -e
cd $GRAFT_REPO_ROOT
K=$1; shift
cp vtm_amd/libvtmhip.so /tmp/libvtmhip_orig.so
trap 'cp /tmp/libvtmhip_orig.so vtm_amd/libvtmhip.so' EXIT
for t in "$@"; do
  cp vtm_amd/libvtmhip_$t.so vtm_amd/libvtmhip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/time_${t}_bench.json 2> gpurun_out/time_${t}_bench.err || (tail -5 gpurun_out/time_${t}_bench.err; exit 1)
  python - <<PY
import json
d = json.load(open("gpurun_out/time_${t}_bench.json"))
print("${t}", round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if "$K" in k})
PY
done
