# GPU box: A/B of prebuilt library variants vtm_amd/libvtmhip_<tag>.so (built in the container with other -D settings): for each tag the variant is copied over
# vtm_amd/libvtmhip.so, the parity tests of the touched kernels run, then the default bench without the CPU chain.
# usage: gpurun -- 'bash scripts/gpu_lib_variants.sh "<pytest files>" tag1 tag2 ...'
set -e
cd $GRAFT_REPO_ROOT
TESTS=$1; shift
cp vtm_amd/libvtmhip.so /tmp/libvtmhip_orig.so
trap 'cp /tmp/libvtmhip_orig.so vtm_amd/libvtmhip.so' EXIT      # whatever happens below (a failing test, a bench timeout, a missing variant): the tree gets its own library back
for t in "$@"; do
  cp vtm_amd/libvtmhip_$t.so vtm_amd/libvtmhip.so
  timeout -k 10 500 python -m pytest $TESTS -m gpu -x -q > gpurun_out/var_${t}_tests.log 2>&1 || (tail -20 gpurun_out/var_${t}_tests.log; exit 1)
  tail -1 gpurun_out/var_${t}_tests.log
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/var_${t}_bench.json 2> gpurun_out/var_${t}_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/var_${t}_bench.json"))
print("${t}", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stages_ms"].items()})
print("${t}", {k: round(v["ms_per_step"] if isinstance(v, dict) else v, 3) for k, v in d["kernels"].items()})
PY
done
