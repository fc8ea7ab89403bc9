# GPU box: parity tests on the tree's library, then its bench beside prebuilt variants vtm_amd/libvtmhip_<tag>.so (bench only: no tests on the variants).
# usage: gpurun -- 'bash scripts/gpu_ab_lib.sh "<pytest files>" "<kernel substring>" tag1 tag2 ...'
set -e
cd $GRAFT_REPO_ROOT
TESTS=$1; K=$2; shift; shift
timeout -k 10 700 python -m pytest $TESTS -m gpu -x -q > gpurun_out/ab_tests.log 2>&1 || (grep -n "^E " gpurun_out/ab_tests.log | head -20; tail -5 gpurun_out/ab_tests.log; exit 1)
tail -1 gpurun_out/ab_tests.log
cp vtm_amd/libvtmhip.so /tmp/libvtmhip_orig.so
trap 'cp /tmp/libvtmhip_orig.so vtm_amd/libvtmhip.so' EXIT
for t in tree "$@"; do
  if [ $t != tree ]; then cp vtm_amd/libvtmhip_$t.so vtm_amd/libvtmhip.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/ab_${t}.json 2> gpurun_out/ab_${t}.err || (tail -5 gpurun_out/ab_${t}.err; exit 1)
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_${t}.json"))
print("${t}", round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if "$K" in k}, {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
done
