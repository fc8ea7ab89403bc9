# GPU box: rectangular fractional-search kernels at 4 waves / SIMD with spills (default) against 3 waves / SIMD without (libvtmhip_rect3.so): the split-shape partition
cd $GRAFT_REPO_ROOT
cp vtm_amd/libvtmhip.so /tmp/libvtmhip_orig.so
trap 'cp /tmp/libvtmhip_orig.so vtm_amd/libvtmhip.so' EXIT
for t in orig rect3; do
  if [ $t != orig ]; then cp vtm_amd/libvtmhip_$t.so vtm_amd/libvtmhip.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --partition btt > gpurun_out/rect_$t.json 2> gpurun_out/rect.err || { tail -5 gpurun_out/rect.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/rect_$t.json"))
print("$t btt ms_per_step %.3f" % d["ms_per_step"], {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if "frac" in k})
PY
done
cp /tmp/libvtmhip_orig.so vtm_amd/libvtmhip.so
timeout -k 10 300 python -m pytest tests/test_gpu_interp.py -m gpu -x -q 2>&1 | tail -2
