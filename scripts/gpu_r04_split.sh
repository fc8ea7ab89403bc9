# GPU box: the finest level as K levels (VTM_AMD_SPLIT_LAST): whole picture and 1/8, 1/4 shares; the split must not change a result (tests/test_gpu_pis.py under each K)
cd $GRAFT_REPO_ROOT
for k in 1 2 3 4; do
  for sim in 0 8 4; do
    VTM_AMD_SPLIT_LAST=$k VTM_BENCH_SIMULATE_WORLD=$sim timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/split${k}_sim$sim.json 2> gpurun_out/split.err || { tail -5 gpurun_out/split.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/split${k}_sim$sim.json"))
print("split $k sim $sim ms_per_step %.3f" % d["ms_per_step"])
PY
  done
done
VTM_AMD_SPLIT_LAST=2 timeout -k 10 400 python -m pytest tests/test_gpu_pis.py -m gpu -x -q 2>&1 | tail -2
