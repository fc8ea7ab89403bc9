cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 6; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 36 --inflight $k > gpurun_out/inflight$k.json 2> gpurun_out/inflight.err || { tail -5 gpurun_out/inflight.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/inflight$k.json"))
print("inflight $k ms_per_step %.3f value %.1f" % (d["ms_per_step"], d["value"]))
PY
done
