# GPU box: VERDICT r3 item 6 experiments on the picture driver's issue order and the number of hardware queues (whole picture and a 1/8 share)
cd $GRAFT_REPO_ROOT
run() { # tag, env...
  tag=$1; shift
  for sim in 0 8 4; do
    env "$@" VTM_BENCH_SIMULATE_WORLD=$sim timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/q_${tag}_sim$sim.json 2> gpurun_out/q_${tag}_sim$sim.err || { tail -5 gpurun_out/q_${tag}_sim$sim.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/q_${tag}_sim$sim.json"))
print("${tag} sim$sim ms_per_step %.3f" % d["ms_per_step"])
PY
  done
}
run base
run hwq8 GPU_MAX_HW_QUEUES=8
run inter VTMHIP_ISSUE_ORDER=interleave
run inter_hwq8 VTMHIP_ISSUE_ORDER=interleave GPU_MAX_HW_QUEUES=8
