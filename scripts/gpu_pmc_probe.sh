# GPU box: stall / unit-activity counters of the kernels matching a regex (several small --pmc passes).
# (TA_* / TCP_* counter passes hung the profiler on this pool: SQ counters only.)
# usage: gpurun -- "bash scripts/gpu_pmc_probe.sh <tag> '<kernel regex>' [bench args]"; output: gpurun_out/probe_<tag>/pass*/ (csv)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
RE=$2
shift; shift
EXTRA="$@"
i=0
for CNTS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  mkdir -p gpurun_out/probe_$TAG/pass$i
  rocprofv3 --pmc $CNTS --kernel-trace --kernel-include-regex "$RE" --output-format csv -d gpurun_out/probe_$TAG/pass$i -o p -- python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/probe_$TAG/pass$i/stdout.json 2> gpurun_out/probe_$TAG/pass$i/stderr.txt || echo "pass $i failed"
done
echo probed
