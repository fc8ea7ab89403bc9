# GPU box: full bench line + rocprofv3 kernel trace + PMC passes.  usage: gpurun -- "bash scripts/gpu_measure.sh <tag>", then
# python3 scripts/summarize_profiles.py <tag> here to condense gpurun_out/ into profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
shift
EXTRA="$@"
python bench.py $EXTRA > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || (tail -30 gpurun_out/bench_$TAG.err; exit 1)
cat gpurun_out/bench_$TAG.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','satd_gblocks_per_s')}); print(d['stages_ms']); print(d['kernels']); print(d.get('roofline')); print(d.get('cpu_baseline'))"
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 bench.py --serial --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > gpurun_out/prof_$TAG/bench_stdout.json 2> gpurun_out/prof_$TAG/bench_stderr.txt
mkdir -p gpurun_out/pmc_${TAG}_INSTS
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_INSTS -o bench -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/pmc_${TAG}_INSTS/stdout.json 2> gpurun_out/pmc_${TAG}_INSTS/stderr.txt
for CNT in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/pmc_${TAG}_$CNT
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$CNT -o bench -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/pmc_${TAG}_$CNT/stdout.json 2> gpurun_out/pmc_${TAG}_$CNT/stderr.txt
done
echo profiled
