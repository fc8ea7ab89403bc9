set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
for CNT in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/pmc_${TAG}_$CNT
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$CNT -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$CNT/stdout.json 2> gpurun_out/pmc_${TAG}_$CNT/stderr.txt || (tail -20 gpurun_out/pmc_${TAG}_$CNT/stderr.txt; exit 1)
  ls gpurun_out/pmc_${TAG}_$CNT
done
