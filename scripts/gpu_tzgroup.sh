# GPU box: the lane-per-candidate integer search (tz_group_kernel) -- parity tests that reach it, the bench with it on / off / taking 32x32 too, instruction counters of the kernel.
# usage: gpurun -- 'bash scripts/gpu_tzgroup.sh <tag>'
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
timeout -k 10 500 python -m pytest tests/test_gpu_mest.py tests/test_gpu_pis.py tests/test_gpu_me.py -m gpu -x -q > gpurun_out/tzg_tests_$TAG.txt 2>&1 || (grep -n "^E " gpurun_out/tzg_tests_$TAG.txt | head -30; tail -5 gpurun_out/tzg_tests_$TAG.txt; exit 1)
tail -2 gpurun_out/tzg_tests_$TAG.txt
for g in 1 0 64; do
  if [ $g = 64 ]; then export VTMHIP_TZ_GROUP=1 VTMHIP_TZ_GROUP_ITEMS=64; else export VTMHIP_TZ_GROUP=$g; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/tzg_err_$TAG.txt | tail -1 > gpurun_out/tzg_bench_${TAG}_g$g.json || (tail -30 gpurun_out/tzg_err_$TAG.txt; exit 1)
  python - <<P
import json; d=json.load(open('gpurun_out/tzg_bench_${TAG}_g$g.json'))
print('group=$g', round(d['value'],2), round(d['ms_per_step'],3), {k:(round(v['ms_per_step'],3),v['launches_per_step']) for k,v in d['kernels'].items() if 'tz_' in k}, round(d['stages_ms']['uni_me'],3))
P
done
unset VTMHIP_TZ_GROUP_ITEMS
export VTMHIP_TZ_GROUP=1
mkdir -p gpurun_out/tzg_pmc_$TAG
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --kernel-include-regex "tz_" --output-format csv -d gpurun_out/tzg_pmc_$TAG -o p -- python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/tzg_pmc_$TAG/stdout.json 2> gpurun_out/tzg_pmc_$TAG/stderr.txt
python - <<P
import csv, glob, collections
f = glob.glob('gpurun_out/tzg_pmc_$TAG/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'][:40], r['Grid_Size'])][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items()): print(k, dict(v))
P
