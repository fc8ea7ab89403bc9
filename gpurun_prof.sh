set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$TAG/bench_stdout.json 2> gpurun_out/prof_$TAG/bench_stderr.txt || (tail -20 gpurun_out/prof_$TAG/bench_stderr.txt; exit 1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_$TAG/bench_kernel_trace.csv')))
ours=[r for r in rows if 'anonymous' in r['Kernel_Name']]
import collections
last={}
for r in ours:
    key=(r['Kernel_Name'].split('::')[1].split('(')[0], r['Grid_Size_X'])
    last.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in last.items(): print(k, 'n=%d'%len(v), 'median_us=%.1f'%sorted(v)[len(v)//2], 'vgpr', [r['VGPR_Count'] for r in ours if r['Grid_Size_X']==k[1]][0])
PY
