# GPU box: dynamic instruction counters of the bench kernels (separate --pmc pass, kernel-trace only).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_insts
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_insts -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_insts/stdout.json 2> gpurun_out/pmc_insts/stderr.txt || (tail -20 gpurun_out/pmc_insts/stderr.txt; exit 1)
python3 - <<'PY'
import csv, collections, re
acc = collections.OrderedDict()
for r in csv.DictReader(open('gpurun_out/pmc_insts/bench_counter_collection.csv')):
    n = r['Kernel_Name']
    if 'at::native' in n: continue
    m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', n)
    k = (m.group(1) if m else n[:40], r['Grid_Size'])
    d = acc.setdefault(k, collections.defaultdict(float))
    d[r['Counter_Name']] += float(r['Counter_Value']); d['_n'] += 1
for k, d in acc.items():
    nd = d['_n'] / 5.0
    w = d['SQ_WAVES'] / nd
    print(k, 'waves %.0f' % w, ' per wave: valu %.0f salu %.0f vmem_rd %.0f lds %.0f' % (d['SQ_INSTS_VALU'] / nd / w, d['SQ_INSTS_SALU'] / nd / w, d['SQ_INSTS_VMEM_RD'] / nd / w, d['SQ_INSTS_LDS'] / nd / w))
PY
