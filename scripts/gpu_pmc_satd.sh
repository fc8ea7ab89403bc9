# GPU box: where the SATD-grid kernel spends its cycles (separate --pmc passes; kernel-trace only).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS"; do
  D=gpurun_out/pmc_satd_$(echo $SET | cut -d' ' -f1)
  mkdir -p $D
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -o s -- python3 scripts/satd_variants.py > $D/stdout.txt 2> $D/stderr.txt || (tail -5 $D/stderr.txt; exit 1)
  python3 - $D <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1] + '/s_counter_collection.csv')):
    if 'satd8_grid' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
done
