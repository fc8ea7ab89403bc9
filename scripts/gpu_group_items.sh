# GPU box: the group kernel's block-size cap (VTMHIP_TZ_GROUP_ITEMS) on the quadtree and the split-shape partitions.  usage: gpurun -- 'bash scripts/gpu_group_items.sh'
cd $GRAFT_REPO_ROOT
for it in 16 32; do for part in qt btt; do
VTMHIP_TZ_GROUP_ITEMS=$it timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --partition $part > gpurun_out/gi.json 2> gpurun_out/gi.err || { tail -3 gpurun_out/gi.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/gi.json')); print('items $it $part', round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items() if 'tz_' in k})"
done; done
