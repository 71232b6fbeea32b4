# usage (through gpurun): bash scripts/gpu_flags_ab.sh <batch> <variant> <variant> ...   (hesaff_amd/variants/<variant>.so, scripts/build_variant.sh)
# every kernel's serial time per variant (batch 32, rocprofv3 --stats, one table), then the overlapped step per variant, twice, interleaved
cd $GRAFT_REPO_ROOT
BATCH=$1; shift
for v in "$@"; do
(cd /tmp && export TMPDIR=/tmp && HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so HESAFF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/flags_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-host-path > /dev/null 2>&1)
done
python3 - "$@" <<'PY'
import csv, sys, collections
names = sys.argv[1:]
tab = collections.OrderedDict()
for v in names:
    for r in csv.DictReader(open('gpurun_out/flags_%s/p_kernel_stats.csv' % v)):
        k = r['Name'].replace('void ', '').split('(')[0]
        tab.setdefault(k, {})[v] = float(r['TotalDurationNs']) / 3e6
print('%-58s' % 'kernel (ms per 32 images, serial)' + ''.join('%10s' % v for v in names))
tot = {v: 0.0 for v in names}
for k, row in sorted(tab.items(), key=lambda kv: -kv[1].get(names[0], 0)):
    if not k.startswith('k_'): continue
    for v in names: tot[v] += row.get(v, 0)
    if row.get(names[0], 0) >= 0.5: print('%-58s' % k[:58] + ''.join('%10.2f' % row.get(v, float('nan')) for v in names))
print('%-58s' % 'all k_* kernels' + ''.join('%10.2f' % tot[v] for v in names))
PY
for i in 1 2; do for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch $BATCH --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s B=$BATCH step %.1f' % ('$v', d['ms_per_step']))"
done; done
