# usage (through gpurun): bash scripts/gpu_ab_serial.sh <batch> <variant> <variant> ...   (hesaff_amd/variants/<variant>.so, from scripts/build_variant.sh)
# every kernel alone on the device (HESAFF_OVERLAP=0, rocprofv3 --stats), one column per variant, ms per step of <batch> UHD images;
# then the overlapped step at B = 256 per variant, two interleaved rounds (STEP_ROUNDS=0 skips it)
cd $GRAFT_REPO_ROOT
BATCH=$1; shift
mkdir -p gpurun_out/abs
for v in "$@"; do
  (cd /tmp && export TMPDIR=/tmp && HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so HESAFF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abs/$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline --no-host-path $BENCH_EXTRA > /dev/null 2>&1)
  find gpurun_out/abs/$v -name "*kernel_trace.csv" -delete; find gpurun_out/abs/$v -name "*.db" -delete
done
python3 - "$@" <<'PY'
import csv, sys, glob
vs = sys.argv[1:]
tab = {}
for v in vs:
    f = glob.glob('gpurun_out/abs/%s/**/p_kernel_stats.csv' % v, recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n = r['Name'].split('(')[0].replace('void ', '')
        if not n.startswith('k_'): continue
        tab.setdefault(n, {})[v] = float(r['TotalDurationNs']) / 3e6
print('%-62s' % 'kernel (ms per step, serial)' + ''.join('%10s' % v[:9] for v in vs))
tot = {v: 0.0 for v in vs}
for n, d in sorted(tab.items(), key=lambda kv: -max(kv[1].values())):
    for v in vs: tot[v] += d.get(v, 0.0)
    if max(d.values()) >= 0.5: print('%-62s' % n[:61] + ''.join('%10.3f' % d.get(v, float('nan')) for v in vs))
print('%-62s' % 'all kernels' + ''.join('%10.2f' % tot[v] for v in vs))
PY
for i in $(seq 1 ${STEP_ROUNDS:-2}); do for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch 256 --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s B=256 step %.1f ms' % ('$v', d['ms_per_step']))"
done; done
