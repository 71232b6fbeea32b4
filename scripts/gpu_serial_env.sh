# usage (through gpurun): bash scripts/gpu_serial_env.sh <batch> "<variant>[:ENV=val[,ENV=val...]]" ...
# per-kernel totals (rocprofv3 --kernel-trace --stats) of the device-resident bench step per configuration, one column each, ms per step of <batch> UHD
# images.  OVERLAP=0 (default): every kernel alone on the device (HESAFF_OVERLAP=0); OVERLAP=1: the overlapped product schedule (durations under contention)
cd $GRAFT_REPO_ROOT
BATCH=$1; shift
mkdir -p gpurun_out/sev
i=0
for cfg in "$@"; do
  v=${cfg%%:*}; e=""; [ "$cfg" != "$v" ] && e=${cfg#*:}
  ( for kv in ${e//,/ }; do export "$kv"; done
    cd /tmp && export TMPDIR=/tmp && HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so HESAFF_OVERLAP=${OVERLAP:-0} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sev/c$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch $BATCH --density ${DENSITY:-dense} --no-cpu-baseline --no-host-path > /dev/null 2>&1 )
  find gpurun_out/sev/c$i -name "*kernel_trace.csv" -delete; find gpurun_out/sev/c$i -name "*.db" -delete
  i=$((i+1))
done
python3 - "$@" <<'PY'
import csv, sys, glob
cfgs = sys.argv[1:]
tab = {}
for i, c in enumerate(cfgs):
    f = glob.glob('gpurun_out/sev/c%d/**/p_kernel_stats.csv' % i, recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n = r['Name'].split('(')[0].replace('void ', '')
        if not n.startswith('k_'): continue
        tab.setdefault(n, {})[i] = float(r['TotalDurationNs']) / 3e6
for i, c in enumerate(cfgs): print('c%d = %s' % (i, c))
print('%-62s' % 'kernel (ms per step)' + ''.join('%10s' % ('c%d' % i) for i in range(len(cfgs))))
tot = {i: 0.0 for i in range(len(cfgs))}
for n, d in sorted(tab.items(), key=lambda kv: -max(kv[1].values())):
    for i in tot: tot[i] += d.get(i, 0.0)
    if max(d.values()) >= 0.5: print('%-62s' % n[:61] + ''.join('%10.3f' % d.get(i, float('nan')) for i in range(len(cfgs))))
print('%-62s' % 'all kernels' + ''.join('%10.2f' % tot[i] for i in range(len(cfgs))))
PY
