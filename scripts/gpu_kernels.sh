# usage (through gpurun): bash scripts/gpu_kernels.sh <tag> [batch] [lib]
# per-kernel durations with every kernel alone on the device (tuning build, HESAFF_OVERLAP=0), rocprofv3 --stats
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-32}; LIB=${3:-$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HESAFF_AMD_LIB=$LIB HESAFF_OVERLAP=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline --no-host-path $BENCH_EXTRA > $OUT/bench.json 2> $OUT/log.txt
cd $GRAFT_REPO_ROOT
python3 - $OUT $BATCH <<'PY'
import csv, sys, glob, json
out, batch = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + '/**/p_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
lines = []
for r in rows:
    n = r['Name'].split('(')[0].replace('void ', '')
    if not (n.startswith('k_') or n.startswith('hsfast::k_')): continue
    ms = float(r['TotalDurationNs']) / 1e6 / 3.0   # per step (2 timed + 1 warm-up)
    tot += ms
    lines.append((ms, n, int(r['Calls']) // 3))
print('per step, batch %d, serial: %.1f ms of kernels' % (batch, tot))
for ms, n, calls in sorted(lines, reverse=True)[:24]:
    print('%9.3f ms  %5d calls  %s' % (ms, calls, n[:90]))
try:
    print(open(out + '/bench.json').read()[:400])
except Exception as e:
    print(e)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
