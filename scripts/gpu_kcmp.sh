# usage: bash scripts/gpu_kcmp.sh <kernel-name-prefix> lib1.so lib2.so ... : serial-mode time of one kernel per library build
cd $GRAFT_REPO_ROOT
K=$1; shift
for lib in "$@"; do
(cd /tmp && export TMPDIR=/tmp && HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/$lib HESAFF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kcmp_x -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline > /dev/null 2>&1)
python3 - "$K" $lib <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/kcmp_x/p_kernel_stats.csv')):
    if r['Name'].replace('void ', '').startswith(sys.argv[1]): print(sys.argv[2], r['Name'].split('(')[0], 'total/3 ms %.2f' % (float(r['TotalDurationNs'])/3e6))
PY
done
