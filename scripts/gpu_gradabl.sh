cd $GRAFT_REPO_ROOT
for a in 0; do
(cd /tmp && export TMPDIR=/tmp && HESAFF_ABLATE=$a HESAFF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ga_$a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline > /dev/null 2>&1)
python3 - $a <<'PY'
import csv, sys
a = sys.argv[1]
for r in csv.DictReader(open(f'gpurun_out/ga_{a}/p_kernel_stats.csv')):
    if r['Name'].startswith('k_sift_grad'): print('ablate', a, 'k_sift_grad avg us', float(r['AverageNs'])/1e3, 'total/3 ms', float(r['TotalDurationNs'])/3e6)
PY
done
