cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch ${BATCH:-16} --no-cpu-baseline > /dev/null 2>&1)
python3 - $1 <<'PY'
import csv, sys
rows=list(csv.DictReader(open(f'gpurun_out/ks_{sys.argv[1]}/p_kernel_stats.csv')))
for r in rows[:22]:
    n=r['Name']
    if n.startswith('void at::') or 'igemm' in n or 'Tensor' in n: continue
    print('%-60s calls=%4s total/3=%8.2f ms avg=%9.1f us'%(n.split('(')[0][:60], r['Calls'], float(r['TotalDurationNs'])/3e6, float(r['AverageNs'])/1e3))
PY
