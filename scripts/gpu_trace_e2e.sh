# usage (through gpurun): bash scripts/gpu_trace_e2e.sh <format 1|2> : kernel trace of the file path (scripts/dbg_e2e.py, 2 + 2 threads, 256 files); prints what runs
# beside the copy-out blits and how long the same kernels take with and without one beside them
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e2e_trace -o p -- python3 $GRAFT_REPO_ROOT/scripts/dbg_e2e.py $1 2 256 32 2>/dev/null | tail -1)
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/e2e_trace/p_kernel_trace.csv')))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name'].replace('void ', '').split('(')[0][:44]
blits = [r for r in rows if 'copyBuffer' in r['n'] and r['e'] - r['s'] > 500000]
print('long blits:', len(blits), 'durations ms', sorted(set(round((b['e'] - b['s']) / 1e6, 1) for b in blits)))
inside = collections.defaultdict(list); outside = collections.defaultdict(list)
for r in rows:
    if 'copyBuffer' in r['n']: continue
    ov = any(b['s'] < r['e'] and b['e'] > r['s'] for b in blits)
    (inside if ov else outside)[r['n']].append((r['e'] - r['s']) / 1e3)
print('%-46s %8s %10s %8s %10s' % ('kernel', 'n beside', 'avg us', 'n alone', 'avg us'))
for k in sorted(inside, key=lambda k: -sum(inside[k])):
    if len(inside[k]) >= 2 and outside.get(k):
        print('%-46s %8d %10.1f %8d %10.1f' % (k, len(inside[k]), sum(inside[k]) / len(inside[k]), len(outside[k]), sum(outside[k]) / len(outside[k])))
PY
