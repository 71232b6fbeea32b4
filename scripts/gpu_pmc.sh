# usage: bash scripts/gpu_pmc.sh <tag> "<counters>" [kernel-regex]
cd $GRAFT_REPO_ROOT
TAG=$1; CTRS=$2; KRE=${3:-.}
mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1)
tail -2 gpurun_out/pmc_$TAG.log
python3 - $TAG "$KRE" <<'PY'
import csv, sys, re, collections
tag, kre = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f'gpurun_out/pmc_{tag}/p_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    if not re.search(kre, n): continue
    n = n.split('(')[0][:40]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
for n, d in agg.items():
    print(n, {k: ('%.4g' % v) for k, v in d.items()})
PY
