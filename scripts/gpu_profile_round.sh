# Round-end evidence: rocprofv3 kernel stats of the bench command + HBM traffic PMC passes.
# usage: bash scripts/gpu_profile_round.sh <round tag>     (run through gpurun; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/fetch.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/write.log
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
def agg(path, counter):
    tot = 0.0; n = 0
    for r in csv.DictReader(open(path)):
        if 'k_blur_hess_march' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            # the initial blur launch (template argument WRITE_R = false: no response) is not one of the 58 B/px launches
            targs = r['Kernel_Name'].split('<', 1)[1].split('>', 1)[0].split(',')
            if targs[2].strip() == 'false': continue
            tot += float(r['Counter_Value']); n += 1
    return tot, n
f, nf = agg(out + '/fetch/p_counter_collection.csv', 'FETCH_SIZE')
w, nw = agg(out + '/write/p_counter_collection.csv', 'WRITE_SIZE')
bench = json.loads(open(out + '/bench_under_rocprof.json').read().strip().splitlines()[-1])
cfg = bench['config']
res = {"fetch_kb_total": f, "write_kb_total": w, "launches": nf,
       "bytes_per_launch_avg": (2.0 * f + w) * 1024.0 / max(nf, 1),
       "bytes_per_launch_raw": (f + w) * 1024.0 / max(nf, 1),
       "algorithmic_bytes_per_launch_avg": bench['roofline']['bytes_per_launch_avg'],
       "batch": cfg['images_per_gpu_per_step'], "width": cfg['width'], "height": cfg['height'],
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path (separate passes)",
       "note": "FETCH_SIZE / WRITE_SIZE (KB) from two separate rocprofv3 --pmc passes, summed over the 58 B/px k_blur_hess_march launches of one bench step. bytes_per_launch_avg applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B line fills as 64 B: x2 on the read side); bytes_per_launch_raw is the uncorrected sum."}
print(json.dumps(res))
json.dump(res, open(out + '/pmc_traffic_raw.json', 'w'), indent=1)
rows = list(csv.DictReader(open(out + '/stats/p_kernel_stats.csv')))
for r in rows[:14]:
    print('%-70s calls=%5s total=%9.3f ms avg=%9.1f us' % (r['Name'].split('(')[0][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
cat $OUT/bench_under_rocprof.json | head -c 1500
# raw traces are large (gpurun merges at most 64 MiB back): keep the summaries only
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
