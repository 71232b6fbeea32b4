# usage: bash scripts/gpu_pmc_all.sh <tag> : per-kernel PMC table (serial mode, batch 8): VALU busy, LDS, HBM bytes
cd $GRAFT_REPO_ROOT
TAG=$1
run() { # name counters
  (cd /tmp && export TMPDIR=/tmp && HESAFF_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcall_${TAG}_$1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 8 --no-cpu-baseline > /dev/null 2>&1)
}
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
run b "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
# (a third pass with FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE over all kernels hung the profiler once: not collected here)
python3 - $TAG <<'PY'
import csv, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); calls = collections.Counter()
for part in 'ab':
    d = f'gpurun_out/pmcall_{tag}_{part}'
    for r in csv.DictReader(open(d + '/p_counter_collection.csv')):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if part == 'a':
        for r in csv.DictReader(open(d + '/p_kernel_trace.csv')):
            n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
            dur[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; calls[n] += 1
print('| %-34s | %5s | %8s | %9s | %6s | %9s | %8s | %8s |' % ('kernel', 'calls', 'ms', 'VALU inst', 'VALU %', 'SALU/VALU', 'LDS/VALU', 'LDS conf'))
print('|---|---|---|---|---|---|---|---|')
for n in sorted(dur, key=lambda k: -dur[k])[:22]:
    a = agg[n]
    if not n.startswith('k_'): continue
    ms = dur[n]
    # SIMD-cycles available: 1024 SIMDs * 2.4e6 cycles/ms ; a wave64 VALU op holds a SIMD 4 cycles
    valu_pct = 100.0 * a['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * 2.4e6 * ms) if ms else 0
    print('| %-34s | %5d | %8.2f | %9.3g | %6.1f | %9.2f | %8.2f | %8.2f |' % (n, calls[n], ms, a['SQ_INSTS_VALU'], valu_pct,
          a['SQ_INSTS_SALU'] / max(a['SQ_INSTS_VALU'], 1), a['SQ_INSTS_LDS'] / max(a['SQ_INSTS_VALU'], 1),
          a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_ACTIVE_INST_LDS'], 1)))
PY
