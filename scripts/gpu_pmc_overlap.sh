# usage (through gpurun): bash scripts/gpu_pmc_overlap.sh <tag> [batch]
# Counters over the OVERLAPPED device-resident step of the PRODUCT library (no HESAFF_OVERLAP=0): one rocprofv3 --kernel-trace --pmc pass with the
# issue counters, FETCH_SIZE and WRITE_SIZE in passes of their own, and one plain --kernel-trace pass (no counters) for the wall-clock window of the
# per-keypoint stage.  Writes gpurun_out/stage_util_<tag>.md: VALU-issue, LDS and HBM time of the per-keypoint kernels against the window.
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmco_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-host-path"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/plain -o p -- $B > $OUT/plain.json 2> $OUT/plain.log
timeout 420 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- $B > $OUT/sq.json 2> $OUT/sq.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- $B > $OUT/$C.json 2> $OUT/$C.log
done
cd $GRAFT_REPO_ROOT
python3 - $OUT $BATCH > gpurun_out/stage_util_$TAG.md <<'PY'
import csv, sys, glob, json, collections
out, batch = sys.argv[1], int(sys.argv[2])
KP = ("k_affine", "k_prepare_patch", "k_patch_", "k_large_prefix", "k_sift_")
def short(n): return n.split('(')[0].replace('void ', '')
def is_kp(n): return any(n.startswith(p) for p in KP)
def trace(d):
    f = glob.glob(out + '/' + d + '/**/p_kernel_trace.csv', recursive=True)[0]
    rows = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
    return [r for r in rows if r[0].startswith('k_')]
def window(rows):
    kp = sorted((s, e) for n, s, e in rows if is_kp(n))
    if not kp: return 0.0, 0.0, 0.0
    span = (max(e for s, e in kp) - kp[0][0]) / 1e6
    union, cs, ce = 0, kp[0][0], kp[0][1]
    for s, e in kp[1:]:
        if s > ce: union += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    union += ce - cs
    return span, union / 1e6, sum(e - s for s, e in kp) / 1e6
def counters(d):
    f = glob.glob(out + '/' + d + '/**/p_counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
    return agg
def bench(d):
    try: return json.loads(open(out + '/' + d + '.json').read().strip().splitlines()[-1])
    except Exception: return {}
pl, sq = trace('plain'), trace('sq')
sp, un, sm = window(pl)
sp2, un2, sm2 = window(sq)
bj, bsq = bench('plain'), bench('sq')
print('# Utilisation of the per-keypoint stage, overlapped step, B = %d x 3840x2160 (scripts/gpu_pmc_overlap.sh)\n' % batch)
print('Per-keypoint kernels = k_affine, k_prepare_patch, k_patch_*, k_large_prefix, k_sift_* of ONE step (`bench.py --steps 1 --warmup 0`, product library).\n')
print('| pass | step ms (bench) | stage window ms (first start .. last end) | time with >= 1 such kernel running | sum of kernel durations | overlap factor |')
print('|---|---|---|---|---|---|')
print('| `--kernel-trace` only | %.1f | %.1f | %.1f | %.1f | %.2f |' % (bj.get('ms_per_step', 0), sp, un, sm, sm / max(un, 1e-9)))
print('| `--kernel-trace --pmc` (7 SQ counters) | %.1f | %.1f | %.1f | %.1f | %.2f |' % (bsq.get('ms_per_step', 0), sp2, un2, sm2, sm2 / max(un2, 1e-9)))
print("\n(An overlap factor of 1.00 in the counter pass means the profiler serialises dispatches while it counts: the counters below are then per-kernel WORK - instruction issue cycles, LDS cycles, bytes - which does not depend on what runs beside a kernel; the window they are set against is the plain pass's.)\n")
a = counters('sq')
tot = collections.defaultdict(float)
for n, c in a.items():
    if is_kp(n):
        for k, v in c.items(): tot[k] += v
fe, wr = counters('FETCH_SIZE'), counters('WRITE_SIZE')
fkb = sum(c['FETCH_SIZE'] for n, c in fe.items() if is_kp(n)); wkb = sum(c['WRITE_SIZE'] for n, c in wr.items() if is_kp(n))
cyc_busy = tot['GRBM_GUI_ACTIVE'] / 8.0          # shader-clock cycles with the GPU busy, summed over these kernels' dispatches (counter kept per XCD)
clock = cyc_busy / (sm2 * 1e6) if sm2 else 0.0   # GHz while these kernels ran in the counter pass
win = sp / 1e3
nk = bj.get('config', {}).get('hessian_keypoints_timed_all_ranks', 0)
valu_s = tot['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (clock * 1e9) if clock else 0
lds_s = tot['SQ_LDS_IDX_ACTIVE'] / 256 / (clock * 1e9) if clock else 0
rd, wrb = 2 * fkb * 1024, wkb * 1024             # FETCH_SIZE doubled: gfx950 counts 128-byte fills as 64 (MI355X_MICROARCH.md); see the note
print('| quantity (per-keypoint kernels of one step) | value | as time | share of the %.1f ms window |' % sp)
print('|---|---|---|---|')
print('| SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs (cycles a SIMD spends issuing vector ALU instructions) | %.3g SIMD-cycles | %.3f s at the %.2f GHz of the pass | %.0f %% |' % (tot['SQ_ACTIVE_INST_VALU'] * 4 / 1024, valu_s, clock, 100 * valu_s / win if win else 0))
print('| SQ_INSTS_VALU | %.3g wavefront instructions = %.0f per Hessian keypoint | | |' % (tot['SQ_INSTS_VALU'], tot['SQ_INSTS_VALU'] / max(nk, 1)))
print("| SQ_LDS_IDX_ACTIVE / 256 CUs (cycles a CU's LDS spends on indexed accesses) | %.3g CU-cycles | %.3f s | %.0f %% |" % (tot['SQ_LDS_IDX_ACTIVE'] / 256, lds_s, 100 * lds_s / win if win else 0))
print('| FETCH_SIZE x 2 (bytes read past L2) | %.1f GB = %.1f KB per Hessian keypoint | %.3f s at 5 TB/s (the copy rate measured on the boxes of the pool) | %.0f %% |' % (rd / 1e9, rd / max(nk, 1) / 1e3, rd / 5e12, 100 * rd / 5e12 / win if win else 0))
print('| FETCH_SIZE as counted | %.1f GB | %.3f s | %.0f %% |' % (rd / 2e9, rd / 2 / 5e12, 100 * rd / 2 / 5e12 / win if win else 0))
print('| WRITE_SIZE (bytes written past L2) | %.1f GB = %.1f KB per Hessian keypoint | %.3f s | %.0f %% |' % (wrb / 1e9, wrb / max(nk, 1) / 1e3, wrb / 5e12, 100 * wrb / 5e12 / win if win else 0))
print('| SQ_BUSY_CYCLES / 8 XCDs... (raw) | %.3g | | |' % tot['SQ_BUSY_CYCLES'])
print('| SQ_WAVE_CYCLES x 4 / 1024 (resident wavefronts per SIMD, averaged over the time these kernels ran) | %.2f | | |' % (tot['SQ_WAVE_CYCLES'] * 4 / 1024 / max(cyc_busy, 1)))
print('\nHessian keypoints of the step: %d.  Floor of the stage = the largest of the three times: **%.3f s = %.0f %% of the window**.\n' % (nk, max(valu_s, lds_s, (rd + wrb) / 5e12), 100 * max(valu_s, lds_s, (rd + wrb) / 5e12) / win if win else 0))
print('| kernel | launches | ms in the plain pass (sum) | ms in the counter pass | VALU issue ms | LDS ms | read GB (2 x FETCH) | written GB |')
print('|---|---|---|---|---|---|---|---|')
dp, dq, cp = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
for n, s, e in pl: dp[n] += (e - s) / 1e6; cp[n] += 1
for n, s, e in sq: dq[n] += (e - s) / 1e6
for n in sorted(dp, key=lambda k: -dp[k]):
    if not is_kp(n) or dp[n] < 0.5: continue
    c = a.get(n, {})
    print('| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |' % (n[:48], cp[n], dp[n], dq[n], 1e3 * c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / (clock * 1e9) if clock else 0,
          1e3 * c.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / (clock * 1e9) if clock else 0, 2 * fe.get(n, {}).get('FETCH_SIZE', 0) * 1024 / 1e9, wr.get(n, {}).get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
[ -s gpurun_out/stage_util_$TAG.md ] && { find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; }
find $OUT -name "*.db" -delete
cat gpurun_out/stage_util_$TAG.md
