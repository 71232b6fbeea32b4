"""Text timeline of the last bench step from a rocprofv3 kernel trace: per 5 ms slice, the busy
time of each kernel family (sum over concurrent launches), so that overlap and bubbles show."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Queue_Id', '')) for r in rows]
ks.sort()
# the last step starts at its first pyramid launch: the initial blur that also converts the 8-bit source (template flag SRC8 = true,
# the only k_blur_hess_march instantiation whose last argument is true), or k_gray on the non-default paths
starts = [s for s, e, n, q in ks if n.startswith('k_gray') or (n.startswith('k_blur_hess_march') and n.rstrip('>').endswith('true') and ', false, false, false, true' in n)]
t0 = max(starts)
ks = [k for k in ks if k[0] >= t0]
# ... and ends with its k_pack launch (what follows in the trace belongs to other legs of the bench)
ends = [e for s, e, n, q in ks if n.startswith('k_pack')]
t1 = min(ends) if ends else max(e for s, e, n, q in ks)
ks = [k for k in ks if k[0] < t1]
print('step span %.1f ms, %d launches' % ((t1 - t0) / 1e6, len(ks)))
def fam(n):
    for key, f in (('k_patch_extract_small<0', 'small0'), ('k_patch_extract_small<1', 'small1'), ('k_patch_small<0', 'small0'), ('k_patch_small<1', 'small1'), ('k_patch_mid<128', 'mid128'), ('k_patch_mid<512', 'mid512'),
                   ('k_patch_large', 'large'), ('k_sift_grad', 'grad'), ('k_sift_hist', 'hist'), ('k_sift_meanvar', 'meanvar'), ('k_sift_quant', 'quant'),
                   ('k_affine', 'affine'), ('k_blur_hess', 'pyr'), ('k_extrema', 'extrema'), ('k_localize', 'extrema'), ('k_prepare', 'prep')):
        if n.startswith(key): return f
    return 'other'
SL = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
nsl = int((t1 - t0) / SL) + 1
acc = [collections.Counter() for _ in range(nsl)]
# union busy time
ev = []
for s, e, n, q in ks:
    f = fam(n)
    a = s
    while a < e:
        i = int((a - t0) / SL)
        b = min(e, t0 + (i + 1) * SL)
        acc[i][f] += (b - a) / 1e6
        a = b
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print('time with >= 1 kernel running: %.1f ms (idle %.1f ms)' % (busy / 1e6, (t1 - t0 - busy) / 1e6))
fams = ['pyr', 'extrema', 'affine', 'prep', 'small0', 'small1', 'mid128', 'mid512', 'large', 'meanvar', 'grad', 'hist', 'quant', 'other']
print('slice   ' + ' '.join('%7s' % f for f in fams))
for i, a in enumerate(acc):
    print('%5.0f ms ' % (i * SL / 1e6) + ' '.join('%7.1f' % a[f] if a[f] > 0.05 else '      .' for f in fams))
tot = collections.Counter()
for a in acc: tot.update(a)
print('total    ' + ' '.join('%7.1f' % tot[f] for f in fams))
