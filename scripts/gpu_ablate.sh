cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ab in "$@"; do
  (cd /tmp && export TMPDIR=/tmp && HESAFF_ABLATE=$ab timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$ab -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 4 --no-cpu-baseline > /dev/null 2>&1)
done
python3 - "$@" <<'PY'
import csv, sys
keys=('k_patch_small<0>','k_patch_small<1>','k_patch_mid<128>','k_patch_mid<512>','k_patch_large_rows','k_patch_large_finish','k_affine','k_sift_hist','k_sift_meanvar')
for ab in sys.argv[1:]:
    rows=list(csv.DictReader(open(f'gpurun_out/abl_{ab}/p_kernel_stats.csv')))
    d={}
    for r in rows:
        for key in keys:
            if key in r['Name']: d[key]=float(r['TotalDurationNs'])/1e6/2
    print(ab, ' '.join(f"{k.replace('k_patch_','')}={v:6.2f}" for k,v in d.items()))
PY
