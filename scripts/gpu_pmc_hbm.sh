# usage (through gpurun): bash scripts/gpu_pmc_hbm.sh <tag> [batch]
# HBM bytes per kernel (every kernel alone on the device: tuning build, HESAFF_OVERLAP=0): FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
# (KB; FETCH doubled as in profiles/README.md: gfx950 counts 128-byte fills as 64), per Hessian keypoint of the batch
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/hbm_$TAG
mkdir -p $OUT
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so HESAFF_OVERLAP=0
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-host-path > $OUT/$C.json 2> $OUT/$C.log
done
cd $GRAFT_REPO_ROOT
python3 - $OUT $BATCH <<'PY'
import csv, sys, glob, json, collections
out, batch = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: [0.0, 0.0])
for k, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob(out + "/" + c + "/**/p_counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][k] += float(r["Counter_Value"])
d = json.loads(open(out + "/FETCH_SIZE.json").read().strip().splitlines()[-1])
nk = d["config"]["hessian_keypoints_timed_all_ranks"]
print("batch %d: %d Hessian keypoints; HBM bytes per Hessian keypoint (read = 2 x FETCH_SIZE KB, written = WRITE_SIZE KB)" % (batch, nk))
print("| kernel | read GB | written GB | read B/keypoint | written B/keypoint |\n|---|---|---|---|---|")
for n, (f, w) in sorted(acc.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    if not n.startswith("k_") or 2 * f + w < 1e4: continue
    print("| %s | %.2f | %.2f | %.0f | %.0f |" % (n[:60], 2 * f * 1024 / 1e9, w * 1024 / 1e9, 2 * f * 1024 / nk, w * 1024 / nk))
PY
