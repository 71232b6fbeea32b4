cd $GRAFT_REPO_ROOT
run() { (cd /tmp && export TMPDIR=/tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hx -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps $S --warmup 1 --batch $B --no-cpu-baseline > /dev/null 2>&1)
python3 - "$*" <<'PY'
import csv, sys
rows=list(csv.DictReader(open('gpurun_out/hx/p_kernel_trace.csv')))
out=[]
for name in ('k_sift_hist','k_sift_grad','k_sift_meanvar'):
    v=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows if name in r['Kernel_Name']]
    out.append(name+': '+' '.join('%.1f'%x for x in v))
print(sys.argv[1], '|', ' | '.join(out))
PY
}
B=16 S=1 run HESAFF_X=1


B=16 S=1 run HESAFF_X=1
