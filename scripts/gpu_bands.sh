cd $GRAFT_REPO_ROOT
for nb in "$@"; do
 (cd /tmp && export TMPDIR=/tmp && HESAFF_BANDS=$nb timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bands_$nb -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline > /dev/null 2>&1)
 python3 - $nb <<'PY'
import csv, sys, collections
nb=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/bands_{nb}/p_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'march' in n:
        key=(n.split('<')[1].split(',')[0], int(r['Grid_Size_X'])//256, r['Grid_Size_Y'])
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out=[]
for k,v in sorted(agg.items(), key=lambda kv:(-kv[0][1], kv[0][0])):
    if k[1]>=2: out.append('K%s gx%d gy%s: %.0fus'%(k[0],k[1],k[2],sorted(v)[len(v)//2]))
print('bands',nb,' | '.join(out))
PY
done
