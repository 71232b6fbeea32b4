# usage (through gpurun): bash scripts/gpu_pyr.sh <lib> ...   roofline numbers of the pyramid kernels for several builds (bench --batch 64, 3 steps)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/$lib python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', 'ms/step', round(d['ms_per_step'],1), 'kernel frac', round(r['frac'],3), 'stage frac', round(r['stage']['frac'],3), 'pyramid_ms', round(d['stage_ms_per_step']['serial_on_main_stream']['pyramid_ms'],2))"
done
