# usage (through gpurun): bash scripts/gpu_sweep_sift2.sh   -- descriptor kernels of even / odd groups on one or two streams (tuning build)
cd $GRAFT_REPO_ROOT
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so
run() { python bench.py --no-cpu-baseline --no-host-path --batch $2 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s B=$2 ms_per_step %.1f img/s %.1f' % ('$1', d['ms_per_step'], d['images_per_s']))"; }
HESAFF_SIFT2=0 run one_sift_stream 128
HESAFF_SIFT2=1 run two_sift_streams 128
HESAFF_SIFT2=0 run one_sift_stream 128
HESAFF_SIFT2=1 run two_sift_streams 128
HESAFF_SIFT2=0 run one_sift_stream 256
HESAFF_SIFT2=1 run two_sift_streams 256
