cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ab in 0 2 4 6 8 16 30; do
  (cd /tmp && export TMPDIR=/tmp && HESAFF_ABLATE=$ab timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$ab -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 4 --no-cpu-baseline > /dev/null 2>&1)
  echo "== ablate $ab"; grep -E "k_patch_sift|k_affine|k_extrema" gpurun_out/abl_$ab/p_kernel_stats.csv | cut -d, -f1-4 | sed 's/(HessList.*)"//'
done
