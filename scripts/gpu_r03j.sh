# usage (through gpurun): bash scripts/gpu_r03j.sh  -- k_affine with two keypoints per wavefront (HS_AFFP_G=2): kernel alone, parity tests, step
cd $GRAFT_REPO_ROOT
V=hesaff_amd/variants
bash scripts/gpu_kcmp.sh k_affine $V/base.so
for nb in 8 12 16; do HESAFF_AFF_BLOCKS=$nb bash scripts/gpu_kcmp.sh k_affine $V/affg2.so | sed "s/^/blocks=$nb /"; done
for nb in 16; do HESAFF_AFF_BLOCKS=$nb bash scripts/gpu_kcmp.sh k_affine $V/affg2w4.so | sed "s/^/blocks=$nb /"; done
HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/$V/affg2.so timeout 900 python -m pytest tests -m gpu -x -q -k "affine or end_to_end or golden or full_size" 2>&1 | tail -3
bash scripts/gpu_variant_bench2.sh "base" "HESAFF_AFF_BLOCKS=12 affg2" "HESAFF_AFF_BLOCKS=16 affg2" "HESAFF_AFF_BLOCKS=8 affg2" "base"
