# usage (through gpurun): bash scripts/gpu_prof_jpeg.sh : rocprofv3 kernel summary of hesaff_process_files on 256 colour UHD JPEG photographs (8 + 8 host threads)
cd $GRAFT_REPO_ROOT
D=/dev/shm/hesaff_jpgprof; rm -rf $D; mkdir -p $D
python3 - <<PY
import os, shutil, sys
sys.path.insert(0, ".")
from PIL import Image
from hesaff_amd import synth
photos = synth.load_sample_photos(); paths = []
for i in range(256):
    q = os.path.join("$D", "p%04d.jpg" % i)
    if i < 32: Image.fromarray(synth.photo_mosaic(2160, 3840, i, photos=photos)).save(q, quality=90, subsampling=2)
    else: shutil.copyfile(paths[i % 32], q)
    paths.append(q)
open(os.path.join("$D", "l.txt"), "w").write("\n".join(paths) + "\n")
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/jpeg_prof -o p -- python3 $GRAFT_REPO_ROOT/scripts/jpeg_list_rate.py --case $D/l.txt 8 8 2 2>/dev/null | tail -1)
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/jpeg_prof/p_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%9.2f ms %6s calls %5.1f %%  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], 100 * float(r['TotalDurationNs']) / tot, r['Name'].replace('void ', '').split('(')[0][:60]))
PY
rm -rf $D
