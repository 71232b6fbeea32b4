# usage (through gpurun): python scripts/dbg_jpeg.py <decode threads> <write threads> [files]: per-chunk timings (tuning build, HESAFF_DEBUG) of a JPEG list
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, ".")
from PIL import Image
from hesaff_amd import synth
n = int(sys.argv[3]) if len(sys.argv) > 3 else 192
tmp = tempfile.mkdtemp(prefix="hesaff_jpg_", dir="/dev/shm")
try:
    photos = synth.load_sample_photos(); paths = []
    for i in range(n):
        q = os.path.join(tmp, "p%04d.jpg" % i)
        Image.fromarray(synth.photo_mosaic(2160, 3840, i, photos=photos)).save(q, quality=90, subsampling=2); paths.append(q)
    lst = os.path.join(tmp, "l.txt"); open(lst, "w").write("\n".join(paths) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HESAFF_AMD_LIB=os.path.join(root, "hesaff_amd", "libhesaff_amd_tuning.so"), HESAFF_DEBUG="1")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "jpeg_list_rate.py"), "--case", lst, sys.argv[1], sys.argv[2], "2"], env=env, capture_output=True, text=True)
    print("\n".join(l[:250] for l in r.stderr.splitlines() if "chunk" in l and "march" not in l)); print(r.stdout[-400:])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
