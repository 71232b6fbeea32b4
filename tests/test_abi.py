"""CPU suite: the C-ABI library loads and exports every symbol include/hesaff_amd.h declares;
host-only entry points work; computing entry points fail loudly without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import hesaff_amd
from hesaff_amd import _binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hesaff_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hesaff_[a-z0-9_]+)\s*\(", txt)))


def test_library_built_in_tree():
    assert os.path.exists(hesaff_amd.lib_path()), "run __graft_entry__.build()"
    assert os.path.dirname(hesaff_amd.lib_path()) == os.path.join(ROOT, "hesaff_amd")


def test_every_header_symbol_is_exported():
    L = hesaff_amd.load_library()
    syms = header_symbols()
    assert len(syms) >= 28
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_binding.ABI_SYMBOLS) == syms, "python binding list out of sync with the header"


def test_no_oracle_dependency_in_product():
    """The product library must not link or reference anything under oracle/."""
    out = subprocess.run(["ldd", hesaff_amd.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hesaff_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hesaff_oracle" not in src and "tests._oracle" not in src and "from tests" not in src, os.path.join(dirpath, f)


def test_product_library_reads_no_environment():
    """A drop-in library must not change its schedule or results because of an environment variable: the product
    build imports no getenv (the schedule knobs exist only in the optional -DHESAFF_TUNING build)."""
    out = subprocess.run(["nm", "-D", "--undefined-only", hesaff_amd.lib_path()], capture_output=True, text=True).stdout
    assert "getenv" not in out
    for f in ("capi_impl.h", "pipeline.hip", "hostio.cpp"):
        src = open(os.path.join(ROOT, "hesaff_amd", "csrc", f)).read()
        src = re.sub(r"#ifdef HESAFF_TUNING.*?#(else|endif)", "", src, flags=re.S)
        assert "getenv" not in src, f


def test_bad_png_header_is_an_error_not_an_abort(tmp_path):
    """A tiny PNG claiming 65535 x 65535 RGBA16 (34 GB unpacked): HESAFF_ERR_IO, no allocation of that size, no abort."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 65535, 65535, 16, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b"")
    p = tmp_path / "bomb.png"
    p.write_bytes(png)
    with pytest.raises(hesaff_amd.HesaffError) as e:
        hesaff_amd.read_image(str(p))
    assert e.value.code == -4


def test_keypoint_layout_matches_reference_struct():
    # struct Keypoint hesaff.cpp:41-48: 8 floats, int, 128 bytes
    assert _binding.KEYPOINT_DTYPE.itemsize == 8 * 4 + 4 + 128
    assert _binding.KEYPOINT_DTYPE.fields["desc"][1] == 36


def test_default_params_are_reference_defaults():
    p = hesaff_amd.default_params()
    assert np.float32(p.threshold) == np.float32(16.0) / np.float32(3.0)          # pyramid.h:37
    assert p.edgeEigenValueRatio == 10.0 and np.float32(p.initialSigma) == np.float32(1.6)
    assert p.maxIterations == 16                                                    # affine.h:39
    assert np.float32(p.convergenceThreshold) == np.float32(0.05)                   # affine.h:41
    assert np.float32(p.mrSize) == np.float32(3.0) * np.sqrt(np.float32(3.0))       # hesaff.cpp:32
    assert np.float32(p.maxBinValue) == np.float32(0.2)                             # siftdesc.h:29
    assert p.upscaleInputImage == 0 and p.fast == 0                                 # pyramid.h:34 ; parity mode


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hesaff_amd.HesaffError) as e:
        hesaff_amd.HesaffContext()
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_cli_usage_text_matches_reference():
    exe = os.path.join(ROOT, "hesaff_amd", "bin", "hesaff")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0
    # hesaff.cpp:178
    assert r.stdout == ("\nUsage: hesaff image_name.ppm\nDetects Hessian Affine points and describes them using SIFT descriptor.\n"
                        "The detector assumes that the vertical orientation is preserved.\n\n")
