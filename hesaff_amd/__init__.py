"""hesaff_amd -- MI355X-native Hessian-Affine + SIFT (drop-in for perdoch/hesaff's detect+describe path).

Python host mirror over the C ABI (include/hesaff_amd.h -> hesaff_amd/libhesaff_amd.so).
The library is HIP only: there is no CPU fallback, creating a context without a gfx950
device raises HesaffError.
"""
from ._binding import (  # noqa: F401
    HesaffError,
    HesaffContext,
    Params,
    KEYPOINT_DTYPE,
    default_params,
    format_sift,
    format_sift_mt,
    write_sift,
    write_sift_batch,
    write_bin,
    read_bin,
    BIN_ROW_DTYPE,
    read_image,
    read_pnm,
    read_jpeg_coefficients,
    JpegLayout,
    ellipse,
    lib_path,
    load_library,
    host_plan,
    table_gauss_mask,
    table_circ_gauss_mask,
    table_sift_bins,
    table_gauss_kernel,
)
