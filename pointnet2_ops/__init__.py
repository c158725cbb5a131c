"""Drop-in `pointnet2_ops` package: lets the reference's
`import pointnet2_ops._ext as _ext` (pointnet2/utils/pointnet2_utils.py:17) resolve to the
MI355X-native HIP library of open3dsot_amd -- no CUDA extension, no upstream package."""
__version__ = "3.0.0+o3dsot.gfx950"
