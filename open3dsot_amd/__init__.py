"""open3dsot_amd -- MI355X (gfx950) native implementation of Open3DSOT's PointNet++ /
BAT / P2B feature-extraction hot path.

    open3dsot_amd.csrc/          hand-written HIP kernels + the C-ABI (include/o3dsot.h)
    open3dsot_amd.capi           ctypes binding of the C-ABI (fails loudly if the .so is absent)
    open3dsot_amd.ext            the nine `pointnet2_ops._ext` operators on torch tensors
    open3dsot_amd.pointnet2_utils / pointnet2_modules / pytorch_utils
                                 host-side mirror of the reference's pointnet2.utils API
    open3dsot_amd.backbone / xcorr / rpn / trackers
                                 host-side mirror of models/backbone, models/head, BAT / P2B
    open3dsot_amd.dist           one-process-per-GPU data-parallel step (RCCL all-reduce)
"""
__version__ = "0.1.0"
