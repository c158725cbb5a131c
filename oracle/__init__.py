"""CPU oracle for the Open3DSOT PointNet++ hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product package ``open3dsot_amd`` never does.

``oracle.ops``        numpy front-end of the plain-C restatement (pointnet2_oracle.c)
``oracle.ext_shim``   a CPU ``pointnet2_ops._ext`` look-alike on torch tensors (used to
                      import the reference's own Python layers in the build container)
``oracle.torch_ref``  pure-PyTorch fp32 restatement of the SA / XCorr / RPN / BAT graph
"""
