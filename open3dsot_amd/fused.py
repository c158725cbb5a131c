"""Fused grouped-MLP path (gather + 1x1 conv + BatchNorm + ReLU + max-pool on fp32 MFMA).
Placeholder until csrc/mlp.hip lands: reports 'unsupported' so callers use the composed path."""


def supports(grouper, mlp, features):
    return False


def supports_mlp(mlp):
    return False
