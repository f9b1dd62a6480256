"""rigl.rigl_tf2.mask_updaters -> rigl_amd.mask_updaters (HIP-backed)."""
from rigl_amd.mask_updaters import (  # noqa: F401
    ConstantUpdateSchedule, CosineUpdateSchedule, MaskUpdater, RigL, RigLInverted,
    ScaledLRUpdateSchedule, SET, UpdateSchedule, get_mask_updater)
