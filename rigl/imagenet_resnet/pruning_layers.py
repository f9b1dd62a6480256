"""rigl.imagenet_resnet.pruning_layers -> rigl_amd.pruning_layers."""
from rigl_amd.pruning_layers import (  # noqa: F401
    l2_regularizer, sparse_conv2d, sparse_fully_connected)
