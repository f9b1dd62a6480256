"""Drop-in alias: the reference's module paths (``rigl.sparse_optimizers``,
``rigl.sparse_utils``, ``rigl.imagenet_resnet.pruning_layers``) re-exported
from the MI355X-native implementation in ``rigl_amd``."""
name = 'rigl'
