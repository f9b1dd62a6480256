"""tf.contrib.model_pruning.python.pruning getters over the default graph
(what PruningGetterTf1Mixin calls, rigl/sparse_optimizers.py:46-56)."""
from rigl_amd import variables as V


def get_weights(graph=None):
  return (graph or V.get_default_graph()).get_weights()


def get_masks(graph=None):
  return (graph or V.get_default_graph()).get_masks()


def get_masked_weights(graph=None):
  """mask * W per layer (materialised on request; the kernels never need it)."""
  g = graph or V.get_default_graph()
  return [l.mask.data * l.weights.data for l in g.masked_layers()]
