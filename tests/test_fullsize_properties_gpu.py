"""Size-independent properties of the hot-path kernels at the FULL BASELINE sizes
(ResNet-50 layer shapes at per-GPU batch 128, all 54 masked tensors), where the
CPU oracle is too slow to be the checker: exact homogeneity, per-image
independence, batch additivity, conservation and idempotence."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
B = 128

# (H, W, Cin, Cout, k, stride, pad) -- one of each kind of ResNet-50 layer, at batch 128
LAYERS = [
    (56, 56, 64, 256, 1, 1, 0),      # 1x1 expand, HBM-bound
    (56, 56, 256, 64, 1, 1, 0),      # 1x1 reduce
    (28, 28, 128, 128, 3, 1, 1),     # 3x3
    (56, 56, 256, 512, 1, 2, 0),     # projection shortcut, stride 2 (parity classes)
    (14, 14, 256, 256, 3, 1, 1),     # 3x3, 392 tiles
    (7, 7, 512, 512, 3, 1, 1),       # 3x3, 196 tiles (narrow-tile path)
    (224, 224, 3, 64, 7, 2, 3),      # stem (tiny-Cin path, no dX)
]


def _setup(shape, seed):
  from rigl_amd import ops
  H, W, Cin, Cout, k, s, p = shape
  Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
  g = torch.Generator(device=DEV).manual_seed(seed)
  x = torch.randn(B, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
  dy = torch.randn(B, Ho, Wo, Cout, generator=g, device=DEV).to(torch.bfloat16)
  w = (torch.randn(k * k * Cin * Cout, generator=g, device=DEV) * (2.0 / (k * k * Cin)) ** 0.5)
  bits = ops.mask_pack((torch.rand(k * k * Cin * Cout, generator=g, device=DEV) < 0.2).float())
  hwio = torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV)
  ohwi = torch.empty_like(hwio)
  ops.pack_weights(w, bits, k * k * Cin, Cout, hwio, ohwi)

  def desc(n):
    return ops.conv_desc(n, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
  return x, dy, hwio, ohwi, desc


@pytest.mark.parametrize('shape', LAYERS)
def test_conv_homogeneity_and_image_independence(shape):
  from rigl_amd import ops
  x, dy, hwio, ohwi, desc = _setup(shape, 1)
  d, dh = desc(B), desc(B // 2)
  y = ops.conv_fwd(d, x, ohwi)
  # exact homogeneity: scaling an operand by 2 only changes exponents
  assert torch.equal(ops.conv_fwd(d, x * 2, ohwi), y * 2)
  # images do not interact, and the result does not depend on where a tile falls in the batch
  lo, hi = ops.conv_fwd(dh, x[:B // 2].contiguous(), ohwi), ops.conv_fwd(dh, x[B // 2:].contiguous(), ohwi)
  assert torch.equal(torch.cat([lo, hi]), y)
  # the epilogue statistics add up to the statistics of y
  y2, part = ops.conv_fwd(d, x, ohwi, stats=True)
  assert torch.equal(y2, y)
  yf = y.double().reshape(-1, y.shape[-1])
  s = part.double().sum(0)
  np.testing.assert_allclose(s[0].cpu().numpy(), yf.sum(0).cpu().numpy(), rtol=0, atol=2e-6 * float(yf.abs().sum(0).max()))
  np.testing.assert_allclose(s[1].cpu().numpy(), (yf * yf).sum(0).cpu().numpy(), rtol=2e-6)
  if shape[2] % 8 == 0:
    dx = ops.conv_dgrad(d, dy, hwio)
    assert torch.equal(ops.conv_dgrad(d, dy * 2, hwio), dx * 2)
    dlo = ops.conv_dgrad(dh, dy[:B // 2].contiguous(), hwio)
    dhi = ops.conv_dgrad(dh, dy[B // 2:].contiguous(), hwio)
    assert torch.equal(torch.cat([dlo, dhi]), dx)


@pytest.mark.parametrize('shape', LAYERS)
def test_wgrad_homogeneity_additivity_and_single_launch_backward(shape):
  from rigl_amd import ops
  x, dy, hwio, ohwi, desc = _setup(shape, 2)
  d, dh = desc(B), desc(B // 2)
  dw = ops.conv_wgrad(d, x, dy)
  assert torch.equal(ops.conv_wgrad(d, x * 2, dy), dw * 2)                       # same summation tree, exponents shift
  assert torch.equal(ops.conv_wgrad(d, x, dy), dw)                                # deterministic split-K reduction
  half = ops.conv_wgrad(dh, x[:B // 2].contiguous(), dy[:B // 2].contiguous()) + \
      ops.conv_wgrad(dh, x[B // 2:].contiguous(), dy[B // 2:].contiguous())
  scale = (x.float().abs().mean() * dy.float().abs().mean() * x.shape[0] * dy.shape[1] * dy.shape[2]).item()
  assert float((half - dw).abs().max()) <= 1e-5 * scale                           # reassociated fp32 sums
  # the training step's one-launch backward: dX has the same bits as the separate dgrad kernel; dW is the same
  # sum split over fewer pixel ranges (the weight-gradient workgroups share the launch with the dgrad tiles), so
  # it agrees to fp32 reassociation and is itself deterministic
  dw1, dw2 = torch.empty_like(dw), torch.empty_like(dw)
  need_dx = shape[2] % 8 == 0
  dx1 = ops.conv_bwd(d, x, dy, hwio, dw1, need_dx=need_dx)
  assert float((dw1 - dw).abs().max()) <= 1e-5 * scale
  ops.conv_bwd(d, x, dy, hwio, dw2, need_dx=need_dx)
  assert torch.equal(dw1, dw2)
  if need_dx:
    assert torch.equal(dx1, ops.conv_dgrad(d, dy, hwio))


def test_prune_regrow_and_momentum_properties_on_the_whole_model():
  """All 54 ResNet-50 tensors (25.5 M weights) in one K2 call: connections are
  conserved, masks stay 0/1-disjoint bookkeeping-wise (counts), a zero drop
  fraction is the identity; K3 with lr = 0 moves nothing but the accumulator."""
  from rigl_amd import _lib, ops
  from tests.golden import layer_shapes
  g = torch.Generator(device=DEV).manual_seed(3)
  layers, ones0 = [], []
  for sh in layer_shapes.resnet50().values():
    n = int(np.prod(sh))
    w = torch.randn(n, generator=g, device=DEV)
    # an ACTIVE weight that is exactly 0 ties with the inactive ones and loses to the lowest index -- the
    # reference's semantics, reproduced by the kernel (3 such draws in 25.5 M); the identity needs none
    w = torch.where(w == 0, torch.ones_like(w), w)
    m = (torch.rand(n, generator=g, device=DEV) < 0.2).float()
    layers.append(dict(w=w, mask_bits=ops.mask_pack(m), momentum=torch.randn(n, generator=g, device=DEV),
                       dense_grad=torch.randn(n, generator=g, device=DEV)))
    ones0.append(int(m.sum()))
  assert sum(l['w'].numel() for l in layers) == 25502912
  snap = [(l['w'].clone(), l['mask_bits'].clone(), l['momentum'].clone()) for l in layers]
  counts = ops.prune_regrow(layers, 0.0).cpu().numpy()
  for l, (w0, b0, a0) in zip(layers, snap):                                      # f = 0: identity
    assert torch.equal(l['w'], w0) and torch.equal(l['mask_bits'], b0) and torch.equal(l['momentum'], a0)
  counts = ops.prune_regrow(layers, 0.3).cpu().numpy()
  for i, (l, (w0, b0, a0)) in enumerate(zip(layers, snap)):
    n = l['w'].numel()
    n_ones, n_prune, n_keep, n_new, overlap, _, _, pop = counts[i]
    assert n_ones == ones0[i] and n_prune == int(np.float32(ones0[i]) * np.float32(0.3)) and n_keep == n_ones - n_prune
    assert overlap == 0 and pop == ones0[i]                                       # disjoint masks, connections conserved
    m1 = ops.mask_unpack(l['mask_bits'], (n,))
    m0 = ops.mask_unpack(b0, (n,))
    assert int(m1.sum()) == ones0[i]
    grown = (m1 > m0)
    assert int(grown.sum()) == n_new
    assert float(l['w'][grown].abs().max() if n_new else 0.0) == 0.0              # grow_init = zeros
    assert torch.equal(l['w'][~grown], w0[~grown])                                 # everything else untouched
  # K3: lr = 0 leaves the weights alone, the accumulator still integrates the masked gradient
  l = layers[10]
  w0, a0 = l['w'].clone(), l['momentum'].clone()
  ops.masked_sgd_momentum(l['w'], l['dense_grad'], 0.0, momentum=l['momentum'], mask_bits=l['mask_bits'], mu=0.9,
                          weight_decay=0.0, nesterov=True)
  assert torch.equal(l['w'], w0)
  m = ops.mask_unpack(l['mask_bits'], (l['w'].numel(),))
  assert torch.equal(l['momentum'], a0 * 0.9 + m * l['dense_grad'])
