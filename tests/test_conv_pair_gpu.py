"""The two readers of a ResNet group's first block input as ONE autograd node (pruning_layers.conv_pair /
_MaskedConvPairFn -> rigl_masked_conv2d_bwd_sub): the strided 1x1 projection's input gradient is computed on its own
[n, ho, wo] grid and added inside conv1's dgrad epilogue at the pixels (2i, 2j) -- never a full-size tensor.

Checked against (a) stock ops in float64 on the bf16 operands (resnet_model.py:456-501: projection shortcut =
conv2d_fixed_padding(kernel 1, strides 2) beside conv1 = 1x1 stride 1 on the same tensor) and (b) the two-node form
(conv_sub.fork + conv_main, RIGL_CONV_PAIR=0) at the three ResNet-50 shapes and a ragged one."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# (n, h, w, cin, projection units, conv1 units)
CASES = [
    (16, 56, 56, 256, 512, 128),       # group 2, block 0
    (16, 28, 28, 512, 1024, 256),      # group 3
    (32, 14, 14, 1024, 2048, 512),     # group 4
    (3, 15, 13, 64, 96, 40),           # odd sizes: the last row / column of the subsampled grid is pixel h - 1 / w - 1
]


def _run(case, pair, monkeypatch):
  from rigl_amd import pruning_layers as PL, variables as V
  n, h, w, cin, cs, cm = case
  monkeypatch.setattr(PL, '_CONV_PAIR', bool(pair))
  g = V.reset_default_graph(DEV)
  PL.set_init_seed(7)
  sub = PL.MaskedConv2d(g, 'proj', cin, cs, (1, 1), (2, 2), 'SAME', 'threshold', 0.0)
  main = PL.MaskedConv2d(g, 'c1', cin, cm, (1, 1), (1, 1), 'SAME', 'threshold', 0.0)
  g.finalize()
  gen = torch.Generator(device=DEV).manual_seed(n + h + cin)
  for l in (sub, main):                                  # random 30 % masks
    l.mask.assign((torch.rand(l.weights.shape, generator=gen, device=DEV) < 0.3).float())
  x = torch.randn(n, h, w, cin, generator=gen, device=DEV).to(torch.bfloat16).requires_grad_(True)
  xin = x * 1                                            # a non-leaf, like a block input inside the network
  xin.retain_grad()
  ys, ym = PL.conv_pair(sub, main, xin, bn_stats=True)
  assert (ys.grad_fn is ym.grad_fn) == pair              # one node | two nodes
  gs = torch.randn(ys.shape, generator=gen, device=DEV).to(torch.bfloat16)
  gm = torch.randn(ym.shape, generator=gen, device=DEV).to(torch.bfloat16)
  torch.autograd.backward([ys, ym], [gs, gm])
  torch.cuda.synchronize()
  out = dict(ys=ys.detach(), ym=ym.detach(), dx=xin.grad.detach().clone(),
             dws=sub.weights.grad.detach().clone().reshape(sub.weights.shape),
             dwm=main.weights.grad.detach().clone().reshape(main.weights.shape),
             ps=getattr(ys, 'bn_partials', None), pm=getattr(ym, 'bn_partials', None))
  ops = dict(x=x.detach(), gs=gs, gm=gm, ws=sub.vars.hwio.detach().float().reshape(1, 1, cin, cs),
             wm=main.vars.hwio.detach().float().reshape(1, 1, cin, cm))
  return out, ops


def _reference(o):
  x = o['x'].double().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
  ws = o['ws'].double().cpu().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
  wm = o['wm'].double().cpu().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
  ys = F.conv2d(x, ws, stride=2)
  ym = F.conv2d(x, wm)
  torch.autograd.backward([ys, ym], [o['gs'].double().cpu().permute(0, 3, 1, 2), o['gm'].double().cpu().permute(0, 3, 1, 2)])
  return dict(ys=ys.detach().permute(0, 2, 3, 1), ym=ym.detach().permute(0, 2, 3, 1), dx=x.grad.permute(0, 2, 3, 1),
              dws=ws.grad.permute(2, 3, 1, 0), dwm=wm.grad.permute(2, 3, 1, 0))


def _err(got, ref):
  return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_pair_node_against_float64_and_the_two_node_form(case, monkeypatch):
  pair, o = _run(case, True, monkeypatch)
  ref = _reference(o)
  two, o2 = _run(case, False, monkeypatch)
  assert torch.equal(o['x'], o2['x']) and torch.equal(o['gs'], o2['gs']) and torch.equal(o['ws'], o2['ws'])
  for name, got in (('pair', pair), ('two nodes', two)):
    assert _err(got['ys'], ref['ys']) <= 2.0 ** -8 and _err(got['ym'], ref['ym']) <= 2.0 ** -8, name
    assert _err(got['dws'], ref['dws']) <= 1e-5 and _err(got['dwm'], ref['dwm']) <= 1e-5, name
    # dX: each dgrad rounded to bf16, then their sum rounded once more
    assert _err(got['dx'], ref['dx']) <= 2.0 ** -7, (name, _err(got['dx'], ref['dx']))
  # the forward kernels are the same calls in both forms
  assert torch.equal(pair['ys'], two['ys']) and torch.equal(pair['ym'], two['ym'])
  for k in ('ps', 'pm'):
    assert (pair[k] is None) == (two[k] is None)
    if pair[k] is not None:
      assert torch.equal(pair[k], two[k])
  # off the projection's grid dX is conv1's dgrad alone -- in the pair form nothing was ever added there
  n, h, w, cin, cs, cm = case
  off = torch.ones(h, w, dtype=torch.bool)
  off[::2, ::2] = False
  dgm = F.conv_transpose2d(o['gm'].double().cpu().permute(0, 3, 1, 2), o['wm'].double().cpu().permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
  assert float((pair['dx'].double().cpu()[:, off] - dgm[:, off]).abs().max()) <= 2.0 ** -8 * float(dgm.abs().max())
