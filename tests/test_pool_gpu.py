"""Max-pool glue kernels vs torch's max_pool2d (fp32 on the same bf16 values),
forward and backward, incl. TF-SAME's asymmetric padding and tie order."""
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (3, 15, 13, 8), (1, 7, 9, 24), (4, 112, 112, 64), (2, 2, 2, 16)])
def test_maxpool_3x3_s2_same(shape):
  from rigl_amd.workloads import nn as gnn
  n, h, w, c = shape
  g = torch.Generator(device=DEV).manual_seed(sum(shape))
  x = torch.randn(shape, generator=g, device=DEV).to(torch.bfloat16)      # bf16 normals tie often: order matters
  xr = x.float().requires_grad_(True)
  ho, wo = -(-h // 2), -(-w // 2)
  ph, pw = max((ho - 1) * 2 + 3 - h, 0), max((wo - 1) * 2 + 3 - w, 0)
  xp = F.pad(xr.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float('-inf'))
  yr = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
  xg = x.clone().requires_grad_(True)
  y = gnn.max_pool_3x3_s2_same(xg)
  assert y.shape == (n, ho, wo, c) and torch.equal(y.float(), yr.detach())
  dy = torch.randn(y.shape, generator=g, device=DEV).to(torch.bfloat16)
  y.backward(dy)
  yr.backward(dy.float())
  # up to 4 windows add into one input: bf16 output of an fp32 sum
  gr = xr.grad
  assert ((xg.grad.float() - gr).abs() <= 2.0**-7 * gr.abs() + 1e-6).all()
  assert ((xg.grad == 0) == (gr == 0)).all()          # same winners


def test_maxpool_general_window_and_errors():
  from rigl_amd import ops, _lib
  x = torch.randn(2, 9, 11, 8, device=DEV).to(torch.bfloat16)
  d = ops.conv_desc(2, 9, 11, 8, 8, 2, 3, (2, 1), 0, 1, 4, 11)
  y, arg = ops.maxpool_fwd(d, x)
  xp = F.pad(x.float().permute(0, 3, 1, 2), (1, 1, 0, 0), value=float('-inf'))
  yr = F.max_pool2d(xp, (2, 3), (2, 1))[:, :, :4, :11].permute(0, 2, 3, 1)
  assert torch.equal(y.float(), yr)
  assert int(arg.max()) < 6
  with pytest.raises(_lib.RiglError):
    ops.maxpool_fwd(ops.conv_desc(2, 9, 11, 8, 16, 2, 3, (2, 1), 0, 1, 4, 11), x)   # channel change
