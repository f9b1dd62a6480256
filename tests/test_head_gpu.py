"""Classifier-head glue kernels (global average pool, softmax cross-entropy with label smoothing)
against their PyTorch fp32 formulations."""
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('shape', [(128, 7, 7, 2048), (3, 8, 8, 640), (5, 1, 1, 64), (2, 4, 3, 10)])
def test_global_avg_pool_fwd_bwd(shape):
  from rigl_amd.workloads import nn as gnn
  g = torch.Generator().manual_seed(sum(shape))
  x = torch.randn(*shape, generator=g).to(torch.bfloat16).to(DEV).requires_grad_(True)
  dy = torch.randn(shape[0], shape[3], generator=g).to(torch.bfloat16).to(DEV)
  y = gnn.global_avg_pool(x)
  y.backward(dy)
  xr = x.detach().clone().requires_grad_(True)
  yr = xr.float().mean(dim=(1, 2)).to(torch.bfloat16)
  yr.backward(dy)
  # fp32 accumulation in a different order, then one bf16 rounding: at most one bf16 ulp apart
  assert y.shape == yr.shape and y.dtype == torch.bfloat16
  assert (y.float() - yr.float()).abs().max() <= 2.0**-8 * yr.float().abs().max() + 1e-30
  assert torch.equal(x.grad.view(torch.int16), xr.grad.view(torch.int16))      # elementwise: bit-exact


@pytest.mark.parametrize('rows,k,eps', [(128, 1000, 0.1), (7, 10, 0.0), (33, 1000, 0.0), (4, 37, 0.3), (2, 5000, 0.1)])
def test_softmax_cross_entropy(rows, k, eps):
  from rigl_amd import ops
  from rigl_amd.workloads import nn as gnn
  g = torch.Generator().manual_seed(rows * k)
  z = (torch.randn(rows, k, generator=g) * 3.0).to(torch.bfloat16).to(DEV)
  labels = torch.randint(0, k, (rows,), generator=g).to(DEV)
  zf = z.float().requires_grad_(True)
  ref = F.cross_entropy(zf, labels, label_smoothing=eps)          # == tf.losses.softmax_cross_entropy(label_smoothing)
  ref.backward()
  per_row, dz = ops.softmax_xent(z, labels, eps)
  ref_rows = F.cross_entropy(z.float(), labels, label_smoothing=eps, reduction='none')
  assert (per_row - ref_rows).abs().max() <= 1e-5 * max(1.0, float(ref_rows.abs().max()))   # north star: 1e-5 fp32
  # gradient: bf16 of (softmax - target) / rows -- half a bf16 ulp of the value plus fp32 noise
  assert (dz.float() - zf.grad).abs().max() <= 2.0**-8 * float(zf.grad.abs().max()) + 1e-9
  # through autograd, with an upstream factor
  zz = z.clone().requires_grad_(True)
  loss = gnn.softmax_cross_entropy(zz, labels, eps)
  (loss * 2.0).backward()
  assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
  assert (zz.grad.float() - 2.0 * zf.grad).abs().max() <= 2.0**-7 * 2.0 * float(zf.grad.abs().max()) + 1e-9


def test_head_argument_errors():
  from rigl_amd import ops
  from rigl_amd._lib import RiglError
  z = torch.zeros(4, 10, dtype=torch.bfloat16, device=DEV)
  with pytest.raises(ValueError):
    ops.softmax_xent(z, torch.zeros(3, dtype=torch.int64, device=DEV))
  with pytest.raises(RiglError):
    ops.softmax_xent(z, torch.zeros(4, dtype=torch.int64, device=DEV), label_smoothing=1.5)
  with pytest.raises(RiglError):
    ops.global_avgpool_fwd(torch.zeros(2, 3, 3, 5, dtype=torch.bfloat16, device=DEV))    # odd channel count
  with pytest.raises(TypeError):
    ops.global_avgpool_fwd(torch.zeros(2, 3, 3, 4, dtype=torch.float32, device=DEV))
