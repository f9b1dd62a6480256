"""The index arithmetic of the channel-sliced single-pass backward (rigl_amd/csrc/bwdslice.hpp) restated per lane in NumPy
(tools/experiments/emu/bs_emu.py: workgroup -> slice / row group, LDS-DMA piece placement with the source-side XOR
swizzle, the row-major and the transposing fragment reads, MFMA operand / accumulator layouts, the addend fragments, the dX
staging tile and its flush, the slab rows) against plain matrix products on integer data, and the bank layout of the dY
tile under that swizzle.  No GPU: pins the emulator the kernel was developed against (it found nothing to fix on the first
GPU run because the arithmetic had been checked here).  Reference: the autodiff of layers.masked_conv2d
(rigl/imagenet_resnet/pruning_layers.py:139-157)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emu():
  spec = importlib.util.spec_from_file_location('bs_emu', os.path.join(ROOT, 'tools', 'experiments', 'emu', 'bs_emu.py'))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  return m


def test_bank_layout_of_the_dual_use_tile():
  m = _emu()
  for co in (128, 256):
    m.bank_check(co)


def test_one_slice_ragged_rows_with_addend():
  # 8 row groups x 1 slice of 128 channels, cout 128, five tiles per group minus a ragged tail, addend through the ring
  _emu().check(8 * 5 * 32 - 7, 128, 128, 8, 4, add=True, seed=3)


def test_the_64_channel_slices_of_cout_512():
  # k_bwdslice64: the (channel fragment, output-channel half) split of the dgrad, the LDS hand-over of the partials, the
  # wave-private staging tile, the 128-byte-row swizzles; 8 row groups x 1 slice, ragged tail, addend through the ring
  m = _emu()
  m.bank_check64()
  m.check64(8 * 4 * 32 + 19, 64, 8, add=True, seed=5)
