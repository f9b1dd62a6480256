#!/usr/bin/env python3
"""K1 parity runner (GPU): every conv entry point of the C ABI against tests/convref.py's fp64 reference on the same
bf16 operands, with the per-element bound  2^-8 |ref| + 1e-5 sum|a||b|  (fwd / dgrad)  and  1e-5 sum|a||b|  (wgrad).

Run as a script so that the kernel-selection environment (RIGL_PP_FWD, RIGL_PP_DGRAD, RIGL_PP_PH, RIGL_CONV_BIG, ...),
which the library reads once per process, can be chosen per invocation:

  python tests/k1_check.py --set small         # small 1x1 / 3x3 shapes (both column widths, ragged N, halo rows)
  python tests/k1_check.py --set pp            # edge cases of the 8-wave ping-pong body (1..9 K-tiles, row tails, stride 2)
  python tests/k1_check.py --set rs            # the row-streaming 1x1 body (run with RIGL_ROWSTREAM=2: every legal shape takes it)
  python tests/k1_check.py --set bs            # the channel-sliced single-pass 1x1 backward (bwdslice.hpp): 1 .. 32 slices, ragged tiles
  python tests/k1_check.py --set c3            # the slab-resident 3x3 kernels (64 -> 64 channels): tile heights, widths, partial tiles
  python tests/k1_check.py --set resnet50 --batch 128     # the 23 distinct ResNet-50 layer shapes at the benchmarked batch

Prints one line per case and a final JSON line {"ok": true, "cases": n, "worst": ratio}; exit code 1 on a mismatch.
Test infrastructure: the pytest files call it in a subprocess.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests import convref  # noqa: E402

DEV = 'cuda:0'

# N, H, W, Cin, Cout, k, stride, pad_top, pad_left, Ho, Wo
SMALL_CASES = [
    (4, 7, 7, 32, 128, 1, 1, 0, 0, 7, 7),         # 1x1, one K-tile (the peeled tail alone)
    (4, 7, 7, 64, 64, 1, 1, 0, 0, 7, 7),          # 1x1, two K-tiles, 64 columns (K split)
    (8, 7, 7, 96, 96, 1, 1, 0, 0, 7, 7),          # three K-tiles, ragged N (96 = 64 + 32)
    (4, 14, 14, 256, 128, 1, 1, 0, 0, 14, 14),    # 1x1, 8 K-tiles, four row tiles
    (2, 14, 14, 160, 64, 1, 1, 0, 0, 14, 14),     # odd number of K-tiles with the K split (zero batch in the tail)
    (4, 7, 7, 32, 128, 3, 1, 1, 1, 7, 7),         # 3x3, one channel block, one tile spanning four images
    (4, 7, 7, 64, 64, 3, 1, 1, 1, 7, 7),          # 3x3, two channel blocks (slab double buffer), K split
    (2, 14, 14, 96, 72, 3, 1, 1, 1, 14, 14),      # 3x3, three channel blocks, ragged N
    (1, 28, 28, 128, 128, 3, 1, 1, 1, 28, 28),    # four row bands of one image: real halo rows between tiles
    (2, 14, 28, 64, 256, 3, 1, 1, 1, 14, 28),     # H != W, two column tiles
    (16, 7, 7, 512, 512, 3, 1, 1, 1, 7, 7),       # the late 3x3 of ResNet-50 (16 channel blocks)
    (4, 14, 14, 256, 256, 3, 1, 1, 1, 14, 14),    # the group-3 3x3
]


# Ping-pong body (convpp.hpp): prologue / tail branches by K-tile count, row tails, strides, column widths
PP_CASES = [
    (2, 14, 14, 64, 256, 1, 1, 0, 0, 14, 14),     # one K-tile, 392 rows (a ragged second 256-row tile)
    (2, 14, 14, 128, 256, 1, 1, 0, 0, 14, 14),    # two K-tiles
    (3, 14, 14, 192, 128, 1, 1, 0, 0, 14, 14),    # three K-tiles, 128 columns, 588 rows
    (4, 7, 7, 320, 256, 1, 1, 0, 0, 7, 7),        # five K-tiles, 196 rows (only the 128-row tile is legal)
    (2, 14, 14, 448, 512, 1, 1, 0, 0, 14, 14),    # seven K-tiles, two column tiles
    (2, 14, 14, 64, 256, 3, 1, 1, 1, 14, 14),     # 3x3: nine K-tiles, image borders in every tile
    (2, 28, 28, 64, 128, 3, 2, 1, 1, 14, 14),     # 3x3 stride 2 forward (fixed padding)
    (2, 14, 14, 64, 256, 3, 2, 0, 0, 7, 7),       # 3x3 stride 2, TF SAME on an even size (pad 0 top/left, 1 bottom/right)
    (1, 16, 20, 128, 384, 3, 1, 1, 1, 16, 20),    # H != W, 384 columns (128-wide tiles only)
    (2, 9, 9, 64, 512, 3, 1, 1, 1, 9, 9),         # odd sizes, 162 rows
    (3, 14, 14, 256, 256, 3, 1, 1, 1, 14, 14),    # 36 K-tiles: steady state of the ring
    (2, 28, 28, 512, 256, 1, 2, 0, 0, 14, 14),    # 1x1 stride 2 forward
]


# The ImageNet stem kernels (stem.hpp): image borders on every side, odd tile counts, the TF-SAME top padding of 2
STEM_CASES = [
    (2, 64, 64, 3, 64, 7, 2, 3, 3, 32, 32),        # 2 x 2 tiles per image
    (3, 32, 96, 3, 64, 7, 2, 3, 3, 16, 48),        # H != W, one tile row
    (1, 96, 32, 3, 64, 7, 2, 3, 3, 48, 16),        # one tile column: both side borders in every tile
    (5, 64, 64, 3, 64, 7, 2, 2, 3, 32, 32),        # 2 rows of padding above, 3 + 1 below
    (4, 224, 224, 3, 64, 7, 2, 3, 3, 112, 112),    # the benchmarked image size (7 x 7 tiles)
    (2, 60, 64, 3, 64, 7, 2, 3, 3, 30, 32),        # not a whole number of tiles: the generic path
]


# The slab-resident 3x3 kernels for 64 -> 64 channels (c3x3.hpp): whole and partial bottom tiles, odd and maximal widths,
# tile heights from 6 to 15 rows, one- and many-image batches
C3_CASES = [
    (2, 56, 56, 64, 64, 3, 1, 1, 1, 56, 56),      # the ResNet-50 plane: eight tiles of seven rows per image
    (3, 57, 56, 64, 64, 3, 1, 1, 1, 57, 56),      # a one-row ninth tile at the bottom of every image
    (4, 40, 33, 64, 64, 3, 1, 1, 1, 40, 33),      # odd width (35-pixel patch rows), 13 + 13 + 13 + 1 rows
    (5, 30, 28, 64, 64, 3, 1, 1, 1, 30, 28),      # two tiles of 15 rows
    (1, 64, 62, 64, 64, 3, 1, 1, 1, 64, 62),      # the widest legal plane (64-pixel patch rows), one image
    (9, 24, 24, 64, 64, 3, 1, 1, 1, 24, 24),      # the narrowest, odd batch
]


# The row-streaming 1x1 body (rowstream.hpp): every (reduction, slice width) instantiation in both directions -- run with
# RIGL_ROWSTREAM=2 so that the "reduce" shapes take it too -- fragments of fewer than 32 rows, ragged last fragments,
# waves without a fragment, 1 .. 32 column slices
RS_CASES = [
    (16, 56, 56, 64, 256, 1, 1, 0, 0, 56, 56),     # fwd K = 64 / 128 columns (2 slices); dgrad K = 256 / 64 columns (1 slice)
    (130, 23, 19, 64, 256, 1, 1, 0, 0, 23, 19),    # ... 56 810 rows: ragged
    (16, 56, 56, 64, 64, 1, 1, 0, 0, 56, 56),      # K = 64 / 64 columns both ways
    (8, 28, 28, 128, 64, 1, 1, 0, 0, 28, 28),      # fwd K = 128 / 64 columns; dgrad K = 64 / 128 columns; 6 272 rows on 2 048 waves
    (64, 28, 28, 128, 512, 1, 1, 0, 0, 28, 28),    # fwd K = 128 / 128 columns (4 slices); dgrad K = 512 / 64 columns (2 slices)
    (50, 13, 17, 128, 512, 1, 1, 0, 0, 13, 17),    # ... 11 050 rows
    (200, 14, 14, 256, 1024, 1, 1, 0, 0, 14, 14),  # fwd K = 256 (8 slices); dgrad (K = 1024) on the other bodies
    (128, 14, 14, 1024, 256, 1, 1, 0, 0, 14, 14),  # dgrad K = 256 / 1024 columns, the benchmarked shape (25 rows per fragment)
    (16, 56, 56, 256, 64, 1, 1, 0, 0, 56, 56),     # fwd K = 256 / 64 columns; dgrad K = 64 / 256 columns
    (64, 28, 28, 512, 128, 1, 1, 0, 0, 28, 28),    # fwd K = 512 / 64 columns; dgrad K = 128 / 512 columns
    (128, 7, 7, 512, 2048, 1, 1, 0, 0, 7, 7),      # fwd K = 512, 32 slices of 64 columns, 8 workgroups per slice
    (128, 7, 7, 2048, 512, 1, 1, 0, 0, 7, 7),      # dgrad K = 512 / 2048 columns
    (90, 7, 7, 256, 1024, 1, 1, 0, 0, 7, 7),       # 4 410 rows: 18 rows per fragment, one pass
]


# The channel-sliced single-pass 1x1 backward (bwdslice.hpp): cout 128 / 256, 1 .. 32 slices of 128 input channels, row
# groups with and without a ragged last tile, the benchmarked shapes
BS_CASES = [
    (128, 14, 14, 1024, 256, 1, 1, 0, 0, 14, 14),  # the benchmarked group-3 layer: 8 slices x 32 row groups
    (128, 28, 28, 512, 128, 1, 1, 0, 0, 28, 28),   # the group-2 layer: 4 slices x 64 row groups, cout 128 (1 x 2 fragments per wave)
    (32, 28, 28, 512, 256, 1, 1, 0, 0, 28, 28),    # 4 slices, cout 256
    (16, 56, 56, 256, 128, 1, 1, 0, 0, 56, 56),    # 2 slices x 128 row groups
    (130, 13, 15, 1024, 256, 1, 1, 0, 0, 13, 15),  # 25 350 rows: a six-row last tile
    (130, 23, 19, 128, 128, 1, 1, 0, 0, 23, 19),   # one slice, 256 row groups, ragged
    (67, 14, 14, 2048, 128, 1, 1, 0, 0, 14, 14),   # 16 slices x 16 row groups, ragged
    (20, 14, 14, 4096, 128, 1, 1, 0, 0, 14, 14),   # 32 slices x 8 row groups
    (128, 7, 7, 2048, 512, 1, 1, 0, 0, 7, 7),      # cout 512: 32 slices of 64 channels x 8 row groups (k_bwdslice64)
    (128, 14, 14, 1024, 512, 1, 1, 0, 0, 14, 14),  # 16 slices x 16 row groups
    (37, 11, 13, 256, 512, 1, 1, 0, 0, 11, 13),    # 4 slices x 64 row groups, 5 291 rows: ragged
    (9, 14, 14, 64, 512, 1, 1, 0, 0, 14, 14),      # one slice x 256 row groups... too few tiles: stays on the former body
]


def _c3_tile_rows(H, W):
  """Tile height of the c3x3.hpp forward (c3x3_geom restated): the most rows whose patch + zero tail fit the 544-pixel
  LDS budget of one of the two patch buffers, one fewer where that divides H."""
  pw = W + 2
  for th in range(H, 0, -1):
    mt = (th * pw + 31) // 32
    if (mt * 32 + 2 * pw + 2 + 7) // 8 * 8 <= 544:
      break
  if H % th and th > 1 and H % (th - 1) == 0:
    th -= 1
  return th


def resnet50_shapes(batch):
  seen, out = set(), []
  out.append((batch, 224, 224, 3, 64, 7, 2, 3, 3, 112, 112))
  in_ch, hw = 64, 56
  for g, nb in enumerate([3, 4, 6, 3], start=1):
    f = 64 * 2 ** (g - 1)
    stride = 1 if g == 1 else 2
    ohw = hw // stride
    p3 = 1
    cand = [(batch, hw, hw, in_ch, 4 * f, 1, stride, 0, 0, ohw, ohw),
            (batch, hw, hw, in_ch, f, 1, 1, 0, 0, hw, hw),
            (batch, hw, hw, f, f, 3, stride, p3, p3, ohw, ohw),
            (batch, ohw, ohw, f, 4 * f, 1, 1, 0, 0, ohw, ohw),
            (batch, ohw, ohw, 4 * f, f, 1, 1, 0, 0, ohw, ohw),
            (batch, ohw, ohw, f, f, 3, 1, 1, 1, ohw, ohw)]
    for c in cand:
      if c not in seen:
        seen.add(c)
        out.append(c)
    in_ch, hw = 4 * f, ohw
  return out


def run_case(case, seed, check_bwd_call=True, check_stats=True):
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  g = torch.Generator(device=DEV).manual_seed(seed)
  x = torch.randn(N, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
  dy = torch.randn(N, Ho, Wo, Cout, generator=g, device=DEV).to(torch.bfloat16)
  w = torch.randn(k, k, Cin, Cout, generator=g, device=DEV) * (2.0 / (k * k * Cin)) ** 0.5
  add = torch.randn(N, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
  n = w.numel()
  hwio = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ohwi = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  # every second case packs through a 1-bit mask (80 % of the weights off, as in the benchmarked ERK layers): the bf16
  # shadows must be exactly bf16(mask * w) in both layouts -- pack -> conv is pinned at these shapes, not only in situ
  mask_bits = None
  wflat = w.reshape(-1).contiguous()
  if seed % 2 == 1:
    m01 = (torch.rand(n, generator=g, device=DEV) < 0.2).float()
    mask_bits = ops.mask_pack(m01)
    expect = torch.where(m01 > 0, wflat, torch.zeros_like(wflat)).to(torch.bfloat16)   # (+0 where masked, not -0)
  else:
    expect = wflat.to(torch.bfloat16)
  ops.pack_weights(wflat, mask_bits, k * k * Cin, Cout, hwio, ohwi)
  assert torch.equal(hwio.view(torch.int16), expect.view(torch.int16)), 'pack: HWIO shadow != bf16(mask * w)'
  assert torch.equal(ohwi.view(torch.int16).reshape(Cout, k * k * Cin),
                     expect.view(torch.int16).reshape(k * k * Cin, Cout).t()), 'pack: OHWI shadow != transpose of the HWIO shadow'
  wm = hwio.float().reshape(k, k, Cin, Cout)
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  has_dx = Cin % 8 == 0
  want = ('y', 'dx', 'dw') if has_dx else ('y', 'dw')
  ref = convref.conv_fp64(x, wm, dy, stride, pt, pl, Ho, Wo, want)
  ab = convref.conv_fp64(x.abs(), wm.abs(), dy.abs(), stride, pt, pl, Ho, Wo, want)
  worst = 0.0
  # forward (with the batch-norm statistics epilogue: y must not change, the partials must add up)
  y = ops.conv_fwd(d, x, ohwi)
  worst = max(worst, convref.check_close('fwd', y, ref['y'], ab['y'], 1e-5, 2.0 ** -8))
  if check_stats and Cout % 8 == 0:
    y2, part = ops.conv_fwd(d, x, ohwi, stats=True)
    assert torch.equal(y2.view(torch.int16), y.view(torch.int16)), 'fwd_stats changed y'
    if part is not None:
      yf = y.double().reshape(-1, Cout)
      s = part.double().sum(0)
      assert part.shape[0] == d._stats_parts
      tol0 = 2e-6 * float(yf.abs().sum(0).max()) + 1e-6
      assert float((s[0] - yf.sum(0)).abs().max()) <= tol0, 'statistics: sum y'
      q = (yf * yf).sum(0)
      assert float(((s[1] - q).abs() / (q.abs() + 1e-6)).max()) <= 2e-6, 'statistics: sum y^2'
      rows = 128                                     # one partial per 128 output rows, whatever the tile
      c3 = (k == 3 and stride == 1 and Cin == 64 and Cout == 64 and pt == 1 and pl == 1 and
            part.shape[0] != (yf.shape[0] + rows - 1) // rows)
      x1 = (k == 1 and stride == 1 and Cin in (64, 128, 256, 512) and
            part.shape[0] != (yf.shape[0] + rows - 1) // rows)
      stem_direct = (k == 7 and stride == 2 and Cin == 3 and Cout == 64 and pl == 3 and Ho % 16 == 0 and Wo % 16 == 0 and
                     W % 4 == 0 and ops.tune_get('stem_direct', 1) != 0)
      blk = None
      if x1:
        # rowstream.hpp: one partial per persistent workgroup of a column slice (rigl_conv2d_stats_parts says how many) -- the
        # totals above are the whole check
        assert part.shape[0] <= torch.cuda.get_device_properties(0).multi_processor_count, \
            'rowstream: more statistics parts than workgroups per slice'
      elif c3:
        # c3x3.hpp: one partial per persistent workgroup (rigl_conv2d_stats_parts says how many), each the sum over the
        # tiles that workgroup walked -- the totals above are the whole check
        th = _c3_tile_rows(H, W)
        assert part.shape[0] <= N * ((H + th - 1) // th)
      else:
        assert part.shape[0] == (yf.shape[0] + rows - 1) // rows
        t = min(part.shape[0] - 1, 1)
        if stem_direct:
          # the stem kernel's parts are halves of 16 x 16 output tiles (include/rigl_hip.h): check one in the interior
          t = min(part.shape[0] - 1, 2 * (Wo // 16 + 1) + 1)
          tile, h = t // 2, t % 2
          per = (Ho // 16) * (Wo // 16)
          ni, r = tile // per, tile % per
          oh0, ow0 = (r // (Wo // 16)) * 16 + 8 * h, (r % (Wo // 16)) * 16
          blk = y.double()[ni, oh0:oh0 + 8, ow0:ow0 + 16].reshape(-1, Cout)
        else:
          blk = yf[t * rows:(t + 1) * rows]
      if blk is not None:
        assert float((part[t, 0].double() - blk.sum(0)).abs().max()) <= 1e-5 * float(blk.abs().sum(0).max()) + 1e-6, \
            'statistics: a tile partial is not the sum of its own rows'
        assert float((part[t, 1].double() - (blk * blk).sum(0)).abs().max()) <= 1e-5 * float((blk * blk).sum(0).max()) + 1e-6, \
            'statistics: a tile partial is not the sum of squares of its own rows'
  del y
  dw = ops.conv_wgrad(d, x, dy).reshape(k, k, Cin, Cout)
  worst = max(worst, convref.check_close('wgrad', dw, ref['dw'], ab['dw'], 1e-5))
  if has_dx:
    dx = ops.conv_dgrad(d, dy, hwio)
    worst = max(worst, convref.check_close('dgrad', dx, ref['dx'], ab['dx'], 1e-5, 2.0 ** -8))
    dxa = ops.conv_dgrad(d, dy, hwio, addend=add)
    assert torch.equal(dxa.view(torch.int16), (dx + add).view(torch.int16)), 'dgrad_acc != dgrad + addend'
    del dxa
  if check_bwd_call:
    dw1 = torch.empty(n, dtype=torch.float32, device=DEV)
    dx1 = ops.conv_bwd(d, x, dy, hwio, dw1, need_dx=has_dx, addend=add if has_dx else None)
    worst = max(worst, convref.check_close('bwd.dw', dw1.reshape(k, k, Cin, Cout), ref['dw'], ab['dw'], 1e-5))
    if has_dx:
      # the one-call backward adds the addend in bf16 after rounding dgrad to bf16, like dgrad_acc
      assert torch.equal(dx1.view(torch.int16), (dx + add).view(torch.int16)), 'bwd.dx != dgrad + addend'
    # run-to-run determinism of the dense gradient (masks are a function of it)
    dw2 = torch.empty_like(dw1)
    ops.conv_bwd(d, x, dy, hwio, dw2, need_dx=has_dx, addend=add if has_dx else None)
    assert torch.equal(dw1.view(torch.int32), dw2.view(torch.int32)), 'bwd.dw not deterministic'
    if has_dx and ops.conv_bwd_takes_masked_addend(d):
      # the addend handed over unmasked with a 1-bit mask (rigl_masked_conv2d_bwd_masked): exactly dgrad + (bit ? addend : 0)
      m01 = torch.rand(add.numel(), generator=g, device=DEV) < 0.4
      abits = torch.zeros(add.numel() // 8, dtype=torch.uint8, device=DEV)
      for j in range(8):
        abits |= (m01.view(-1, 8)[:, j].to(torch.uint8) << j)
      dw5 = torch.empty_like(dw1)
      dx5 = ops.conv_bwd(d, x, dy, hwio, dw5, need_dx=True, addend=add, addend_bits=abits)
      want5 = dx + torch.where(m01.view(add.shape), add, torch.zeros_like(add))
      assert torch.equal(dx5.view(torch.int16), want5.view(torch.int16)), 'bwd_masked.dx != dgrad + masked addend'
      assert torch.equal(dw5.view(torch.int32), dw1.view(torch.int32)), 'bwd_masked.dw != bwd.dw'
      del dx5, dw5, want5, abits, m01
    if has_dx:
      # the addend as the gradient of a SUBSAMPLED view (rigl_masked_conv2d_bwd_sub): added at the pixels (2i, 2j) only.
      # That call runs on the implicit-GEMM body whatever the layer's default kernel, so its dgrad is taken from the same
      # call with a zero addend (and checked against the reference), and the addition must then be exact.
      addc = add[:, ::2, ::2, :].contiguous()
      full = torch.zeros_like(add)
      full[:, ::2, ::2, :] = addc
      dw3 = torch.empty_like(dw1)
      dx0 = ops.conv_bwd(d, x, dy, hwio, dw3, need_dx=True, addend=torch.zeros_like(addc), addend_sub=(2, 2))
      worst = max(worst, convref.check_close('bwd_sub.dgrad', dx0, ref['dx'], ab['dx'], 1e-5, 2.0 ** -8))
      worst = max(worst, convref.check_close('bwd_sub.dw', dw3.reshape(k, k, Cin, Cout), ref['dw'], ab['dw'], 1e-5))
      dx3 = ops.conv_bwd(d, x, dy, hwio, dw3, need_dx=True, addend=addc, addend_sub=(2, 2))
      assert torch.equal(dx3.view(torch.int16), (dx0 + full).view(torch.int16)), 'bwd_sub.dx != dgrad + scattered addend'
      del dx0, dx3, full, addc
    if has_dx and k == 1 and stride > 1 and pt == 0 and pl == 0 and Ho == -(-H // stride) and Wo == -(-W // stride):
      # a strided 1x1 conv: dX on its own grid only (rigl_masked_conv2d_bwd_grid) = the reference's dX at those pixels
      dw4 = torch.empty_like(dw1)
      dxg = ops.conv_bwd_grid(d, x, dy, hwio, dw4)
      worst = max(worst, convref.check_close('bwd_grid.dx', dxg, ref['dx'][:, ::stride, ::stride, :], ab['dx'][:, ::stride, ::stride, :],
                                             1e-5, 2.0 ** -8))
      worst = max(worst, convref.check_close('bwd_grid.dw', dw4.reshape(k, k, Cin, Cout), ref['dw'], ab['dw'], 1e-5))
      off = torch.ones(H, W, dtype=torch.bool, device=DEV)
      off[::stride, ::stride] = False
      assert float(ref['dx'][:, off].abs().max()) == 0.0        # (what is not materialised is exactly zero)
  torch.cuda.synchronize()
  return worst


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--set', default='small', choices=['small', 'pp', 'stem', 'c3', 'rs', 'bs', 'resnet50'])
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--only', type=int, default=-1)
  a = ap.parse_args()
  cases = {'small': SMALL_CASES, 'pp': PP_CASES, 'stem': STEM_CASES, 'c3': C3_CASES, 'rs': RS_CASES, 'bs': BS_CASES}.get(a.set) or resnet50_shapes(a.batch)
  worst, n = 0.0, 0
  for i, c in enumerate(cases):
    if a.only >= 0 and i != a.only:
      continue
    try:
      r = run_case(c, 100 + i)
    except AssertionError as e:
      print('FAIL case %d %s: %s' % (i, c, e), flush=True)
      print(json.dumps({'ok': False, 'case': i, 'shape': list(c), 'error': str(e)[:500]}))
      return 1
    print('ok case %d %s worst ratio %.3f' % (i, c, r), flush=True)
    worst, n = max(worst, r), n + 1
    torch.cuda.empty_cache()
  print(json.dumps({'ok': True, 'cases': n, 'worst': worst, 'set': a.set,
                    'env': {k: v for k, v in os.environ.items() if k.startswith('RIGL_')}}))
  return 0


if __name__ == '__main__':
  sys.exit(main())
