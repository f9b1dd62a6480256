#!/usr/bin/env python3
"""Lane-level emulator of the all-taps 3x3 weight-gradient kernel (rigl_amd/csrc/wgrad9.hpp) -- development tool.

Restates per wave and per lane: the DMA lane -> (pixel row, channel chunk) mapping with its source-side swizzle,
the circular X ring, `ds_read_b64_tr_b16` (per 16-lane group: lane j supplies 4 consecutive channels of pixel j/4
and receives channel j for the group's 4 pixels), the tap shift + per-pixel validity masks + zero-row redirect,
the MFMA operand / accumulator layouts and the slab store.  Loads are applied in program order (what the counted
vmcnt + barriers guarantee).  Keep the formulas textually close to the HIP source.
  python tools/emu/w9_emu.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from t196_emu import Lds, OOB  # noqa: E402

XROWS = 256
XRING_B = XROWS * 128
DY_ST = 32 * 128
OFF_DY = XRING_B
OFF_ZERO = OFF_DY + 3 * DY_ST
SMEM = OFF_ZERO + 64


def tr_read(lds, addr):
  """ds_read_b64_tr_b16: addr[64] byte addresses (8-byte aligned) -> [64, 4] values."""
  sup = np.empty((64, 4), np.float32)
  for l in range(64):
    a = int(addr[l])
    assert a % 8 == 0
    sup[l] = lds.v[a // 2:a // 2 + 4]
  out = np.empty((64, 4), np.float32)
  for l in range(64):
    g, j = l >> 4, l & 15
    for i in range(4):
      out[l, i] = sup[16 * g + 4 * i + j // 4, j % 4]
  return out


def tr_pair(lds, a0, a1):
  return np.concatenate([tr_read(lds, a0), tr_read(lds, a1)], axis=1)     # [64, 8]


def mfma(aop, bop, acc):
  """A rows i = l&31 (k = 8(l>>5)..), B cols j = l&31; D lane l: col l&31, row (e&3)+8(e>>2)+4(l>>5)."""
  A = np.zeros((32, 16))
  B = np.zeros((16, 32))
  for l in range(64):
    A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = aop[l]
    B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = bop[l]
  D = A @ B
  for l in range(64):
    for e in range(16):
      acc[l, e] += D[(e & 3) + 8 * (e >> 2) + 4 * (l >> 5), l & 31]


def run_wg(X, DY, M, Cin, Cout, H, W, tci, tco, split, kt_per_split, hb, out):
  lane = np.arange(64)
  ci0, co0 = tci * 64, tco * 64
  p_begin = split * kt_per_split * 32
  p_end = min(p_begin + kt_per_split * 32, M)
  KT = (p_end - p_begin + 31) >> 5 if p_end > p_begin else 0
  HB = hb
  xb0 = p_begin - 32 * HB
  lds = Lds(SMEM)
  lds.v[OFF_ZERO // 2:OFF_ZERO // 2 + 32] = 0.0
  l8, slot = lane >> 3, lane & 7
  dsw = ((l8 >> 1) & 1) << 2
  Xf, Yf = X.reshape(-1), DY.reshape(-1)

  def issue_x(c):
    for wave in range(4):
      row_w = wave * 8 + l8
      pix = xb0 + c * 32 + row_w
      x_lane = row_w * Cin + ci0 + ((slot ^ dsw) << 3)
      off = np.where((pix >= 0) & (pix < M), ((xb0 + c * 32) * Cin + x_lane) * 2, OOB)
      lds.dma(Xf, ((c * 32 + wave * 8) & (XROWS - 1)) * 128, off)

  def issue_y(t, st):
    for wave in range(4):
      row_w = wave * 8 + l8
      pix = p_begin + t * 32 + row_w
      y_lane = row_w * Cout + co0 + ((slot ^ dsw) << 3)
      off = np.where(pix < p_end, ((p_begin + t * 32) * Cout + y_lane) * 2, OOB)
      lds.dma(Yf, OFF_DY + st * DY_ST + wave * 1024, off)

  g, j = lane >> 4, lane & 15
  prow = 8 * (g >> 1) + (j >> 2)
  sub = (j & 1) * 8

  def mask(p):
    w_ = p % W
    h_ = (p // W) % H
    m = np.full(64, 0x1FF, np.int64)
    m = np.where(h_ == 0, m & ~0x007, m)
    m = np.where(h_ == H - 1, m & ~0x1C0, m)
    m = np.where(w_ == 0, m & ~0x049, m)
    m = np.where(w_ == W - 1, m & ~0x124, m)
    return m

  def read(t, st, ks, wave):
    wm, wn = wave >> 1, wave & 1
    chunkA = wm * 4 + 2 * (g & 1) + ((j >> 1) & 1)
    chunkB = wn * 4 + 2 * (g & 1) + ((j >> 1) & 1)
    b_rd = prow * 128 + ((chunkB ^ (((prow >> 1) & 1) << 2)) << 4) + sub
    Bs = OFF_DY + st * DY_ST + ks * 2048
    fb = tr_pair(lds, Bs + b_rd, Bs + b_rd + 512)
    p0 = p_begin + t * 32 + ks * 16 + prow
    mlo, mhi = mask(p0), mask(p0 + 4)
    q0 = t * 32 + 32 * HB + ks * 16 + prow
    fa = []
    for tp in range(9):
      sh = (tp // 3 - 1) * W + (tp % 3 - 1)
      r0, r1 = (q0 + sh) & (XROWS - 1), (q0 + 4 + sh) & (XROWS - 1)
      a0 = r0 * 128 + ((chunkA ^ (((r0 >> 1) & 1) << 2)) << 4) + sub
      a1 = r1 * 128 + ((chunkA ^ (((r1 >> 1) & 1) << 2)) << 4) + sub
      a0 = np.where((mlo >> tp) & 1, a0, OFF_ZERO + sub)
      a1 = np.where((mhi >> tp) & 1, a1, OFF_ZERO + sub)
      f = tr_pair(lds, a0, a1)
      assert not np.isnan(f).any(), 'read of never-written LDS (tap %d)' % tp
      fa.append(f)
    assert not np.isnan(fb).any()
    return fb, fa

  acc = np.zeros((4, 9, 64, 16))

  def mfma_batch(F):
    for wave in range(4):
      fb, fa = F[wave]
      for tp in range(9):
        mfma(fa[tp], fb, acc[wave, tp])

  if KT > 0:
    for c in range(2 * HB):
      issue_x(c)
    for t in range(min(3, KT)):
      issue_y(t, t)
      issue_x(2 * HB + t)
    st = 0
    F0 = [read(0, 0, 0, w) for w in range(4)]
    for kt in range(KT):
      F1 = [read(kt, st, 1, w) for w in range(4)]
      mfma_batch(F0)
      if kt + 1 < KT:
        if kt + 3 < KT:
          issue_y(kt + 3, st)
          issue_x(kt + 3 + 2 * HB)
        st = 0 if st == 2 else st + 1
        F0 = [read(kt + 1, st, 0, w) for w in range(4)]
      mfma_batch(F1)
  for wave in range(4):
    wm, wn = wave >> 1, wave & 1
    for tp in range(9):
      for l in range(64):
        co = co0 + wn * 32 + (l & 31)
        for e in range(16):
          ci = ci0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)
          out[split, tp, ci, co] = acc[wave, tp, l, e]


def check(Nimg, H, W, Cin, Cout, splits_target, seed=0):
  rs = np.random.RandomState(seed)
  M = Nimg * H * W
  X = rs.randint(-3, 4, size=(M, Cin)).astype(np.float32)
  DY = rs.randint(-3, 4, size=(M, Cout)).astype(np.float32)
  # reference dW[r][s][ci][co]
  x4, d4 = X.reshape(Nimg, H, W, Cin), DY.reshape(Nimg, H, W, Cout)
  ref = np.zeros((9, Cin, Cout))
  for r in range(3):
    for s in range(3):
      for h in range(H):
        for w in range(W):
          hh, ww = h + r - 1, w + s - 1
          if 0 <= hh < H and 0 <= ww < W:
            ref[r * 3 + s] += x4[:, hh, ww, :].T @ d4[:, h, w, :]
  tiles_ci, tiles_co = Cin // 64, Cout // 64
  kt_all = (M + 31) // 32
  splits = max(1, min(splits_target, kt_all))
  kt_per = (kt_all + splits - 1) // splits
  splits = (kt_all + kt_per - 1) // kt_per
  hb = (W + 1 + 31) // 32
  out = np.full((splits, 9, Cin, Cout), np.nan)
  for sp in range(splits):
    for tci in range(tiles_ci):
      for tco in range(tiles_co):
        run_wg(X, DY, M, Cin, Cout, H, W, tci, tco, sp, kt_per, hb, out)
  got = out.sum(0)
  assert np.array_equal(got, ref), 'mismatch: max err %g' % np.nanmax(np.abs(got - ref))
  print('ok N=%d %dx%d cin=%d cout=%d splits=%d kt/split=%d hb=%d' % (Nimg, H, W, Cin, Cout, splits, kt_per, hb))


if __name__ == '__main__':
  check(2, 7, 7, 64, 64, 1)          # M = 98: ragged last K-tile, images inside a K-step
  check(3, 7, 7, 64, 128, 2)         # two splits: the split boundary cuts an image; two co tiles
  check(1, 14, 14, 128, 64, 3)       # two ci tiles, three splits
  check(1, 12, 40, 64, 64, 2)        # W + 1 > 32: two halo chunks, H != W
  print('all wgrad9 emulator checks passed')
