#!/usr/bin/env python3
"""Lane-level emulator of the 3x3 stride-1 conv kernel with an LDS-resident zero-padded slab whose MFMA rows
enumerate PADDED ENTRIES (rigl_amd/csrc/conv3x3.hpp) -- development tool, see t196_emu.py for the conventions.

Layout: pixel (n, h, w) lives at padded entry  PG = (n (H+1) + h) RP + w + 1  with RP = W + 1: entry (row, 0) is a
zero pixel that serves as w = -1 of its row and w = W of the row above; one all-zero row follows every image.  A tile
is RT consecutive padded rows (whole image rows, or whole images incl. their gap rows); its MFMA row r IS the entry
tile_P0 + r, so a 16-lane group always reads 16 consecutive 80-byte entries (bank-conflict-free) and tap (dh, dw) is
the constant entry offset dh RP + dw.  Rows that are pads / gap rows / beyond RT RP are computed and discarded.
Waves: WN x WM x 2 (K split: wave pair wk multiplies K-step wk of every 32-channel K-tile; pair sums added at the end).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from t196_emu import Lds, OOB, mfma_32x32x16  # noqa: E402

PITCH = 80
SLAB_BYTES = 24576


def plan(H, W, nimg, BMC):
  RP = W + 1
  if (H + 1) * RP <= BMC:
    k = BMC // ((H + 1) * RP)
    while k > 1 and nimg % k:
      k -= 1
    RT = k * (H + 1)
    tiles = nimg // k
    step_rows = RT                      # padded rows per tile
  else:
    RT = max(d for d in range(1, H + 1) if H % d == 0 and d * RP <= BMC)
    tiles = nimg * (H // RT)
    step_rows = None                    # tiles inside an image: padded row of tile t = n (H+1) + (t % (H/RT)) RT
  return RP, RT, tiles, step_rows


def tile_prow0(t, H, RT, step_rows):
  if step_rows is not None:
    return t * step_rows
  per = H // RT
  return (t // per) * (H + 1) + (t % per) * RT


def run_tile(mode, BN, RBT, A, B, nimg, H, W, Cred, N, b_row_stride, b_tap_stride, tile, tiles_n, out):
  BMC = RBT * 32
  RP, RT, tiles_m, step_rows = plan(H, W, nimg, BMC)
  WN = BN // 64
  WM = 1 if BN == 128 else 2
  RB = (RBT + WM - 1) // WM              # row blocks per wave (7 or 8 blocks over 2 waves -> 4, a block index >= RBT is all discarded rows)
  LB = BN // 64
  tile_m, n0 = tile // tiles_n, (tile % tiles_n) * BN
  prow0 = tile_prow0(tile_m, H, RT, step_rows)
  P0 = prow0 * RP                        # global padded entry of the tile's row 0
  S0 = P0 - RP - 1                       # ... of slab entry 0
  lane = np.arange(64)
  l4 = lane >> 2
  dchunk = (lane & 3) ^ ((lane >> 4) & 3)
  hi = lane >> 5
  CB = Cred // 32
  KT = CB * 9
  SLAB_B = SLAB_BYTES
  B_STAGE = BN * 64
  OFF_B = 2 * SLAB_B
  lds = Lds(OFF_B + 3 * B_STAGE)
  LA = SLAB_B // 4096

  def voff_b(wave, j):
    row = (j * 4 + wave) * 16 + l4
    n = n0 + row
    ok = (row < BN) & (n < N)
    return np.where(ok, (n * b_row_stride + dchunk * 8) * 2, OOB), (j * 4 + wave) * 1024

  def voff_s(wave, j):
    o = (j * 4 + wave) * 1024 + lane * 16
    e, col = o // PITCH, (o % PITCH) // 16
    pg = S0 + e
    prow, c = pg // RP, pg % RP           # floor semantics (pg may be negative for the first tile)
    n_, h_ = prow // (H + 1), prow % (H + 1)
    ok = (col < 4) & (c >= 1) & (h_ < H) & (pg >= 0) & (n_ < nimg)
    pix = (n_ * H + h_) * W + c - 1
    return np.where(ok, (pix * Cred + col * 8) * 2, OOB), (j * 4 + wave) * 1024

  def issue_b(kt):
    cb, tap = kt // 9, kt % 9
    st = kt % 3
    for wave in range(4):
      for j in range(LB):
        vo, dst = voff_b(wave, j)
        add = (tap * b_tap_stride + cb * 32) * 2
        lds.dma(B, OFF_B + st * B_STAGE + dst, np.where(vo >= OOB, OOB, vo + add))

  def issue_slab(cb):
    for wave in range(4):
      for j in range(LA):
        vo, dst = voff_s(wave, j)
        lds.dma(A, (cb & 1) * SLAB_B + dst, np.where(vo >= OOB, OOB, vo + cb * 64))

  def wave_ids(wave):
    wk = wave & 1
    rest = wave >> 1
    return rest % WN, rest // WN, wk      # wn, wm, wk   (WN * WM == 2)

  lr0 = lane & 31

  def read_frags(kt, wave):
    wn, wm, wk = wave_ids(wave)
    cb, tap = kt // 9, kt % 9
    st = kt % 3
    r, s = tap // 3, tap % 3
    dh, dw = (r - 1, s - 1) if mode == 0 else (1 - r, 1 - s)
    tapoff = (cb & 1) * SLAB_B + (dh * RP + dw) * PITCH
    kx = wk * 32
    a_base = (RP + 1 + wm * RB * 32 + lr0) * PITCH + hi * 16
    afs = [lds.read16(a_base + i * 32 * PITCH + tapoff + kx) for i in range(RB)]
    bfs = []
    for jb in range(2):
      row = wn * 64 + jb * 32 + lr0
      b_rd = row * 64 + (((hi ^ (row >> 2)) & 3) << 4)
      bfs.append(lds.read16(OFF_B + st * B_STAGE + (b_rd ^ kx)))
    # dummy rows may read stale / never-written LDS: they are discarded, so NaN there is tolerated; check valid rows only
    return afs, bfs

  acc = np.zeros((4, RB, 2, 64, 16), np.float64)

  def mfma_batch(F, wave):
    afs, bfs = F
    for i in range(RB):
      for jb in range(2):
        a = np.nan_to_num(afs[i], nan=1e30)      # garbage stays garbage (and must never reach a stored row)
        mfma_32x32x16(bfs[jb], a, acc[wave, i, jb])

  issue_slab(0)
  for t in range(3):
    issue_b(t)
  F = [read_frags(0, w) for w in range(4)]
  for kt in range(KT):
    cb, tap = kt // 9, kt % 9
    FN = None
    if kt + 1 < KT:
      if kt + 3 < KT:
        issue_b(kt + 3)
      if tap == 0 and cb + 1 < CB:
        issue_slab(cb + 1)
      FN = [read_frags(kt + 1, w) for w in range(4)]
    for w in range(4):
      mfma_batch(F[w], w)
    F = FN

  # K split: pair 1 hands over, pair 0 adds (fixed order), then the epilogue stages / stores valid rows
  for wave in range(4):
    wn, wm, wk = wave_ids(wave)
    if wk == 0:
      acc[wave] += acc[wave + 1]
  per = None if step_rows is not None else H // RT
  for wave in range(4):
    wn, wm, wk = wave_ids(wave)
    if wk:
      continue
    for i in range(RB):
      for jb in range(2):
        for l in range(64):
          rrow = (wm * RB + i) * 32 + (l & 31)
          pr, c = rrow // RP, rrow % RP
          if pr >= RT or c == 0:
            continue
          prow = prow0 + pr
          n_, h_ = prow // (H + 1), prow % (H + 1)
          if h_ >= H or n_ >= nimg:
            continue
          m = (n_ * H + h_) * W + c - 1
          for e in range(16):
            col = n0 + wn * 64 + jb * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)
            if col < N:
              assert np.isnan(out[m, col]), 'output written twice'
              out[m, col] = acc[wave, i, jb, l, e]
  return tiles_m


def conv_ref(x, w_hwio, mode):
  N, H, W, C = x.shape
  co = w_hwio.shape[3] if mode == 0 else w_hwio.shape[2]
  y = np.zeros((N, H, W, co))
  for r in range(3):
    for s in range(3):
      for h in range(H):
        for ww in range(W):
          hh, w2 = (h + r - 1, ww + s - 1) if mode == 0 else (h + 1 - r, ww + 1 - s)
          if 0 <= hh < H and 0 <= w2 < W:
            y[:, h, ww, :] += x[:, hh, w2, :] @ (w_hwio[r, s] if mode == 0 else w_hwio[r, s].T)
  return y


def check(mode, BN, RBT, nimg, H, W, Cin, Cout, seed=0):
  rs = np.random.RandomState(seed)
  w = rs.randint(-3, 4, size=(3, 3, Cin, Cout)).astype(np.float32)
  if mode == 0:
    x = rs.randint(-3, 4, size=(nimg, H, W, Cin)).astype(np.float32)
    A, B, Cred, N = x.reshape(-1), np.ascontiguousarray(w.transpose(3, 0, 1, 2)).reshape(-1), Cin, Cout
    brs, bts = 9 * Cin, Cin
  else:
    x = rs.randint(-3, 4, size=(nimg, H, W, Cout)).astype(np.float32)
    A, B, Cred, N = x.reshape(-1), w.reshape(-1), Cout, Cin
    brs, bts = Cout, Cin * Cout
  M = nimg * H * W
  ref = conv_ref(x.astype(np.float64), w.astype(np.float64), mode).reshape(M, N)
  out = np.full((M, N), np.nan)
  tiles_n = (N + BN - 1) // BN
  RP, RT, tiles_m, _ = plan(H, W, nimg, RBT * 32)
  for tile in range(tiles_m * tiles_n):
    run_tile(mode, BN, RBT, A, B, nimg, H, W, Cred, N, brs, bts, tile, tiles_n, out)
  assert not np.isnan(out).any(), 'rows never written: %d' % int(np.isnan(out).sum())
  assert np.array_equal(out, ref), 'mismatch: max err %g' % np.max(np.abs(out - ref))
  print('ok mode=%d BN=%d RBT=%d  n=%d %dx%d cin=%d cout=%d  (RP=%d RT=%d tiles_m=%d)' % (mode, BN, RBT, nimg, H, W, Cin, Cout, RP, RT, tiles_m))


if __name__ == '__main__':
  check(0, 128, 7, 2, 14, 14, 32, 128)       # one image per tile (210 of 224 rows)
  check(1, 64, 8, 8, 7, 7, 64, 64)           # four 7x7 images per 256-row tile, gap rows inside the tile
  check(0, 64, 7, 1, 28, 28, 32, 96)         # four 7-row bands of one image, ragged N
  check(1, 128, 7, 3, 14, 14, 64, 128)       # dgrad, two channel blocks (slab double buffer)
  check(0, 128, 8, 2, 7, 7, 32, 128)         # 2 images: k falls back to a divisor of the batch
  print('all c3 emulator checks passed')
