#!/usr/bin/env python3
"""Lane-level emulator of the tile196 conv kernels (rigl_amd/csrc/conv196.hpp) -- development tool.

There is no GPU in the build container, so the index arithmetic of the kernel (which lane fetches
which 16 bytes, where `buffer_load ... lds` puts them, the XOR swizzle on the source side and on the
fragment reads, the shifted 3x3 views of the LDS-resident input slab with their zero-row redirect,
the 32x32x16 MFMA operand / accumulator layouts, the row-block split between wave pairs and the epilogue's
row / column mapping) is restated here per wave and per lane with NumPy, and the result is compared
with a plain convolution.  What it cannot model is timing (vmcnt counts, barriers): loads are applied
in program order, which is what the counted waits + barriers guarantee on the device.

The formulas are kept textually close to the HIP source; change them together.
  python tools/emu/t196_emu.py        # runs the self-checks
"""
import numpy as np

BMV, BMC, RB = 196, 224, 7
SLAB_BYTES = 24576          # 3x3 slab buffer: up to 307 padded pixel entries of 80 bytes
PITCH = 80
OOB = 1 << 31


def lanes():
  return np.arange(64)


class Lds:
  """LDS as an array of bf16 *values* (float32 here), addressed in bytes."""

  def __init__(self, nbytes):
    self.v = np.full(nbytes // 2, np.nan, dtype=np.float32)   # NaN = never written

  def dma(self, src_flat, base_byte, voff_bytes):
    """one wave instruction `buffer_load_dwordx4 ... lds`: lane l writes 16 B at base + 16 l."""
    for l in range(64):
      d = (base_byte + 16 * l) // 2
      o = int(voff_bytes[l])
      if o >= OOB or o + 16 > src_flat.size * 2:
        self.v[d:d + 8] = 0.0
      else:
        assert o % 16 == 0 or o % 2 == 0
        self.v[d:d + 8] = src_flat[o // 2:o // 2 + 8]

  def read16(self, addr_bytes):
    """ds_read_b128 per lane -> [64, 8] values"""
    out = np.empty((64, 8), np.float32)
    for l in range(64):
      a = int(addr_bytes[l])
      assert a % 16 == 0
      out[l] = self.v[a // 2:a // 2 + 8]
    return out


def mfma_32x32x16(aop, bop, acc):
  """acc[lane][e] += ...  with aop rows i = l&31 / k = 8(l>>5).., bop cols j = l&31 / same k,
  D lane l: col j = l&31, row i = (e&3) + 8(e>>2) + 4(l>>5)."""
  A = np.zeros((32, 16), np.float64)
  B = np.zeros((16, 32), np.float64)
  for l in range(64):
    A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = aop[l]
    B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = bop[l]
  D = A @ B
  for l in range(64):
    for e in range(16):
      acc[l, e] += D[(e & 3) + 8 * (e >> 2) + 4 * (l >> 5), l & 31]


def run_tile(mode, k3, BN, A, B, M, N, Cred, H, W, b_row_stride, b_tap_stride, tile, tiles_n):
  """Emulates one workgroup.  A: flat activations [M * Cred]; B: flat weights.  Returns C tile [224, BN]."""
  WN = BN // 32
  RBW = RB if WN == 4 else 4               # row blocks per wave; BN = 64: blocks 4 wm .. 4 wm + 3 (block 7 = zeros)
  LB = BN // 64
  tile_m, n0 = tile // tiles_n, (tile % tiles_n) * BN
  m0 = tile_m * BMV
  lane = lanes()
  l4, slot = lane >> 2, lane & 3
  dchunk = slot ^ ((lane >> 4) & 3)
  hi = lane >> 5
  KT_taps = 9 if k3 else 1
  CB = Cred // 32
  if k3:
    SLAB_B = SLAB_BYTES
    B_STAGE = BN * 64
    OFF_B = 2 * SLAB_B
    lds = Lds(OFF_B + 3 * B_STAGE)
    # zero-padded slab geometry: image rows are stored as W + 2 entries (a zero pixel at either end), one all-zero
    # row separates consecutive images; the slab starts one padded row above the tile's first image row
    Nimg = M // (H * W)
    gr0 = m0 // W
    pr_base = gr0 + gr0 // H - 1
  else:
    A_ST = 256 * 64
    STAGE = A_ST + BN * 64
    lds = Lds(3 * STAGE)

  # ---- per-lane DMA offsets ---------------------------------------------------
  def voff_b(wave, j):
    row = (j * 4 + wave) * 16 + l4
    n = n0 + row
    ok = (row < BN) & (n < N)
    return np.where(ok, (n * b_row_stride + dchunk * 8) * 2, OOB), (j * 4 + wave) * 1024

  def voff_a(wave, j):                      # 1x1: A rows straight into the ring
    row = (j * 4 + wave) * 16 + l4
    m = m0 + row
    ok = (row < BMV) & (m < M)
    return np.where(ok, (m * Cred + dchunk * 8) * 2, OOB), (j * 4 + wave) * 1024

  def voff_s(wave, j):                      # 3x3: padded slab, 80-byte entries (lane 4 of every 5 writes the unused tail)
    o = (j * 4 + wave) * 1024 + lane * 16
    e, col = o // PITCH, (o % PITCH) // 16
    pr = pr_base + e // (W + 2)
    w_ = e % (W + 2) - 1
    n_, h_ = pr // (H + 1), pr % (H + 1)
    ok = (col < 4) & (w_ >= 0) & (w_ < W) & (h_ < H) & (pr >= 0) & (n_ < Nimg)
    pix = (n_ * H + h_) * W + w_
    return np.where(ok, (pix * Cred + col * 8) * 2, OOB), (j * 4 + wave) * 1024

  def issue_b(kt):
    cb, tap = (kt // 9, kt % 9) if k3 else (kt, 0)
    st = kt % 3
    for wave in range(4):
      for j in range(LB):
        vo, dst = voff_b(wave, j)
        add = (tap * b_tap_stride + cb * 32) * 2
        base = (OFF_B + st * B_STAGE) if k3 else (st * STAGE + A_ST)
        lds.dma(B, base + dst, np.where(vo >= OOB, OOB, vo + add))

  def issue_a(kt):                          # 1x1
    st = kt % 3
    for wave in range(4):
      for j in range(4):
        vo, dst = voff_a(wave, j)
        lds.dma(A, st * STAGE + dst, np.where(vo >= OOB, OOB, vo + kt * 64))

  def issue_slab(cb):
    for wave in range(4):
      for j in range(6):
        vo, dst = voff_s(wave, j)
        lds.dma(A, (cb & 1) * SLAB_B + dst, np.where(vo >= OOB, OOB, vo + cb * 64))

  # ---- per-lane fragment state (per wave: its own row blocks) ----------------------
  blk0 = [(wave // WN) * 4 for wave in range(4)]
  lr = [[(blk0[wave] + i) * 32 + (lane & 31) for i in range(RBW)] for wave in range(4)]
  if k3:
    a_base = []
    for wave in range(4):
      aw = []
      for i in range(RBW):
        m = m0 + np.minimum(lr[wave][i], BMV - 1)      # rows beyond 196 repeat row 195 (computed, never stored)
        gr = m // W
        P = (gr + gr // H - pr_base) * (W + 2) + m % W + 1
        aw.append(P * PITCH + hi * 16)
      a_base.append(aw)
  b_rd = [None] * 4
  for wave in range(4):
    wn = wave % WN
    row = wn * 32 + (lane & 31)
    b_rd[wave] = row * 64 + (((hi ^ (row >> 2)) & 3) << 4)

  acc = np.zeros((4, RBW, 64, 16), np.float64)
  KT = CB * KT_taps

  # ---- main loop, in the program order of the device code (software pipeline across the barrier):
  #   read LAST K-step of tile kt | MFMA batch requested earlier | (barrier) | DMA tile kt+3 -> stage kt%3 [+ slab] |
  #   read first K-step of tile kt+1 | MFMA on the last K-step of tile kt
  def tile_ct(kt):
    return (kt // 9, kt % 9) if k3 else (kt, 0)

  def a_offsets(kt, wave):
    cb, tap = tile_ct(kt)
    st = kt % 3
    if k3:
      r, s = tap // 3, tap % 3
      dh, dw = (r - 1, s - 1) if mode == 0 else (1 - r, 1 - s)
      tapoff = (cb & 1) * SLAB_B + (dh * (W + 2) + dw) * PITCH
      return [a_base[wave][i] + tapoff for i in range(RBW)], OFF_B + st * B_STAGE
    return [st * STAGE + lr[wave][i] * 64 + (((hi ^ (lr[wave][i] >> 2)) & 3) << 4) for i in range(RBW)], st * STAGE + A_ST

  def read_frags(kt, kx):
    out = []
    for wave in range(4):
      a_addr, bbase = a_offsets(kt, wave)
      bfr = lds.read16(bbase + (b_rd[wave] ^ kx))
      afs = [lds.read16((a_addr[i] + kx) if k3 else (a_addr[i] ^ kx)) for i in range(RBW)]
      assert not np.isnan(bfr).any() and not any(np.isnan(a).any() for a in afs), 'read of never-written LDS'
      out.append((bfr, afs))
    return out

  def mfma_batch(frags):
    for wave in range(4):
      bfr, afs = frags[wave]
      for i in range(RBW):
        mfma_32x32x16(bfr, afs[i], acc[wave, i])

  def issue_tile(kt):
    if not k3:
      issue_a(kt)
    issue_b(kt)

  if k3:
    issue_slab(0)
  for t in range(min(3, KT)):
    issue_tile(t)
  F0 = read_frags(0, 0)
  for kt in range(KT):
    cb, tap = tile_ct(kt)
    F1 = read_frags(kt, 32)
    mfma_batch(F0)
    if kt + 1 < KT:
      if kt + 3 < KT:
        issue_tile(kt + 3)
      if k3 and tap == 0 and cb + 1 < CB:
        issue_slab(cb + 1)
      F0 = read_frags(kt + 1, 0)
    mfma_batch(F1)

  # ---- epilogue mapping -----------------------------------------------------------
  Cs = np.zeros((BMC, BN), np.float64)
  for wave in range(4):
    wn = wave % WN
    for i in range(RBW):
      if blk0[wave] + i >= RB:
        assert k3 or np.all(acc[wave, i] == 0), '1x1: the eighth row block multiplies zeros'
        continue
      for q in range(4):
        for t in range(4):
          row = (blk0[wave] + i) * 32 + (lane & 31)
          col = wn * 32 + 8 * q + 4 * (lane >> 5) + t
          Cs[row, col] = acc[wave, i, lane, 4 * q + t]
  return m0, n0, Cs


def conv_ref(x, w_hwio, mode):
  """x [N,H,W,C] ; w [3|1,3|1,ci,co]; SAME stride 1.  mode 0: y = conv(x); mode 1: dx = conv_backprop_input(dy=x)."""
  N, H, W, C = x.shape
  kh, kw, ci, co = w_hwio.shape
  p = kh // 2
  if mode == 0:
    y = np.zeros((N, H, W, co))
    for r in range(kh):
      for s in range(kw):
        for h in range(H):
          for ww in range(W):
            hh, w2 = h + r - p, ww + s - p
            if 0 <= hh < H and 0 <= w2 < W:
              y[:, h, ww, :] += x[:, hh, w2, :] @ w_hwio[r, s]
    return y
  dx = np.zeros((N, H, W, ci))
  for r in range(kh):
    for s in range(kw):
      for h in range(H):
        for ww in range(W):
          hh, w2 = h + p - r, ww + p - s
          if 0 <= hh < H and 0 <= w2 < W:
            dx[:, h, ww, :] += x[:, hh, w2, :] @ w_hwio[r, s].T
  return dx


def check(mode, k, BN, Nimg, H, W, Cin, Cout, seed=0):
  rs = np.random.RandomState(seed)
  k3 = k == 3
  w = rs.randint(-3, 4, size=(k, k, Cin, Cout)).astype(np.float32)
  if mode == 0:
    x = rs.randint(-3, 4, size=(Nimg, H, W, Cin)).astype(np.float32)
    ohwi = np.ascontiguousarray(w.transpose(3, 0, 1, 2)).reshape(-1)     # [co][r][s][ci]
    A, B, Cred, N = x.reshape(-1), ohwi, Cin, Cout
    brs, bts = k * k * Cin, Cin
  else:
    x = rs.randint(-3, 4, size=(Nimg, H, W, Cout)).astype(np.float32)   # dy
    hwio = w.reshape(-1)                                                # [(r,s)][ci][co]
    A, B, Cred, N = x.reshape(-1), hwio, Cout, Cin
    brs, bts = Cout, Cin * Cout
  M = Nimg * H * W
  assert M % BMV == 0 and Cred % 32 == 0
  ref = conv_ref(x.astype(np.float64), w.astype(np.float64), mode).reshape(M, N)
  tiles_n = (N + BN - 1) // BN
  out = np.full((M, N), np.nan)
  for tile in range((M // BMV) * tiles_n):
    m0, n0, Cs = run_tile(mode, k3, BN, A, B, M, N, Cred, H, W, brs, bts, tile, tiles_n)
    nn = min(BN, N - n0)
    out[m0:m0 + BMV, n0:n0 + nn] = Cs[:BMV, :nn]
    assert k3 or np.all(Cs[BMV:] == 0), '1x1: rows beyond the valid 196 are exact zeros'
  assert np.array_equal(out, ref), 'mismatch mode=%d k=%d BN=%d: max err %g' % (mode, k, BN, np.nanmax(np.abs(out - ref)))
  print('ok mode=%d k=%d BN=%d  N=%d %dx%d cin=%d cout=%d' % (mode, k, BN, Nimg, H, W, Cin, Cout))


if __name__ == '__main__':
  check(0, 1, 128, 4, 7, 7, 32, 128)
  check(0, 1, 64, 4, 7, 7, 64, 64)
  check(1, 1, 64, 4, 7, 7, 64, 96)          # ragged N (96 = 64 + 32): columns beyond N are never stored
  check(0, 3, 128, 4, 7, 7, 32, 128)        # one tile spanning 4 images: every image edge inside the tile
  check(1, 3, 64, 4, 7, 7, 64, 64)
  check(0, 3, 64, 2, 14, 14, 64, 64)        # two tiles, one image each
  check(1, 3, 128, 1, 28, 28, 32, 128)      # four row bands of one image: real halo rows across tiles
  check(0, 3, 64, 1, 14, 28, 32, 64)        # H != W
  print('all tile196 emulator checks passed')
