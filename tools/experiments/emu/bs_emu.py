#!/usr/bin/env python3
"""Lane-level emulator of the channel-sliced single-pass 1x1 backward (rigl_amd/csrc/bwdslice.hpp) -- development tool.

No GPU in the build container: the index arithmetic of the kernel (workgroup -> slice / row group, which lane's LDS-DMA
piece lands where with the source-side XOR swizzle, the row-major ds_read_b128 fragments of the dgrad product, the
ds_read_b64_tr_b16 fragments of the weight-gradient product, MFMA operand / accumulator layouts, the dX staging tile and
its flush, the slab rows) is restated per wave and per lane with NumPy and compared with plain matrix products on
integer-valued data (every sum exact).  Timing (vmcnt counts, barriers) is not modelled: loads apply in program order.
The formulas are kept textually close to the HIP source; change them together.
  python tools/experiments/emu/bs_emu.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from t196_emu import Lds, OOB, mfma_32x32x16  # noqa: E402

SC, PX = 128, 32


def bs_swz(row):
  return ((row & 3) << 2) | ((row >> 2) & 3)


def read_tr_pair(lds, a0, a1):
  """two ds_read_b64_tr_b16 -> [64, 8]: per 16-lane group, out[j][e] = in[lane 4e + j/4][j%4] (tools/probes/tr_read_probe.hip)"""
  out = np.empty((64, 8), np.float32)
  for half, addr in enumerate((a0, a1)):
    src = np.empty((64, 4), np.float32)
    for l in range(64):
      a = int(addr[l])
      assert a % 8 == 0
      src[l] = lds.v[a // 2:a // 2 + 4]
    for l in range(64):
      g, j = l & ~15, l & 15
      for e in range(4):
        out[l, half * 4 + e] = src[g + 4 * e + j // 4, j % 4]
  return out


def run_workgroup(block, X, DY, W, ADD, DX, SLAB, M, CI, CO, slices, G, NST, do_w=True):
  YROWB, XROWB = CO * 2, SC * 2
  Y_BYTES, X_BYTES, A_BYTES = PX * YROWB, (PX * XROWB if do_w else 0), PX * XROWB
  STAGE = Y_BYTES + X_BYTES + A_BYTES
  DXROWB = XROWB + 8
  DX_BYTES = PX * DXROWB
  YPW = Y_BYTES // 1024 // 4
  KS, NFO = CO // 16, CO // 32
  TI, TO = (2 if NFO >= 8 else 1), 2
  lds = Lds(NST * STAGE + 2 * DX_BYTES)
  dxs = NST * STAGE
  lane = np.arange(64)
  hi, r31 = lane >> 5, lane & 31
  xcd, idx = block & 7, block >> 3
  slice_, g = idx % slices, xcd + 8 * (idx // slices)
  KT_all = (M + PX - 1) // PX
  KT = (KT_all - g + G - 1) // G if g < KT_all else 0
  Xf, Yf, Wf = X.reshape(-1), DY.reshape(-1), W.reshape(-1)
  Af = ADD.reshape(-1) if ADD is not None else None

  def issue(kt, stage, par):
    """the four waves of half `par` bring in tile kt"""
    assert ((kt & 1) ^ (NST & 1)) == par, "tile t is issued by half (t + NST) & 1: the half not multiplying NST - 1 iterations earlier"
    p0 = (g + kt * G) * PX
    for cf in range(4):
      for q in range(YPW):
        i = q * 4 + cf
        row = i * (1024 // YROWB) + lane // (YROWB // 16)
        slot = lane % (YROWB // 16)
        col = (slot ^ bs_swz(row)) * 8
        p = p0 + row
        off = np.where(p < M, (p * CO + col) * 2, OOB)
        lds.dma(Yf, stage * STAGE + (q * 4 + cf) * 1024, off)
      for q in range(2):
        i = q * 4 + cf
        row = i * (1024 // XROWB) + lane // (XROWB // 16)
        col = slice_ * SC + (((lane % (XROWB // 16)) ^ bs_swz(row)) * 8)
        p = p0 + row
        off = np.where(p < M, (p * CI + col) * 2, OOB)
        if do_w:
          lds.dma(Xf, stage * STAGE + Y_BYTES + (q * 4 + cf) * 1024, off)
        if Af is not None:
          lds.dma(Af, stage * STAGE + Y_BYTES + X_BYTES + (q * 4 + cf) * 1024, off)

  # W fragments
  wfr = {}
  for wave in range(8):
    cf = wave & 3
    for ks in range(KS):
      ci = slice_ * SC + cf * 32 + r31
      co = ks * 16 + hi * 8
      wfr[wave, ks] = np.stack([Wf[c * CO + o:c * CO + o + 8] for c, o in zip(ci, co)])
  d_base, d_swz = r31 * YROWB, bs_swz(r31)
  gq, j16 = lane >> 4, lane & 15
  t_row = 8 * (gq >> 1) + (j16 >> 2)
  t_low = 2 * (gq & 1) + ((j16 >> 1) & 1)
  t_half = (j16 & 1) * 8

  def tr_off(rowb, chunk, plus4):
    return (t_row + plus4) * rowb + (((chunk + t_low) ^ bs_swz(t_row + plus4)) << 4) + t_half

  acc2 = {(w, i, j): np.zeros((64, 16)) for w in range(8) for i in range(TI) for j in range(TO)}

  def flush(ktp):
    for ft in range(256):
      for q in range(2):
        pc = q * 256 + ft
        row, ch = pc >> 4, pc & 15
        p = (g + ktp * G) * PX + row
        a = dxs + (ktp & 1) * DX_BYTES + row * DXROWB + ch * 16
        v = lds.v[a // 2:a // 2 + 8].copy()
        if p < M:
          assert np.all(np.isnan(DX[p, slice_ * SC + ch * 8:slice_ * SC + ch * 8 + 8])), "written twice"
          DX[p, slice_ * SC + ch * 8:slice_ * SC + ch * 8 + 8] = v

  for t in range(NST - 1):
    if t < KT:
      issue(t, t, (t & 1) ^ (NST & 1))
  for kt in range(KT):
    Ys = (kt % NST) * STAGE
    Xs = Ys + Y_BYTES
    As = Xs + X_BYTES
    light = 1 - (kt & 1)                   # the half whose waves have par != kt & 1
    if kt + NST - 1 < KT:
      issue(kt + NST - 1, (kt + NST - 1) % NST, light)
    if kt > 0:
      flush(kt - 1)
    for wave in range(8):
      cf, par = wave & 3, wave >> 2
      mine = (kt & 1) == par
      if mine:
        a0 = np.zeros((64, 16))
        a1 = np.zeros((64, 16))
        av = None
        if Af is not None:
          av = [np.stack([lds.v[int(a) // 2:int(a) // 2 + 4] for a in (As + r31 * XROWB + (((cf * 4 + q) ^ d_swz) << 4) + hi * 8)])
                for q in range(4)]
        for ks in range(0, KS, 2):
          y0 = lds.read16(Ys + d_base + (((2 * ks + hi) ^ d_swz) << 4))
          y1 = lds.read16(Ys + d_base + (((2 * ks + 2 + hi) ^ d_swz) << 4))
          mfma_32x32x16(wfr[wave, ks], y0, a0)
          mfma_32x32x16(wfr[wave, ks + 1], y1, a1)
      if do_w:
        fi0 = (wave & 1) * 2 if TI == 2 else (wave & 3)
        fo0 = (wave >> 1) * 2 if TI == 2 else (wave >> 2) * 2
        for k2 in range(2):
          fa = [read_tr_pair(lds, Xs + k2 * 16 * XROWB + tr_off(XROWB, (fi0 + i) * 4, 0), Xs + k2 * 16 * XROWB + tr_off(XROWB, (fi0 + i) * 4, 4))
                for i in range(TI)]
          fb = [read_tr_pair(lds, Ys + k2 * 16 * YROWB + tr_off(YROWB, (fo0 + j) * 4, 0), Ys + k2 * 16 * YROWB + tr_off(YROWB, (fo0 + j) * 4, 4))
                for j in range(TO)]
          for i in range(TI):
            for j in range(TO):
              mfma_32x32x16(fa[i], fb[j], acc2[wave, i, j])
      if mine:
        for l in range(64):
          dst = dxs + (kt & 1) * DX_BYTES + int(r31[l]) * DXROWB
          for q in range(4):
            a = dst + (cf * 32 + 8 * q + 4 * int(hi[l])) * 2
            v = a0[l, 4 * q:4 * q + 4] + a1[l, 4 * q:4 * q + 4]
            if av is not None:
              v = v + av[q][l]
            lds.v[a // 2:a // 2 + 4] = v
  if KT > 0:
    flush(KT - 1)
  if do_w:
    for wave in range(8):
      fi0 = (wave & 1) * 2 if TI == 2 else (wave & 3)
      fo0 = (wave >> 1) * 2 if TI == 2 else (wave >> 2) * 2
      for i in range(TI):
        for j in range(TO):
          for l in range(64):
            for e in range(16):
              ci = (fi0 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * int(hi[l])
              co = (fo0 + j) * 32 + int(r31[l])
              assert np.isnan(SLAB[g, slice_ * SC + ci, co]), "slab element written twice"
              SLAB[g, slice_ * SC + ci, co] = acc2[wave, i, j][l, e]


def swz_x64(row):
  return ((row >> 1) & 1) << 2


def swz_a64(row):
  return (row >> 1) & 7


def run_workgroup64(block, X, DY, W, ADD, DX, SLAB, M, CI, CO, slices, G, do_w=True):
  """k_bwdslice64: slices of 64 input channels, dgrad split (cf, kh) over the four waves of a half, NST = 3"""
  SC64, NST = 64, 3
  YROWB, XROWB = CO * 2, SC64 * 2
  assert YROWB == 1024 and XROWB == 128
  Y_BYTES, X_BYTES, A_BYTES = PX * YROWB, (PX * XROWB if do_w else 0), PX * XROWB
  STAGE = Y_BYTES + X_BYTES + A_BYTES + 256
  KSH = CO // 32
  lds = Lds(NST * STAGE)
  lane = np.arange(64)
  hi, r31 = lane >> 5, lane & 31
  xcd, idx = block & 7, block >> 3
  slice_, g = idx % slices, xcd + 8 * (idx // slices)
  KT_all = (M + PX - 1) // PX
  KT = (KT_all - g + G - 1) // G if g < KT_all else 0
  Xf, Yf, Wf = X.reshape(-1), DY.reshape(-1), W.reshape(-1)
  Af = ADD.reshape(-1) if ADD is not None else None

  def issue(kt, stage, par):
    assert ((kt & 1) ^ 1) == par, "tile t is issued by the half that is not multiplying two iterations earlier"
    p0 = (g + kt * G) * PX
    for w4 in range(4):                    # the four waves of the half
      for q in range(PX // 4):
        row = q * 4 + w4
        p = p0 + row
        col = (lane ^ bs_swz(row)) * 8
        off = np.where(p < M, (p * CO + col) * 2, OOB) if p < M else np.full(64, OOB)
        lds.dma(Yf, stage * STAGE + row * 1024, off)
      row = w4 * 8 + lane // 8
      slot = lane % 8
      p = p0 + row
      if do_w:
        col = slice_ * SC64 + ((slot ^ swz_x64(row)) * 8)
        lds.dma(Xf, stage * STAGE + Y_BYTES + w4 * 1024, np.where(p < M, (p * CI + col) * 2, OOB))
      if Af is not None:
        col = slice_ * SC64 + ((slot ^ swz_a64(row)) * 8)
        lds.dma(Af, stage * STAGE + Y_BYTES + X_BYTES + w4 * 1024, np.where(p < M, (p * CI + col) * 2, OOB))

  wfr = {}
  for wave in range(8):
    kh, cf = (wave >> 1) & 1, wave & 1
    for ks in range(KSH):
      ci = slice_ * SC64 + cf * 32 + r31
      co = (kh * KSH + ks) * 16 + hi * 8
      wfr[wave, ks] = np.stack([Wf[c * CO + o:c * CO + o + 8] for c, o in zip(ci, co)])
  gq, j16 = lane >> 4, lane & 15
  t_row = 8 * (gq >> 1) + (j16 >> 2)
  t_low = 2 * (gq & 1) + ((j16 >> 1) & 1)
  t_half = (j16 & 1) * 8

  def tr_y(chunk, plus4):
    return (t_row + plus4) * YROWB + (((chunk + t_low) ^ bs_swz(t_row + plus4)) << 4) + t_half

  def tr_x(chunk, plus4):
    return (t_row + plus4) * XROWB + (((chunk + t_low) ^ swz_x64(t_row + plus4)) << 4) + t_half

  acc2 = {(w, i, j): np.zeros((64, 16)) for w in range(8) for i in range(2) for j in range(2)}
  a0s = {w: np.zeros((64, 16)) for w in range(8)}
  avs = {w: None for w in range(8)}
  parts = {}

  def combine(ktp, wave):
    par, cf = wave >> 2, wave & 1
    tot = a0s[wave] + parts[par, cf]
    stg = np.full((32, 32), np.nan, np.float32)
    for l in range(64):
      for q in range(4):
        v = tot[l, 4 * q:4 * q + 4].copy()
        if avs[wave] is not None:
          v = v + avs[wave][q][l]
        c = 8 * q + 4 * int(hi[l])
        stg[int(r31[l]), c:c + 4] = v
    for q in range(2):
      for l in range(64):
        pc = q * 64 + l
        row, ch = pc >> 2, pc & 3
        p = (g + ktp * G) * PX + row
        if p < M:
          c0 = slice_ * SC64 + cf * 32 + ch * 8
          assert np.all(np.isnan(DX[p, c0:c0 + 8])), "written twice"
          DX[p, c0:c0 + 8] = stg[row, ch * 8:ch * 8 + 8]

  for t in range(NST - 1):
    if t < KT:
      issue(t, t, (t & 1) ^ 1)
  for kt in range(KT):
    Ys = (kt % NST) * STAGE
    Xs = Ys + Y_BYTES
    As = Xs + X_BYTES
    light = (kt & 1) ^ 1
    if kt + NST - 1 < KT:
      issue(kt + NST - 1, (kt + NST - 1) % NST, light)
    if kt > 0:
      for wave in range(8):
        if (wave >> 2) == light and ((wave >> 1) & 1) == 0:
          combine(kt - 1, wave)
    for wave in range(8):
      par, kh, cf = wave >> 2, (wave >> 1) & 1, wave & 1
      if (kt & 1) == par:
        if kh == 0 and Af is not None:
          avs[wave] = [np.stack([lds.v[int(a) // 2:int(a) // 2 + 4] for a in (As + r31 * XROWB + (((cf * 4 + q) ^ swz_a64(r31)) << 4) + hi * 8)])
                       for q in range(4)]
        a0 = np.zeros((64, 16))
        a1 = np.zeros((64, 16))
        for ks in range(0, KSH, 2):
          c0 = 2 * (kh * KSH + ks) + hi
          y0 = lds.read16(Ys + r31 * YROWB + ((c0 ^ bs_swz(r31)) << 4))
          y1 = lds.read16(Ys + r31 * YROWB + (((c0 + 2) ^ bs_swz(r31)) << 4))
          mfma_32x32x16(wfr[wave, ks], y0, a0)
          mfma_32x32x16(wfr[wave, ks + 1], y1, a1)
        a0 = a0 + a1
        if kh == 1:
          parts[par, cf] = a0
        else:
          a0s[wave] = a0
      if do_w:
        fo0 = wave * 2
        for k2 in range(2):
          fa = [read_tr_pair(lds, Xs + k2 * 16 * XROWB + tr_x(i * 4, 0), Xs + k2 * 16 * XROWB + tr_x(i * 4, 4)) for i in range(2)]
          fb = [read_tr_pair(lds, Ys + k2 * 16 * YROWB + tr_y((fo0 + j) * 4, 0), Ys + k2 * 16 * YROWB + tr_y((fo0 + j) * 4, 4)) for j in range(2)]
          for i in range(2):
            for j in range(2):
              mfma_32x32x16(fa[i], fb[j], acc2[wave, i, j])
  if KT > 0:
    for wave in range(8):
      if (wave >> 2) == ((KT - 1) & 1) and ((wave >> 1) & 1) == 0:
        combine(KT - 1, wave)
  if do_w:
    for wave in range(8):
      fo0 = wave * 2
      for i in range(2):
        for j in range(2):
          for l in range(64):
            for e in range(16):
              ci = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * int(hi[l])
              co = (fo0 + j) * 32 + int(r31[l])
              assert np.isnan(SLAB[g, slice_ * SC64 + ci, co]), "slab element written twice"
              SLAB[g, slice_ * SC64 + ci, co] = acc2[wave, i, j][l, e]


def check64(M, CI, G, add=True, seed=0):
  CO = 512
  rng = np.random.RandomState(seed)
  X = rng.randint(-2, 3, (M, CI)).astype(np.float32)
  DY = rng.randint(-2, 3, (M, CO)).astype(np.float32)
  W = rng.randint(-2, 3, (CI, CO)).astype(np.float32)
  ADD = rng.randint(-2, 3, (M, CI)).astype(np.float32) if add else None
  slices = CI // 64
  DX = np.full((M, CI), np.nan, np.float32)
  SLAB = np.full((G, CI, CO), np.nan, np.float32)
  for block in range(slices * G):
    run_workgroup64(block, X, DY, W, ADD, DX, SLAB, M, CI, CO, slices, G)
  ref_dx = DY.astype(np.float64) @ W.T.astype(np.float64) + (ADD if add else 0)
  ref_dw = X.T.astype(np.float64) @ DY.astype(np.float64)
  assert not np.isnan(DX).any() and not np.isnan(SLAB).any()
  assert np.array_equal(DX, ref_dx), (M, CI, CO, "dX")
  assert np.array_equal(SLAB.sum(0), ref_dw), (M, CI, CO, "dW")
  print(f"ok  (64-channel slices) M={M} cin={CI} cout={CO} G={G} addend={add}")


def bank_check64():
  """128-byte rows: the X tile under the transposing reads (4 rows x 64 bytes per 32-lane pass -> four bank quarters), the addend
  tile under the row-per-lane ds_read_b64 (rows 0 .. 15 of a 32-lane pass -> sixteen different 16-byte slots of the window)"""
  for base in (0, 4):
    for p0 in range(0, 32, 4):
      quarters = {((r * 128 + ((base ^ swz_x64(r)) << 4)) // 64) % 4 for r in range(p0, p0 + 4)}
      assert len(quarters) == 4, (base, p0)
  for c in range(8):
    for r0 in (0, 16):
      slots = {((r * 128 + ((c ^ swz_a64(r)) << 4)) // 16) % 16 for r in range(r0, r0 + 16)}
      assert len(slots) == 16, (c, r0)
  print("bank layout ok for the 64-channel slices")


def check(M, CI, CO, G, NST, add=True, seed=0):
  rng = np.random.RandomState(seed)
  X = rng.randint(-2, 3, (M, CI)).astype(np.float32)
  DY = rng.randint(-2, 3, (M, CO)).astype(np.float32)
  W = rng.randint(-2, 3, (CI, CO)).astype(np.float32)
  ADD = rng.randint(-2, 3, (M, CI)).astype(np.float32) if add else None
  slices = CI // SC
  DX = np.full((M, CI), np.nan, np.float32)
  SLAB = np.full((G, CI, CO), np.nan, np.float32)
  for block in range(slices * G):
    run_workgroup(block, X, DY, W, ADD, DX, SLAB, M, CI, CO, slices, G, NST)
  ref_dx = DY.astype(np.float64) @ W.T.astype(np.float64) + (ADD if add else 0)
  ref_dw = X.T.astype(np.float64) @ DY.astype(np.float64)
  assert not np.isnan(DX).any() and not np.isnan(SLAB).any()
  assert np.array_equal(DX, ref_dx), (M, CI, CO, "dX")
  assert np.array_equal(SLAB.sum(0), ref_dw), (M, CI, CO, "dW")
  print(f"ok  M={M} cin={CI} cout={CO} G={G} NST={NST} addend={add}")


def bank_check(CO):
  """the dY tile: ds_read_b128 lane groups and transposing passes are conflict-free under bs_swz"""
  YROWB = CO * 2
  groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
  for ks in range(CO // 16):
    for hi in (0, 1):
      for grp in groups:
        slots = {((r * YROWB + (((2 * ks + hi) ^ bs_swz(r)) << 4)) // 16) % 16 for r in grp}
        assert len(slots) == 16, (CO, ks, hi)
  for rowb in (YROWB, 256):
    for chunk in range(0, rowb // 16, 4):
      for p0 in list(range(0, 16, 4)):
        quarters = {((r * rowb + ((chunk ^ bs_swz(r)) << 4)) // 64) % 4 for r in range(p0, p0 + 4)}
        assert len(quarters) == 4, (rowb, chunk, p0)
  print(f"bank layout ok for cout={CO}")


if __name__ == "__main__":
  for co in (128, 256):
    bank_check(co)
  check(8 * 8 * 32 + 40, 256, 256, 8, 4)
  check(8 * 5 * 32 - 7, 128, 128, 8, 5, add=False)
  check(16 * 3 * 32, 256, 128, 16, 5)
  bank_check64()
  check64(8 * 5 * 32 + 19, 128, 8)
  check64(8 * 4 * 32, 64, 8, add=False)
