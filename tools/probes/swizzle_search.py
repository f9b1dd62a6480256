#!/usr/bin/env python3
"""Exhaustive search for an XOR swizzle of a [rows][512-byte] bf16 LDS tile that is bank-conflict-free for BOTH
  * ds_read_b128 of a 16x16x32 MFMA operand fragment (lane l: row l & 15, 16-byte chunk c0 + (l >> 4); four 16-lane
    groups {0-3,12-15,20-27}, ... of MI355X_MICROARCH.md) and
  * ds_read_b64_tr_b16 (a 32-lane pass reads four pixel rows x one 64-byte quad: the quads must sit in four different
    bank quarters),
over all XOR-linear maps row bits -> chunk bits (rigl_amd/csrc/bwd1x1.hpp: the dY tile is the k = channel operand of
the dX product and the k = pixel operand of the dW product).  First solution printed: chunk ^= (r&1)<<1 | (r>>1&1)<<2 |
((r ^ r>>2)&1)<<3.  The 128-byte-row variant (second search below) gives (r>>1&1)<<1 | ((r>>1 ^ r>>2)&1)<<2.  Host only."""
import itertools, sys
# ds_read_b128 lane groups
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)),
        list(range(36,44))+list(range(48,52))+list(range(60,64))]
def swz(M,row):
    v=0
    for b in range(4):
        bit=0
        for k in range(5):
            if (M[b]>>k)&1: bit^=(row>>k)&1
        v|=bit<<b
    return v
def ok_b128(M, rows_base):
    # 16x16x32 A fragment: lane l -> row rows_base + (l&15), logical chunk c0 + (l>>4)
    for c0 in (0,):
        for g in G128:
            seen=set()
            for l in g:
                row=rows_base+(l&15); ch=c0+(l>>4)
                s=(ch ^ swz(M,row))&15
                if s in seen: return False
                seen.add(s)
    return True
def ok_tr(M):
    # 32-lane pass: rows p0..p0+3, quad Q: physical quad ((Q*4 + x) ^ swz(row))>>2 &3 distinct
    for p0 in range(0,32,4):
        qs=set()
        for r in range(4):
            qs.add((swz(M,p0+r)>>2)&3)
        if len(qs)!=4: return False
    return True
sol=[]
for M in itertools.product(range(32), repeat=4):
    if not ok_tr(M): continue
    if ok_b128(M,0) and ok_b128(M,16):
        sol.append(M)
        if len(sol)>=5: break
print(sol)
for M in sol[:2]:
    print([swz(M,r) for r in range(32)])


# ---- 128-byte rows ------------------------------------------------------------------------------------------------
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)),
        list(range(36,44))+list(range(48,52))+list(range(60,64))]
# 128-byte rows (8 chunks of 16 B): swizzle = 3-bit XOR mask from 5 row bits
def swz(M,row):
    v=0
    for b in range(3):
        bit=0
        for k in range(5):
            if (M[b]>>k)&1: bit^=(row>>k)&1
        v|=bit<<b
    return v
def ok_b128(M, base):
    # 16x16x32 A frag: lane l: row base+(l&15), chunk c0+(l>>4), c0 in {0,4}
    for c0 in (0,4):
        for g in G128:
            seen=set()
            for l in g:
                row=base+(l&15); ch=c0+(l>>4)
                slot=(row&1)*8 + ((ch ^ swz(M,row))&7)
                if slot in seen: return False
                seen.add(slot)
    return True
def ok_tr(M):
    # 32-lane pass: rows p0..p0+3, logical quad Q (chunks 4Q..4Q+3): physical 64-B unit = (row&1)*2 + quadphys ; need 4 distinct among 4 rows
    for p0 in range(0,32,4):
        for Q in (0,1):
            seen=set()
            for r in range(4):
                row=p0+r
                qp=((4*Q) ^ swz(M,row))>>2 &1
                unit=(row&1)*2+qp
                if unit in seen: return False
                seen.add(unit)
    return True
sol=[]
for M in itertools.product(range(32), repeat=3):
    if ok_tr(M) and ok_b128(M,0) and ok_b128(M,16):
        sol.append(M)
        if len(sol)>=5: break
print(sol)
for M in sol[:2]: print([swz(M,r) for r in range(16)])
