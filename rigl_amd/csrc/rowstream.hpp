// K1, the row-streaming body for the 1x1 / stride-1 GEMMs: C[M][N] = A[M][K] x B[N][K]^T (included inside namespace rigl::k1
// of conv.hip).  The A operand (activation rows for the forward, output-gradient rows for dgrad: 16 bytes per lane,
// contiguous in NHWC, exactly one v_mfma_f32_32x32x16_bf16 fragment) goes global memory -> VGPRs and never touches LDS; only
// the filter passes through LDS, and it is STATIONARY there: a workgroup owns one slice of NS = 32 * TN output columns, brings
// that slice's NS x K filter rows into LDS once (LDS-DMA, XOR-swizzled for conflict-free ds_read_b128) and then streams rows.
// After the one barrier behind the filter load the eight waves of a workgroup never meet again: each walks its own row
// fragments (F <= 32 rows x K, see below), so one wave's epilogue (convert, stage, add, store, statistics) runs under its
// SIMD partner's MFMAs and the waves of a CU drift apart instead of moving in lockstep phases.
//   * A ring: four chunks of 64 reduction channels (4 x 4 VGPRs each) per wave.  After the MFMAs of chunk s are issued the
//     registers of its slot are re-loaded with chunk s + 4 -- the next chunks of this fragment (K = 512) or the first
//     chunks of the wave's next fragment(s) (K <= 256): 16 KB per wave, 128 KB per CU in flight at all times, no second
//     register set.  The reduction of one fragment is K / 16 k-steps x TN MFMAs into TN accumulators.
//   * Row fragments are F rows, not 32: F = ceil(M / (slots x passes)) so that every one of the slots = 8 x workgroups-per-
//     slice waves gets the same number of fragments (+- 1).  These GEMMs move bytes, not flops (256 -> 1024 at 14x14: 13
//     GFLOP against 64 MB): idle MFMA rows cost nothing, an uneven split of the rows costs bandwidth.  25 088 rows over 256
//     waves are 3.06 fragments of 32 rows (one wave in sixteen runs a fourth pass: 77 %) but 3.92 of 25.  Fragment fr
//     belongs to slot fr % slots: at any moment the grid streams ONE window of the tensors.
//   * Epilogue per fragment (operands swapped: a lane holds 4 consecutive output channels of one row): bf16, staged in a
//     wave-private LDS tile (rows padded by 8 bytes: conflict-free ds_write_b64), read back 16 bytes per lane (two
//     ds_read_b64) = whole 64 * TN-byte row segments, dgrad adds the shortcut
//     gradient (bf16(bf16(acc) + addend), requested at the top of the fragment), stores.  Forward: the batch-norm
//     statistics of the bf16 outputs are summed from the read-back chunks -- a lane sees the SAME eight channels in
//     every iteration of every fragment, so sixteen fp32 registers per lane collect them for the whole kernel; one
//     partial row per workgroup at the end (fixed order: deterministic).
//   * Workgroup b -> XCD b % 8; the `slices` workgroups that stream the same rows against different filter slices sit on
//     the same XCD, so a row fragment comes from HBM once and from that XCD's L2 for the other slices.
// Where the time goes on a small layer (14x14 256 -> 1024 forward, 27.4 us per call; -DRIGL_RS_ABLATE builds, gpurun r5h-r5j):
// without the output stores 22.6 us, without the ring refills 25.7, without statistics 27.9, without MFMAs 27.7, without
// any of loads / stores / statistics 20.4, additionally without MFMAs 17.2, without the filter reads 18.8, with everything
// off 10.1 (launch, filter load, barrier, statistics hand-off).  No single pipe paces it: ~10 us are fixed per launch, ~7 the
// memory operations, ~3 the MFMAs.  Tried on top and measured level (removed): starting the second wave of every SIMD one
// MFMA phase late (the waves are not in lockstep: no effect), every workgroup starting its filter load at another piece
// (the load is latency, not L2 queueing), conflict-free staging rows and filter reads pipelined across chunk
// boundaries (both kept: LDS conflicts 25-39 % -> see profiles/r5/pmc_sq_k1.txt, time unchanged).
// Reference: layers.masked_conv2d with a 1x1 kernel (pruning_layers.py:139-157), the bottleneck's first / third conv and
// the projection shortcut (resnet_model.py:396-501), and their autodiff.
#pragma once

struct RsArgs {
  const uint16_t* A;     // [M][K] bf16
  const uint16_t* B;     // [N][K] bf16
  uint16_t* C;           // [M][N] bf16
  const uint16_t* ADD;   // dgrad: optional [M][N]
  float* STATS;          // forward: optional [gprime][2][N]
  int M, N;
  int F, nfrag;          // rows per fragment, ceil(M / F)
  int slices, gprime;    // N / NS column slices, workgroups per slice (a multiple of 8); grid = slices * gprime
  uint32_t a_bytes, b_bytes, c_bytes;
  unsigned long long* TRACE;   // development (-DRIGL_RS_TRACE): [grid][64] s_memtime stamps of wave 0
  const float* BNP;            // BNL kernels: [2][K] scale, shift of the batch norm in front of this conv
  uint16_t* A2;                // BNL kernels: [M][K] bf16, the activated operand (side output, the backward reads it)
};
#ifdef RIGL_RS_TRACE
#define RS_STAMP(i_) { if (tid == 0 && P.TRACE && (i_) < 64) P.TRACE[blockIdx.x * 64 + (i_)] = __builtin_amdgcn_s_memtime(); }
#else
#define RS_STAMP(i_) { }
#endif

constexpr int RS_THREADS = 512;

// bf16(a + b) per half, round to nearest even on the hardware converter (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t rs_add_bf16x2(uint32_t a, uint32_t b) {
  const f32x2 s = {__uint_as_float(a << 16) + __uint_as_float(b << 16),
                   __uint_as_float(a & 0xFFFF0000u) + __uint_as_float(b & 0xFFFF0000u)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(s, bf16x2));
}

// BNL kernels: the batch-norm apply + ReLU of bn.hip's k_fwd_apply on one A fragment (8 consecutive reduction channels of one
// row) in registers, bf16(max(fma(x, scale, shift), 0)): ps -> this lane's 8 scales in LDS, ps + K its 8 shifts; rows that do
// not exist become zeros (they read zeros, and max(shift, 0) would enter the statistics)
typedef __attribute__((ext_vector_type(4))) float rs_f32x4;
__device__ __forceinline__ u32x4 rs_bnrelu(bf16x8 v, const float* ps, int K, bool rowok) {
  const rs_f32x4 s0 = *reinterpret_cast<const rs_f32x4*>(ps), s1 = *reinterpret_cast<const rs_f32x4*>(ps + 4);
  const rs_f32x4 h0 = *reinterpret_cast<const rs_f32x4*>(ps + K), h1 = *reinterpret_cast<const rs_f32x4*>(ps + K + 4);
  u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float sa = d < 2 ? s0[2 * d] : s1[2 * d - 4], sb = d < 2 ? s0[2 * d + 1] : s1[2 * d - 3];
    const float ha = d < 2 ? h0[2 * d] : h1[2 * d - 4], hb = d < 2 ? h0[2 * d + 1] : h1[2 * d - 3];
    const f32x2 y = {fmaxf(fmaf(__uint_as_float(w[d] << 16), sa, ha), 0.f), fmaxf(fmaf(__uint_as_float(w[d] & 0xFFFF0000u), sb, hb), 0.f)};
    w[d] = rowok ? __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2)) : 0u;
  }
  return w;
}

template <int KC, int TN>
struct RsGeom {
  static constexpr int K = 64 * KC, NS = 32 * TN, RB = K * 2, CPR = K / 8;
  static constexpr int WBYTES = NS * RB;                         // the filter slice
  static constexpr int SROW = NS * 2 + 8, STG_WAVE = 32 * SROW;  // a wave's staging tile: 32 rows, 8 bytes of padding (below)
  static constexpr int SMEM = WBYTES + 8 * STG_WAVE;
  static constexpr int SMEM_BNL = SMEM + 2 * K * 4;               // + the batch-norm parameters of the BNL kernels
  static_assert(8 * STG_WAVE >= 8 * 64 * 16 * 4, "the statistics hand-off re-uses the staging tiles");
};

// MODE 0 = forward (statistics always summed, written if STATS is given), 1 = dgrad, 2 = dgrad + addend
// BNL (forward only; VERDICT r5 item 2, "batch-norm apply + ReLU on the operand load"): A is the PRE-batch-norm tensor; every A
// fragment becomes bf16(max(x * scale[k] + shift[k], 0)) in its ring registers one k-step before its MFMAs (under the previous
// step's), so the product is conv(relu(bn(x))) with the rounding points of the separate apply pass, which is then not run;
// the activated tensor the backward needs (weight-gradient operand) leaves as a side output from the same registers: chunk
// c of a row is stored by the workgroup of column slice c % slices.
template <int KC, int TN, int MODE, bool BNL = false>
__global__ __launch_bounds__(RS_THREADS) void k_rowstream(RsArgs P) {
  static_assert(!BNL || MODE == 0, "the on-load transform belongs to the forward");
  constexpr bool DGRAD = MODE != 0, ADDEND = MODE == 2;
  using G = RsGeom<KC, TN>;
  constexpr int NS = G::NS, RB = G::RB, CPR = G::CPR, SROW = G::SROW;
  constexpr int CHR = 4 * TN, RPI = 64 / CHR, ITERS = 32 / RPI;    // 16-byte chunks per staged row, rows per store iteration
  constexpr int U = KC >= 4 ? 1 : 4 / KC;                          // fragments per unrolled body (U * KC chunks, a multiple of 4)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_rs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, r31 = lane & 31;
  // workgroup -> (column slice, row group): the slices of one row group share an XCD (block b runs on XCD b % 8)
  RS_STAMP(0);
  const int xcd = (int)(blockIdx.x & 7u), idx = (int)(blockIdx.x >> 3);
  const int slice = idx % P.slices, gp = xcd + 8 * (idx / P.slices);
  const int slots = 8 * P.gprime, ws = gp * 8 + wave;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);
  const __amdgpu_buffer_rsrc_t rsrcD = make_rsrc(ADDEND ? P.ADD : P.A, ADDEND ? P.c_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsrcC = make_rsrc(P.C, P.c_bytes);

  float* const prm = reinterpret_cast<float*>(smem_rs + G::SMEM);           // BNL: [2][K]
  const __amdgpu_buffer_rsrc_t rsrcA2 = make_rsrc(BNL ? P.A2 : P.C, BNL ? P.a_bytes : 0u);
  if (BNL) {
    for (int i = tid; i < 2 * G::K; i += RS_THREADS) prm[i] = P.BNP[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the raw barrier below does not wait for LDS stores)
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- the filter slice -> LDS: pieces of 1 KB, position p = piece * 64 + lane -> (row, 16-byte slot); the slot holds the
  // row's chunk slot ^ swizzle(row)
#define RS_SWZ(row_) (CPR == 8 ? (((row_) >> 1) & 7) : ((row_) & 15))
  {
    constexpr int PIECES = G::WBYTES / 1024, PPW = PIECES / 8;
    static_assert(PIECES % 8 == 0, "whole rounds of eight waves");
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = j * 8 + wave, p = piece * 64 + lane, row = p / CPR, ch = p % CPR;
      const int src = ((slice * NS + row) * CPR + (ch ^ RS_SWZ(row))) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(smem_rs + piece * 1024), 16, src, 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // byte offset of this lane's 16 bytes of chunk 0, k-step 0 of the wave's ii-th fragment: a per-lane constant (out of range for
  // the lanes beyond F rows: they read zeros) + the fragment's first row (uniform); fragments beyond the tensor are out of
  // range by themselves (offsets stay below 2^32: rs_plan keeps the tensors under 2^30 bytes)
  const uint32_t a_lane = r31 < P.F ? (uint32_t)(r31 * RB + hi * 16) : OOB;
  auto row_off = [&](int ii) -> uint32_t {
    // (readfirstlane: the uniform part stays one scalar per use instead of becoming per-lane induction registers)
    return a_lane + (uint32_t)__builtin_amdgcn_readfirstlane((ws + ii * slots) * P.F * RB);
  };
  bf16x8 a[4][4];
#define RS_LOAD_CHUNK(slot_, ii_, c_)                                                                    \
  {                                                                                                      \
    const uint32_t o_ = row_off(ii_) + (uint32_t)((c_) * 128);                                           \
    _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4)                                                     \
      a[slot_][k4] = __builtin_bit_cast(bf16x8, buf_load16(rsrcA, o_ + (uint32_t)(k4 * 32)));            \
  }
  // the first four chunks of the wave's sequence s = ii * KC + c
#pragma unroll
  for (int s = 0; s < 4; ++s) RS_LOAD_CHUNK(s, s / KC, s % KC);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // the filter pieces (issued before the sixteen row loads) have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  RS_STAMP(1);

  const int swz = RS_SWZ(r31);
  const unsigned char* const bs = smem_rs + r31 * RB;
  unsigned char* const stg = smem_rs + G::WBYTES + wave * G::STG_WAVE;
  float sS[8], sQ[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sS[e] = sQ[e] = 0.f;
  const int ni = ws < P.nfrag ? (P.nfrag - ws + slots - 1) / slots : 0;   // fragments of this wave
  const int srow = lane / CHR, sch = lane % CHR;                          // this lane's row / chunk within a store iteration
  uint32_t c_lane[ITERS];                                                 // (rows beyond F belong to the next fragment: out of range)
#pragma unroll
  for (int it = 0; it < ITERS; ++it)
    c_lane[it] = it * RPI + srow < P.F ? (uint32_t)(((it * RPI + srow) * P.N + slice * NS + sch * 8) * 2) : OOB;

  for (int i0 = 0; i0 < ni; i0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i0 + u;
      const int fr = ws + ii * slots;
      const int row0 = fr * P.F;
      RS_STAMP(2 + 3 * ii);
      // this lane's 16 bytes of the output rows it stores: c_lane[it] + the fragment's first row (rows beyond the tensor
      // are out of the buffer's range: dropped)
      uint32_t coff[ITERS];
      const uint32_t c_base = (uint32_t)__builtin_amdgcn_readfirstlane(row0 * P.N * 2);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) coff[it] = c_lane[it] + c_base;
      uint4 addv[ITERS];
      if (ADDEND) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) addv[it] = buf_load16(rsrcD, coff[it]);   // (non-temporal measured level: profiles/r6)
      }
      f32x16 acc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      // K / 16 k-steps of TN MFMAs each.  The filter fragments of k-step s + 2 are requested right behind the MFMAs of
      // k-step s (two register sets, also across the chunk boundaries: no exposed LDS latency inside a fragment); the ring
      // slot of a chunk is re-loaded with chunk s + 4 of the wave's sequence behind the chunk's last MFMAs.
      {
        constexpr int S = KC * 4;
        bf16x8 bq[2][TN];
#define RS_READ(s_, buf_)                                                                                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
          bq[buf_][j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * RB + ((((2 * (s_) + hi)) ^ swz) << 4));
        RS_READ(0, 0);
        RS_READ(1, 1);
        const bool rowok = BNL && r31 < P.F && row0 + r31 < P.M;
#define RS_BNL(st_)                                                                                      \
        {                                                                                                \
          const int c_ = (st_) / 4, g_ = (st_) % 4, slot_ = (u * KC + c_) & 3;                           \
          const u32x4 w_ = rs_bnrelu(a[slot_][g_], prm + c_ * 64 + g_ * 16 + hi * 8, G::K, rowok);       \
          a[slot_][g_] = __builtin_bit_cast(bf16x8, w_);                                                 \
          if ((c_ % P.slices) == slice)                                                                  \
            __builtin_amdgcn_raw_buffer_store_b128(w_, rsrcA2, (int)(row_off(ii) + (uint32_t)(c_ * 128 + g_ * 32)), 0, 0); \
        }
        if (BNL) RS_BNL(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < S; ++st) {
          const int c = st / 4, g4 = st % 4;
          const int t = u * KC + c;                      // chunk of the unrolled body: its ring slot is static
          const int slot = t & 3;
#pragma unroll
          for (int j = 0; j < TN; ++j)
#if defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 8)        // timing experiment 8: no MFMAs (operands stay live through one add)
            acc[j][0] += (float)bq[st & 1][j][0] + (float)a[slot][g4][0];
#else
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][j], a[slot][g4], acc[j], 0, 0, 0);
#endif
#if !(defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 64))      // timing experiment 64: the filter fragments are read once per fragment
          if (st + 2 < S) RS_READ(st + 2, st & 1);
#endif
          if (BNL && st + 1 < S) RS_BNL(st + 1);       // the next k-step's A fragment, transformed under this step's MFMAs
          if (g4 == 3) {
            const int t4 = t + 4;
#if !(defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 1))      // timing experiment 1: the ring is never refilled (wrong results)
            RS_LOAD_CHUNK(slot, i0 + t4 / KC, t4 % KC);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#undef RS_READ
#undef RS_BNL
      }
      RS_STAMP(3 + 3 * ii);
      // ---- epilogue: D row = (e & 3) + 8 * (e >> 2) + 4 * hi -> channel of n-tile j, column = lane & 31 -> row of the fragment
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 lo = {acc[j][4 * q], acc[j][4 * q + 1]}, hi2 = {acc[j][4 * q + 2], acc[j][4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          // rows are 8 bytes longer than the tile: row r starts 2 r banks further, so the 16 lanes of a ds_write_b64 group (16
          // rows, one column) cover all 32 banks of their window (16 bytes of padding left rows r and r + 8 on the same banks:
          // 2-way conflicts on every staging write, 25-39 % of the kernel's LDS cycles in profiles/r5/pmc_sq_k1.txt)
#if defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 32)       // timing experiment 32: no staging writes
          if (pk.x == 0x12345678u)
#endif
          *reinterpret_cast<uint2*>(stg + r31 * SROW + (j * 32 + 8 * q + 4 * hi) * 2) = pk;
        }
      // read back / add / store / sum in groups of four iterations (16 registers of read-back data live at a time)
#pragma unroll
      for (int g0 = 0; g0 < ITERS; g0 += 4) {
        uint4 v[4];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          // this lane's 16 bytes as two ds_read_b64 (rows are only 8-byte aligned): the 16 lanes of row R and the 16 of
          // row R + 1 interleave on the banks
          const unsigned char* const p = stg + ((g0 + i4) * RPI + srow) * SROW + sch * 16;
          const uint2 lo8 = *reinterpret_cast<const uint2*>(p), hi8 = *reinterpret_cast<const uint2*>(p + 8);
          v[i4] = make_uint4(lo8.x, lo8.y, hi8.x, hi8.y);
        }
        if (ADDEND) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const uint4 q4 = addv[g0 + i4];
            v[i4].x = rs_add_bf16x2(v[i4].x, q4.x); v[i4].y = rs_add_bf16x2(v[i4].y, q4.y);
            v[i4].z = rs_add_bf16x2(v[i4].z, q4.z); v[i4].w = rs_add_bf16x2(v[i4].w, q4.w);
          }
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const u32x4 o = {v[i4].x, v[i4].y, v[i4].z, v[i4].w};
#if defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 2)        // timing experiment 2: no output stores
          if (o.x == 0x12345678u)
#endif
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrcC, (int)coff[g0 + i4], 0, 0);
        }
#if defined(RIGL_RS_ABLATE) && (RIGL_RS_ABLATE & 4)        // timing experiment 4: no statistics
        if (false) {
#else
        if (!DGRAD) {
#endif
          // rows beyond the fragment were multiplied as zeros: they add nothing
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const uint32_t w2 = dword_of(v[i4], d);
              const float y0 = __uint_as_float(w2 << 16), y1 = __uint_as_float(w2 & 0xFFFF0000u);
              sS[2 * d] += y0; sQ[2 * d] += y0 * y0;
              sS[2 * d + 1] += y1; sQ[2 * d + 1] += y1 * y1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the next fragment of the unrolled body starts with this one's registers free, and re-reads its filter fragments:
      // kept across the K = 64 fragments of one body they are 64 more live registers -- the A ring spilled)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      RS_STAMP(4 + 3 * ii);
    }
  }
  RS_STAMP(62);
#undef RS_LOAD_CHUNK
#undef RS_SWZ
  if (!DGRAD && P.STATS) {
    // every wave is done with its staging tile; the lanes' sums meet in LDS and leave in a fixed order
    __syncthreads();
    float* const area = reinterpret_cast<float*>(smem_rs + G::WBYTES);       // [8 waves][64 lanes][16]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      area[(wave * 64 + lane) * 16 + e] = sS[e];
      area[(wave * 64 + lane) * 16 + 8 + e] = sQ[e];
    }
    __syncthreads();
    if (tid < 2 * NS) {
      const int k = tid / NS, n = tid % NS, ch = n / 8, e = n % 8;
      float v[8 * RPI];         // (all reads first, then the additions in the fixed order: one wait instead of one per value)
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8)
#pragma unroll
        for (int q = 0; q < RPI; ++q) v[w8 * RPI + q] = area[(w8 * 64 + q * CHR + ch) * 16 + k * 8 + e];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8 * RPI; ++i) s += v[i];
      P.STATS[((int64_t)gp * 2 + k) * P.N + slice * NS + n] = s;
    }
  }
  RS_STAMP(63);
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Legal: 1x1, stride 1, no padding; reduction K in {64, 128, 256, 512}; the filter slice NS x K x 2 bytes <= 64 KB next to the
// eight staging tiles (TN = 4 -> NS = 128 for K <= 256, TN = 2 -> NS = 64 for K = 512 or N = 64); N a multiple of NS with
// N / NS a power of two <= 32; at least 4 096 rows.  MODE 0 = forward (K = cin, N = cout), 1 = dgrad (K = cout, N = cin).
struct RsPlan { int kc, tn, slices, gprime, F, nfrag; };
template <int MODE>
static bool rs_plan(const RiglConvDesc* d, RsPlan& p) {
  if (d->kh != 1 || d->kw != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_top || d->pad_left) return false;
  if (d->ho != d->h || d->wo != d->w) return false;        // (a cropped output grid: the generic bodies walk ho x wo)
  const int k = MODE == 0 ? d->cin : d->cout, n = MODE == 0 ? d->cout : d->cin;
  if (k != 64 && k != 128 && k != 256 && k != 512) return false;
  p.kc = k / 64;
  p.tn = (k <= 256 && n % 128 == 0) ? 4 : 2;
  const int ns = 32 * p.tn;
  if (n % ns) return false;
  p.slices = n / ns;
  if (p.slices > 32 || (p.slices & (p.slices - 1))) return false;
  const int64_t M = (int64_t)d->n * d->h * d->w;
  if (M < 4096 || M * (k > n ? k : n) * 2 >= (1ll << 30)) return false;
  int groups = num_cus() / (8 * p.slices);
  if (groups < 1) groups = 1;
  p.gprime = 8 * groups;
  const int slots = 8 * p.gprime;
  const int passes = (int)((M + 32ll * slots - 1) / (32ll * slots));
  p.F = (int)((M + (int64_t)slots * passes - 1) / ((int64_t)slots * passes));
  if (p.F > 32) p.F = 32;
  if (p.F < 8) p.F = 8;
  p.nfrag = (int)((M + p.F - 1) / p.F);
  return true;
}
template <int KC, int TN, int MODE>
static bool rs_ready_i() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowstream<KC, TN, MODE>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, RsGeom<KC, TN>::SMEM) == hipSuccess;
  return ready;
}
template <int KC, int TN>
static bool rs_ready_bnl() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowstream<KC, TN, 0, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, RsGeom<KC, TN>::SMEM_BNL) == hipSuccess;
  return ready;
}
#define RS_DISPATCH(p_, DG_, WHAT_)                                                                      \
  switch ((p_).kc * 8 + (p_).tn) {                                                                       \
    case 1 * 8 + 2: { WHAT_(1, 2, DG_); } break;                                                         \
    case 1 * 8 + 4: { WHAT_(1, 4, DG_); } break;                                                         \
    case 2 * 8 + 2: { WHAT_(2, 2, DG_); } break;                                                         \
    case 2 * 8 + 4: { WHAT_(2, 4, DG_); } break;                                                         \
    case 4 * 8 + 2: { WHAT_(4, 2, DG_); } break;                                                         \
    case 4 * 8 + 4: { WHAT_(4, 4, DG_); } break;                                                         \
    case 8 * 8 + 2: { WHAT_(8, 2, DG_); } break;                                                         \
    default: break;                                                                                      \
  }
// "rowstream": 0 = off, 1 = on for the layers of rs_default (measured faster than the bodies they replace), 2 = every legal layer
// The layers the body takes by default: where the ResNet-50 step at batch 128 got faster (tools/instep_table.py, gpurun r5c; us
// per layer in the step, the body it replaces -> this one; alone from HBM in brackets, tools/rs_bench.py, gpurun r5a):
//   forward + statistics, reductions of up to 256 channels:
//     28x28 128->512 37.5 -> 31.3 (40.6 -> 37.4)   14x14 256->1024 30.1 -> 25.1 (33.7 -> 27.8)   56x56 256->64 53.8 -> 47.3
//     56x56 256->128 78.8 -> 71.0   56x56 64->64 23.8 -> 19.1   56x56 64->256 54.4 -> 53.0 (the layers of the former x1x1.hpp)
//     512-channel reductions stay on the ping-pong body (28x28 512->128 29.4 -> 32.6, 512->256 45.2 -> 55.1, 7x7 512->2048
//     21.7 -> 33.9: 32 column slices re-read the rows 32 times)
//   dgrad + addend: faster ALONE on every expand shape (14x14 1024<-256 42.7 -> 33.1, 28x28 512<-128 60.3 -> 44.5, 56x56
//     256<-128 120 -> 103) but the one-call backward then loses the overlap of the shared launch with the weight gradient
//     (in the step 14x14 1024->256 70.3 -> 72.8, 28x28 128->512 61.0 -> 72.0, 7x7 2048->512 63.8 -> 72.6); only the
//     256 <- 64 dgrad of the 56x56 layers keeps it (backward 139.8 -> 133.0)
template <int MODE>
static bool rs_default(const RiglConvDesc* d, const RsPlan& p) {
  const int k = MODE == 0 ? d->cin : d->cout, n = MODE == 0 ? d->cout : d->cin;
  // (measured on the ResNet-50 shapes at batch 128 only: up to 8 column slices -- 32 slices re-read the rows 32 times, the
  // 512 -> 2048 regression above -- and the row counts of that table; anything else keeps its former body unless knob = 2)
  const int64_t M = (int64_t)d->n * d->h * d->w;
  if (MODE == 0) return k <= 256 && p.slices <= 8 && M >= 25088;
  return k == 64 && n == 256;
}
template <int MODE>
static bool rs_use(const RiglConvDesc* d, RsPlan* out = nullptr) {
  RsPlan p;
  if (!rs_plan<MODE>(d, p)) return false;
  const int knob = RIGL_TUNE("rowstream", 1);
  if (knob == 0 || (knob == 1 && !rs_default<MODE>(d, p))) return false;
  bool ready = false;
#define RS_READY(KC_, TN_, DG_) ready = rs_ready_i<KC_, TN_, DG_>();
  if (MODE == 0) { RS_DISPATCH(p, 0, RS_READY) } else { RS_DISPATCH(p, 1, RS_READY) if (ready) { RS_DISPATCH(p, 2, RS_READY) } }
#undef RS_READY
  if (ready && out) *out = p;
  return ready;
}
template <int MODE>
static void launch_rs(const RiglConvDesc* d, const RsPlan& p, const rigl_bf16* a_act, const rigl_bf16* b_w, const rigl_bf16* addend,
                      rigl_bf16* c_out, float* stats, hipStream_t st) {
  RsArgs a = {};
#ifdef RIGL_RS_TRACE
  { const char* e = getenv("RIGL_RS_TRACE_PTR"); a.TRACE = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  const int k = MODE == 0 ? d->cin : d->cout, n = MODE == 0 ? d->cout : d->cin;
  a.A = a_act; a.B = b_w; a.C = c_out; a.ADD = addend; a.STATS = stats;
  a.M = d->n * d->h * d->w; a.N = n; a.F = p.F; a.nfrag = p.nfrag; a.slices = p.slices; a.gprime = p.gprime;
  a.a_bytes = (uint32_t)((size_t)a.M * k * 2); a.b_bytes = (uint32_t)((size_t)n * k * 2); a.c_bytes = (uint32_t)((size_t)a.M * n * 2);
  const dim3 grid((unsigned)(p.slices * p.gprime)), blk(RS_THREADS);
#define RS_LAUNCH(KC_, TN_, DG_) RIGL_K_LAUNCH((k_rowstream<KC_, TN_, DG_>), grid, blk, (unsigned)(RsGeom<KC_, TN_>::SMEM), st, a);
  if (MODE == 0) { RS_DISPATCH(p, 0, RS_LAUNCH) } else if (addend) { RS_DISPATCH(p, 2, RS_LAUNCH) } else { RS_DISPATCH(p, 1, RS_LAUNCH) }
#undef RS_LAUNCH
}

// ---- the forward with the batch-norm apply + ReLU on the operand load (BNL kernels) -----------------------------------------
// Taken where the layer's forward is the row-streaming body anyway, except the K = 64 / TN = 4 variant (its four-fragment
// unrolled body spills 79 registers with the transform: measured twice as slow); knob "bn_on_load": 1 = these layers
// (default), 0 = none.
static bool rs_bnl_use(const RiglConvDesc* d, RsPlan* out = nullptr) {
  RsPlan p;
  if (!RIGL_TUNE("bn_on_load", 1) || !rs_use<0>(d, &p)) return false;
  if (p.kc == 1 && p.tn == 4) return false;
  bool ready = false;
#define RS_READY(KC_, TN_, DG_) ready = rs_ready_bnl<KC_, TN_>();
  RS_DISPATCH(p, 0, RS_READY)
#undef RS_READY
  if (ready && out) *out = p;
  return ready;
}
static void launch_rs_bnl(const RiglConvDesc* d, const RsPlan& p, const rigl_bf16* x_pre, const float* scale_shift, rigl_bf16* a_out,
                          const rigl_bf16* w_ohwi, rigl_bf16* y, float* stats, hipStream_t st) {
  RsArgs a = {};
  a.A = x_pre; a.B = w_ohwi; a.C = y; a.STATS = stats; a.BNP = scale_shift; a.A2 = a_out;
  a.M = d->n * d->h * d->w; a.N = d->cout; a.F = p.F; a.nfrag = p.nfrag; a.slices = p.slices; a.gprime = p.gprime;
  a.a_bytes = (uint32_t)((size_t)a.M * d->cin * 2); a.b_bytes = (uint32_t)((size_t)d->cout * d->cin * 2);
  a.c_bytes = (uint32_t)((size_t)a.M * d->cout * 2);
  const dim3 grid((unsigned)(p.slices * p.gprime)), blk(RS_THREADS);
#define RS_LAUNCH(KC_, TN_, DG_) RIGL_K_LAUNCH((k_rowstream<KC_, TN_, 0, true>), grid, blk, (unsigned)(RsGeom<KC_, TN_>::SMEM_BNL), st, a);
  RS_DISPATCH(p, 0, RS_LAUNCH)
#undef RS_LAUNCH
}
