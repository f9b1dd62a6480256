// K1, the "expand" 1x1 GEMMs: C[M][N] = A[M][K] x B[N][K]^T with a SHORT reduction (K = 64 / 128 / 256 channels) and a
// WIDE output (N = 256 .. 2048), stride 1, no padding (included inside namespace rigl::k1 of conv.hip):
//   forward of the bottleneck's third conv and of the group-1 projection (resnet_model.py:484-501: 64 -> 256 at 56x56,
//     128 -> 512 at 28x28, 256 -> 1024 at 14x14; pruning_layers.py:139-157),                  A = x,  B = OHWI shadow
//   dgrad of the bottleneck's first conv (256 <- 64, 512 <- 128, 1024 <- 256; autodiff of the same op),  A = dY, B = HWIO shadow
// These GEMMs move bytes, not flops (256 -> 1024 at 14x14: 13 GFLOP against 64 MB, 51 MB of it OUTPUT), and on the
// generic igemm body they ran at 2.5-2.8x their HBM bound: 1 568 tiles of 128x128 with an 8-step K loop each -- a
// workgroup spends ~14 us of prologue / DMA latency / epilogue on ~1 us of MFMA work.  Here:
//   * persistent 4-wave workgroups, TWO per CU, own (128-row tile of A, half of the N columns) units; a wave's 32 rows x K
//     of A live in REGISTERS (16 .. 64 VGPRs, fetched straight from global memory, the next unit's rows already in
//     flight in a second set).  Two independent workgroups per CU because every chunk ends in an epilogue as long as its
//     MFMA phase and a barrier: one 8-wave workgroup ran MFMA -> stores -> statistics in lockstep on all waves (2 950
//     cycles per chunk, 1 050 of them MFMA); two workgroups drift apart and fill each other's gaps;
//   * B streams through LDS in chunks of BN = 32 * (256 / K) rows x K (16 KB whatever K is: two stages, 4 LDS-DMA pieces
//     per wave per chunk, chunk c + 1 requested while chunk c is multiplied; at K = 64, N = 256 a workgroup's half of the
//     filter is one chunk and stays resident: no barrier at all in the unit loop), rows XOR-swizzled for conflict-free
//     ds_read_b128; workgroups walk their chunks in different rotations;
//   * every chunk is 16 MFMAs (v_mfma_f32_32x32x16_bf16, operands swapped: a lane holds 4 consecutive output channels) per
//     wave, then the wave stages its 32 x (BN / 2) bf16 outputs in a private LDS area and stores 16 bytes per lane -- row
//     segments of 64 / 128 / 256 bytes;
//   * forward: batch-norm statistics of the bf16 outputs.  The columns change with every chunk, so per-lane accumulators
//     would need 16 x 16 registers; instead the column sums of the staged tile come from four MFMAs on its transposing
//     read (see below) and go into a per-wave-row array in LDS (wave-private: plain read-modify-write, deterministic);
//     one partial row per workgroup at the end (rigl_conv2d_stats_parts = the grid).  dgrad: the addend
//     (bf16(bf16(acc) + addend)).
#pragma once

struct X1Args {
  const uint16_t* A;     // [M][K] bf16
  const uint16_t* B;     // [N][K] bf16
  uint16_t* C;           // [M][N] bf16
  const uint16_t* ADD;   // dgrad: optional [M][N]
  float* STATS;          // forward: optional [grid][2][N]
  int M, N, tiles_m;
  uint32_t a_bytes, b_bytes;
  unsigned long long* TRACE;   // development (-DRIGL_X1_TRACE): [grid][64] s_memtime stamps of wave 0
};
#ifdef RIGL_X1_TRACE
#define X1_STAMP(i_) { if (tid == 0 && P.TRACE && (i_) < 64) P.TRACE[blockIdx.x * 64 + (i_)] = __builtin_amdgcn_s_memtime(); }
#else
#define X1_STAMP(i_) { }
#endif

constexpr int X1_THREADS = 256;

template <int KC>
struct X1Geom {
  static constexpr int K = 64 * KC, TN = 4 / KC, BN = 32 * TN, CPR = K / 8, KS = K / 16;
  static constexpr int STAGE = BN * K * 2;                       // 16 384 bytes whatever KC
  static constexpr int SROW = 64 * TN + 16, STG_WAVE = 32 * SROW;
  static constexpr int STG_OFF = 2 * STAGE, STATS_OFF = STG_OFF + 4 * STG_WAVE;
  static int smem(int n, bool stats) { return STATS_OFF + (stats ? 4 * n * 2 * 4 : 0); }
};

template <int KC, bool DGRAD>
__global__ __launch_bounds__(X1_THREADS, 2) void k_x1x1(X1Args P) {
  using G = X1Geom<KC>;
  constexpr int K = G::K, TN = G::TN, BN = G::BN, CPR = G::CPR, KS = G::KS, STAGE = G::STAGE, SROW = G::SROW;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_x1[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;                                 // 4 wave rows; a wave covers all BN columns of a chunk
  const int hi = lane >> 5;
  unsigned char* const stg = smem_x1 + G::STG_OFF + wave * G::STG_WAVE;
  float* const stl = reinterpret_cast<float*>(smem_x1 + G::STATS_OFF) + wm * P.N * 2;      // this wave row's [N][2]
  const bool stats = !DGRAD && P.STATS != nullptr;
  const __amdgpu_buffer_rsrc_t rsrcA = make_rsrc(P.A, P.a_bytes), rsrcB = make_rsrc(P.B, P.b_bytes);
  const int NC = (P.N / 2) / BN;                        // chunks per unit (a unit = a 128-row tile x half of the columns)
  const bool resident = NC == 1;                        // this workgroup's half of the filter is one chunk: it stays in stage 0
  const int nh = (int)(blockIdx.x & 1u);                // (the grid is even: a workgroup keeps its column half)
  // Every workgroup walks the chunks in its OWN rotation (c0, c0 + 1, ... mod NC): in lockstep -- all ~200 workgroups
  // requesting the same 32 KB of the filter at the same moment -- the L2 channels holding that chunk serialised the chip
  // (1.5 us per chunk whatever was ablated).
  const int c0 = (int)((blockIdx.x >> 1) % (unsigned)NC);
#define X1_CHUNK(c_) (nh * NC + ((c_) + c0) % NC)

  // B chunk c -> stage: 16 pieces of 1 KB, 4 per wave; piece position p = piece * 64 + lane -> (row, chunk) of the LDS
  // image; the source chunk is XORed with the row's swizzle (3 bits at 8 chunks per row, else 4)
#define X1_SWZ(row_) (CPR == 8 ? (((row_) >> 1) & 7) : ((row_) & 15))
#define X1_ISSUE_B(c_, st_)                                                                              \
  {                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
      const int piece = j * 4 + wave, p = piece * 64 + lane, row = p / CPR, ch = p % CPR;                \
      const int src = ((c_) * BN + row) * CPR + (ch ^ X1_SWZ(row));                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(smem_x1 + (st_) * STAGE + piece * 1024), \
                                               16, src * 16, 0, 0, 0);                                   \
    }                                                                                                    \
  }
  // A rows of a tile -> registers (MFMA "B" operand: lane l = column l & 31 = row of the tile, 8 consecutive k at
  // 16 * ks + 8 * (l >> 5)); rows beyond M read zeros (buffer range)
#define X1_LOAD_A(dst_, t_)                                                                              \
  {                                                                                                      \
    const uint32_t r_ = (uint32_t)((t_) * 128 + wm * 32 + (lane & 31));                                  \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                  \
      const uint4 v_ = buf_load16(rsrcA, (r_ * (uint32_t)K + (uint32_t)(ks * 16 + hi * 8)) * 2u);        \
      dst_[ks] = __builtin_bit_cast(bf16x8, v_);                                                         \
    }                                                                                                    \
  }
  if (stats) {
    for (int i = tid; i < 4 * P.N * 2; i += X1_THREADS) reinterpret_cast<float*>(smem_x1 + G::STATS_OFF)[i] = 0.f;
  }
  int t = blockIdx.x >> 1;                              // units u = blockIdx.x, + gridDim.x, ...: tile u >> 1, column half u & 1
  const int t_step = (int)(gridDim.x >> 1);
  X1_STAMP(0);
  bf16x8 a_cur[KS], a_nxt[KS];
  if (t < P.tiles_m) {
    X1_ISSUE_B(X1_CHUNK(0), 0);
    X1_LOAD_A(a_cur, t);
  }
  // fragment row offsets of this lane in a B stage (the swizzle depends on the row's low bits only = the lane)
  const int n_lane = lane & 31;
  const int swz = X1_SWZ(n_lane);
  int stage = 0;
  for (; t < P.tiles_m; t += t_step) {
    const int tn = t + t_step;
    const bool more = tn < P.tiles_m;
    if (more && resident) X1_LOAD_A(a_nxt, tn);        // (no waits in the resident tile loop: in flight for the whole tile)
    for (int c = 0; c < NC; ++c) {
      X1_STAMP(1 + c * 6);
      if (!resident || (t == (int)(blockIdx.x >> 1) && c == 0)) {
        // chunk c has landed (every wave waits for its own pieces, then all meet); the other stage is free again
#if defined(RIGL_X1_ABLATE) && (RIGL_X1_ABLATE & 2)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // timing experiment: does not wait for the stores (racy)
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        X1_STAMP(2 + c * 6);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        X1_STAMP(3 + c * 6);
        if (!resident) {
          const bool last = c + 1 == NC;
          if (!last) X1_ISSUE_B(X1_CHUNK(c + 1), stage ^ 1)
          else if (more) X1_ISSUE_B(X1_CHUNK(0), stage ^ 1)
          // the next tile's rows are requested behind chunk 0's barrier: the vmcnt(0) of chunk 1 finds them an MFMA phase old
          if (c == 0 && more) X1_LOAD_A(a_nxt, tn);
        }
      }
      const unsigned char* const bs = smem_x1 + stage * STAGE;
      // accumulators: one per n-tile, and at TN == 1 (K = 256) TWO for the one n-tile (even / odd k-steps, added at the end):
      // a single dependent chain of 16 MFMAs issued every ~62 cycles instead of every 32
      constexpr int NACC = TN == 1 ? 2 : TN;
      f32x16 acc[NACC];
#pragma unroll
      for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      // 16 MFMAs per chunk in groups of four; the fragments of group g + 1 are requested above the MFMAs of group g (left to
      // itself the compiler read one fragment, waited, multiplied: 2 750 cycles per chunk)
      {
        bf16x8 bq[2][4];
#define X1_READ_GROUP(g_, buf_)                                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                  \
          const int f_ = (g_) * 4 + u, ks_ = f_ / TN, j_ = f_ % TN;                                      \
          bq[buf_][u] = *reinterpret_cast<const bf16x8*>(bs + (j_ * 32 + n_lane) * (K * 2) + (((2 * ks_ + hi) ^ swz) << 4)); \
        }
#define X1_MFMA_GROUP(g_, buf_)                                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                  \
          const int f_ = (g_) * 4 + u, ks_ = f_ / TN, j_ = f_ % TN, a_ = TN == 1 ? (ks_ & 1) : j_;       \
          acc[a_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[buf_][u], a_cur[ks_], acc[a_], 0, 0, 0);  \
        }
        X1_READ_GROUP(0, 0);
        X1_READ_GROUP(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        X1_MFMA_GROUP(0, 0);
        X1_READ_GROUP(2, 0);
        __builtin_amdgcn_sched_barrier(0);
        X1_MFMA_GROUP(1, 1);
        X1_READ_GROUP(3, 1);
        __builtin_amdgcn_sched_barrier(0);
        X1_MFMA_GROUP(2, 0);
        X1_MFMA_GROUP(3, 1);
#undef X1_READ_GROUP
#undef X1_MFMA_GROUP
      }
      if (TN == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] += acc[1][e];
      }
      X1_STAMP(4 + c * 6);
      // ---- epilogue of the chunk: D row = (e & 3) + 8 * (e >> 2) + 4 * hi -> channel of n-tile j, column = lane & 31 -> row
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 lo = {acc[j][4 * q], acc[j][4 * q + 1]}, hi2 = {acc[j][4 * q + 2], acc[j][4 * q + 3]};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          *reinterpret_cast<uint2*>(stg + (lane & 31) * SROW + (j * 32 + 8 * q + 4 * hi) * 2) = pk;
        }
      const int ncol0 = X1_CHUNK(c) * BN;                 // first output column of the chunk
      const int row0 = t * 128 + wm * 32;
      constexpr int CHR = 4 * TN, ITERS = 2 * TN;       // 16-byte chunks per staged row, store iterations
      uint4 addv[ITERS];
      if (DGRAD && P.ADD) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int idx = it * 64 + lane, row = idx / CHR, ch = idx % CHR;
          const int m = row0 + row;
          addv[it] = m < P.M ? *reinterpret_cast<const uint4*>(P.ADD + (int64_t)m * P.N + ncol0 + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = it * 64 + lane, row = idx / CHR, ch = idx % CHR;
        const int m = row0 + row;
        uint4 v = *reinterpret_cast<const uint4*>(stg + row * SROW + ch * 16);
        if (m < P.M) {
          if (DGRAD && P.ADD) {
            const uint4 q4 = addv[it];
            v.x = add_bf16x2(v.x, q4.x); v.y = add_bf16x2(v.y, q4.y); v.z = add_bf16x2(v.z, q4.z); v.w = add_bf16x2(v.w, q4.w);
          }
#if !(defined(RIGL_X1_ABLATE) && (RIGL_X1_ABLATE & 1))
          store16(P.C + (int64_t)m * P.N + ncol0 + ch * 8, v);
#else
          if (v.x == 0x12345678u) store16(P.C + (int64_t)m * P.N + ncol0 + ch * 8, v);   // timing experiment: no output stores
#endif
        }
      }
      X1_STAMP(5 + c * 6);
#if defined(RIGL_X1_ABLATE) && (RIGL_X1_ABLATE & 4)
      if (false) {                                      // timing experiment: no statistics
#else
      if (stats) {
#endif
        // Column sums and sums of squares of the staged bf16 tile on the matrix cores: a transposing read gives the
        // fragment Yf with k = the 32 rows (lane = column); sum_m Y[m][n] = (ones x Yf)[.][n], and
        // sum_m Y[m][n]^2 = the diagonal of Yf^T x Yf -- the SAME fragment serves as both MFMA operands (products of
        // two bf16 values are exact in fp32, the accumulation order is the hardware's fixed one: deterministic).  The
        // columns change with every chunk, so the two numbers of a column are added into this wave row's LDS array by the
        // one lane that holds the column's diagonal element.  (One lane per column reading its 32 staged rows cost 10 us
        // of the 39 us of the 256 -> 1024 layer.)
        const int g = lane >> 4, j16 = lane & 15;
        const int t_row = 8 * (g >> 1) + (j16 >> 2), t_ch = 2 * (g & 1) + ((j16 >> 1) & 1), t_half = (j16 & 1) * 8;
        bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
        const int n31 = lane & 31;
        const int esel = (n31 & 3) | ((n31 >> 3) << 2);
        const bool owner = hi == ((n31 >> 2) & 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 as, aq;
#pragma unroll
          for (int e = 0; e < 16; ++e) as[e] = aq[e] = 0.f;
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned char* const p0 = stg + (k2 * 16 + t_row) * SROW + (j * 4 + t_ch) * 16 + t_half;
            const bf16x8 yf = lds_read_tr_pair(p0, p0 + 4 * SROW);
            as = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yf, as, 0, 0, 0);
            aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf, yf, aq, 0, 0, 0);
          }
          float qd = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) qd = (e == esel) ? aq[e] : qd;
          if (owner) {
            float* const p = stl + (ncol0 + j * 32 + n31) * 2;
            p[0] += as[0]; p[1] += qd;
          }
        }
      }
      X1_STAMP(6 + c * 6);
      if (!resident) stage ^= 1;
    }
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
    }
  }
#undef X1_ISSUE_B
#undef X1_LOAD_A
#undef X1_SWZ
#undef X1_CHUNK
  if (stats) {
    __syncthreads();
    const float* const all = reinterpret_cast<const float*>(smem_x1 + G::STATS_OFF);
    for (int i = tid; i < 2 * P.N; i += X1_THREADS) {
      const int k = i / P.N, n = i % P.N;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += all[(w4 * P.N + n) * 2 + k];            // wave rows in a fixed order
      P.STATS[((int64_t)blockIdx.x * 2 + k) * P.N + n] = s;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Legal: 1x1, stride 1, no padding; reduction 64 / 128 / 256 channels; output channels a multiple of two chunks and at least
// 4x the reduction ("expand"); at least 8 192 rows.  MODE 0 = forward (reduction = cin), 1 = dgrad (reduction = cout).
// Measured at batch 128, operands from HBM (tools/x1_bench.py, gpurun r4n-r4r; us, igemm body -> this kernel):
//   forward + statistics  56x56 64->256 65.4 -> 59.7   28x28 128->512 50.1 -> 41.7   14x14 256->1024 33.2 -> 32.6
//   dgrad + addend        56x56 256<-64 99.6 -> 106    28x28 512<-128 58.8 -> 69.3   14x14 1024<-256 41.8 -> 45.2
// so the default ("x1x1" = 1) takes the FORWARDS with reductions of 64 and 128 channels only; "x1x1" = 2 adds the 256-channel
// forwards, "x1x1_dgrad" = 1 the dgrads (parity-tested, slower: the addend doubles the bytes and the shared launch's
// overlap with the weight gradient is lost).  Why not more (tools/x1_trace.py, 14x14 256->1024): a chunk of 32 columns
// is ~2 800 cycles per workgroup -- ~500 of MFMA, ~500 of issuing the next chunk's four DMA pieces, ~750 of staging and
// stores, ~500 of statistics, ~400 of wait + barrier -- and a workgroup walks 16 of them behind a 10 000-cycle prologue
// (first rows from HBM): the chain per workgroup, not the pipes, sets the time; two workgroups per CU did not shorten it.
// "x1x1" = 0 turns the kernel off (the layer then runs on the igemm bodies in every entry point).
static inline int x1x1_kc(int k) { return k == 64 ? 1 : (k == 128 ? 2 : (k == 256 ? 4 : 0)); }
template <int MODE>
static bool x1x1_legal(const RiglConvDesc* d) {
  if (d->kh != 1 || d->kw != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_top || d->pad_left) return false;
  const int k = MODE == 0 ? d->cin : d->cout, n = MODE == 0 ? d->cout : d->cin;
  const int kc = x1x1_kc(k);
  if (!kc || n < 4 * k || n % (2 * 32 * (4 / kc)) || n > 4096) return false;
  if ((int64_t)d->n * d->h * d->w < 128 * 64) return false;
  const int knob = RIGL_TUNE("x1x1", 1);
  if (knob == 0) return false;
  if (MODE == 1) return RIGL_TUNE("x1x1_dgrad", 0) != 0;
  return kc < 4 || knob >= 2;
}
// units = (128-row tile, column half); persistent workgroups, two per CU (an even number: a workgroup keeps its half)
static inline int x1x1_grid(const RiglConvDesc* d) {
  const int units = 2 * (int)(((int64_t)d->n * d->h * d->w + 127) / 128);
  return units < 2 * num_cus() ? units : 2 * num_cus();
}
template <int KC, bool DGRAD>
static bool x1x1_ready_i() {
  static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_x1x1<KC, DGRAD>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2) == hipSuccess;
  return ready;
}
template <int MODE>
static bool x1x1_use(const RiglConvDesc* d) {
  if (!x1x1_legal<MODE>(d)) return false;
  const int n = MODE == 0 ? d->cout : d->cin;
  switch (x1x1_kc(MODE == 0 ? d->cin : d->cout)) {
    case 1: return X1Geom<1>::smem(n, MODE == 0) <= 160 * 1024 / 2 && x1x1_ready_i<1, MODE == 1>();
    case 2: return X1Geom<2>::smem(n, MODE == 0) <= 160 * 1024 / 2 && x1x1_ready_i<2, MODE == 1>();
    case 4: return X1Geom<4>::smem(n, MODE == 0) <= 160 * 1024 / 2 && x1x1_ready_i<4, MODE == 1>();
    default: return false;
  }
}
template <int MODE>
static void launch_x1x1(const RiglConvDesc* d, const rigl_bf16* a_act, const rigl_bf16* b_w, const rigl_bf16* addend, rigl_bf16* c_out,
                        float* stats, hipStream_t st) {
  X1Args a = {};
#ifdef RIGL_X1_TRACE
  { const char* e = getenv("RIGL_X1_TRACE_PTR"); a.TRACE = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
  const int k = MODE == 0 ? d->cin : d->cout, n = MODE == 0 ? d->cout : d->cin;
  a.A = a_act; a.B = b_w; a.C = c_out; a.ADD = addend; a.STATS = stats;
  a.M = d->n * d->h * d->w; a.N = n; a.tiles_m = (a.M + 127) / 128;
  a.a_bytes = (uint32_t)((size_t)a.M * k * 2); a.b_bytes = (uint32_t)((size_t)n * k * 2);
  const dim3 grid((unsigned)x1x1_grid(d)), blk(X1_THREADS);
  switch (x1x1_kc(k)) {
    case 1: RIGL_K_LAUNCH((k_x1x1<1, MODE == 1>), grid, blk, (unsigned)X1Geom<1>::smem(n, MODE == 0 && stats), st, a); break;
    case 2: RIGL_K_LAUNCH((k_x1x1<2, MODE == 1>), grid, blk, (unsigned)X1Geom<2>::smem(n, MODE == 0 && stats), st, a); break;
    default: RIGL_K_LAUNCH((k_x1x1<4, MODE == 1>), grid, blk, (unsigned)X1Geom<4>::smem(n, MODE == 0 && stats), st, a); break;
  }
}
