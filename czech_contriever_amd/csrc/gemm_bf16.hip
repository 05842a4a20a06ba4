// gemm_bf16.hip — bf16 MFMA GEMM for gfx950, fp32 accumulate.
//
// One 256x256 output tile at a time per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
// K-step 64, v_mfma_f32_16x16x32_bf16.  Operand tiles go HBM -> LDS with global_load_lds_dwordx4
// (no VGPR staging), double-buffered: the DMA of K-tile t+1 is in flight while tile t is
// multiplied; one barrier per K-tile.  LDS: 2 buffers x (A 32 KiB + B 32 KiB) = 128 KiB.
// For the bf16 epilogues a workgroup walks over several output tiles (one workgroup per CU): the
// epilogue stages through the second buffer while the next tile's first K-tile lands in the first.
// Two K-loops share everything else: the plain one (software-pipelined fragment reads, one barrier per
// K-tile) and, for the K-major forward products with long K or wide N, the ping-pong one (PP = true:
// two wave groups one barrier apart, quadrant phases, half-tile LDS refills; described at its top).
//
// Each operand may be stored with the reduction index contiguous ("K-major": X[M][K], W[N][K]) or
// with the free index contiguous ("MN-major": the operand of the weight-gradient and input-
// gradient products).  K-major tiles are read with ds_read_b128, MN-major tiles with the
// transposing ds_read_b64_tr_b16, so no operand is ever transposed in HBM.
//   LDS images (conflict-free for the reads used):
//     K-major  : [256 rows][8 chunks of 16 B], physical chunk = c ^ ((row >> 1) & 7)
//     MN-major : 2 blocks of [64 k-rows][16 chunks], physical chunk = c ^ (((k&3)<<2) | ((k>>2)&3))
//   global_load_lds writes LDS linearly (wave base + lane*16), so the swizzle is applied to the
//   per-lane SOURCE address and again on the read.
// 16-B chunks that fall outside the matrix are fetched from a zero page instead (per-lane source),
// which gives exact tails in M, N and K without a second code path.
//
// The MFMA is issued as D = B_frag x A_frag, i.e. the accumulator tile is C^T: a lane owns one
// output row m and four consecutive columns n, so bias loads are float4 and the bf16 tile is put
// into LDS with 8-byte writes, then streamed out as whole 128-B row segments.

#include "czc_common.h"
#include "gemm_args.h"

namespace czc {

constexpr int BK = 64;
// Tile configuration: WM x WN waves, each owning (TM*16) x 64 outputs.
//   big   = 2x4 waves, TM 8: 256x256 tile, 512 threads, 128 KiB LDS, one workgroup per CU
//   small = 2x2 waves, TM 4: 128x128 tile, 256 threads,  64 KiB LDS, two workgroups per CU (a workgroup's
//           epilogue stores and tile prologue overlap the other's MFMA work)
//   (measured and dropped: 2x2 waves with TM 8, TN 8 -- one wave per SIMD owning 128x128 outputs in 256
//    accumulator registers -- ran 0.65-0.75x the speed of `big` under compiler scheduling; `small` is 3-7 % slower
//    than `big` on these shapes and is kept for small problems)
template <int WM_, int WN_, int TM_, int TN_ = 4>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  static constexpr int TILE_A = BM * BK * 2, TILE_B = BN * BK * 2;
  static constexpr int BUF = TILE_A + TILE_B;
  static constexpr int LDS = 2 * BUF;
};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// 16 B per lane, global -> LDS (destination = dst + 16*lane, dst wave-uniform).  Issued as inline asm on purpose:
// given the builtin, the compiler tracks the pending LDS DMA and puts an s_waitcnt vmcnt(0) in front of the next
// ds_read_b64_tr_b16 (it cannot prove that read does not alias the DMA's destination), which exposes the whole
// global latency once per K-tile in the MN-major kernels.  The K-loop orders the two itself: vmcnt(0) + barrier
// before a staged buffer is first read.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is "reserved": it is ours to set here, nothing else in this file uses it
__device__ __forceinline__ void glds16(const void* src, lptr_t dst) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
}
// The same with a position in the instruction stream: `pin` (an MFMA operand) is declared read and written, so the
// DMA is placed before the MFMAs that consume it, and `after` (an operand of the MFMAs before it, dead afterwards)
// likewise, so the DMA follows the last of those.  An LDS-DMA piece costs its
// wave 60-180 issue cycles; eight of them back to back after the barrier left the MFMA pipe idle on both waves of a
// SIMD, spread between the rows of the last phase they issue under the other wave's MFMAs.
template <class PIN, class AFTER>
__device__ __forceinline__ void glds16_pin(const void* src, lptr_t dst, PIN& pin, AFTER& after) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)dst);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" : "+v"(pin), "+v"(after) : "v"(src), "s"(l) : "memory", "m0");
}
// the same as a streaming (nt) read (gemm_kernel<..., ANT>: the A rows of a long-K product with three column tiles)
template <class PIN, class AFTER>
__device__ __forceinline__ void glds16_pin_nt(const void* src, lptr_t dst, PIN& pin, AFTER& after) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)dst);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt" : "+v"(pin), "+v"(after) : "v"(src), "s"(l) : "memory", "m0");
}
// The same from a wave-uniform 64-bit base (SGPRs) plus a 32-bit per-lane byte offset: the offsets of a workgroup's
// DMA pieces do not change along K, only the base does, so the K-loop spends scalar adds instead of ~13 vector
// instructions per piece on addresses (stamps: the eight pieces after the barrier took a wave 650-1250 cycles of a
// 3400-cycle K-tile).
template <class PIN, class AFTER>
__device__ __forceinline__ void glds16_sbase(unsigned off, const void* base, lptr_t dst, PIN& pin, AFTER& after) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)dst);
  asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" : "+v"(pin), "+v"(after) : "v"(off), "s"(base), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

// per-lane byte offset of DMA piece i of a full, interior tile, relative to (operand base + the K-tile's advance)
template <int ROWS, int NW>
__device__ __forceinline__ unsigned piece_off_kmajor(int ld, int row0, int wave, int lane, int i) {
  constexpr int PER = ROWS / 8 / NW;
  const int q = wave * PER + i, r = 8 * q + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
  return (unsigned)(((size_t)(row0 + r) * ld + c * 8) * 2);
}
template <int COLS, int NW>
__device__ __forceinline__ unsigned piece_off_mnmajor(int ld, int col0, int wave, int lane, int i) {
  constexpr int PER = COLS / 128 * 16 / NW;
  const int q = wave * PER + i, blk = q >> 4, kr = 4 * (q & 15) + (lane >> 4);
  const int ch = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
  return (unsigned)(((size_t)kr * ld + col0 + blk * 128 + ch * 8) * 2);
}

// ---- staging -------------------------------------------------------------------------------
// K-major operand: X[rows][ld], tile = rows [row0, row0+256) x k [k0, k0+64)
// 1-KiB piece q of a K-major tile: rows 8q .. 8q+7
template <class PIN, class AFTER>
__device__ __forceinline__ void stage_kmajor_q(const bf16_t* X, int ld, int row0, int nrows, int k0, int kdim,
                                               unsigned char* tile, int q, int lane, PIN& pin, AFTER& after) {
  const int r = 8 * q + (lane >> 3);
  const int c = (lane & 7) ^ ((r >> 1) & 7);
  const int gr = row0 + r, gk = k0 + c * 8;
  const void* src = (gr < nrows && gk < kdim) ? (const void*)(X + (size_t)gr * ld + gk) : (const void*)g_zero16;
  glds16_pin(src, (lptr_t)(tile + q * 1024), pin, after);
}
// 1-KiB piece q of an MN-major tile: block q>>4 (128 columns), k-rows 4*(q&15) .. +3
template <class PIN, class AFTER>
__device__ __forceinline__ void stage_mnmajor_q(const bf16_t* X, int ld, int col0, int ncols, int k0, int kdim,
                                                unsigned char* tile, int q, int lane, PIN& pin, AFTER& after) {
  const int blk = q >> 4;
  const int kr = 4 * (q & 15) + (lane >> 4);
  const int ch = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
  const int gk = k0 + kr, gc = col0 + blk * 128 + ch * 8;
  const void* src = (gk < kdim && gc < ncols) ? (const void*)(X + (size_t)gk * ld + gc) : (const void*)g_zero16;
  glds16_pin(src, (lptr_t)(tile + q * 1024), pin, after);
}
// byte offsets of those pieces for interior tiles (see piece_off_*): relative to operand base + the K-tile's advance
__device__ __forceinline__ unsigned piece_off_kmajor_q(int ld, int row0, int q, int lane) {
  const int r = 8 * q + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
  return (unsigned)(((size_t)(row0 + r) * ld + c * 8) * 2);
}
__device__ __forceinline__ unsigned piece_off_mnmajor_q(int ld, int col0, int q, int lane) {
  const int blk = q >> 4, kr = 4 * (q & 15) + (lane >> 4);
  const int ch = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
  return (unsigned)(((size_t)kr * ld + col0 + blk * 128 + ch * 8) * 2);
}

template <int ROWS, int NW, class PIN, class AFTER>
__device__ __forceinline__ void stage_kmajor_piece(const bf16_t* X, int ld, int row0, int nrows, int k0, int kdim,
                                                   unsigned char* tile, int wave, int lane, int i, PIN& pin, AFTER& after) {
  constexpr int PER = ROWS / 8 / NW;
  const int q = wave * PER + i;          // 1-KiB piece: rows 8q .. 8q+7
  const int r = 8 * q + (lane >> 3);
  const int c = (lane & 7) ^ ((r >> 1) & 7);
  const int gr = row0 + r, gk = k0 + c * 8;
  const void* src = (gr < nrows && gk < kdim) ? (const void*)(X + (size_t)gr * ld + gk) : (const void*)g_zero16;
  glds16_pin(src, (lptr_t)(tile + q * 1024), pin, after);
}
template <int ROWS, int NW>
__device__ __forceinline__ void stage_kmajor(const bf16_t* X, int ld, int row0, int nrows, int k0, int kdim,
                                             unsigned char* tile, int wave, int lane) {
  constexpr int PER = ROWS / 8 / NW;
  int nopin = 0, nopin2 = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) stage_kmajor_piece<ROWS, NW>(X, ld, row0, nrows, k0, kdim, tile, wave, lane, i, nopin, nopin2);
}
// MN-major operand: X[kdim][ld] (free index contiguous), tile = k [k0,k0+64) x cols [col0, col0+256)
template <int COLS, int NW, class PIN, class AFTER>
__device__ __forceinline__ void stage_mnmajor_piece(const bf16_t* X, int ld, int col0, int ncols, int k0, int kdim,
                                                    unsigned char* tile, int wave, int lane, int i, PIN& pin, AFTER& after) {
  constexpr int PER = COLS / 128 * 16 / NW;
  const int q = wave * PER + i;          // piece q: block q>>4, k-rows 4*(q&15) .. +3
  const int blk = q >> 4;
  const int kr = 4 * (q & 15) + (lane >> 4);
  const int ch = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
  const int gk = k0 + kr, gc = col0 + blk * 128 + ch * 8;
  const void* src = (gk < kdim && gc < ncols) ? (const void*)(X + (size_t)gk * ld + gc) : (const void*)g_zero16;
  glds16_pin(src, (lptr_t)(tile + q * 1024), pin, after);
}
template <int COLS, int NW>
__device__ __forceinline__ void stage_mnmajor(const bf16_t* X, int ld, int col0, int ncols, int k0, int kdim,
                                              unsigned char* tile, int wave, int lane) {
  constexpr int PER = COLS / 128 * 16 / NW;
  int nopin = 0, nopin2 = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) stage_mnmajor_piece<COLS, NW>(X, ld, col0, ncols, k0, kdim, tile, wave, lane, i, nopin, nopin2);
}

// ---- fragment reads --------------------------------------------------------------------------
// K-major: rows [rbase + 16*i + (lane&15)], k-step ks (32 wide): lane reads k = 32ks + 8(lane>>4) .. +7
__device__ __forceinline__ bf16x8 frag_kmajor(const unsigned char* tile, int rbase, int i, int ks, int lane) {
  const int r = rbase + 16 * i + (lane & 15);
  const int c = (4 * ks + (lane >> 4)) ^ (((lane & 15) >> 1) & 7);
  return *(const bf16x8*)(tile + r * 128 + (c << 4));
}
// MN-major: columns [cbase + 16*i + (lane&15)] of the 256-wide tile, k-step ks
__device__ __forceinline__ bf16x8 frag_mnmajor(const unsigned char* tile, int cbase, int i, int ks, int lane) {
  const int col = cbase + 16 * i;                 // multiple of 16
  const int blk = col >> 7, cb = col & 127;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row1 = 32 * ks + 8 * g + q, row2 = row1 + 4;
  const int ch = (cb >> 3) + (p >> 1);
  const int f1 = ((row1 & 3) << 2) | ((row1 >> 2) & 3);
  const int f2 = ((row2 & 3) << 2) | ((row2 >> 2) & 3);
  const unsigned char* b = tile + blk * 16384 + 8 * (p & 1);
  typedef s16x4 __attribute__((address_space(3)))* trp_t;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(b + 256 * row1 + 16 * (ch ^ f1)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(b + 256 * row2 + 16 * (ch ^ f2)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void swizzle_tile(int bid, int nwg, int tiles_m, int tiles_n, int group_n, int& tm, int& tn) {
  // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of the tile order, so the
  // blocks that run together on an XCD share operand panels in its L2.  The order is N-grouped: group_n column tiles
  // at a time, all row tiles under them, then the next group -- the group's B panels (group_n x 256 x K) stay in
  // the 4 MiB L2 for the whole sweep instead of being re-fetched for every row of tiles.  Bijective for any shape.
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per_group = group_n * tiles_m;
  const int g = swz / per_group, rem = swz - g * per_group;
  const int gw = min(group_n, tiles_n - g * group_n);   // the last group may be narrower
  tm = rem / gw;
  tn = g * group_n + (rem - tm * gw);
}

// 16-byte output store; pol = 1: sc1 (the line is not kept in this XCD's L2: a tile's 64-128 KiB per CU would push the
// B panels and the shared A rows out of the 4 MiB the K-loop lives on), 2: nt
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_st;
__device__ __forceinline__ void store16(u32x4_st v, __amdgpu_buffer_rsrc_t rc, unsigned off, int pol) {
#ifdef CZC_EPI_NOSTORE   // diagnostic build: everything of the epilogue but the global stores (pol is a run-time value >= 0)
  if (pol >= 0) return;
#endif
  if (pol == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 16);
  else if (pol == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 0);
}

// ANT (with PP, K-major operands): the A operand's full interior K-tiles are requested as streaming (nt) reads.  For the
// long-K products with three column tiles (FFN-down, the input-gradient products): their A rows (402 / 302 MB) are read
// once per launch and shared by three workgroups only; read with the default policy they push out of the Infinity Cache
// what the kernels behind them read next (the residual stream's rows).  A compile-time parameter: a wave-uniform branch
// per DMA piece cost every ping-pong product 2-3 %, and on the wide K = 768 products nt evicts rows that 9-12 column tiles
// share.
template <int AL, int BL, int EPI, class CF, bool PP = false, bool ANT = false>
__global__ __launch_bounds__(CF::NT, 2) void gemm_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = CF::BM, BN = CF::BN, TM = CF::TM, TN = CF::TN, WN = CF::WN;
  constexpr int WNC = TN * 16;  // columns per wave
  constexpr int TILE_BYTES_A = CF::TILE_A, BUF_BYTES = CF::BUF;

  // Split-K launches (f32 slabs, one workgroup per (tile, split); this covers EVERY split-K EPI_F32 caller -- the weight
  // gradients and the head's dq alike -- whose tiles inside a split are then walked row-major instead of by swizzle_tile):
  // workgroups are dispatched round-robin over the 8 XCDs in linear order, so with split = blockIdx.y the 36 tiles of one split -- which share that split's K-range of BOTH
  // operands -- were spread over all 8 L2s, and every L2 fetched nearly every panel (PMC: 1.43 GB from beyond L2 per
  // weight-gradient launch against 0.50 GB of operands).  Here each XCD takes a contiguous run of the split-major
  // (split, tile) list instead: an L2 sees one or two splits' panels, each tile row's A block and the B blocks once.
  int split = blockIdx.y, tile0 = blockIdx.x;
#ifdef CZC_NO_XCD_SPLIT   // A/B builds: the mapping before round 3's last change
  const bool by_xcd = false;
#else
  // (a.xcd8: set by the host from the CU count.  The remap below is a bijection of (split, tile) for any grid, so results
  //  do not depend on it; on a partitioned device or another part it would only scatter locality, so it stays off there)
  const bool by_xcd = EPI == EPI_F32 && gridDim.y > 1 && a.xcd8;
#endif
  if (by_xcd) {
    const int nt_ = a.tiles_m * a.tiles_n;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    split = swz / nt_;
    tile0 = swz - split * nt_;
  }
  const int kt_total = (a.K + BK - 1) / BK;
  const int kt_begin = split * a.ktiles_per_split;
  int kt_end = kt_begin + a.ktiles_per_split;
  if (kt_end > kt_total) kt_end = kt_total;
  const int nt = kt_end - kt_begin;

  // Persistent over output tiles: the workgroup takes tiles blockIdx.x, + gridDim.x, ... (the host launches one
  // workgroup per resident slot for the bf16 epilogues, one per tile otherwise).  A bf16 epilogue stages through the
  // second operand buffer only, so the first K-tile of the workgroup's NEXT output tile is already on its way into
  // the first buffer while this tile is written out: the ~3 us a tile otherwise waits for its first operands
  // (12 K-tiles of 1.7 us at K = 768) are hidden behind the stores.
  constexpr bool PREFETCH = (EPI != EPI_F32 && EPI != EPI_LSE && EPI != EPI_FILTER);
  const int ntiles = a.tiles_m * a.tiles_n;
  bool primed = false;   // K-tile 0 of the coming output tile was requested during the previous epilogue
  for (int tile = tile0; tile < ntiles; tile += gridDim.x) {
  // lane-derived addresses are recomputed per tile from an opaque copy of the thread id: hoisted out of the tile loop
  // they would stay live through the epilogue, where the accumulators, the prefetched second operand and the staged
  // segments already fill the register file (the hoisted form spilled ~100 dwords per lane and tile)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int tm, tn;
  if (by_xcd) { tm = tile / a.tiles_n; tn = tile - tm * a.tiles_n; }   // row-major inside a split: a tile row's A block is shared by its column tiles
  else swizzle_tile(tile, ntiles, a.tiles_m, a.tiles_n, a.group_n, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  // Two products in one launch (the f32 slab epilogue only; gemm_args.h): output rows from g_rows on are the SECOND
  // product's, with operands, leading dimensions and row count of their own (same N and K).  gA .. aM are what the
  // operand staging below uses; the output side keeps row0 and a.M.
  const bool grp2 = EPI == EPI_F32 && a.A2 != nullptr && row0 >= a.g_rows;
  const bf16_t* const gA = grp2 ? a.A2 : a.A;
  const bf16_t* const gB = grp2 ? a.B2 : a.B;
  const int glda = grp2 ? a.lda2 : a.lda, gldb = grp2 ? a.ldb2 : a.ldb;
  const int arow0 = grp2 ? row0 - a.g_rows : row0;
  const int aM = (EPI == EPI_F32 && a.A2 != nullptr) ? (grp2 ? a.M - a.g_rows : a.g_rows) : a.M;
  auto stage_at = [&](int r0, int c0, int t, int buf) {   // (r0: operand rows, i.e. arow0 for the current tile)
    unsigned char* ta = smem + buf * BUF_BYTES;
    unsigned char* tb = ta + TILE_BYTES_A;
    const int k0 = (kt_begin + t) * BK;
    if (AL == 0) stage_kmajor<BM, CF::NW>(gA, glda, r0, aM, k0, a.K, ta, wave, lane);
    else         stage_mnmajor<BM, CF::NW>(gA, glda, r0, aM, k0, a.K, ta, wave, lane);
    if (BL == 0) stage_kmajor<BN, CF::NW>(gB, gldb, c0, a.N, k0, a.K, tb, wave, lane);
    else         stage_mnmajor<BN, CF::NW>(gB, gldb, c0, a.N, k0, a.K, tb, wave, lane);
  };
  if (!PREFETCH && tile != tile0) __syncthreads();   // nobody still reads what the first DMA overwrites

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // piece p of K-tile t's DMA (4 pieces of A, then 4 of B), positioned by `pin` (glds16_pin)
  constexpr int PER_A = (AL == 0 ? BM / 8 : BM / 128 * 16) / CF::NW, PER_B = (BL == 0 ? BN / 8 : BN / 128 * 16) / CF::NW;
  // interior tiles of matrices that fit 32-bit byte offsets take the scalar-base addressing (glds16_sbase) for their
  // full K-tiles; edge tiles, a ragged last K-tile and huge matrices keep the bounds-checked per-lane pointers
  // (measured A/B: scalar base + lane offset: -6 % for the weight-gradient products, both operands MN-major, but
  //  +3..9 % with a K-major operand; those add the offset to the base on the VALU instead)
  // (measured A/B: -2..-5 %; the dGELU epilogue, which ran out of registers with it while it still evaluated gelu'
  //  itself, is a plain multiply now)
  constexpr bool USE_FAST = true;
  constexpr bool SADDR = (AL == 1 && BL == 1);   // else: the same offsets added to the base on the VALU (2 instructions per piece)
  const bool fast = USE_FAST && a.fast_addr && arow0 + BM <= aM && col0 + BN <= a.N;
  unsigned off_a[PER_A], off_b[PER_B];
#pragma unroll
  for (int i = 0; i < PER_A; ++i)
    off_a[i] = (AL == 0) ? piece_off_kmajor<BM, CF::NW>(glda, arow0, wave, lane, i) : piece_off_mnmajor<BM, CF::NW>(glda, arow0, wave, lane, i);
#pragma unroll
  for (int i = 0; i < PER_B; ++i)
    off_b[i] = (BL == 0) ? piece_off_kmajor<BN, CF::NW>(gldb, col0, wave, lane, i) : piece_off_mnmajor<BN, CF::NW>(gldb, col0, wave, lane, i);
  auto fast_piece = [&](int t, int buf, int p, auto& pin, auto& after) {
    const int k0 = (kt_begin + t) * BK;
    if (p < PER_A) {
      const char* base = (const char*)gA + ((AL == 0) ? (size_t)k0 * 2 : (size_t)k0 * glda * 2);
      if (SADDR) glds16_sbase(off_a[p], base, (lptr_t)(smem + buf * BUF_BYTES + (wave * PER_A + p) * 1024), pin, after);
      else glds16_pin(base + off_a[p], (lptr_t)(smem + buf * BUF_BYTES + (wave * PER_A + p) * 1024), pin, after);
    } else {
      const char* base = (const char*)gB + ((BL == 0) ? (size_t)k0 * 2 : (size_t)k0 * gldb * 2);
      if (SADDR) glds16_sbase(off_b[p - PER_A], base, (lptr_t)(smem + buf * BUF_BYTES + TILE_BYTES_A + (wave * PER_B + p - PER_A) * 1024), pin, after);
      else glds16_pin(base + off_b[p - PER_A], (lptr_t)(smem + buf * BUF_BYTES + TILE_BYTES_A + (wave * PER_B + p - PER_A) * 1024), pin, after);
    }
  };
  // K-tiles [0, ntf) of this split are full ones and may take the scalar-base form; at most the last one is ragged
  const int ntf = fast ? max(0, min(nt, a.K / BK - kt_begin)) : 0;
  auto stage_fast = [&](int t, int buf) {
    int nopin = 0, nopin2 = 0;
#pragma unroll
    for (int p = 0; p < PER_A + PER_B; ++p) fast_piece(t, buf, p, nopin, nopin2);
  };
  auto stage = [&](int t, int buf) {
    if (USE_FAST && t < ntf) stage_fast(t, buf);
    else stage_at(arow0, col0, t, buf);
  };
  auto stage_piece = [&](int t, int buf, int p, bf16x8& pin, bf16x8& after) {
    unsigned char* ta = smem + buf * BUF_BYTES;
    unsigned char* tb = ta + TILE_BYTES_A;
    const int k0 = (kt_begin + t) * BK;
    if (p < PER_A) {
      if (AL == 0) stage_kmajor_piece<BM, CF::NW>(gA, glda, arow0, aM, k0, a.K, ta, wave, lane, p, pin, after);
      else         stage_mnmajor_piece<BM, CF::NW>(gA, glda, arow0, aM, k0, a.K, ta, wave, lane, p, pin, after);
    } else {
      if (BL == 0) stage_kmajor_piece<BN, CF::NW>(gB, gldb, col0, a.N, k0, a.K, tb, wave, lane, p - PER_A, pin, after);
      else         stage_mnmajor_piece<BN, CF::NW>(gB, gldb, col0, a.N, k0, a.K, tb, wave, lane, p - PER_A, pin, after);
    }
  };

  // Software pipeline, 4 phases per K-tile (2 k-steps x 2 halves of the wave's rows), one barrier per K-tile.
  // While the MFMAs of one phase run, the LDS reads of the next phase's fragments are in flight:
  //   phase 0: MFMA(t, ks0, rows lo)   || read A(t, ks0, hi)
  //   phase 1: MFMA(t, ks0, rows hi)   || read A(t, ks1, lo), B(t, ks1)
  //   phase 2: MFMA(t, ks1, rows lo)   || read A(t, ks1, hi)
  //   ---- vmcnt(0): tile t+1 landed; barrier: every wave has finished reading tile t ----
  //   global_load_lds of tile t+2 into tile t's buffer (lands during the next K-tile)
  //   phase 3: MFMA(t, ks1, rows hi)   || read A(t+1, ks0, lo), B(t+1, ks0)
  constexpr int HM = TM / 2;
  bf16x8 fa0[HM], fa1[HM], fb0[TN], fb1[TN];
  auto read_a = [&](bf16x8 (&fa)[HM], int buf, int ks, int half) {
    const unsigned char* ta = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < HM; ++i)
      fa[i] = (AL == 0) ? frag_kmajor(ta, wm * TM * 16, half * HM + i, ks, lane)
                        : frag_mnmajor(ta, wm * TM * 16, half * HM + i, ks, lane);
  };
  auto read_b = [&](bf16x8 (&fb)[TN], int buf, int ks) {
    const unsigned char* tb = smem + buf * BUF_BYTES + TILE_BYTES_A;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[j] = (BL == 0) ? frag_kmajor(tb, wn * WNC, j, ks, lane) : frag_mnmajor(tb, wn * WNC, j, ks, lane);
  };
#define CZC_MFMA_HALF(FA, FB, HALF)                                                                  \
  _Pragma("unroll") for (int i = 0; i < HM; ++i)                                                     \
  _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                     \
    acc[(HALF) * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j], FA[i], acc[(HALF) * HM + i][j], 0, 0, 0)

  if constexpr (PP) {
  // ---------------------------------------------------------------------------------------------------------------
  // Ping-pong K-loop.  The waves form two groups, G0 = waves 0-3 (wm = 0) and G1 = waves 4-7 (wm = 1): one wave of
  // each per SIMD.  A K-tile is four quadrant phases (64 x 32 outputs x K = 64: 16 MFMAs); a phase is
  //     LOAD: this quadrant's fragment reads + two LDS-DMA pieces      | s_barrier |
  //     MFMA: the 16 MFMAs                                              | s_barrier |
  // and G1 runs one barrier behind G0, so on every SIMD one wave's MFMA cluster covers the other wave's LDS round
  // trip and DMA issue -- the two never stall together (stamps of the plain loop: a K-tile took 3400 cycles for 2048
  // of MFMA work; both waves of a SIMD issued DMA / waited at the same points).
  // LDS: two K-tile buffers, each refilled by HALF operand tiles (128 rows, 16 pieces, 2 per wave) as soon as
  // every wave's last read of that half has retired and a further barrier has passed:
  //   segment of tile t   G0 does                      G1 does
  //   s0                  LOAD Q0 + DMA A0(t+1)        MFMA Q3(t-1)
  //   s1                  MFMA Q0                      LOAD Q0 + DMA A1(t+1)
  //   s2                  LOAD Q1 + DMA A1(t+1)        MFMA Q0
  //   s3                  MFMA Q1                      LOAD Q1 + DMA B1(t+1)     (last B reads of tile t)
  //   s4                  LOAD Q2 + DMA B1(t+1)        MFMA Q1                   (last A0 reads of tile t)
  //   s5                  MFMA Q2                      LOAD Q2 + DMA B0(t+2)     (last A1 reads; B of tile t is free)
  //   s6                  LOAD Q3 + DMA B0(t+2)        MFMA Q2                   (A0 of tile t is free)
  //   s7                  MFMA Q3, vmcnt(2)            LOAD Q3 + DMA A0(t+2), vmcnt(4)
  // Q0 = rows lo x cols lo (reads A lo, B lo), Q1 = lo x hi (B hi), Q2 = hi x hi (A hi), Q3 = hi x lo (B lo kept).
  // The counted waits leave exactly the pieces of tile t+2 in flight: everything of tile t+1 has landed before the
  // barrier that ends s7, one segment before its first read.
  static_assert(CF::NW == 8 && TM == 8 && TN == 4, "the ping-pong schedule is written for the 256x256 / 8-wave tile");
  const int grp = wm;
  auto stage_half = [&](int r0, int c0, int opnd, int t, int h) {   // this wave's two pieces of half h of A (0) / B (1)
    unsigned char* ta = smem + (t & 1) * BUF_BYTES;
    unsigned char* tb = ta + TILE_BYTES_A;
    const int k0 = (kt_begin + t) * BK;
    int np = 0, np2 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = 16 * h + 2 * wave + i;
      if (opnd == 0) {
        if (AL == 0) stage_kmajor_q(gA, glda, r0, aM, k0, a.K, ta, q, lane, np, np2);
        else         stage_mnmajor_q(gA, glda, r0, aM, k0, a.K, ta, q, lane, np, np2);
      } else {
        if (BL == 0) stage_kmajor_q(gB, gldb, c0, a.N, k0, a.K, tb, q, lane, np, np2);
        else         stage_mnmajor_q(gB, gldb, c0, a.N, k0, a.K, tb, q, lane, np, np2);
      }
    }
  };
  // interior tiles: piece offsets fixed along K (see fast_piece above), index [2*h + i]
  unsigned poff_a[4], poff_b[4];
#pragma unroll
  for (int hi = 0; hi < 4; ++hi) {
    const int q = 16 * (hi >> 1) + 2 * wave + (hi & 1);
    poff_a[hi] = (AL == 0) ? piece_off_kmajor_q(glda, arow0, q, lane) : piece_off_mnmajor_q(glda, arow0, q, lane);
    poff_b[hi] = (BL == 0) ? piece_off_kmajor_q(gldb, col0, q, lane) : piece_off_mnmajor_q(gldb, col0, q, lane);
  }
  const bool pfast = a.fast_addr && arow0 + BM <= aM && col0 + BN <= a.N;
  const int pntf = pfast ? max(0, min(nt, a.K / BK - kt_begin)) : 0;   // K-tiles [0, pntf) are full
  auto stage_half_fast = [&](int opnd, int t, int h) {
    const int k0 = (kt_begin + t) * BK;
    int np = 0, np2 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = 16 * h + 2 * wave + i;
      if (opnd == 0) {
        const char* base = (const char*)gA + ((AL == 0) ? (size_t)k0 * 2 : (size_t)k0 * glda * 2);
        lptr_t dst = (lptr_t)(smem + (t & 1) * BUF_BYTES + q * 1024);
        if (AL == 1 && BL == 1) glds16_sbase(poff_a[2 * h + i], base, dst, np, np2);
        else if (ANT) glds16_pin_nt(base + poff_a[2 * h + i], dst, np, np2);
        else glds16_pin(base + poff_a[2 * h + i], dst, np, np2);
      } else {
        const char* base = (const char*)gB + ((BL == 0) ? (size_t)k0 * 2 : (size_t)k0 * gldb * 2);
        lptr_t dst = (lptr_t)(smem + (t & 1) * BUF_BYTES + TILE_BYTES_A + q * 1024);
        if (AL == 1 && BL == 1) glds16_sbase(poff_b[2 * h + i], base, dst, np, np2);
        else glds16_pin(base + poff_b[2 * h + i], dst, np, np2);
      }
    }
  };
  auto sh = [&](int opnd, int t, int h) {
    if (t < pntf) stage_half_fast(opnd, t, h);
    else if (t < nt) stage_half(arow0, col0, opnd, t, h);
  };
  bf16x8 fa[8], fbl[4], fbh[4];   // A: [ks*4 + i] of the current row half; B lo / hi column pairs: [ks*2 + j]
  auto load_a = [&](int t, int qi) {
    const unsigned char* ta = smem + (t & 1) * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[ks * 4 + i] = (AL == 0) ? frag_kmajor(ta, wm * TM * 16, 4 * qi + i, ks, lane) : frag_mnmajor(ta, wm * TM * 16, 4 * qi + i, ks, lane);
  };
  auto load_b = [&](bf16x8 (&fb)[4], int t, int qj) {
    const unsigned char* tb = smem + (t & 1) * BUF_BYTES + TILE_BYTES_A;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[ks * 2 + j] = (BL == 0) ? frag_kmajor(tb, wn * WNC, 2 * qj + j, ks, lane) : frag_mnmajor(tb, wn * WNC, 2 * qj + j, ks, lane);
  };
#define CZC_BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define CZC_QUAD(QI, QJ, FB)                                                                          \
  do {                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
      acc[4 * (QI) + i][2 * (QJ) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                     \
          FB[ks * 2 + j], fa[ks * 4 + i], acc[4 * (QI) + i][2 * (QJ) + j], 0, 0, 0);                 \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)
  constexpr int NST = (EPI == EPI_BIAS_GELU ? 2 : 1) * (TM * 16 * TN * 2 / 64);
  auto wait_vm = [&](int n) {   // at most the n youngest vector-memory instructions of this wave stay outstanding
    switch (n >> 1) {
#define CZC_W(N) case N / 2: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
      CZC_W(2) CZC_W(4) CZC_W(8) CZC_W(10) CZC_W(12) CZC_W(16) CZC_W(18) CZC_W(20) CZC_W(32) CZC_W(34) CZC_W(36)
#undef CZC_W
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  if (nt > 0) {
    // tile 0 whole (unless the previous epilogue requested it), then what "tile -1" would have issued for tile 1
    if (!primed) { sh(0, 0, 0); sh(0, 0, 1); sh(1, 0, 0); sh(1, 0, 1); }
    else CZC_BAR();   // the previous epilogue staged through the second buffer: every wave has read its part back
    int young = primed ? NST : 0;
    if (nt > 1) { sh(1, 1, 0); young += 2; if (grp == 1) { sh(0, 1, 0); young += 2; } }
    wait_vm(young);
    CZC_BAR();
    if (grp == 1) CZC_BAR();   // G1 runs one barrier behind
    // STEADY: tiles t+1 and t+2 are full interior tiles: unconditional fast staging and constant wait counts
#define CZC_PP_TILE(STEADY)                                                                          \
    {                                                                                                \
      auto st_ = [&](int opnd, int tt, int h) { if (STEADY) stage_half_fast(opnd, tt, h); else sh(opnd, tt, h); }; \
      /* ---- Q0 */                                                                                  \
      load_a(t, 0); load_b(fbl, t, 0);                                                               \
      if (grp == 0) st_(0, t + 1, 0); else st_(0, t + 1, 1);                                         \
      CZC_BAR();                                                                                     \
      CZC_QUAD(0, 0, fbl);                                                                           \
      CZC_BAR();                                                                                     \
      /* ---- Q1 */                                                                                  \
      load_b(fbh, t, 1);                                                                             \
      if (grp == 0) st_(0, t + 1, 1); else st_(1, t + 1, 1);                                         \
      CZC_BAR();                                                                                     \
      CZC_QUAD(0, 1, fbh);                                                                           \
      CZC_BAR();                                                                                     \
      /* ---- Q2 */                                                                                  \
      load_a(t, 1);                                                                                  \
      if (grp == 0) st_(1, t + 1, 1); else st_(1, t + 2, 0);                                         \
      CZC_BAR();                                                                                     \
      CZC_QUAD(1, 1, fbh);                                                                           \
      CZC_BAR();                                                                                     \
      /* ---- Q3 */                                                                                  \
      if (grp == 0) st_(1, t + 2, 0); else st_(0, t + 2, 0);                                         \
      if (grp == 1) { if (STEADY) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else wait_vm(t + 2 < nt ? 4 : 0); } \
      CZC_BAR();                                                                                     \
      CZC_QUAD(1, 0, fbl);                                                                           \
      if (grp == 0) { if (STEADY) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else wait_vm(t + 2 < nt ? 2 : 0); } \
      CZC_BAR();                                                                                     \
    }
    int t = 0;
    for (; t + 2 < pntf; ++t) CZC_PP_TILE(true)
    for (; t < nt; ++t) CZC_PP_TILE(false)
#undef CZC_PP_TILE
    if (grp == 0) CZC_BAR();
  }
#undef CZC_QUAD
#undef CZC_BAR
  } else {
  if (nt > 0) {
    // counted wait when primed: the epilogue's stores were issued after that DMA and need not have been acknowledged
    constexpr int NST = (EPI == EPI_BIAS_GELU ? 2 : 1) * (TM * 16 * TN * 2 / 64);
    static_assert(NST <= 32, "vmcnt immediate");
    if (!primed) { stage(0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else if (NST == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (NST == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (NST == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nt > 1) stage(1, 1);
    read_a(fa0, 0, 0, 0);
    read_b(fb0, 0, 0);
  }
  // pin the issue order inside each phase: one LDS read, then a few MFMAs, ... (the scheduler otherwise sinks the
  // reads to just before their first use and the MFMA pipe waits out every LDS round trip)
  constexpr int RA = (AL == 0) ? 1 : 2, RB = (BL == 0) ? 1 : 2;   // ds instructions per fragment
  // K-major fragments (one ds_read_b128 each): reads spread evenly over the phase's MFMAs.  Both operands MN-major
  // (two transposing reads per fragment, the weight-gradient products): one read per MFMA from the start of the phase,
  // so the last of them has the rest of the phase to come back (measured A/B on one device: -4.5 % on those products,
  // +1..4 % on the K-major ones, which therefore keep the even spread).
  constexpr bool FRONT = (AL == 1 && BL == 1);
#define CZC_INTERLEAVE(NREADS, NMFMA)                                                      \
  _Pragma("unroll") for (int q_ = 0; q_ < (NREADS); ++q_) {                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, FRONT ? 1 : (NMFMA) / (NREADS), 0);         \
  }                                                                                         \
  if (FRONT) __builtin_amdgcn_sched_group_barrier(0x008, (NMFMA) - (NREADS), 0);
  constexpr int NM = HM * TN;  // MFMAs per phase
  constexpr int PPR = (PER_A + PER_B) / HM;   // DMA pieces per row of the last phase
  static_assert((PER_A + PER_B) % HM == 0, "DMA pieces do not divide over the last phase's rows");
  // STAGE: K-tile t+2 exists.  SPREAD: its DMA pieces go between the rows of the last phase instead of in one burst
  // after the barrier (measured A/B on one device: -2..-4 % for the input-gradient products, K-major A x MN-major B;
  // no change for K-major x K-major; the weight-gradient products run out of registers with it)
  constexpr bool SPREAD = (AL == 0 && BL == 1);
#define CZC_KTILE(STAGE, FAST)                                                                       \
  {                                                                                                  \
    const int buf = t & 1;                                                                           \
    read_a(fa1, buf, 0, 1);                                                                          \
    CZC_MFMA_HALF(fa0, fb0, 0);                                                                      \
    CZC_INTERLEAVE(HM * RA, NM)                                                                      \
    read_a(fa0, buf, 1, 0);                                                                          \
    read_b(fb1, buf, 1);                                                                             \
    CZC_MFMA_HALF(fa1, fb0, 1);                                                                      \
    CZC_INTERLEAVE(HM * RA + TN * RB, NM)                                                            \
    read_a(fa1, buf, 1, 1);                                                                          \
    CZC_MFMA_HALF(fa0, fb1, 0);                                                                      \
    CZC_INTERLEAVE(HM * RA, NM)                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
    __syncthreads();                                                                                 \
    __builtin_amdgcn_sched_barrier(0);   /* phases 0-2 and phase 3 are scheduled as separate regions */ \
    if ((STAGE) && !SPREAD) { if (FAST) stage_fast(t + 2, buf); else stage_at(arow0, col0, t + 2, buf); } \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    /* (after the last tile these read the other buffer's stale bytes: never used) */               \
    read_a(fa0, buf ^ 1, 0, 0);                                                                      \
    read_b(fb0, buf ^ 1, 0);                                                                         \
    _Pragma("unroll") for (int i = 0; i < HM; ++i) {                                                 \
      if ((STAGE) && SPREAD) {                                                                       \
        _Pragma("unroll") for (int pp = 0; pp < PPR; ++pp) {                                         \
          if (FAST) fast_piece(t + 2, buf, i * PPR + pp, fa1[i], i ? fa1[i - 1] : fa0x);               \
          else stage_piece(t + 2, buf, i * PPR + pp, fa1[i], i ? fa1[i - 1] : fa0x);                   \
        }                                                                                            \
      }                                                                                              \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[HM + i][j], 0, 0, 0); \
    }                                                                                                \
    CZC_INTERLEAVE(HM * RA + TN * RB, NM)                                                            \
  }
  bf16x8 fa0x = {};   // nothing precedes the first row's pieces
  int t = 0;
  if (USE_FAST) { for (; t + 2 < ntf; ++t) CZC_KTILE(true, USE_FAST) }
  for (; t + 2 < nt; ++t) CZC_KTILE(true, false)
  for (; t < nt; ++t) CZC_KTILE(false, false)
#undef CZC_KTILE
#undef CZC_INTERLEAVE
#undef CZC_MFMA_HALF
  }   // !PP

  // ---------------- epilogue ----------------
  // acc[i][j][r] = C[row0 + wm*128 + 16i + (lane&15)][col0 + wn*64 + 16j + 4(lane>>4) + r]
  const int mrow = lane & 15, nq = lane >> 4;

  if (EPI == EPI_F32) {
    float* C = (float*)a.C + (size_t)split * a.M * a.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = row0 + wm * TM * 16 + 16 * i + mrow;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = col0 + wn * WNC + 16 * j + 4 * nq;
        if (m < a.M && n < a.N) *(f32x4*)(C + (size_t)m * a.ldc + n) = acc[i][j];
      }
    }
    continue;
  }

  if (EPI == EPI_LSE) {
    // per-row partial (max, sum exp, sum) over this tile's valid columns of z = alpha*acc
    __syncthreads();
    float* red = (float*)smem;  // [3][WN][BM rows]
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float mx = -1e30f, sm = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = col0 + wn * WNC + 16 * j + 4 * nq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z = (n + r < a.N) ? acc[i][j][r] * a.alpha : -1e30f;
          acc[i][j][r] = z;
          mx = fmaxf(mx, z);
          sm += (n + r < a.N) ? z : 0.f;
        }
      }
      mx = xor32_max(xor16_max(mx));
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) se += __expf(acc[i][j][r] - mx);
      se = xor32_sum(xor16_sum(se));
      sm = xor32_sum(xor16_sum(sm));
      if (nq == 0) {
        const int ml = wm * TM * 16 + 16 * i + mrow;
        red[(0 * WN + wn) * BM + ml] = mx;
        red[(1 * WN + wn) * BM + ml] = se;
        red[(2 * WN + wn) * BM + ml] = sm;
      }
    }
    __syncthreads();
    if (tid < BM) {
      const int m = row0 + tid;
      float mx = -1e30f;
#pragma unroll
      for (int w = 0; w < WN; ++w) mx = fmaxf(mx, red[(0 * WN + w) * BM + tid]);
      float se = 0.f, sm = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        se += red[(1 * WN + w) * BM + tid] * __expf(red[(0 * WN + w) * BM + tid] - mx);
        sm += red[(2 * WN + w) * BM + tid];
      }
      if (m < a.M) {
        const size_t pl = (size_t)a.tiles_n * a.M;
        a.part[0 * pl + (size_t)tn * a.M + m] = mx;
        a.part[1 * pl + (size_t)tn * a.M + m] = se;
        a.part[2 * pl + (size_t)tn * a.M + m] = sm;
      }
    }
    continue;
  }

  if (EPI == EPI_FILTER) {
    // flat-index coarse scan: survivors are rare once the lower bounds have settled, so every lane appends its own
    // (one atomic per survivor); the common case is 2 VALU ops per accumulator and no memory traffic at all
    f32x4 dn[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = col0 + wn * WNC + 16 * j + 4 * nq;
#pragma unroll
      for (int r = 0; r < 4; ++r) dn[j][r] = (n + r < a.f_nvalid) ? a.f_dn[n + r] : -INFINITY;   // padded rows never pass
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = row0 + wm * TM * 16 + 16 * i + mrow;
      const bool mok = m < a.M;
      const float lb = mok ? a.f_lb[m] : INFINITY;
      const float qe = mok ? a.f_qe[m] : 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = col0 + wn * WNC + 16 * j + 4 * nq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float c = acc[i][j][r];
          const float e = qe * dn[j][r];
          if (c + e > lb) {           // e = -inf for padded columns, lb = +inf for padded rows
            const long long pos = atomicAdd(&a.f_cnt[m], 1);
            if (pos < a.f_cap) {
              const long long at = (long long)m * a.f_cap + pos;
              a.f_cs[at] = c;
              a.f_ce[at] = e;
              a.f_ci[at] = a.f_idbase + n + r;
            }
          }
        }
      }
    }
    continue;
  }

  // bf16 outputs: registers -> LDS (per-wave rows x 64 bf16 region, 16-B chunks XOR-swizzled by
  // row&7) -> 16-B row segments -> global
  // bias / row vector first, as bounds-checked buffer loads (out of range reads 0): all in flight together and
  // hidden behind the barrier, instead of one dependent load per fragment
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  f32x4 bv[TN];
  float rvv[TM];
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU1) {
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (unsigned)a.N * 4u, 0x00020000);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(col0 + wn * WNC + 16 * j + 4 * nq) * 4u, 0, 0));
  }
  if (EPI == EPI_SOFTGRAD) {
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.rowv, 0, (unsigned)a.M * 4u, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
      rvv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (unsigned)(row0 + wm * TM * 16 + 16 * i + mrow) * 4u, 0, 0));
  }
  // the readback below hands lane l of pass `it` the 16-B segment idx = 64*it + l of the wave's 128 x 64 region
  constexpr int ROWB = TN * 32, NCH = TN * 2;   // bytes / 16-B chunks per staged row
  constexpr int NIT = TM * 16 * NCH / 64;
  const unsigned cbytes = (unsigned)min((size_t)a.M * (size_t)a.ldc * 2u, (size_t)0xfffffff0u);
  auto seg_off = [&](int it) -> unsigned {   // byte offset of that segment in C / C2 / aux, out of range past N
    const int idx = it * 64 + lane;
    const int m = row0 + wm * TM * 16 + idx / NCH, n = col0 + wn * WNC + (idx % NCH) * 8;
    return (n < a.N) ? (unsigned)(((size_t)m * a.ldc + n) * 2u) : 0xfffffff0u;
  };
  // DGELU / ADD: the second operand's segments, all requested before the barrier (a load issued inside the store
  // loop would be kept behind the preceding store and pay the full memory latency NIT times over)
  u32x4 au[NIT];
  if (EPI == EPI_DGELU || EPI == EPI_ADD) {
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, cbytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < NIT; ++it) au[it] = __builtin_amdgcn_raw_buffer_load_b128(ra, seg_off(it), 0, 0);
  }
  __syncthreads();  // all waves are done reading the operand tiles
  // the first operand buffer is free now: request K-tile 0 of this workgroup's next output tile into it
  if (tile + (int)gridDim.x < ntiles && nt > 0) {
    int tm2, tn2;
    swizzle_tile(tile + gridDim.x, ntiles, a.tiles_m, a.tiles_n, a.group_n, tm2, tn2);
    stage_at(tm2 * BM, tn2 * BN, 0, 0);
    primed = true;
  } else {
    primed = false;
  }
  // the tile leaves through the second operand buffer, NPASS slices of the wave's rows at a time
  constexpr int NPASS = (CF::NW * TM * 16 * ROWB) / BUF_BYTES;
  static_assert(NPASS >= 1 && TM % NPASS == 0 && NIT % NPASS == 0, "epilogue staging does not tile the operand buffer");
  constexpr int TMP = TM / NPASS, NITP = NIT / NPASS;
  unsigned char* reg = smem + BUF_BYTES + wave * (TMP * 16 * ROWB);
  // read the staged rows back as 16-B row segments and store them with raw buffer ops: rows beyond M fall outside
  // the descriptor and columns beyond N get an out-of-range offset, so the loop is branch-free and the LDS reads of
  // later segments are issued while earlier stores drain
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, cbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rc2 = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_BIAS_GELU ? a.C2 : a.C, 0, cbytes, 0x00020000);
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // DGELU: this lane's 8 columns over its 2*TM rows
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int ii = 0; ii < TMP; ++ii) {
      const int i = ps * TMP + ii;
      const int rl = 16 * ii + mrow;   // row inside this pass's slice
      const float rv = (EPI == EPI_SOFTGRAD) ? rvv[i] : 0.f;
      if (EPI == EPI_BIAS_GELU1) {
        // one output: gelu on the fp32 values, before the staging (no unpacking of the staged bf16 pre-activation and
        // one rounding less; the two-output form below needs u itself rounded once for both results)
        static_assert(EPI != EPI_BIAS_GELU1 || TN % 2 == 0, "gelu8 takes two column groups at a time");
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const f32x4 v0 = acc[i][j] + bv[j], v1 = acc[i][j + 1] + bv[j + 1];
          float xin[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, gl[8], dg[8];
          gelu8<0>(xin, gl, dg);
          acc[i][j] = (f32x4){gl[0], gl[1], gl[2], gl[3]};
          acc[i][j + 1] = (f32x4){gl[4], gl[5], gl[6], gl[7]};
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[i][j];
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) v += bv[j];
        if (EPI == EPI_SOFTGRAD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __expf(v[r] * a.alpha - rv) - a.beta;
        }
        uint2 w;
        w.x = pack2bf(v[0], v[1]);
        w.y = pack2bf(v[2], v[3]);
        const int chunk = (2 * j + (nq >> 1)) ^ (rl & 7);
        *(uint2*)(reg + rl * ROWB + chunk * 16 + (nq & 1) * 8) = w;
      }
    }
    u32x4 sv[NITP];
#pragma unroll
    for (int it = 0; it < NITP; ++it) {
      const int idx = it * 64 + lane;
      const int rl = idx / NCH, ch = idx % NCH;
      sv[it] = *(const u32x4*)(reg + rl * ROWB + ((ch ^ (rl & 7)) << 4));
    }
#pragma unroll
    for (int itp = 0; itp < NITP; ++itp) {
      const int it = ps * NITP + itp;
      const int m = row0 + wm * TM * 16 + (it * 64 + lane) / NCH;
      const unsigned off = seg_off(it);
      const u32x4 v = sv[itp];
      if (EPI == EPI_BIAS_GELU) {
        // two outputs: C = gelu'(u) -- all the backward needs of the pre-activation (dX = dH . gelu'(u); dW1 reads x1, dW2
        // reads h), so its epilogue is a multiply instead of a second exp / rcp / polynomial per element -- and C2 = gelu(u)
        float xin[8], gl[8], dg[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xin[2 * e] = bflo(v[e]); xin[2 * e + 1] = bfhi(v[e]); }
        gelu8<2>(xin, gl, dg);
        u32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = pack2bf(dg[2 * e], dg[2 * e + 1]);
        store16(g, rc, off, a.store_pol);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = pack2bf(gl[2 * e], gl[2 * e + 1]);
        store16(g, rc2, off, a.store_pol);
      } else if (EPI == EPI_DGELU) {
        const u32x4 u = au[it];   // gelu'(pre-activation), as the forward stored it
        u32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bflo(v[e]) * bflo(u[e]), hi = bfhi(v[e]) * bfhi(u[e]);
          g[e] = pack2bf(lo, hi);
          if (m < a.M) { csum[2 * e] += lo; csum[2 * e + 1] += hi; }
        }
        store16(g, rc, off, a.store_pol);
      } else if (EPI == EPI_ADD) {
        const u32x4 u = au[it];
        u32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = pack2bf(bflo(v[e]) + bflo(u[e]), bfhi(v[e]) + bfhi(u[e]));
        store16(g, rc, off, a.store_pol);
      } else {
        store16(v, rc, off, a.store_pol);
      }
    }
  }
  if (EPI == EPI_DGELU && a.colsum) {
    // lanes with equal (lane & 7) own the same 8 columns: fold the 8 row groups, then 8 lanes x 8 atomics
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      static_assert(NCH == 8, "the fold below pairs lanes l, l^8, l^16, l^32");
      float t = csum[j];
      t += dpp_mov<0x128>(t);   // row_ror:8 = lane ^ 8 within a row of 16
      csum[j] = xor32_sum(xor16_sum(t));
    }
    // one partial row per (M-tile, wave row): colsum[(tm * WM + wm)][N], written whole (no atomics)
    const int n = col0 + wn * WNC + (lane % NCH) * 8;
    if ((lane / NCH) == 0 && n < a.N) {
      float* row = a.colsum + (size_t)(tm * CF::WM + wm) * a.N + n;
      *(f32x4*)row = (f32x4){csum[0], csum[1], csum[2], csum[3]};
      *(f32x4*)(row + 4) = (f32x4){csum[4], csum[5], csum[6], csum[7]};
    }
  }
  }  // tile loop
}

// ---- host launcher ---------------------------------------------------------------------------
typedef Cfg<2, 4, 8> CfgBig;
typedef Cfg<2, 2, 4> CfgSmall;
static int g_tile_override = 0;  // 0 = heuristic, 1 = 256x256, 2 = 128x128
static bool g_one_wg_per_tile = false;
static int g_pingpong = 0;             // 0 = built-in choice; tuning knob czc_gemm_set_tile: +32 = always, +64 = never  // tuning knob: switch the persistent tile loop off

static int g_store_pol = -1;   // -1 = built-in choice; czc_gemm_set_tile +512: sc1 output stores, +1024: nt, +2048: plain
template <int AL, int BL, int EPI, class CF, bool PP = false, bool ANT = false>
static int launch_cfg(GemmArgs a, int splits, hipStream_t st) {
  static bool attr_done = false;
  auto kern = gemm_kernel<AL, BL, EPI, CF, PP, ANT>;
  if (!attr_done) {
    CZC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS));
    attr_done = true;
  }
  a.tiles_m = ceil_div(a.M, CF::BM);
  a.tiles_n = ceil_div(a.N, CF::BN);
  // bf16 output stores of wide products are streaming (nt): the output is read next by another kernel, from HBM, and a
  // tile's 128 KiB per CU otherwise competes with the B panels and the shared A rows for the XCD's 4 MiB of L2
  // (interleaved A/B: QKV 234 -> 215 us, FFN-up 389 -> 377, its backward 348 -> 335; no difference at N = 768)
  a.store_pol = g_store_pol >= 0 ? g_store_pol : (a.N >= 2048 ? 2 : 0);
  // bf16 epilogues run persistent (one workgroup per resident slot, see the kernel); the split-K and head-LSE
  // forms keep one workgroup per tile
  static int slots = 0, xcd8 = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    CZC_HIP(hipGetDevice(&dev));
    CZC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    slots = cus * (160 * 1024 / CF::LDS);
    xcd8 = cus == 256 ? 1 : 0;   // MI355X in SPX mode: 8 XCDs x 32 CUs (HIP has no XCD-count attribute; a CPX partition shows 32 CUs)
#ifdef CZC_SLOTS_OVERRIDE
    slots = CZC_SLOTS_OVERRIDE;   // diagnostic builds: a persistent launch on part of the chip
#endif
  }
  {
    const size_t ea = (AL == 0) ? (size_t)a.M * a.lda : (size_t)a.K * a.lda;   // elements spanned by A / B
    const size_t eb = (BL == 0) ? (size_t)a.N * a.ldb : (size_t)a.K * a.ldb;
    const size_t ea2 = a.A2 ? ((AL == 0) ? (size_t)a.M * a.lda2 : (size_t)a.K * a.lda2) : 0;
    const size_t eb2 = a.A2 ? ((BL == 0) ? (size_t)a.N * a.ldb2 : (size_t)a.K * a.ldb2) : 0;
    a.fast_addr = (ea * 2 < (1ull << 31) && eb * 2 < (1ull << 31) && ea2 * 2 < (1ull << 31) && eb2 * 2 < (1ull << 31)) ? 1 : 0;
  }
  // wide outputs (FFN-up, its backward): six column tiles per sweep keep that group's B panels (6 x 256 x K) in L2 for
  // all the row tiles under them (PMC: FETCH 558 -> 433 MB per FFN-up launch, -2 % time); split-K products and
  // narrow outputs keep the plain row-major order
  a.xcd8 = xcd8;
  a.group_n = (a.tiles_n >= 12 && splits == 1) ? 6 : a.tiles_n;
  const int ntiles = a.tiles_m * a.tiles_n;
  // (measured, interleaved A/B at 65536 x {768, 2304, 3072}: persistent -1..-3 % for the bias / store / add / dGELU
  //  epilogues, +2 % for the two-output bias+GELU epilogue, which therefore keeps one workgroup per tile)
  const bool persistent = (EPI != EPI_F32 && EPI != EPI_LSE && EPI != EPI_BIAS_GELU && EPI != EPI_FILTER) && splits == 1 && !g_one_wg_per_tile;
  dim3 grid(persistent ? min(ntiles, slots) : ntiles, splits, 1);
  hipLaunchKernelGGL(kern, grid, dim3(CF::NT), CF::LDS, st, a);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

template <int AL, int BL, int EPI>
static int launch(const GemmArgs& a, int splits, hipStream_t st) {
  int cfg = a.cfg ? a.cfg : g_tile_override;
  if (EPI == EPI_LSE || EPI == EPI_SOFTGRAD || a.colsum) cfg = 1;  // partial-sum rows are laid out for 256-row tiles
  if (cfg == 0) cfg = (ceil_div(a.M, 256) * ceil_div(a.N, 256) * splits < 128 && !a.colsum) ? 2 : 1;
  if (cfg == 2) return launch_cfg<AL, BL, EPI, CfgSmall>(a, splits, st);
  // the ping-pong K-loop (see the kernel) where it measured faster than the plain one in interleaved A/B runs:
  // K-major x K-major products with a long K or a wide N (FFN-up / FFN-down / QKV forward: -1..-5 %); it loses on
  // the transposing-read kernels (their LOAD slots are twice as long) and on the 768 x 768 projections
  const bool pp_auto = (AL == 0 && BL == 0 && splits == 1 && (a.K >= 1536 || a.N >= 2304));
  if (g_pingpong == 1 || (g_pingpong == 0 && pp_auto)) {
#ifndef CZC_NO_ANT
    if constexpr (AL == 0 && BL == 0 && (EPI == EPI_BIAS || EPI == EPI_ADD)) {
      if (a.K >= 1536 && a.N <= 1024) return launch_cfg<AL, BL, EPI, CfgBig, true, true>(a, splits, st);
    }
#endif
    return launch_cfg<AL, BL, EPI, CfgBig, true>(a, splits, st);
  }
  return launch_cfg<AL, BL, EPI, CfgBig>(a, splits, st);
}

template <int EPI>
static int launch_layout(int al, int bl, const GemmArgs& a, int splits, hipStream_t st) {
  if (al == 0 && bl == 0) return launch<0, 0, EPI>(a, splits, st);
  if (al == 0 && bl == 1) return launch<0, 1, EPI>(a, splits, st);
  if (al == 1 && bl == 1) return launch<1, 1, EPI>(a, splits, st);
  set_error("czc_gemm: layout combination A=%d B=%d is not built (A MN-major needs B MN-major)", al, bl);
  return CZC_ERR_INVALID;
}

// ---- optional timing probe: hipEvents around every launch of ONE (layouts, epilogue, M, N, K) shape ----
struct Probe {
  bool armed = false;
  int al, bl, epi, M, N, K;
  int max_samples = 0, count = 0;
  hipEvent_t* ev = nullptr;  // 2 per sample
};
static Probe g_probe;

static int gemm_dispatch_inner(int al, int bl, int epi, GemmArgs a, int splitk, hipStream_t st);

int gemm_dispatch(int al, int bl, int epi, GemmArgs a, int splitk, hipStream_t st) {
  Probe& p = g_probe;
  const bool hit = p.armed && p.count < p.max_samples && al == p.al && bl == p.bl && epi == p.epi && a.M == p.M &&
                   a.N == p.N && a.K == p.K;
  if (hit) CZC_HIP(hipEventRecord(p.ev[2 * p.count], st));
  char pname[96];
  if (g_prof_on) snprintf(pname, sizeof(pname), "gemm a%d b%d epi%d %dx%dx%d%s%s", al, bl, epi, a.M, a.N, a.K, splitk > 1 ? " splitk" : "", a.A2 ? " (two products)" : "");
  const double outb = (epi == EPI_F32 ? 4.0 * splitk : (epi == EPI_LSE || epi == EPI_FILTER) ? 0.0 : 2.0) *
                      (epi == EPI_BIAS_GELU ? 2.0 : 1.0);
  ProfScope prof(st, g_prof_on ? pname : "", 2.0 * a.M * (double)a.N * a.K,
                 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + outb * a.M * (double)a.N +
                     ((epi == EPI_DGELU || epi == EPI_ADD) ? 2.0 * a.M * (double)a.N : 0.0));
  const int rc = gemm_dispatch_inner(al, bl, epi, a, splitk, st);
  if (hit) {
    CZC_HIP(hipEventRecord(p.ev[2 * p.count + 1], st));
    ++p.count;
  }
  return rc;
}

static int gemm_dispatch_inner(int al, int bl, int epi, GemmArgs a, int splitk, hipStream_t st) {
  CZC_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "czc_gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  CZC_REQUIRE((a.lda % 8) == 0 && (a.ldb % 8) == 0 && (a.ldc % 8) == 0,
              "czc_gemm: leading dimensions must be multiples of 8 (lda=%d ldb=%d ldc=%d)", a.lda, a.ldb, a.ldc);
  CZC_REQUIRE((a.N % 8) == 0, "czc_gemm: N=%d must be a multiple of 8", a.N);
  CZC_REQUIRE(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.B % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
              "czc_gemm: operand pointers must be 16-byte aligned");
  if (al == 0) CZC_REQUIRE((a.K % 8) == 0, "czc_gemm: K=%d must be a multiple of 8 for a K-major A", a.K);
  if (al == 1) CZC_REQUIRE((a.M % 8) == 0, "czc_gemm: M=%d must be a multiple of 8 for an M-major A", a.M);
  if (bl == 0) CZC_REQUIRE((a.K % 8) == 0, "czc_gemm: K=%d must be a multiple of 8 for a K-major B", a.K);
  CZC_REQUIRE(splitk >= 1, "czc_gemm: splitk=%d", splitk);
  if (epi != EPI_F32 && epi != EPI_LSE && epi != EPI_FILTER)
    CZC_REQUIRE((size_t)a.M * (size_t)a.ldc * 2u < 0xfffffff0ull,
                "czc_gemm: a bf16 output of %d x %d exceeds the 4 GiB a buffer descriptor addresses", a.M, a.ldc);
  CZC_REQUIRE(splitk == 1 || epi == EPI_F32, "czc_gemm: split-K needs the f32 slab epilogue");
  if (a.A2) {
    CZC_REQUIRE(epi == EPI_F32 && a.B2 && a.g_rows > 0 && a.g_rows < a.M && a.g_rows % 256 == 0 && (a.lda2 % 8) == 0 && (a.ldb2 % 8) == 0 &&
                ((uintptr_t)a.A2 % 16) == 0 && ((uintptr_t)a.B2 % 16) == 0,
                "czc_gemm: the second product needs the f32 slab epilogue, both operands, g_rows = %d a multiple of 256 inside (0, M)", a.g_rows);
    a.cfg = 1;   // (the row split is laid out for the 256-row tile)
  }
  const int kt = ceil_div(a.K, BK);
  a.ktiles_per_split = ceil_div(kt, splitk);
  switch (epi) {
    case EPI_STORE: return launch_layout<EPI_STORE>(al, bl, a, splitk, st);
    case EPI_BIAS: CZC_REQUIRE(a.bias, "czc_gemm: bias epilogue without bias"); return launch_layout<EPI_BIAS>(al, bl, a, splitk, st);
    case EPI_BIAS_GELU:
      CZC_REQUIRE(a.bias && a.C2, "czc_gemm: bias+gelu epilogue needs bias and C2");
      return launch_layout<EPI_BIAS_GELU>(al, bl, a, splitk, st);
    case EPI_BIAS_GELU1:
      CZC_REQUIRE(a.bias && al == 0 && bl == 0, "czc_gemm: bias+gelu(only) epilogue needs bias and K-major operands");
      return launch<0, 0, EPI_BIAS_GELU1>(a, 1, st);
    case EPI_DGELU: CZC_REQUIRE(a.aux, "czc_gemm: dgelu epilogue needs aux"); return launch_layout<EPI_DGELU>(al, bl, a, splitk, st);
    case EPI_ADD: CZC_REQUIRE(a.aux, "czc_gemm: add epilogue needs aux"); return launch_layout<EPI_ADD>(al, bl, a, splitk, st);
    case EPI_F32: return launch_layout<EPI_F32>(al, bl, a, splitk, st);
    case EPI_FILTER:
      CZC_REQUIRE(al == 0 && bl == 0 && a.f_lb && a.f_qe && a.f_dn && a.f_cs && a.f_ce && a.f_ci && a.f_cnt, "czc_gemm: filter epilogue arguments");
      return launch<0, 0, EPI_FILTER>(a, 1, st);
    case EPI_LSE:
      CZC_REQUIRE(al == 0 && bl == 0 && a.part, "czc_gemm: lse epilogue is K-major only");
      return launch<0, 0, EPI_LSE>(a, 1, st);
    case EPI_SOFTGRAD:
      CZC_REQUIRE(al == 0 && bl == 0 && a.rowv, "czc_gemm: softgrad epilogue is K-major only");
      return launch<0, 0, EPI_SOFTGRAD>(a, 1, st);
  }
  set_error("czc_gemm: unknown epilogue %d", epi);
  return CZC_ERR_INVALID;
}

// slabs reduce:  dst (+)= sum_z slabs[z]
// V = 1: four slabs' loads in flight per thread (the plain loop waits for each slab's load before the next: 15.0 -> 14.3 us)
template <int V>
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs, int64_t n4, float* __restrict__ dst,
                                    int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 s = accumulate ? ((const f32x4*)dst)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    int z = 0;
    if (V == 1) {
      for (; z + 4 <= nslabs; z += 4) {   // same order of additions as the plain loop: results stay bit-identical
        const f32x4 a = ((const f32x4*)slabs)[(int64_t)z * n4 + i], b = ((const f32x4*)slabs)[(int64_t)(z + 1) * n4 + i];
        const f32x4 c = ((const f32x4*)slabs)[(int64_t)(z + 2) * n4 + i], d = ((const f32x4*)slabs)[(int64_t)(z + 3) * n4 + i];
        s += a; s += b; s += c; s += d;
      }
    }
    for (; z < nslabs; ++z) s += ((const f32x4*)slabs)[(int64_t)z * n4 + i];
    ((f32x4*)dst)[i] = s;
  }
}

}  // namespace czc

using namespace czc;

extern "C" int czc_gemm_bf16(int a_layout, int b_layout, int epilogue, int M, int N, int K,
                             const uint16_t* A, int lda, const uint16_t* B, int ldb,
                             void* C, int ldc, const float* bias, const uint16_t* aux, void* C2,
                             int splitk, czc_stream_t stream) {
  CZC_REQUIRE((epilogue >= 0 && epilogue <= EPI_F32) || epilogue == EPI_BIAS_GELU1, "czc_gemm_bf16: epilogue %d", epilogue);
  GemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  return gemm_dispatch(a_layout, b_layout, epilogue, a, splitk, (hipStream_t)stream);
}

extern "C" int czc_gemm_wgrad_pair(int M1, int M2, int N, int K, const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1,
                                   const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, float* C, int splitk,
                                   czc_stream_t stream) {
  GemmArgs a{};
  a.A = A1; a.B = B1; a.lda = lda1; a.ldb = ldb1;
  a.A2 = A2; a.B2 = B2; a.lda2 = lda2; a.ldb2 = ldb2; a.g_rows = M1;
  a.C = C; a.M = M1 + M2; a.N = N; a.K = K; a.ldc = N;
  return gemm_dispatch(CZC_LAYOUT_MNMAJOR, CZC_LAYOUT_MNMAJOR, EPI_F32, a, splitk, (hipStream_t)stream);
}

extern "C" int czc_gemm_set_tile(int cfg) {
  g_store_pol = (cfg & 512) ? 1 : (cfg & 1024) ? 2 : (cfg & 2048) ? 0 : -1;
  cfg &= ~3584;
  CZC_REQUIRE((cfg & ~112) >= 0 && (cfg & ~112) <= 2,
              "czc_gemm_set_tile: cfg=%d (0 default, 1 = 256x256, 2 = 128x128; +16 = one workgroup per tile; "
              "+32 / +64 = ping-pong K-loop always / never; +512 / +1024 / +2048 = sc1 / nt / plain output stores)", cfg);
  g_tile_override = cfg & ~112;
  g_one_wg_per_tile = (cfg & 16) != 0;
  g_pingpong = (cfg & 32) ? 1 : ((cfg & 64) ? 2 : 0);
  return CZC_OK;
}

extern "C" int czc_probe_arm(int a_layout, int b_layout, int epilogue, int M, int N, int K, int max_samples) {
  Probe& p = g_probe;
  if (p.ev) {
    for (int i = 0; i < 2 * p.max_samples; ++i) (void)hipEventDestroy(p.ev[i]);
    delete[] p.ev;
    p.ev = nullptr;
  }
  p.armed = false;
  p.count = 0;
  p.max_samples = 0;
  if (max_samples <= 0) return CZC_OK;  // disarm
  p.ev = new hipEvent_t[2 * max_samples];
  for (int i = 0; i < 2 * max_samples; ++i) CZC_HIP(hipEventCreate(&p.ev[i]));
  p.al = a_layout; p.bl = b_layout; p.epi = epilogue; p.M = M; p.N = N; p.K = K;
  p.max_samples = max_samples;
  p.armed = true;
  return CZC_OK;
}

extern "C" int czc_probe_read(float* total_ms, int* count) {
  Probe& p = g_probe;
  float tot = 0.f;
  for (int i = 0; i < p.count; ++i) {
    float ms = 0.f;
    CZC_HIP(hipEventSynchronize(p.ev[2 * i + 1]));
    CZC_HIP(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = p.count;
  return CZC_OK;
}

extern "C" int czc_reduce_slabs(const float* slabs, int nslabs, int64_t n, float* dst, int accumulate,
                                czc_stream_t stream) {
  CZC_REQUIRE(n % 4 == 0 && nslabs >= 1, "czc_reduce_slabs: n=%lld must be a multiple of 4", (long long)n);
  ProfScope prof((hipStream_t)stream, "reduce_slabs", 0.0, 4.0 * n * (nslabs + 1 + (accumulate ? 1 : 0)));
  const int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_slabs_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, n4, dst, accumulate);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
