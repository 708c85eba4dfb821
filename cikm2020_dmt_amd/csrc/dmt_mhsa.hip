// Self-attention block of the sequence encoder in ONE launch (bf16, E64 geometry: d_model 320 = 4 heads x 80):
//
//     y = LN(x + concat_h softmax(mask(Q_h K_h^T / sqrt(80))) V_h),   (Q | K | V) = x Wqkv + b
//
// = multihead_attention(x, x, x, lens, lens) of /root/reference/DMT_code/model/net/TransformerModel_util.py:160-209 with
// scaled_dot_product_attention :11-56, mask :80-108 (key mask before the softmax, query mask after it, both -2^32 + 1), dropout on
// the attention weights :51 and ln :58-78.  There is no output projection in the reference.
//
// The kernel that ships (MCfg<8, 5>, the one shape built; how it got there: DESIGN.md section 3d):
//   * a workgroup = 8 wavefronts of 32 rows each = 256 rows = 256 / Tp examples, every example padded to Tp in {16, 32, 64} rows; ONE
//     workgroup per CU (149 KB of LDS, <= 256 registers), persistent over the row tiles.  There is no loader wavefront: every
//     wavefront computes AND issues its two pieces of every stage of the weight stream (an LDS-DMA instruction holds its wavefront ~230
//     cycles, so the pieces are spread over all eight), a ring of five 10.5 KB stage slots in LDS;
//   * per head h a wavefront multiplies ITS 32 input rows (20 MFMA fragments, loaded once per row tile and resident for the four
//     heads and their residual adds; the next tile's rows are requested before the LayerNorm pass) with the 240 columns
//     (V_h | K_h | Q_h, head-major order, + 16 zero columns) of the packed projection -- eight 32-column tiles of a prebuilt weight image,
//     each streamed in two k halves.  Every tile is computed transposed (lane = row), so that
//       - Q_h stays in REGISTERS: the packed accumulators ARE the B fragments of S^T = K Q^T (k order of a 16-chunk: 0-3, 8-11 | 4-7,
//         12-15 -- the order the 32x32 accumulator hands out; the K rows in LDS are written in the same order, a reduction does not
//         care as long as both operands agree),
//       - K_h goes to LDS as [key][5 chunks x (lower-lane 16 B | upper-lane 16 B)] (one ds_write_b128 per chunk, read back by
//         ds_read_b128 as MFMA A fragments) and V_h row-major [key][80] (ds_write_b64), read back TRANSPOSED by ds_read_b64_tr_b16 as
//         the A fragments of O^T = V^T P^T,
//       - the (Q | K | V) side output for the backward pass leaves straight from the accumulators in 16-byte pieces of the rows;
//   * attention of the wavefront's 32 queries: a lane owns one query, the softmax is an in-lane reduction plus one exchange with
//     lane + 32; P feeds O^T = V^T P^T from the accumulators; s = O + x is stored per head and its row sums kept (shifted by the row's
//     first input element: no cancellation in the variance);
//   * after the four heads the rows' mean / variance are known: the wavefront re-reads its own s pieces (L2) and writes y = LN(s).
// qkv never travels back from memory, the scores never leave registers, LayerNorm is not a separate launch.  Every global access is a
// buffer access with a 32-bit offset (rows that do not exist: out-of-range offsets, zeros on load, dropped on store).
#include "dmt_common.h"
#include <utility>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int MH_D = 320, MH_H = 4, MH_DH = 80, MH_KC = MH_D / 16;   // 20 k chunks
constexpr int MH_KH = MH_KC / 2;                   // k chunks per stage (half the reduction)
constexpr int MH_WSTRIDE = MH_KH * 32 + 16;        // weight row of a stage: 10 chunks x 32 B + one pad slot = 336 B (21 slots: odd)
constexpr int MH_STAGE = 32 * MH_WSTRIDE;          // 10752 B = 10.5 DMA pieces of 1 KB
constexpr int MH_TPH = 8;                          // 32-column tiles per head: 240 columns + 16 zero columns
constexpr int MH_NSTAGE = MH_H * MH_TPH * 2;       // 64 stages per row tile
constexpr long long MH_IMAGE_BYTES = (long long)MH_NSTAGE * MH_STAGE;
constexpr int MH_K_STRIDE = MH_DH * 2 + 16;        // 176 B: 11 slots
constexpr int MH_V_STRIDE = 192;                   // V_h rows [key][80] bf16 + pad: 96 elements = 32 (mod 128), so that the four key rows x 32 B
                                                   // one 16-lane group of a transpose-read touches fall into disjoint bank quarters (dmt_attn.hip)
// A workgroup = NCW wavefronts of 32 rows each around a ring of NS stage slots; every wavefront computes AND issues its share of the
// weight stream.  What shaped this (measured): (1) every 32 rows need the whole 688 KB image through the
// LDS, so more rows per pass of the image is less stream: 8 wavefronts = 256 rows, one workgroup per CU; (2) an LDS-DMA instruction
// holds its wavefront ~230 cycles and a wavefront that is issuing cannot arrive at the next barrier, so the pieces of a stage are
// spread over ALL wavefronts (two pieces each); (3) dedicated loader wavefronts (rounds 2-3: dmt_chain.hip) would make 9-12
// wavefronts per CU = three on a SIMD = 168 registers, 80 of which the resident input fragments take -- the spills that followed
// cost more than the loaders saved; eight wavefronts get 256 registers each.
// vmcnt with loads, stores and DMA pieces in one queue: "my pieces of stage gs have landed" is vmcnt(2 (AHEAD - 1)) -- the pieces of
// the AHEAD - 1 younger stages may be outstanding.  Side-output stores issued since are not counted, which only forces the OLDEST of
// the younger operations (those issued right behind the awaited pieces, AHEAD stages ago) to retire as well: never a recent store.
template <int NCW_, int NS_>
struct MCfg {
  static constexpr int NCW = NCW_, NS = NS_;
  static constexpr int NT = 64 * NCW;
  static constexpr int ROWS = 32 * NCW;
  static constexpr int K_OFF = NS * MH_STAGE;
  static constexpr int V_OFF = K_OFF + ROWS * MH_K_STRIDE;
  static constexpr int BIAS_OFF = V_OFF + ROWS * MH_V_STRIDE;
  static constexpr int LDS = BIAS_OFF + 4 * 256 * 4;        // bias in head-major order, 256 per head
  static constexpr int NP = 2 * NCW, PB = MH_STAGE / NP;    // DMA pieces per stage (two per wavefront), bytes per piece
  static_assert(PB * NP == MH_STAGE && PB % 16 == 0 && PB <= 1024, "stage pieces");
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert(ROWS % 64 == 0, "rows per workgroup: whole 64-row examples");
};

constexpr float MH_PAD = -4294967295.0f;           // -2^32 + 1 (TransformerModel_util.py:86)

template <int... I, typename F>
__device__ __forceinline__ void mfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void mfor(F&& f) { mfor_impl(std::make_integer_sequence<int, N>{}, f); }

__host__ __device__ constexpr int mh_perm16(int p) { return 4 * (p >> 3) + (p & 3) + 8 * ((p >> 2) & 1); }
// column c' (0..255) of head h in head-major order  V_h (80) | 16 zero columns | K_h (80) | Q_h (80)  -> column of the packed
// [Q | K | V] projection, or -1 (padding).  V first: its tiles' epilogue (column-wise side output) needs registers that the Q fragments,
// live from their tile to the attention, would otherwise compete for; Q last keeps them live for the shortest time.
constexpr int MH_CK = 96, MH_CQ = 176;             // first head-major column of K_h / Q_h
__host__ __device__ constexpr int mh_std_col(int h, int cp) {
  return cp < 80 ? 640 + 80 * h + cp : (cp < MH_CK ? -1 : (cp < MH_CQ ? 320 + 80 * h + (cp - MH_CK) : 80 * h + (cp - MH_CQ)));
}

// ------------------------------------------------------------------------------------------------------------ weight image
// stage u = 2 * (8 h + j) + kh: the rows (= output columns 32 j .. + 31 of head h, head-major) of W over the k half kh; a row is 10
// chunks of 32 B (lower-lane 8 | upper-lane 8 bf16, k permuted inside the 16-chunk: dmt_chain.hip) + one pad slot.
__global__ __launch_bounds__(256) void mhsa_image_kernel(const float* __restrict__ w, long long ldw, unsigned char* __restrict__ img) {
  const long long slots = MH_IMAGE_BYTES / 16;
  constexpr int SPS = MH_STAGE / 16, SPR = MH_WSTRIDE / 16;   // slots per stage (672) / per row (21)
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < slots; s += (long long)gridDim.x * 256) {
    const int u = (int)(s / SPS), within = (int)(s % SPS);
    const int row = within / SPR, slot = within % SPR;
    const int g = u >> 1, kh = u & 1;
    const int col = mh_std_col(g / 8, 32 * (g % 8) + row);
    unsigned short hh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * (MH_KH * kh + (slot >> 1)) + mh_perm16(8 * (slot & 1) + e);
      hh[e] = (col >= 0 && slot < 2 * MH_KH) ? f2bf(w[(long long)k * ldw + col]) : (unsigned short)0;
    }
    u32x4_t o = {(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16), (unsigned)hh[4] | ((unsigned)hh[5] << 16),
                 (unsigned)hh[6] | ((unsigned)hh[7] << 16)};
    *reinterpret_cast<u32x4_t*>(img + s * 16) = o;
  }
}

// ------------------------------------------------------------------------------------------------------------ kernel
struct MhsaArgs {
  const bf16_t* x;          // [B, T, 320] contiguous
  const int* lens;          // [B]
  const unsigned char* image;
  const float* bias;        // [960]
  const float* gamma; const float* beta; float eps;
  bf16_t* qkv;              // [B*T, 960] or null
  bf16_t* s_out;            // [B*T, 320] (training: kept for the LayerNorm gradient; inference: == y_out, normalised in place)
  bf16_t* y_out;            // [B*T, 320]
  float* stats;             // [B*T, 2] or null
  int B, T, Tp, lgTp, tiles;
  const int* blocks;        // packed rows (include/dmt_hip.h): [tiles][8 blocks][2] x (example, length, first packed row, log2 Tp); null: dense
  long long n_rows;         // rows of x / s / y / qkv / stats (B * T when dense)
  unsigned drop_seed, drop_thr;   // thr = keep * 2^24, 0: dropout off
  float drop_inv_keep;
  int dbg;   // timing experiments only (make EXPERIMENTS=1, DMT_MHSA_DEBUG): 1 no qkv side stores, 2 no attention, 4 no projection MFMAs,
             // 8 no LayerNorm pass, 16 no weight-fragment reads, 32 no s stores, 64 no DMA, 128 no input-row loads, 256 no per-stage barrier
};

// LDS accesses of the compute wavefronts are inline asm: hipcc makes every LDS access it can see wait vmcnt(0) while an LDS-DMA may be
// in flight, and orders visible LDS reads against the stores of the side outputs.
template <int OFF> __device__ __forceinline__ void mh_read128(bf16x8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF> __device__ __forceinline__ void mh_read128f(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void mh_write64(unsigned addr, unsigned a, unsigned b) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const u32x2_t v = {a, b};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// transpose-read: the 16 lanes of a group point at 4 rows x (4 x 8 bytes) of a row-major bf16 block; lane i of the group receives column
// i: the 4 rows' elements (dmt_gemm.hip, gemm_dw_glds_kernel)
typedef __attribute__((ext_vector_type(2))) unsigned mh_u32x2_t;
template <int OFF> __device__ __forceinline__ void mh_read_tr(mh_u32x2_t& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void mh_write128(unsigned addr, u32x4_t v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void mh_wait4(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}
template <int N> __device__ __forceinline__ void mh_wait3(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(N));
}
template <int N> __device__ __forceinline__ void mh_wait5(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d, bf16x8_t& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "i"(N));
}
// A value the optimiser may not move computations across: address arithmetic that depends on it is done where it is used instead of
// being hoisted to the top of the (fully unrolled) head body and kept in registers -- or spilled -- across all eight tiles.
__device__ __forceinline__ int mh_here(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ void mh_swap(unsigned& a, unsigned& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}

typedef MCfg<8, 5> C;      // the one shape built: 8 wavefronts = 256 rows per workgroup, ring of 5 stage slots (149 KB of LDS)

// PK: packed rows.  The examples of a tile are described by the caller's block table (one 16-byte entry per half block: example, its
// length, its first packed row, log2 of the tile's padded length); the dense instantiation computes all of that from (tile, B, T) as
// before and compiles to the code it was.  A packed example HAS `length` rows: keys past them do not exist (dense: masked, weight 0).
template <bool PK>
__global__ __launch_bounds__(C::NT, 2) void mhsa_fwd_kernel(const MhsaArgs g) {
  constexpr int MH_NT = C::NT, MH_NS = C::NS, MH_K_OFF = C::K_OFF, MH_V_OFF = C::V_OFF, MH_BIAS_OFF = C::BIAS_OFF;
  constexpr int NCW = C::NCW, ROWS = C::ROWS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 31, hi_ = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  float* bias_lds = reinterpret_cast<float*>(smem + MH_BIAS_OFF);

  // bias in head-major order (zero for the padding columns)
  for (int i = tid; i < 4 * 256; i += MH_NT) {
    const int col = mh_std_col(i >> 8, i & 255);
    bias_lds[i] = col >= 0 ? g.bias[col] : 0.f;
  }
  __syncthreads();
  const int G_ = (int)gridDim.x;
#ifdef DMT_TIMING_EXPERIMENTS
  const int dbg = g.dbg;
#else
  constexpr int dbg = 0;
#endif

  // ================================================================ the weight stream: two pieces per wavefront and stage
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.image), 0, (int)MH_IMAGE_BYTES, 0x00020000);
  constexpr int AHEAD = MH_NS - 1;
  static_assert(2 * (AHEAD - 1) < 64, "vmcnt range");
  auto issue = [&](int gsi) {
    unsigned char* sb = smem + (gsi % MH_NS) * MH_STAGE;
    const int u = gsi % MH_NSTAGE;
    if (lane < C::PB / 16 && !(dbg & 64)) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + wave * C::PB), 16, lane * 16, u * MH_STAGE + wave * C::PB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + (wave + NCW) * C::PB), 16, lane * 16, u * MH_STAGE + (wave + NCW) * C::PB, 0, 0);
    }
  };
  for (int s0 = 0; s0 < AHEAD; ++s0) issue(s0);

  // ================================================================ compute wavefronts
  const int rb = wave;                               // row block (32 rows) of the workgroup's ROWS
  // Every global access of a compute wavefront is a BUFFER access: (4 scalar registers of descriptor) + (one 32-bit byte offset per
  // lane).  No 64-bit address pairs (they were what the register allocator spilled, and a reload from scratch waits -- vmcnt is in
  // order -- behind every side-output store issued before it), and a row that does not exist is an offset past the end: its load
  // returns zeros, its store is dropped, without a branch.
  const long long nrow = PK ? g.n_rows : (long long)g.B * g.T;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.x), 0, (int)(nrow * MH_D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rqkv = __builtin_amdgcn_make_buffer_rsrc(g.qkv, 0, g.qkv ? (int)(unsigned)(nrow * 960 * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.s_out, 0, (int)(nrow * MH_D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(g.y_out, 0, (int)(nrow * MH_D * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;              // a byte offset past the end of every tensor here (and far from wrapping around)
  const unsigned a_lane = lds0 + ml * MH_WSTRIDE + 16 * hi_;         // weight fragment (A operand): row ml of the tile, chunk c at + 32 c
  int Tp = g.Tp, lg = g.lgTp;                        // (packed rows: per tile, from the block table)
  const int T = g.T;
  const int epw = ROWS >> lg;                        // examples per workgroup (dense)
  const float scale = 0.11180339887498948f;          // 1 / sqrt(80)
  int gs = 0;                                        // stages consumed so far by this workgroup

  // row info of a lane -- local row -> (example, position), validity, global row -- is RECOMPUTED where it is used (a handful of VALU
  // operations) instead of living in registers across the tile: with the 80 registers of input fragments resident, every value kept
  // is a value spilled, and a reload from scratch waits (vmcnt is in order) behind every side-output store issued before it.
  // Global rows are 32-bit: every address is (uniform base) + (32-bit offset).
  struct RowInfo { int t_pos, ex; bool rvalid; unsigned grow, growc, row640; };
  struct TileInfo { int ex, len, off, lg; };         // packed rows: this lane's example in the tile (kept in registers for the tile)
  auto load_info = [&](int tl) -> TileInfo {
    const int4 v = *reinterpret_cast<const int4*>(g.blocks + (((long long)tl * NCW + rb) * 2 + ((lane & 31) >> 4)) * 4);
    return TileInfo{v.x, v.y, v.z, v.w};
  };
  auto rowinfo_pk = [&](const TileInfo& ti) -> RowInfo {
    const int r_loc = 32 * rb + (mh_here(lane) & 31);
    RowInfo r;
    r.t_pos = r_loc & ((1 << ti.lg) - 1);
    r.ex = ti.ex;
    r.rvalid = (ti.ex >= 0) && (r.t_pos < ti.len);
    r.grow = (unsigned)ti.off + (unsigned)r.t_pos;
    r.growc = r.rvalid ? r.grow : 0u;
    r.row640 = r.rvalid ? r.grow * (unsigned)(MH_D * 2) : OOB;
    return r;
  };
  auto rowinfo_t = [&](int tl) -> RowInfo {
    const int r_loc = 32 * rb + (mh_here(lane) & 31);
    RowInfo r;
    r.t_pos = r_loc & (Tp - 1);
    r.ex = tl * epw + (r_loc >> lg);
    r.rvalid = (r.ex < g.B) && (r.t_pos < T);
    r.grow = (unsigned)r.ex * (unsigned)T + (unsigned)r.t_pos;
    r.growc = r.rvalid ? r.grow : 0u;
    r.row640 = r.rvalid ? r.grow * (unsigned)(MH_D * 2) : OOB;      // byte offset of the row in x / s / y, or out of range
    return r;
  };
  // the raw input rows of a tile: requested one tile ahead (before the LayerNorm pass of the tile in flight, whose latency they share)
  // (ONE set of 80 registers: the raw rows land in the fragment registers and are swapped into fragment order in place)
  bf16x8_t X[MH_KC];
  TileInfo tnext = TileInfo{-1, 0, 0, 6};            // packed rows: the block entry of the tile whose rows were requested last
  auto request_x = [&](int tl) {
    if constexpr (PK) tnext = load_info(tl);         // (a dependent round trip in front of the row requests: once per tile)
    const RowInfo ri = PK ? rowinfo_pk(tnext) : rowinfo_t(tl);
    const unsigned xo = ri.row640 + 16u * (unsigned)hi_;
#pragma unroll
    for (int c = 0; c < MH_KC; ++c) X[c] = __builtin_bit_cast(bf16x8_t, u32x4_t{0u, 0u, 0u, 0u});
    if (!(dbg & 128)) {
#pragma unroll
      for (int c = 0; c < MH_KC; ++c) X[c] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xo + 32u * c, 0, 0));
    }
  };
  request_x((int)blockIdx.x);

  for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
    const TileInfo tcur = tnext;                     // (packed rows: request_x of this tile left it there)
    if constexpr (PK) { lg = __builtin_amdgcn_readfirstlane(tcur.lg); Tp = 1 << lg; }
    auto rowinfo = [&]() -> RowInfo { if constexpr (PK) return rowinfo_pk(tcur); else return rowinfo_t(tile); };
    int len;
    if constexpr (PK) len = tcur.ex >= 0 ? tcur.len : 0;
    else { const RowInfo ri = rowinfo(); len = (ri.ex < g.B) ? g.lens[ri.ex] : 0; }
    float rsum = 0.f, rsq = 0.f;                               // row statistics of s

    // ---- the wavefront's 32 input rows as MFMA fragments (k permuted: lower lane 0-3 | 8-11, upper 4-7 | 12-15 of every 16-chunk),
    //      loaded ONCE per row tile and kept for the four heads' projections and their residual adds.  (A lane reads its own row in
    //      16-byte pieces -- 32 rows x 32 bytes per instruction --: reloading them per head to free their 80 registers during the
    //      attention cost 150 of 450 us.)
#pragma unroll
    for (int c = 0; c < MH_KC; ++c) {     // (swapped in place: rows past the end were out-of-range loads, i.e. zeros)
      const u32x4_t rw = __builtin_bit_cast(u32x4_t, X[c]);
      unsigned a0 = rw[0], a1 = rw[1], a2 = rw[2], a3 = rw[3];
      mh_swap(a0, a2);
      mh_swap(a1, a3);
      X[c] = __builtin_bit_cast(bf16x8_t, u32x4_t{a0, a1, a2, a3});
    }
    // shift of the row statistics: the row's own first input element (lane hi = 0 holds it; both lanes of a row use the same value).
    // sum (s - c) and sum (s - c)^2 lose nothing to cancellation when a row's mean is far larger than its spread -- the plain
    // E[s^2] - mean^2 form does, and with eps = 1e-8 its clamp at zero would then hand the LayerNorm gradient an rstd of 1e4
    const float cshift = __shfl(__uint_as_float(__builtin_bit_cast(u32x4_t, X[0])[0] << 16), lane & 31, 64);

#pragma unroll 1
    for (int h = 0; h < MH_H; ++h) {
      bf16x8_t Qf[5];                                // Q_h of the wavefront's 32 rows: B fragments of S^T = K Q^T

      // ================= projection: (Q_h | K_h | V_h) = x W_h + b, eight 32-column tiles x two k halves =================
      mfor<MH_TPH>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        f32x16_t acc;
        mfor<2>([&](auto khc) {
          constexpr int kh = decltype(khc)::value;
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(2 * (AHEAD - 1)) : "memory");   // my pieces of stage gs have landed
          if (!(dbg & 256)) __builtin_amdgcn_s_barrier();   // ... everybody's have; everybody has left stage gs - 1 = the slot of stage gs + AHEAD
          const unsigned so = (unsigned)(gs % MH_NS) * MH_STAGE;
          const int gs_issue = gs + AHEAD;
          ++gs;
          // ten weight fragments in batches of 3 + 3 + 3 + 1 over two register sets (the next batch is in flight under the current MFMAs)
          bf16x8_t R0[3], R1[3];
          auto mm = [&](const bf16x8_t& w, int c) {
            if (dbg & 4) return;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, X[c], acc, 0, 0, 0);
          };
          const unsigned wa = a_lane + so;
          if (!(dbg & 16)) {
            mh_read128<0 * 32>(R0[0], wa); mh_read128<1 * 32>(R0[1], wa); mh_read128<2 * 32>(R0[2], wa);
            mh_read128<3 * 32>(R1[0], wa); mh_read128<4 * 32>(R1[1], wa); mh_read128<5 * 32>(R1[2], wa);
          }
          f32x4_t b4[4];
          if constexpr (kh == 0) {
            // bias = initial accumulator: register 4 q + i is head-major column 32 j + 8 q + 4 hi + i
            // (inline-asm reads, like every LDS access of this kernel: a visible one would be made to wait for all DMA in flight)
            const unsigned ba = lds0 + MH_BIAS_OFF + (unsigned)(h * 256 + 32 * j + 4 * mh_here(hi_)) * 4u;
            mh_read128f<0>(b4[0], ba); mh_read128f<32>(b4[1], ba); mh_read128f<64>(b4[2], ba); mh_read128f<96>(b4[3], ba);
          }
          // the wavefront's two DMA pieces of stage gs + AHEAD (an LDS-DMA instruction holds its wavefront ~230 cycles: issued here, behind
          // the fragment reads, their latency runs under it)
          issue(gs_issue);
          if constexpr (kh == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[4 * q + i] = b4[q][i];
          }
          mh_wait3<3>(R0[0], R0[1], R0[2]);
          mm(R0[0], MH_KH * kh + 0); mm(R0[1], MH_KH * kh + 1); mm(R0[2], MH_KH * kh + 2);
          if (!(dbg & 16)) { mh_read128<6 * 32>(R0[0], wa); mh_read128<7 * 32>(R0[1], wa); mh_read128<8 * 32>(R0[2], wa); }
          mh_wait3<3>(R1[0], R1[1], R1[2]);
          mm(R1[0], MH_KH * kh + 3); mm(R1[1], MH_KH * kh + 4); mm(R1[2], MH_KH * kh + 5);
          if (!(dbg & 16)) mh_read128<9 * 32>(R1[0], wa);
          mh_wait3<1>(R0[0], R0[1], R0[2]);
          mm(R0[0], MH_KH * kh + 6); mm(R0[1], MH_KH * kh + 7); mm(R0[2], MH_KH * kh + 8);
          mh_wait3<0>(R1[0], R1[1], R1[2]);
          mm(R1[0], MH_KH * kh + 9);
        });
        unsigned pk[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) pk[p] = dmt_pack_bf16(acc[2 * p], acc[2 * p + 1]);
        {
          // lane (row 32 rb + ml, hi): pk[2 q], pk[2 q + 1] = head-major columns 32 j + 8 q + 4 hi .. + 3; pk[4 c .. 4 c + 3] = the lane's 8 k
          // slots of the 16-column chunk c (0, 1) of the tile
          const int hi = mh_here(hi_);
          const int mlv = mh_here(lane) & 31;
          mfor<2>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int cp0 = 32 * j + 16 * c;     // head-major column of the chunk: < 80 V, < MH_CK padding, < MH_CQ K, else Q
            if constexpr (cp0 < 80) {
              // V_h row-major (natural column order: the transpose-read wants 4 contiguous columns per lane): two 8-byte pieces
              const unsigned va = lds0 + MH_V_OFF + (32 * rb + mlv) * MH_V_STRIDE + (cp0 + 4 * hi) * 2;
              mh_write64(va, pk[4 * c], pk[4 * c + 1]);
              mh_write64(va + 16, pk[4 * c + 2], pk[4 * c + 3]);
            } else if constexpr (cp0 >= MH_CQ) {
              Qf[(cp0 - MH_CQ) / 16] = __builtin_bit_cast(bf16x8_t, u32x4_t{pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]});
            } else if constexpr (cp0 >= MH_CK) {
              mh_write128(lds0 + MH_K_OFF + (32 * rb + mlv) * MH_K_STRIDE + ((cp0 - MH_CK) / 16) * 32 + 16 * hi, u32x4_t{pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]});
            }
          });
          if (g.qkv != nullptr && !(dbg & 1)) {
            const RowInfo ri = rowinfo();
            // 16-byte pieces of the row: pair register group q = 2p (lower lane keeps) with q = 2p + 1 (upper lane)
            unsigned ho[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) ho[p] = pk[p];
            mh_swap(ho[0], ho[2]); mh_swap(ho[1], ho[3]);
            mh_swap(ho[4], ho[6]); mh_swap(ho[5], ho[7]);
            constexpr int cA = 32 * j, cB = 32 * j + 16;     // head-major columns of the two 16-column pieces (each inside V, K, Q or the padding)
            const unsigned row = ri.rvalid ? ri.grow * 1920u + (unsigned)(160 * h + 16 * hi) : OOB;
            constexpr int sA = cA < 80 ? 640 + cA : (cA < MH_CK ? -1 : (cA < MH_CQ ? 320 + cA - MH_CK : cA - MH_CQ));
            constexpr int sB = cB < 80 ? 640 + cB : (cB < MH_CK ? -1 : (cB < MH_CQ ? 320 + cB - MH_CK : cB - MH_CQ));
            if constexpr (sA >= 0) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ho[0], ho[1], ho[2], ho[3]}, rqkv, row + 2u * sA, 0, 0);
            if constexpr (sB >= 0) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ho[4], ho[5], ho[6], ho[7]}, rqkv, row + 2u * sB, 0, 0);
          }
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // K_h, V_h complete in LDS

      // ================= attention of head h for the 32 queries of this row block =================
      auto phase_b = [&](auto dropc, auto tpc) {
        const int hi = mh_here(hi_);
        constexpr bool DROP = decltype(dropc)::value;
        constexpr int TPK = decltype(tpc)::value;          // 64, 32 or 16
        constexpr int NKT = TPK == 64 ? 2 : 1;             // key tiles of 32 rows
        const int kwin = (TPK == 64) ? 64 * (rb >> 1) : 32 * rb;   // first local row of the key window
        // lane = query (position t_pos of its example, length len); score register (kt, r) = key 32 kt + (r & 3) + 8 (r >> 2) + 4 hi of the
        // window.  For Tp >= 32 the window IS the query's example, so "the key exists" is key < T and "is valid" is key < len: both
        // compare a compile-time constant with T - 4 hi / len - 4 hi.  For Tp = 16 a window holds two examples.
        const RowInfo ri = rowinfo();
        const bool rvalid = ri.rvalid;
        const bool q_live = rvalid && (ri.t_pos < len);
        const int Tx = PK ? len : T;                       // keys that EXIST (packed rows: the example's own rows, nothing past them)
        int Tm = Tx - 4 * hi, Lm = len - 4 * hi;           // key < Tx  <=>  const(r) < Tm
        const int ehalf = (mh_here(lane) & 31) >> 4;       // (Tp = 16) which example of the 32-row window this query belongs to
        auto kexists = [&](int cr, int tm) -> bool {
          if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr + 4 * hi) & 15) < Tx);
          else return cr < tm;
        };
        auto kvalid_f = [&](int cr, int lm) -> bool {
          if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr + 4 * hi) & 15) < len);
          else return cr < lm;
        };
        // S^T[key, query] = K Q^T, one 32-key tile at a time (the input rows stay resident: registers are short here): A = K rows from
        // LDS, B = Q fragments from the registers (same k order); the masked, scaled scores are all that is kept
        float sv[16 * NKT];
        float mx = -3.0e38f;
        {
          const unsigned ka = lds0 + MH_K_OFF + (kwin + (mh_here(lane) & 31)) * MH_K_STRIDE + 16 * hi;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            bf16x8_t kf[5];
            if (kt == 0) mfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mh_read128<c * 32>(kf[c], ka); });
            else mfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mh_read128<c * 32 + 32 * MH_K_STRIDE>(kf[c], ka); });
            f32x16_t S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            mh_wait5<0>(kf[0], kf[1], kf[2], kf[3], kf[4]);
#pragma unroll
            for (int c = 0; c < 5; ++c) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], Qf[c], S, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cr = 32 * kt + (r & 3) + 8 * (r >> 2);
              float v = S[r] * scale;
              v = kvalid_f(cr, Lm) ? v : MH_PAD;
              v = kexists(cr, Tm) ? v : -3.0e38f;
              sv[16 * kt + r] = v;
              mx = fmaxf(mx, v);
            }
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < 16 * NKT; ++i) { sv[i] = __expf(sv[i] - mx); den += sv[i]; }   // (a key that does not exist: exp(-3e38 - mx) = 0)
        den += __shfl_xor(den, 32, 64);
        const float inv = (den > 0.f && rvalid) ? 1.f / den : 0.f;
        // query mask AFTER the softmax (:43-48): a padded query row holds -2^32 + 1 on every existing key
        const float padv = (rvalid && !q_live) ? MH_PAD : 0.f;
        const float livef = q_live ? 1.f : 0.f;
        asm volatile("" : "+v"(Tm));                       // (a fresh copy: keeps the compiler from holding 32 lane masks across the softmax)
        unsigned pf[8 * NKT];   // P^T as B fragments (kt, chunk): registers r < 8 -> chunk 0, r >= 8 -> chunk 1
        const unsigned qbase = ((unsigned)(ri.ex * MH_H + h) * (unsigned)T + (unsigned)ri.t_pos) * (unsigned)T + (unsigned)(4 * hi);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float pv[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
              const int rr = r + z;
              const int cr = 32 * kt + (rr & 3) + 8 * (rr >> 2);
              float pq = sv[16 * kt + rr] * inv * livef;
              pq = kexists(cr, Tm) ? pq + padv : 0.f;
              if constexpr (DROP) {
                const unsigned kp = (TPK == 16) ? (unsigned)((cr + 4 * hi) & 15) - (unsigned)(4 * hi) : (unsigned)cr;   // key position - 4 hi
                pq = dmt_drop_keep(g.drop_seed, qbase + kp, g.drop_thr) ? pq * g.drop_inv_keep : 0.f;
              }
              pv[z] = pq;
            }
            pf[8 * kt + (r >> 1)] = dmt_pack_bf16(pv[0], pv[1]);
          }
        // O^T[d, query] = V^T P^T, one 32-column tile of the head at a time: A = V^T rows (d), keys in MFMA k order; B = the P fragments.
        // s = O + x on the tile's columns -- register 4 q + i of lane (query, hi) = column 80 h + 32 td + 8 q + 4 hi + i, and the residual
        // sits in the resident input fragments: chunk 5 h + 2 td + (q >> 1), element 4 (q & 1) + i -- is rounded, stored, and summed.
        const unsigned sro = rowinfo().row640 + (unsigned)(160 * h + 16 * hi);      // (out of range for a row that does not exist)
        const unsigned hm0 = h == 0 ? 0xFFFFFFFFu : 0u, hm1 = h == 1 ? 0xFFFFFFFFu : 0u, hm2 = h == 2 ? 0xFFFFFFFFu : 0u, hm3 = h == 3 ? 0xFFFFFFFFu : 0u;
        mfor<3>([&](auto tdc) {
          constexpr int td = decltype(tdc)::value;
          // V^T fragments by transpose-reads of the row-major V_h: the 16-lane group (lane >> 4) covers columns 32 td + 16 ((lane >> 4) & 1)
          // .. + 15 and the keys of k group hi; a lane points at key (lane & 15) >> 2 of its 4-key block, 8 bytes at column 4 (lane & 3);
          // k slots 0-3 = keys 16 ck + 4 hi .. + 3, slots 4-7 = keys 16 ck + 8 + 4 hi .. + 3 (the order of the P fragments)
          const int l16 = mh_here(lane) & 15, gq = (mh_here(lane) >> 4) & 1;
          int dcol = 32 * td + 16 * gq + 4 * (l16 & 3);
          dcol = dcol < MH_DH ? dcol : MH_DH - 4;            // (columns past the head: any valid address; their rows of O^T are not used)
          const unsigned va = lds0 + MH_V_OFF + (kwin + 4 * hi + (l16 >> 2)) * MH_V_STRIDE + dcol * 2;
          mh_u32x2_t vlo[4], vhi[4];
          mh_read_tr<0>(vlo[0], va); mh_read_tr<8 * MH_V_STRIDE>(vhi[0], va);
          mh_read_tr<16 * MH_V_STRIDE>(vlo[1], va); mh_read_tr<24 * MH_V_STRIDE>(vhi[1], va);
          if constexpr (NKT == 2) {
            mh_read_tr<32 * MH_V_STRIDE>(vlo[2], va); mh_read_tr<40 * MH_V_STRIDE>(vhi[2], va);
            mh_read_tr<48 * MH_V_STRIDE>(vlo[3], va); mh_read_tr<56 * MH_V_STRIDE>(vhi[3], va);
          }
          f32x16_t O;
#pragma unroll
          for (int r = 0; r < 16; ++r) O[r] = 0.f;
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
          if constexpr (NKT == 2) asm volatile("" : "+v"(vlo[2]), "+v"(vhi[2]), "+v"(vlo[3]), "+v"(vhi[3]));
#pragma unroll
          for (int ck = 0; ck < 2 * NKT; ++ck) {
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, make_uint4(pf[4 * ck], pf[4 * ck + 1], pf[4 * ck + 2], pf[4 * ck + 3]));
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(vlo[ck][0], vlo[ck][1], vhi[ck][0], vhi[ck][1]));
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, O, 0, 0, 0);
          }
          mfor<2>([&](auto ppc) {                    // pairs of register groups (q even, q odd): columns 32 td + 16 pp .. + 15
            constexpr int pp = decltype(ppc)::value;
            constexpr int pr = 2 * td + pp;
            if constexpr (pr < 5) {
            // the residual chunk X[5 h + pr] (h is a loop variable: selected, not indexed)
            // (by value with bit masks: a conditional LOAD becomes a load through a selected pointer, and the fragments leave the registers)
            const u32x4_t x0 = __builtin_bit_cast(u32x4_t, X[pr]), x1 = __builtin_bit_cast(u32x4_t, X[5 + pr]);
            const u32x4_t x2 = __builtin_bit_cast(u32x4_t, X[10 + pr]), x3 = __builtin_bit_cast(u32x4_t, X[15 + pr]);
            u32x4_t xc;
#pragma unroll
            for (int e = 0; e < 4; ++e) xc[e] = (x0[e] & hm0) | (x1[e] & hm1) | (x2[e] & hm2) | (x3[e] & hm3);
            const int q0 = 2 * pp;
            float va_[4], vb_[4];
            va_[0] = O[4 * q0 + 0] + __uint_as_float(xc[0] << 16); va_[1] = O[4 * q0 + 1] + __uint_as_float(xc[0] & 0xFFFF0000u);
            va_[2] = O[4 * q0 + 2] + __uint_as_float(xc[1] << 16); va_[3] = O[4 * q0 + 3] + __uint_as_float(xc[1] & 0xFFFF0000u);
            vb_[0] = O[4 * q0 + 4] + __uint_as_float(xc[2] << 16); vb_[1] = O[4 * q0 + 5] + __uint_as_float(xc[2] & 0xFFFF0000u);
            vb_[2] = O[4 * q0 + 6] + __uint_as_float(xc[3] << 16); vb_[3] = O[4 * q0 + 7] + __uint_as_float(xc[3] & 0xFFFF0000u);
            unsigned a[2], b[2];
            a[0] = dmt_pack_bf16(va_[0], va_[1]); a[1] = dmt_pack_bf16(va_[2], va_[3]);
            b[0] = dmt_pack_bf16(vb_[0], vb_[1]); b[1] = dmt_pack_bf16(vb_[2], vb_[3]);
            // statistics of the ROUNDED values (what the LayerNorm gradient will read back)
#pragma unroll
            for (int z = 0; z < 2; ++z) {
              const float f0 = __uint_as_float(a[z] << 16), f1 = __uint_as_float(a[z] & 0xFFFF0000u);
              const float f2 = __uint_as_float(b[z] << 16), f3 = __uint_as_float(b[z] & 0xFFFF0000u);
              const float e0 = f0 - cshift, e1 = f1 - cshift, e2 = f2 - cshift, e3 = f3 - cshift;
              rsum += (e0 + e1) + (e2 + e3);
              rsq += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            mh_swap(a[0], b[0]); mh_swap(a[1], b[1]);
            if (!(dbg & 32)) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{a[0], a[1], b[0], b[1]}, rs, sro + 32u * pr, 0, 0);
            }
          });
        });
      };
      const bool drop = g.drop_thr != 0u;
      if (dbg & 2) { }
      else if (Tp == 64) { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 64>{}); else phase_b(std::false_type{}, std::integral_constant<int, 64>{}); }
      else if (Tp == 32) { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 32>{}); else phase_b(std::false_type{}, std::integral_constant<int, 32>{}); }
      else { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 16>{}); else phase_b(std::false_type{}, std::integral_constant<int, 16>{}); }
      // (no barrier here: the next head's first stage barrier comes before any wavefront overwrites K / V^T)
    }

    if (tile + G_ < g.tiles) request_x(tile + G_);      // (the fragments' registers are free from here on)
    // ================= LayerNorm over the row: re-read this lane's own pieces of s =================
    {
      rsum += __shfl_xor(rsum, 32, 64);
      rsq += __shfl_xor(rsq, 32, 64);
      const float dmean = rsum / (float)MH_D;                  // mean - cshift
      const float mean = dmean + cshift;
      float var = rsq / (float)MH_D - dmean * dmean;
      var = var > 0.f ? var : 0.f;
      const float rstd = 1.f / sqrtf(var + g.eps);
      const RowInfo ri = rowinfo();
      const unsigned grow = ri.grow;
      if (ri.rvalid && !(dbg & 8)) {
        const int hi = hi_;
        if (g.stats != nullptr && hi == 0) { g.stats[2u * grow] = mean; g.stats[2u * grow + 1u] = rstd; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's s stores are in the L2
        const unsigned ro = ri.row640 + 16u * (unsigned)hi;
#pragma unroll 5
        for (int c = 0; c < 20; ++c) {
          const int col = 16 * c + 8 * hi;
          const u32x4_t sv1 = __builtin_amdgcn_raw_buffer_load_b128(rs, ro + 32u * c, 0, 1);   // (sc0: past the L1)
          const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(g.gamma + col), g1 = *reinterpret_cast<const f32x4_t*>(g.gamma + col + 4);
          const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(g.beta + col), b1 = *reinterpret_cast<const f32x4_t*>(g.beta + col + 4);
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(sv1[e] << 16), hi_ = __uint_as_float(sv1[e] & 0xFFFF0000u);
            const float ga = (2 * e < 4) ? g0[2 * e] : g1[2 * e - 4], gb = (2 * e + 1 < 4) ? g0[2 * e + 1] : g1[2 * e + 1 - 4];
            const float ba = (2 * e < 4) ? b0[2 * e] : b1[2 * e - 4], bb = (2 * e + 1 < 4) ? b0[2 * e + 1] : b1[2 * e + 1 - 4];
            o[2 * e] = ga * ((lo - mean) * rstd) + ba;
            o[2 * e + 1] = gb * ((hi_ - mean) * rstd) + bb;
          }
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{dmt_pack_bf16(o[0], o[1]), dmt_pack_bf16(o[2], o[3]), dmt_pack_bf16(o[4], o[5]), dmt_pack_bf16(o[6], o[7])}, ry, ro + 32u * c, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring runs ahead: let it land before the LDS goes away
}


}  // namespace

extern "C" int dmt_mhsa_image_bytes(int64_t* bytes) {
  DMT_CHECK_ARG(bytes != nullptr, "dmt_mhsa_image_bytes: null output");
  *bytes = MH_IMAGE_BYTES;
  return DMT_OK;
}

extern "C" int dmt_mhsa_image_build(const float* wqkv, int64_t ldw, void* image, void* stream) {
  DMT_CHECK_ARG(wqkv && image && ldw >= 960, "dmt_mhsa_image_build: bad argument");
  hipLaunchKernelGGL(mhsa_image_kernel, dim3(168), dim3(256), 0, (hipStream_t)stream, wqkv, (long long)ldw, (unsigned char*)image);
  DMT_CHECK_LAUNCH("dmt_mhsa_image_build");
  return DMT_OK;
}

extern "C" int dmt_mhsa_block_fwd(const dmt_mhsa_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_mhsa_block_fwd: null descriptor");
  DMT_CHECK_ARG(d->d_model == MH_D && d->num_heads == MH_H, "dmt_mhsa_block_fwd: built for d_model %d, %d heads (got %d, %d)", MH_D, MH_H, d->d_model, d->num_heads);
  DMT_CHECK_ARG(d->B > 0 && d->T > 0 && d->T <= 64, "dmt_mhsa_block_fwd: 1 <= T <= 64 (got %d)", d->T);
  DMT_CHECK_ARG(d->x && d->lens && d->image && d->bias && d->gamma && d->beta && d->y_out, "dmt_mhsa_block_fwd: null pointer");
  DMT_CHECK_ARG((((uintptr_t)d->x | (uintptr_t)d->s_out | (uintptr_t)d->y_out | (uintptr_t)d->qkv) & 15) == 0, "dmt_mhsa_block_fwd: tensors must be 16-byte aligned");
  DMT_CHECK_ARG((long long)d->B * d->T * d->num_heads * d->T < (1ll << 32), "dmt_mhsa_block_fwd: dropout counter range");
  const bool packed = d->blocks != nullptr;
  DMT_CHECK_ARG(!packed || (d->n_tiles > 0 && d->n_rows > 0 && d->n_rows <= (long long)d->B * d->T && (((uintptr_t)d->blocks) & 15) == 0),
                "dmt_mhsa_block_fwd: packed rows need n_tiles > 0, 0 < n_rows <= B * T and a 16-byte aligned block table");
  DMT_CHECK_ARG((packed ? (long long)d->n_rows : (long long)d->B * d->T) * 1920 < 0x7FFF0000ll, "dmt_mhsa_block_fwd: 32-bit byte offsets (rows * 1920 < 2^31)");
  MhsaArgs a;
  a.x = (const bf16_t*)d->x; a.lens = d->lens; a.image = (const unsigned char*)d->image;
  a.bias = d->bias; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
  a.qkv = (bf16_t*)d->qkv; a.y_out = (bf16_t*)d->y_out; a.stats = d->stats;
  a.s_out = d->s_out ? (bf16_t*)d->s_out : (bf16_t*)d->y_out;     // inference: the pre-norm sum passes through y and is normalised in place
  a.B = d->B; a.T = d->T;
  a.Tp = d->T > 32 ? 64 : (d->T > 16 ? 32 : 16);
  a.lgTp = a.Tp == 64 ? 6 : (a.Tp == 32 ? 5 : 4);
  a.drop_seed = d->drop_seed;
  const bool drop = d->drop_keep > 0.f && d->drop_keep < 1.f;
  a.drop_thr = drop ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv_keep = drop ? 1.0f / d->drop_keep : 1.0f;
  a.dbg = 0;
  int shape = 0;
#ifdef DMT_TIMING_EXPERIMENTS
  { const char* e = getenv("DMT_MHSA_DEBUG"); a.dbg = e ? atoi(e) : 0; }
  { const char* e = getenv("DMT_MHSA_SHAPE"); shape = e ? atoi(e) : 0; }
#endif
  (void)shape;
  const int epw = C::ROWS / a.Tp;
  a.tiles = packed ? d->n_tiles : (d->B + epw - 1) / epw;
  a.blocks = d->blocks;
  a.n_rows = packed ? d->n_rows : (long long)d->B * d->T;
  const int grid = a.tiles < 256 ? a.tiles : 256;     // one workgroup per CU, persistent over the row tiles
  if (packed) {
    hipLaunchKernelGGL(mhsa_fwd_kernel<true>, dim3(grid), dim3(C::NT), 0, (hipStream_t)stream, a);
    DMT_CHECK_LAUNCH("dmt_mhsa_block_fwd(packed)");
    return DMT_OK;
  }
  hipLaunchKernelGGL(mhsa_fwd_kernel<false>, dim3(grid), dim3(C::NT), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_mhsa_block_fwd");
  return DMT_OK;
}
