// Backward of the encoder's self-attention block in ONE launch (bf16, E64 geometry: d_model 320 = 4 heads x 80, T <= 64):
//
//     given ds = d loss / d s  (s = x + concat_h softmax(mask(Q_h K_h^T / sqrt 80)) V_h, the sum the LayerNorm reads) and the saved
//     (Q | K | V) = x Wqkv + b:       dqkv = attention gradient,       dx = dqkv Wqkv^T + ds
//
// = the gradient of multihead_attention(x, x, x, lens, lens) of /root/reference/DMT_code/model/net/TransformerModel_util.py:160-209 with
// scaled_dot_product_attention :11-56 and mask :80-108 (key mask before the softmax, query mask after it, dropout on the weights :51);
// there is no output projection, so d s IS the gradient of the attention output and of the residual.  The LayerNorm
// gradient (:58-78) in front of it is dmt_ln_bwd; the weight gradient dWqkv = x^T dqkv behind it dmt_wgrad320 (it reads the dqkv this
// kernel writes).  Replaces, per sequence: dmt_attn_bwd + the [M, 960] x [960, 320] GEMM with its residual (two launches, dqkv and ds
// read back from HBM by the second).
//
// Shape (the forward's: dmt_mhsa.hip): a workgroup = 8 wavefronts of 32 rows = 256 rows = 256 / Tp examples, every example padded to
// Tp in {16, 32, 64}; one workgroup per CU, persistent over the row tiles; dense [B, T, .] rows or packed rows with the caller's block
// table.  Per tile:
//   per head h (four times):
//     (a) every lane requests its own row's Q_h, K_h, V_h, dO_h pieces (16 bytes each: lane (row, hi) holds columns 16 c + 8 hi .. + 7 --
//         as they lie in memory they ARE the B fragments of a transposed product, lane = row); K_h and V_h rows go to LDS row-major;
//     (i) lane = QUERY: S^T = K Q^T and dP^T = V dO^T from the LDS rows (A) and the lane's own Q / dO (B); softmax statistics, the
//         dropout bits, D = sum P dP and dS^T in registers; (m, 1 / sum, D, kind of row) of every query go to LDS; dS^T, packed, is
//         the B operand of dQ^T = K^T dS^T (A: transposing LDS reads of the K rows); dQ rows leave in 16-byte pieces;
//     (ii) Q_h and dO_h rows replace K_h and V_h in LDS; lane = KEY: S = Q K^T and dP = dO V^T recomputed with the lane's own K / V as
//         B, the queries' statistics read back from LDS, so that P~ (with dropout) and dS come out with the queries in the accumulator's
//         REGISTERS = the reduction index of dV^T = dO^T P~ and dK^T = Q^T dS (A: transposing reads of the dO / Q rows); no transposition
//         of a score tile through memory, no atomics (a wavefront owns its 32 key rows);
//   then dx = dqkv Wqkv^T + ds for the wavefront's 32 rows: the dqkv pieces it has just written come back (L2) as B fragments, ten k
//   chunks at a time, the weights stream through an LDS ring from a prebuilt image (60 stages of 32 output columns x 160 k; the ring
//   lies over the K / V tiles, which are dead by then), two passes of five 32-column output tiles (80 accumulator registers).
// Every global access is a buffer access with a 32-bit offset (rows that do not exist: out-of-range offsets -> zeros / dropped).
#include "dmt_common.h"
#include <utility>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int MB_D = 320, MB_H = 4, MB_DH = 80;
constexpr int MB_NCW = 8, MB_NT = 64 * MB_NCW, MB_ROWS = 32 * MB_NCW;
constexpr int MB_RS = 208;                           // LDS row stride of a head's [row][80] bf16 tile: 13 x 16 B (odd: the 16-byte row reads of 16
                                                     // lanes fall into 16 different bank groups) and 52 dwords (the four rows of a transposing read: 0, 52, 40, 28 mod 64)
constexpr int MB_RA = 0, MB_RB = MB_ROWS * MB_RS;    // tile A: K_h rows, then Q_h rows; tile B: V_h rows, then dO_h rows
constexpr int MB_ST = 2 * MB_ROWS * MB_RS;           // per-query statistics: m | 1 / sum | D | kind, [256] floats each
constexpr int MB_LDS = MB_ST + 4 * MB_ROWS * 4;
// weight stream of the dx GEMM: the forward's stage geometry (32 output columns x 10 k chunks of 16, 32 B per chunk + one pad slot)
constexpr int MB_WSTRIDE = 10 * 32 + 16, MB_STAGE = 32 * MB_WSTRIDE;      // 336 B rows, 10752 B stages
constexpr int MB_NS = 5, MB_AHEAD = MB_NS - 1;
constexpr int MB_NSTAGE = 60;                        // 2 passes x 6 k stages x 5 output tiles
constexpr long long MB_IMAGE_BYTES = (long long)MB_NSTAGE * MB_STAGE;
constexpr int MB_PB = MB_STAGE / (2 * MB_NCW);       // DMA piece: two per wavefront and stage
static_assert(MB_NS * MB_STAGE <= MB_ST, "the ring lies over the K / V tiles");
static_assert(MB_PB * 2 * MB_NCW == MB_STAGE && MB_PB % 16 == 0 && MB_PB <= 1024, "stage pieces");
static_assert(MB_LDS <= 160 * 1024, "LDS");

constexpr float MB_PAD = -4294967295.0f;             // -2^32 + 1 (TransformerModel_util.py:86)
constexpr float MB_LOG2E = 1.4426950408889634f;

template <int... I, typename F>
__device__ __forceinline__ void bfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void bfor(F&& f) { bfor_impl(std::make_integer_sequence<int, N>{}, f); }

// ------------------------------------------------------------------------------------------------------------ weight image
// stage u = (6 p + s) * 5 + jj: rows = output columns 32 (5 p + jj) .. + 31 of dx (= input rows of Wqkv), k = dqkv columns 160 s .. + 159 in
// NATURAL order (the B fragments are 16-byte pieces of the dqkv rows as they lie in memory): chunk c = k 16 c .. + 15, lower lane 8 | upper lane 8
__global__ __launch_bounds__(256) void mhsa_bwd_image_kernel(const float* __restrict__ w, long long ldw, unsigned char* __restrict__ img) {
  const long long slots = MB_IMAGE_BYTES / 16;
  constexpr int SPS = MB_STAGE / 16, SPR = MB_WSTRIDE / 16;   // slots per stage (672) / per row (21)
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < slots; s += (long long)gridDim.x * 256) {
    const int u = (int)(s / SPS), within = (int)(s % SPS);
    const int row = within / SPR, slot = within % SPR;
    const int p = u / 30, st = (u % 30) / 5, jj = u % 5;
    const int n = 32 * (5 * p + jj) + row;
    unsigned short hh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 160 * st + 8 * slot + e;
      hh[e] = (slot < 20) ? f2bf(w[(long long)n * ldw + k]) : (unsigned short)0;
    }
    u32x4_t o = {(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16), (unsigned)hh[4] | ((unsigned)hh[5] << 16),
                 (unsigned)hh[6] | ((unsigned)hh[7] << 16)};
    *reinterpret_cast<u32x4_t*>(img + s * 16) = o;
  }
}

struct MhsaBwdArgs {
  const bf16_t* ds;         // [rows, 320]: d loss / d (pre-LayerNorm sum)
  const bf16_t* qkv;        // [rows, 960]: the forward's side output
  const int* lens;          // [B]
  const unsigned char* image;
  bf16_t* dqkv;             // [rows, 960] out
  bf16_t* dx;               // [rows, 320] out
  int B, T, Tp, lgTp, tiles;
  const int* blocks;        // packed rows: the forward's block table; null: dense
  long long n_rows;
  unsigned drop_seed, drop_thr;
  float drop_inv_keep;
  int dbg;      // timing experiments only (make EXPERIMENTS=1, DMT_MHSA_BWD_DEBUG): 1 no phase (i), 2 no phase (ii), 4 no dx GEMM; results are garbage
};

// LDS accesses are inline asm (as in dmt_mhsa.hip: hipcc makes every LDS access it can see wait vmcnt(0) while an LDS-DMA may be in flight)
template <int OFF> __device__ __forceinline__ void mb_read128(bf16x8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF> __device__ __forceinline__ void mb_read128f(f32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF> __device__ __forceinline__ void mb_read_tr(u32x2_t& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF> __device__ __forceinline__ void mb_write128(unsigned addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void mb_write32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void mb_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int mb_here(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ void mb_swap(unsigned& a, unsigned& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void mb_swapf(float& a, float& b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  mb_swap(ua, ub);
  a = __uint_as_float(ua); b = __uint_as_float(ub);
}

// vm operations issued after the two DMA pieces of stage u and before the wait in front of stage u (the dx GEMM's schedule below): the
// pieces of the AHEAD - 1 younger stages that exist, and the ten B-fragment loads of a k group issued at the end of a group that closed in
// between.  vmcnt(this) at that wait = "my pieces of stage u have landed" (loads return in order; stores in the queue only make it stricter).
__host__ __device__ constexpr int mb_younger(int u) {
  // pieces: issued so far are those of stages 0 .. u - 1 + AHEAD (the prologue's AHEAD, then one stage's per stage)
  const int last = (u + MB_AHEAD - 1 < MB_NSTAGE - 1) ? u + MB_AHEAD - 1 : MB_NSTAGE - 1;
  int n = 2 * (last - u);
  // B loads of group v / 5 + 2 are issued at the END of stage v = 5 g + 4; pieces(u) were issued in the MIDDLE of stage u - AHEAD (u < AHEAD: in
  // the prologue, behind the first two groups' loads)
  const int first = u >= MB_AHEAD ? u - MB_AHEAD : 0;
  for (int v = first; v < u; ++v) n += (v % 5 == 4 && v / 5 + 2 < 12) ? 10 : 0;
  return n;
}

template <bool PK>
__global__ __launch_bounds__(MB_NT, 2) void mhsa_bwd_kernel(const MhsaBwdArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[MB_LDS];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 31, hi_ = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  const int rb = wave;
  const int G_ = (int)gridDim.x;
#ifdef DMT_TIMING_EXPERIMENTS
  const int dbg = g.dbg;
#else
  constexpr int dbg = 0;
#endif

  const long long nrow = PK ? g.n_rows : (long long)g.B * g.T;
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.image), 0, (int)MB_IMAGE_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.ds), 0, (int)(nrow * MB_D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rqkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.qkv), 0, (int)(unsigned)(nrow * 960 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc(g.dqkv, 0, (int)(unsigned)(nrow * 960 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(g.dx, 0, (int)(nrow * MB_D * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  int Tp = g.Tp, lg = g.lgTp;
  const int T = g.T;
  const int epw = MB_ROWS >> lg;
  const float kscale = MB_LOG2E * 0.11180339887498948f;      // log2(e) / sqrt(80)
  const float inv_sc = 0.11180339887498948f;

  struct RowInfo { int t_pos, ex; bool rvalid; unsigned grow; };
  struct TileInfo { int ex, len, off, lg; };
  auto load_info = [&](int tl) -> TileInfo {
    const int4 v = *reinterpret_cast<const int4*>(g.blocks + (((long long)tl * MB_NCW + rb) * 2 + ((lane & 31) >> 4)) * 4);
    return TileInfo{v.x, v.y, v.z, v.w};
  };
  auto rowinfo_pk = [&](const TileInfo& ti) -> RowInfo {
    const int r_loc = 32 * rb + (mb_here(lane) & 31);
    RowInfo r;
    r.t_pos = r_loc & ((1 << ti.lg) - 1);
    r.ex = ti.ex;
    r.rvalid = (ti.ex >= 0) && (r.t_pos < ti.len);
    r.grow = (unsigned)ti.off + (unsigned)r.t_pos;
    return r;
  };
  auto rowinfo_t = [&](int tl) -> RowInfo {
    const int r_loc = 32 * rb + (mb_here(lane) & 31);
    RowInfo r;
    r.t_pos = r_loc & (Tp - 1);
    r.ex = tl * epw + (r_loc >> lg);
    r.rvalid = (r.ex < g.B) && (r.t_pos < T);
    r.grow = (unsigned)r.ex * (unsigned)T + (unsigned)r.t_pos;
    return r;
  };

  // the weight stream: two pieces per wavefront and stage (stage u of the tile -> ring slot u % NS)
  auto issue = [&](int u) {
    unsigned char* sb = smem + (u % MB_NS) * MB_STAGE;
    if (lane < MB_PB / 16) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + wave * MB_PB), 16, lane * 16, u * MB_STAGE + wave * MB_PB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + (wave + MB_NCW) * MB_PB), 16, lane * 16, u * MB_STAGE + (wave + MB_NCW) * MB_PB, 0, 0);
    }
  };

  for (int tile = (int)blockIdx.x; tile < g.tiles; tile += G_) {
    TileInfo tcur = TileInfo{-1, 0, 0, 6};
    if constexpr (PK) { tcur = load_info(tile); lg = __builtin_amdgcn_readfirstlane(tcur.lg); Tp = 1 << lg; }
    auto rowinfo = [&]() -> RowInfo { if constexpr (PK) return rowinfo_pk(tcur); else return rowinfo_t(tile); };
    int len;
    {
      const RowInfo ri = rowinfo();
      if constexpr (PK) len = tcur.ex >= 0 ? tcur.len : 0;
      else len = (ri.ex < g.B) ? g.lens[ri.ex] : 0;
    }
    const unsigned r_loc_b = (unsigned)(32 * rb + ml) * (unsigned)MB_RS + 16u * (unsigned)hi_;     // this lane's row in an LDS tile, its half's 16 bytes

#pragma unroll 1
    for (int h = 0; h < MB_H; ++h) {
      // ================= (a) this lane's own row: Q_h, K_h, V_h, dO_h pieces =================
      u32x4_t Qr[5], Dr[5];
      {
        const RowInfo ri = rowinfo();
        const unsigned r960 = ri.rvalid ? ri.grow * 1920u + (unsigned)(160 * h + 16 * hi_) : OOB;
        const unsigned r320 = ri.rvalid ? ri.grow * 640u + (unsigned)(160 * h + 16 * hi_) : OOB;
        u32x4_t Kr[5], Vr[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          Kr[c] = __builtin_amdgcn_raw_buffer_load_b128(rqkv, r960 + 640u + 32u * c, 0, 0);
          Vr[c] = __builtin_amdgcn_raw_buffer_load_b128(rqkv, r960 + 1280u + 32u * c, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          Qr[c] = __builtin_amdgcn_raw_buffer_load_b128(rqkv, r960 + 32u * c, 0, 0);
          Dr[c] = __builtin_amdgcn_raw_buffer_load_b128(rds, r320 + 32u * c, 0, 0);
        }
        bfor<5>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          mb_write128<MB_RA + 32 * c>(lds0 + r_loc_b, Kr[c]);
          mb_write128<MB_RB + 32 * c>(lds0 + r_loc_b, Vr[c]);
        });
      }
      mb_lgkm0();
      __builtin_amdgcn_s_barrier();           // K_h, V_h of the tile's 256 rows are in LDS

      // ================= (i) lane = query =================
      auto phase_q = [&](auto dropc, auto tpc) {
        constexpr bool DROP = decltype(dropc)::value;
        constexpr int TPK = decltype(tpc)::value;
        constexpr int NKT = TPK == 64 ? 2 : 1;
        const int hi = mb_here(hi_);
        const int kwin = (TPK == 64) ? 64 * (rb >> 1) : 32 * rb;
        const RowInfo ri = rowinfo();
        const bool rvalid = ri.rvalid;
        const bool q_live = rvalid && (ri.t_pos < len);
        const int Tx = PK ? len : T;                      // keys that EXIST
        int Tm = Tx - 4 * hi, Lm = len - 4 * hi;
        const int ehalf = (mb_here(lane) & 31) >> 4;
        auto kexists = [&](int cr, int tm) -> bool {
          if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr + 4 * hi) & 15) < Tx);
          else return cr < tm;
        };
        auto kvalid_f = [&](int cr, int lm) -> bool {
          if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr + 4 * hi) & 15) < len);
          else return cr < lm;
        };
        f32x16_t S[NKT], dP[NKT];
        {
          const unsigned ka = lds0 + MB_RA + (unsigned)(kwin + (mb_here(lane) & 31)) * (unsigned)MB_RS + 16u * (unsigned)hi;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) {
            bf16x8_t kf[5], vf[5];
            if (kt == 0) bfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mb_read128<MB_RA - MB_RA + c * 32>(kf[c], ka); mb_read128<MB_RB - MB_RA + c * 32>(vf[c], ka); });
            else bfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mb_read128<c * 32 + 32 * MB_RS>(kf[c], ka); mb_read128<MB_RB - MB_RA + c * 32 + 32 * MB_RS>(vf[c], ka); });
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[kt][r] = 0.f; dP[kt][r] = 0.f; }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]));
            asm volatile("" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]));
#pragma unroll
            for (int c = 0; c < 5; ++c) {
              S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], __builtin_bit_cast(bf16x8_t, Qr[c]), S[kt], 0, 0, 0);      // S^T = K Q^T
              dP[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c], __builtin_bit_cast(bf16x8_t, Dr[c]), dP[kt], 0, 0, 0);    // dP^T = V dO^T
            }
          }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cr = 32 * kt + (r & 3) + 8 * (r >> 2);
            float v = S[kt][r] * kscale;
            v = kvalid_f(cr, Lm) ? v : MB_PAD * MB_LOG2E;
            v = kexists(cr, Tm) ? v : -3.0e38f;
            S[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(S[kt][r] - mx); S[kt][r] = e; den += e; }
        den += __shfl_xor(den, 32, 64);
        const float inv = (den > 0.f && q_live) ? __builtin_amdgcn_rcpf(den) : 0.f;       // (a padded query row: its weights are constants, no gradient passes)
        asm volatile("" : "+v"(Tm), "+v"(Lm));
        const unsigned qbase = ((unsigned)(ri.ex * MB_H + h) * (unsigned)T + (unsigned)ri.t_pos) * (unsigned)T + (unsigned)(4 * hi);
        float dot = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cr = 32 * kt + (r & 3) + 8 * (r >> 2);
            const float pv = S[kt][r] * inv;
            float gq = dP[kt][r];
            if constexpr (DROP) {
              const unsigned kp = (TPK == 16) ? (unsigned)((cr + 4 * hi) & 15) - (unsigned)(4 * hi) : (unsigned)cr;
              gq = dmt_drop_keep(g.drop_seed, qbase + kp, g.drop_thr) ? gq * g.drop_inv_keep : 0.f;     // gradient w.r.t. the pre-dropout weights
            }
            S[kt][r] = pv;
            dP[kt][r] = gq;
            dot += pv * gq;
          }
        dot += __shfl_xor(dot, 32, 64);
        // statistics for phase (ii): kind 0 = no such row, 1 = live query, 2 = padded query of an existing row (constant weights -2^32 + 1)
        if (hi == 0) {
          const unsigned sa = lds0 + MB_ST + (unsigned)(32 * rb + (mb_here(lane) & 31)) * 4u;
          mb_write32(sa, mx);
          mb_write32(sa + MB_ROWS * 4, inv);
          mb_write32(sa + 2 * MB_ROWS * 4, dot);
          mb_write32(sa + 3 * MB_ROWS * 4, rvalid ? (q_live ? 1.f : 2.f) : 0.f);
        }
        // dS^T, packed: the B fragments (chunk u: keys 16 u .. + 15 of the window in the accumulator's order) of dQ^T = K^T dS^T
        unsigned dsb[8 * NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float d2[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
              const int cr = 32 * kt + ((r + z) & 3) + 8 * ((r + z) >> 2);
              const float v = S[kt][r + z] * (dP[kt][r + z] - dot) * inv_sc;
              d2[z] = kvalid_f(cr, Lm) ? v : 0.f;               // no gradient into masked keys (inv = 0 already silences padded queries)
            }
            dsb[8 * kt + (r >> 1)] = dmt_pack_bf16(d2[0], d2[1]);
          }
        // dQ^T[d, query] = K^T dS^T, one 32-column tile of the head at a time; rows leave as 16-byte pieces
        const unsigned r960 = rvalid ? ri.grow * 1920u + (unsigned)(160 * h + 16 * hi) : OOB;
        bfor<3>([&](auto tdc) {
          constexpr int td = decltype(tdc)::value;
          const int l16 = mb_here(lane) & 15, gq2 = (mb_here(lane) >> 4) & 1;
          int dcol = 32 * td + 16 * gq2 + 4 * (l16 & 3);
          dcol = dcol < MB_DH ? dcol : MB_DH - 4;
          const unsigned va = lds0 + MB_RA + (unsigned)(kwin + 4 * hi + (l16 >> 2)) * (unsigned)MB_RS + (unsigned)dcol * 2u;
          u32x2_t vlo[2 * NKT], vhi[2 * NKT];
          mb_read_tr<0>(vlo[0], va); mb_read_tr<8 * MB_RS>(vhi[0], va);
          mb_read_tr<16 * MB_RS>(vlo[1], va); mb_read_tr<24 * MB_RS>(vhi[1], va);
          if constexpr (NKT == 2) {
            mb_read_tr<32 * MB_RS>(vlo[2], va); mb_read_tr<40 * MB_RS>(vhi[2], va);
            mb_read_tr<48 * MB_RS>(vlo[3], va); mb_read_tr<56 * MB_RS>(vhi[3], va);
          }
          f32x16_t O;
#pragma unroll
          for (int r = 0; r < 16; ++r) O[r] = 0.f;
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]));
          if constexpr (NKT == 2) asm volatile("" : "+v"(vlo[2]), "+v"(vhi[2]), "+v"(vlo[3]), "+v"(vhi[3]));
#pragma unroll
          for (int ck = 0; ck < 2 * NKT; ++ck) {
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, u32x4_t{dsb[4 * ck], dsb[4 * ck + 1], dsb[4 * ck + 2], dsb[4 * ck + 3]});
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, u32x4_t{vlo[ck][0], vlo[ck][1], vhi[ck][0], vhi[ck][1]});
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, O, 0, 0, 0);
          }
          unsigned ho[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) ho[p] = dmt_pack_bf16(O[2 * p], O[2 * p + 1]);
          mb_swap(ho[0], ho[2]); mb_swap(ho[1], ho[3]);
          mb_swap(ho[4], ho[6]); mb_swap(ho[5], ho[7]);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ho[0], ho[1], ho[2], ho[3]}, rdq, r960 + 2u * (32 * td), 0, 0);
          if constexpr (32 * td + 16 < MB_DH) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ho[4], ho[5], ho[6], ho[7]}, rdq, r960 + 2u * (32 * td + 16), 0, 0);
        });
      };
      const bool drop = g.drop_thr != 0u;
      if (dbg & 1) { }
      else if (Tp == 64) { if (drop) phase_q(std::true_type{}, std::integral_constant<int, 64>{}); else phase_q(std::false_type{}, std::integral_constant<int, 64>{}); }
      else if (Tp == 32) { if (drop) phase_q(std::true_type{}, std::integral_constant<int, 32>{}); else phase_q(std::false_type{}, std::integral_constant<int, 32>{}); }
      else { if (drop) phase_q(std::true_type{}, std::integral_constant<int, 16>{}); else phase_q(std::false_type{}, std::integral_constant<int, 16>{}); }

      mb_lgkm0();
      __builtin_amdgcn_s_barrier();           // everybody has read K_h / V_h; the statistics of every query are in LDS
      // ================= Q_h, dO_h rows replace them =================
      bfor<5>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        mb_write128<MB_RA + 32 * c>(lds0 + r_loc_b, Qr[c]);
        mb_write128<MB_RB + 32 * c>(lds0 + r_loc_b, Dr[c]);
      });
      mb_lgkm0();
      __builtin_amdgcn_s_barrier();

      // ================= (ii) lane = key =================
      auto phase_k = [&](auto dropc, auto tpc) {
        constexpr bool DROP = decltype(dropc)::value;
        constexpr int TPK = decltype(tpc)::value;
        constexpr int NQT = TPK == 64 ? 2 : 1;
        const int hi = mb_here(hi_);
        const int qwin = (TPK == 64) ? 64 * (rb >> 1) : 32 * rb;
        const RowInfo ri = rowinfo();
        const bool rvalid = ri.rvalid;                    // this lane's key row exists
        const bool k_valid = rvalid && (ri.t_pos < len);  // ... and is not masked
        const int khalf = (mb_here(lane) & 31) >> 4;
        const unsigned r960 = rvalid ? ri.grow * 1920u + (unsigned)(160 * h + 16 * hi) : OOB;
        bf16x8_t Kf[5], Vf[5];                            // this lane's own K_h / V_h row pieces: B fragments (L2: requested a moment ago by the same lane)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          Kf[c] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rqkv, r960 + 640u + 32u * c, 0, 0));
          Vf[c] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rqkv, r960 + 1280u + 32u * c, 0, 0));
        }
        f32x16_t dK[3], dV[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) { dK[t][r] = 0.f; dV[t][r] = 0.f; }
        // dropout index of (query register, this key): ((ex H + h) T + qpos) T + kpos
        const unsigned kbase = (unsigned)(ri.ex * MB_H + h) * (unsigned)T;
        bfor<NQT>([&](auto qtc) {
          constexpr int qt = decltype(qtc)::value;
          const unsigned qa = lds0 + MB_RA + (unsigned)(qwin + 32 * qt + (mb_here(lane) & 31)) * (unsigned)MB_RS + 16u * (unsigned)hi;
          bf16x8_t qf[5], df[5];
          bfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mb_read128<c * 32>(qf[c], qa); mb_read128<MB_RB - MB_RA + c * 32>(df[c], qa); });
          // the queries' statistics: register 4 j + i of this lane half <-> query 32 qt + 8 j + 4 hi + i of the window
          f32x4_t sm[4], si[4], sd[4], sk[4];
          {
            const unsigned sa = lds0 + MB_ST + (unsigned)(qwin + 32 * qt + 4 * hi) * 4u;
            bfor<4>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              mb_read128f<32 * j>(sm[j], sa); mb_read128f<MB_ROWS * 4 + 32 * j>(si[j], sa);
              mb_read128f<2 * MB_ROWS * 4 + 32 * j>(sd[j], sa); mb_read128f<3 * MB_ROWS * 4 + 32 * j>(sk[j], sa);
            });
          }
          f32x16_t S, dP;
#pragma unroll
          for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]));
          asm volatile("" : "+v"(df[0]), "+v"(df[1]), "+v"(df[2]), "+v"(df[3]), "+v"(df[4]));
          asm volatile("" : "+v"(sm[0]), "+v"(sm[1]), "+v"(sm[2]), "+v"(sm[3]), "+v"(si[0]), "+v"(si[1]), "+v"(si[2]), "+v"(si[3]));
          asm volatile("" : "+v"(sd[0]), "+v"(sd[1]), "+v"(sd[2]), "+v"(sd[3]), "+v"(sk[0]), "+v"(sk[1]), "+v"(sk[2]), "+v"(sk[3]));
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[c], Kf[c], S, 0, 0, 0);       // S[query, key] = Q K^T
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[c], Vf[c], dP, 0, 0, 0);     // dP[query, key] = dO V^T
          }
          unsigned dsq[8], ptq[8];           // dS and P~ packed: B fragments over the QUERIES (chunk u: queries 16 u .. + 15 of the tile, accumulator order)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float d2[2], p2[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
              const int rr = r + z;
              const int ql = 32 * qt + (rr & 3) + 8 * (rr >> 2) + 4 * hi;           // query's row in the window
              const float qm = sm[rr >> 2][rr & 3], qi = si[rr >> 2][rr & 3], qd = sd[rr >> 2][rr & 3], qk = sk[rr >> 2][rr & 3];
              bool same = true;
              if constexpr (TPK == 16) same = ((ql >> 4) == khalf);
              const int qpos = (TPK == 16) ? (ql & 15) : ql;
              float pv = __builtin_amdgcn_exp2f(S[rr] * kscale - qm) * qi;          // (1 / sum is 0 for rows that are not live queries)
              pv = (k_valid && same) ? pv : 0.f;
              float gq = dP[rr];
              bool kept = true;
              if constexpr (DROP) kept = dmt_drop_keep(g.drop_seed, (kbase + (unsigned)qpos) * (unsigned)T + (unsigned)ri.t_pos, g.drop_thr);
              if constexpr (DROP) gq = kept ? gq * g.drop_inv_keep : 0.f;
              const float dsv = pv * (gq - qd) * inv_sc;
              // a padded query of an existing row holds -2^32 + 1 on every key that exists (TransformerModel_util.py:43-48): it reaches V only
              float pt = (qk == 2.f && rvalid && same) ? MB_PAD : pv;
              if constexpr (DROP) pt = kept ? pt * g.drop_inv_keep : 0.f;
              d2[z] = dsv;
              p2[z] = pt;
            }
            dsq[r >> 1] = dmt_pack_bf16(d2[0], d2[1]);
            ptq[r >> 1] = dmt_pack_bf16(p2[0], p2[1]);
          }
          // dK^T[d, key] += Q^T dS, dV^T[d, key] += dO^T P~: A = transposing reads of the Q / dO rows of this query tile (k = query, accumulator order)
          bfor<3>([&](auto tdc) {
            constexpr int td = decltype(tdc)::value;
            const int l16 = mb_here(lane) & 15, gq2 = (mb_here(lane) >> 4) & 1;
            int dcol = 32 * td + 16 * gq2 + 4 * (l16 & 3);
            dcol = dcol < MB_DH ? dcol : MB_DH - 4;
            const unsigned ta = lds0 + MB_RA + (unsigned)(qwin + 32 * qt + 4 * hi + (l16 >> 2)) * (unsigned)MB_RS + (unsigned)dcol * 2u;
            u32x2_t qlo[2], qhi[2], dlo[2], dhi[2];
            mb_read_tr<0>(qlo[0], ta); mb_read_tr<8 * MB_RS>(qhi[0], ta);
            mb_read_tr<16 * MB_RS>(qlo[1], ta); mb_read_tr<24 * MB_RS>(qhi[1], ta);
            mb_read_tr<MB_RB - MB_RA>(dlo[0], ta); mb_read_tr<MB_RB - MB_RA + 8 * MB_RS>(dhi[0], ta);
            mb_read_tr<MB_RB - MB_RA + 16 * MB_RS>(dlo[1], ta); mb_read_tr<MB_RB - MB_RA + 24 * MB_RS>(dhi[1], ta);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qlo[0]), "+v"(qhi[0]), "+v"(qlo[1]), "+v"(qhi[1]));
            asm volatile("" : "+v"(dlo[0]), "+v"(dhi[0]), "+v"(dlo[1]), "+v"(dhi[1]));
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
              const bf16x8_t bs = __builtin_bit_cast(bf16x8_t, u32x4_t{dsq[4 * ck], dsq[4 * ck + 1], dsq[4 * ck + 2], dsq[4 * ck + 3]});
              const bf16x8_t bp = __builtin_bit_cast(bf16x8_t, u32x4_t{ptq[4 * ck], ptq[4 * ck + 1], ptq[4 * ck + 2], ptq[4 * ck + 3]});
              const bf16x8_t aq = __builtin_bit_cast(bf16x8_t, u32x4_t{qlo[ck][0], qlo[ck][1], qhi[ck][0], qhi[ck][1]});
              const bf16x8_t ad = __builtin_bit_cast(bf16x8_t, u32x4_t{dlo[ck][0], dlo[ck][1], dhi[ck][0], dhi[ck][1]});
              dK[td] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bs, dK[td], 0, 0, 0);
              dV[td] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bp, dV[td], 0, 0, 0);
            }
          });
        });
        // this lane's key row: dK_h, dV_h pieces
        bfor<3>([&](auto tdc) {
          constexpr int td = decltype(tdc)::value;
          unsigned hk[8], hv[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) { hk[p] = dmt_pack_bf16(dK[td][2 * p], dK[td][2 * p + 1]); hv[p] = dmt_pack_bf16(dV[td][2 * p], dV[td][2 * p + 1]); }
          mb_swap(hk[0], hk[2]); mb_swap(hk[1], hk[3]); mb_swap(hk[4], hk[6]); mb_swap(hk[5], hk[7]);
          mb_swap(hv[0], hv[2]); mb_swap(hv[1], hv[3]); mb_swap(hv[4], hv[6]); mb_swap(hv[5], hv[7]);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hk[0], hk[1], hk[2], hk[3]}, rdq, r960 + 640u + 2u * (32 * td), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hv[0], hv[1], hv[2], hv[3]}, rdq, r960 + 1280u + 2u * (32 * td), 0, 0);
          if constexpr (32 * td + 16 < MB_DH) {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hk[4], hk[5], hk[6], hk[7]}, rdq, r960 + 640u + 2u * (32 * td + 16), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hv[4], hv[5], hv[6], hv[7]}, rdq, r960 + 1280u + 2u * (32 * td + 16), 0, 0);
          }
        });
      };
      if (dbg & 2) { }
      else if (Tp == 64) { if (drop) phase_k(std::true_type{}, std::integral_constant<int, 64>{}); else phase_k(std::false_type{}, std::integral_constant<int, 64>{}); }
      else if (Tp == 32) { if (drop) phase_k(std::true_type{}, std::integral_constant<int, 32>{}); else phase_k(std::false_type{}, std::integral_constant<int, 32>{}); }
      else { if (drop) phase_k(std::true_type{}, std::integral_constant<int, 16>{}); else phase_k(std::false_type{}, std::integral_constant<int, 16>{}); }
      mb_lgkm0();
      __builtin_amdgcn_s_barrier();           // everybody has read Q_h / dO_h and the statistics: the next head (or the ring) may overwrite them
    }

    // ================= dx = dqkv Wqkv^T + ds for this wavefront's 32 rows =================
    if (!(dbg & 4)) {
      const RowInfo ri = rowinfo();
      const int hi = mb_here(hi_);
      const unsigned r960 = ri.rvalid ? ri.grow * 1920u + (unsigned)(16 * hi) : OOB;
      const unsigned r320 = ri.rvalid ? ri.grow * 640u + (unsigned)(16 * hi) : OOB;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this lane's dqkv stores are in the L2
      bf16x8_t Bf[2][10];                                    // dqkv pieces of k group (= k stage) s: columns 160 s + 16 c + 8 hi .. + 7, double-buffered
      auto load_b = [&](auto bufc, int s) {
        constexpr int bi = decltype(bufc)::value;
#pragma unroll
        for (int c = 0; c < 10; ++c) Bf[bi][c] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rdq, r960 + (unsigned)(320 * s) + 32u * c, 0, 1));   // (sc0: past the L1)
      };
      load_b(std::integral_constant<int, 0>{}, 0);
      load_b(std::integral_constant<int, 1>{}, 1);
      asm volatile("" ::: "memory");
      for (int s0 = 0; s0 < MB_AHEAD; ++s0) issue(s0);
      asm volatile("" ::: "memory");
      const unsigned a_lane = lds0 + (unsigned)ml * (unsigned)MB_WSTRIDE + 16u * (unsigned)hi;
      bfor<2>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        f32x16_t acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        bfor<6>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          constexpr int grp = 6 * p + s;                     // k group in issue order (group grp uses buffer grp & 1, k stage grp % 6)
          bfor<5>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            constexpr int u = 5 * grp + jj;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(mb_younger(u) < 63 ? mb_younger(u) : 63) : "memory");    // my pieces of stage u have landed
            __builtin_amdgcn_s_barrier();                    // ... everybody's have; everybody has left stage u - 1 = the slot of stage u + AHEAD
            const unsigned wa = a_lane + (unsigned)((u % MB_NS) * MB_STAGE);
            bf16x8_t R0[5], R1[5];
            mb_read128<0 * 32>(R0[0], wa); mb_read128<1 * 32>(R0[1], wa); mb_read128<2 * 32>(R0[2], wa); mb_read128<3 * 32>(R0[3], wa); mb_read128<4 * 32>(R0[4], wa);
            mb_read128<5 * 32>(R1[0], wa); mb_read128<6 * 32>(R1[1], wa); mb_read128<7 * 32>(R1[2], wa); mb_read128<8 * 32>(R1[3], wa); mb_read128<9 * 32>(R1[4], wa);
            if constexpr (u + MB_AHEAD < MB_NSTAGE) issue(u + MB_AHEAD);
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(R0[0]), "+v"(R0[1]), "+v"(R0[2]), "+v"(R0[3]), "+v"(R0[4]));
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R0[c], Bf[grp & 1][c], acc[jj], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R1[0]), "+v"(R1[1]), "+v"(R1[2]), "+v"(R1[3]), "+v"(R1[4]));
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R1[c], Bf[grp & 1][5 + c], acc[jj], 0, 0, 0);
          });
          // the buffer this group used takes the pieces of the group after the next one (k stage (grp + 2) % 6).  (Fenced: mb_younger() COUNTS
          // these loads, so they must stay where they are written -- behind this stage's DMA issue, in front of the next stage's wait.)
          asm volatile("" ::: "memory");
          if constexpr (grp + 2 < 12) load_b(std::integral_constant<int, grp & 1>{}, (grp + 2) % 6);
          asm volatile("" ::: "memory");
        });
        // pass p: output columns 160 p .. + 159: + ds, ONE rounding, rows leave as 16-byte pieces
        u32x4_t dsv[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) dsv[q] = __builtin_amdgcn_raw_buffer_load_b128(rds, r320 + (unsigned)(320 * p) + 32u * q, 0, 0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            // register groups q = 2 pp (columns 16 pp + 4 hi + 0..3 of the tile) and 2 pp + 1 (+ 8): after the exchange a lane holds columns 16 pp + 8 hi + 0..7
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = acc[j][8 * pp + i]; b[i] = acc[j][8 * pp + 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) mb_swapf(a[i], b[i]);
            const u32x4_t rv = dsv[2 * j + pp];
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = __uint_as_float(rv[e] << 16), hi2 = __uint_as_float(rv[e] & 0xFFFF0000u);
              const float v0 = (2 * e < 4) ? a[2 * e] : b[2 * e - 4], v1 = (2 * e + 1 < 4) ? a[2 * e + 1] : b[2 * e + 1 - 4];
              o[2 * e] = v0 + lo;
              o[2 * e + 1] = v1 + hi2;
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{dmt_pack_bf16(o[0], o[1]), dmt_pack_bf16(o[2], o[3]), dmt_pack_bf16(o[4], o[5]), dmt_pack_bf16(o[6], o[7])},
                                                  rdx, r320 + (unsigned)(320 * p) + 64u * j + 32u * pp, 0, 0);
          }
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();           // the ring is quiet: the next tile's K / V rows may overwrite it
    }
  }
}

}  // namespace

extern "C" int dmt_mhsa_bwd_image_bytes(int64_t* bytes) {
  DMT_CHECK_ARG(bytes != nullptr, "dmt_mhsa_bwd_image_bytes: null output");
  *bytes = MB_IMAGE_BYTES;
  return DMT_OK;
}

extern "C" int dmt_mhsa_bwd_image_build(const float* wqkv, int64_t ldw, void* image, void* stream) {
  DMT_CHECK_ARG(wqkv && image && ldw >= 960, "dmt_mhsa_bwd_image_build: bad argument");
  hipLaunchKernelGGL(mhsa_bwd_image_kernel, dim3(158), dim3(256), 0, (hipStream_t)stream, wqkv, (long long)ldw, (unsigned char*)image);
  DMT_CHECK_LAUNCH("dmt_mhsa_bwd_image_build");
  return DMT_OK;
}

extern "C" int dmt_mhsa_block_bwd(const dmt_mhsa_bwd_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_mhsa_block_bwd: null descriptor");
  DMT_CHECK_ARG(d->d_model == MB_D && d->num_heads == MB_H, "dmt_mhsa_block_bwd: built for d_model %d, %d heads (got %d, %d)", MB_D, MB_H, d->d_model, d->num_heads);
  DMT_CHECK_ARG(d->B > 0 && d->T > 0 && d->T <= 64, "dmt_mhsa_block_bwd: 1 <= T <= 64 (got %d)", d->T);
  DMT_CHECK_ARG(d->ds && d->qkv && d->lens && d->image && d->dqkv && d->dx, "dmt_mhsa_block_bwd: null pointer");
  DMT_CHECK_ARG((((uintptr_t)d->ds | (uintptr_t)d->qkv | (uintptr_t)d->dqkv | (uintptr_t)d->dx) & 15) == 0, "dmt_mhsa_block_bwd: tensors must be 16-byte aligned");
  DMT_CHECK_ARG((long long)d->B * d->T * d->num_heads * d->T < (1ll << 32), "dmt_mhsa_block_bwd: dropout counter range");
  const bool packed = d->blocks != nullptr;
  DMT_CHECK_ARG(!packed || (d->n_tiles > 0 && d->n_rows > 0 && d->n_rows <= (long long)d->B * d->T && (((uintptr_t)d->blocks) & 15) == 0),
                "dmt_mhsa_block_bwd: packed rows need n_tiles > 0, 0 < n_rows <= B * T and a 16-byte aligned block table");
  DMT_CHECK_ARG((packed ? (long long)d->n_rows : (long long)d->B * d->T) * 1920 < 0x7FFF0000ll, "dmt_mhsa_block_bwd: 32-bit byte offsets (rows * 1920 < 2^31)");
  MhsaBwdArgs a;
  a.ds = (const bf16_t*)d->ds; a.qkv = (const bf16_t*)d->qkv; a.lens = d->lens; a.image = (const unsigned char*)d->image;
  a.dqkv = (bf16_t*)d->dqkv; a.dx = (bf16_t*)d->dx;
  a.B = d->B; a.T = d->T;
  a.Tp = d->T > 32 ? 64 : (d->T > 16 ? 32 : 16);
  a.lgTp = a.Tp == 64 ? 6 : (a.Tp == 32 ? 5 : 4);
  a.drop_seed = d->drop_seed;
  const bool drop = d->drop_keep > 0.f && d->drop_keep < 1.f;
  a.drop_thr = drop ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv_keep = drop ? 1.0f / d->drop_keep : 1.0f;
  a.dbg = 0;
#ifdef DMT_TIMING_EXPERIMENTS
  { const char* e = getenv("DMT_MHSA_BWD_DEBUG"); a.dbg = e ? atoi(e) : 0; }
#endif
  const int epw = MB_ROWS / a.Tp;
  a.tiles = packed ? d->n_tiles : (d->B + epw - 1) / epw;
  a.blocks = d->blocks;
  a.n_rows = packed ? d->n_rows : (long long)d->B * d->T;
  const int grid = a.tiles < 256 ? a.tiles : 256;
  if (packed) {
    hipLaunchKernelGGL(mhsa_bwd_kernel<true>, dim3(grid), dim3(MB_NT), 0, (hipStream_t)stream, a);
    DMT_CHECK_LAUNCH("dmt_mhsa_block_bwd(packed)");
    return DMT_OK;
  }
  hipLaunchKernelGGL(mhsa_bwd_kernel<false>, dim3(grid), dim3(MB_NT), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_mhsa_block_bwd");
  return DMT_OK;
}
