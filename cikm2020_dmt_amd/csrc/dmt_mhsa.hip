// Self-attention block of the sequence encoder in ONE launch (bf16, E64 geometry: d_model 320 = 4 heads x 80):
//
//     y = LN(x + concat_h softmax(mask(Q_h K_h^T / sqrt(80))) V_h),   (Q | K | V) = x Wqkv + b
//
// = multihead_attention(x, x, x, lens, lens) of /root/reference/DMT_code/model/net/TransformerModel_util.py:160-209 with
// scaled_dot_product_attention :11-56, mask :80-108 (key mask before the softmax, query mask after it, both -2^32 + 1), dropout on
// the attention weights :51 and ln :58-78.  There is no output projection in the reference.
//
// A workgroup (8 wavefronts, two per SIMD) owns 128 rows = 128 / Tp examples, every example padded to Tp in {16, 32, 64} rows
// (the padding rows are all-zero inputs and never stored).  Per head h:
//   phase A (all 8 wavefronts): the 240 columns (Q_h | K_h | V_h) of the packed projection, as eight 32-column tiles of a prebuilt
//     weight image streamed through a two-stage LDS ring by DMA; the wavefront's 32 input rows stay in registers as MFMA fragments
//     (the chain-kernel scheme, dmt_chain.hip); results go to LDS -- Q_h, K_h row-major, V_h transposed with its keys in the k order
//     the softmax registers will have -- and, for the backward pass, to the packed qkv tensor in memory;
//   phase B (one wavefront per 32-row block): S^T = K Q^T on MFMA (a lane owns one query: the softmax is an in-lane reduction plus
//     one exchange with lane + 32), masks, softmax, counter dropout, P as MFMA operand straight from the accumulators, O^T = V^T P^T;
//     the head's slice of s = O + x is stored and its row sums kept.
// After the four heads the rows' mean / variance are known; the wavefront re-reads its own s pieces and writes y = LN(s).
// qkv never travels back from memory, the scores never leave registers, LayerNorm is not a separate pass.
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

constexpr int MH_NT = 512;
constexpr int MH_D = 320, MH_H = 4, MH_DH = 80, MH_KC = MH_D / 16;   // 20 k chunks
constexpr int MH_TILE = 32 * MH_D * 2;          // one 32-column weight tile, XOR-swizzled 16-byte slots: 20480 B
constexpr int MH_STAGE = 2 * MH_TILE;           // a stage = two tiles (one per wavefront half): 40960 B = 5 pieces per wavefront
constexpr int MH_PPW = MH_STAGE / (8 * 1024);
constexpr int MH_TILES_PER_HEAD = 8;            // 240 columns + 16 zero columns
constexpr int MH_NSTAGE = MH_H * MH_TILES_PER_HEAD / 2;   // 16 stages per row tile
constexpr int MH_QK_STRIDE = MH_DH * 2 + 16;    // 176 B: 11 slots
constexpr int MH_VT_STRIDE = 128 * 2 + 16;      // 272 B: 17 slots
constexpr int MH_Q_OFF = 2 * MH_STAGE;
constexpr int MH_K_OFF = MH_Q_OFF + 128 * MH_QK_STRIDE;
constexpr int MH_VT_OFF = MH_K_OFF + 128 * MH_QK_STRIDE;
constexpr int MH_BIAS_OFF = MH_VT_OFF + MH_DH * MH_VT_STRIDE;
constexpr int MH_LDS = MH_BIAS_OFF + 4 * 256 * 4;   // bias in head-major order, 256 per head
static_assert(MH_LDS <= 160 * 1024, "LDS");
constexpr long long MH_IMAGE_BYTES = (long long)MH_NSTAGE * MH_STAGE;
constexpr float MH_PAD = -4294967295.0f;        // -2^32 + 1 (TransformerModel_util.py:86)

template <int... I, typename F>
__device__ __forceinline__ void mfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void mfor(F&& f) { mfor_impl(std::make_integer_sequence<int, N>{}, f); }

__host__ __device__ constexpr int mh_perm16(int p) { return 4 * (p >> 3) + (p & 3) + 8 * ((p >> 2) & 1); }
// column c' (0..255) of head h in head-major order -> column of the packed [Q | K | V] projection, or -1 (padding)
__host__ __device__ constexpr int mh_std_col(int h, int cp) {
  return cp < 80 ? 80 * h + cp : (cp < 160 ? 320 + 80 * h + (cp - 80) : (cp < 240 ? 640 + 80 * h + (cp - 160) : -1));
}

// ------------------------------------------------------------------------------------------------------------ weight image
// tile g = 8 h + j, row i = head-major column 32 j + i; a row is 40 16-byte slots (k chunk c, half) of 8 bf16 with k permuted inside
// every 16-chunk (dmt_chain.hip); physical slot = logical slot ^ ((row >> 1) & 7) (conflict-free ds_read_b128 without padding).
__global__ __launch_bounds__(256) void mhsa_image_kernel(const float* __restrict__ w, long long ldw, unsigned char* __restrict__ img) {
  const long long slots = MH_IMAGE_BYTES / 16;
  for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < slots; s += (long long)gridDim.x * 256) {
    const int g = (int)(s / (MH_TILE / 16)), within = (int)(s % (MH_TILE / 16));
    const int row = within / 40, phys = within % 40;
    const int logical = phys ^ ((row >> 1) & 7);
    const int c = logical >> 1, half = logical & 1;
    const int col = mh_std_col(g / 8, 32 * (g % 8) + row);
    unsigned short hh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * c + mh_perm16(8 * half + e);
      hh[e] = col >= 0 ? f2bf(w[(long long)k * ldw + col]) : (unsigned short)0;
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
  bf16_t* s_out;            // [B*T, 320]
  bf16_t* y_out;            // [B*T, 320]
  float* stats;             // [B*T, 2] or null
  int B, T, Tp, lgTp, tiles;
  int dbg;   // timing experiments only (DMT_MHSA_DEBUG): 1 no phase-A multiply, 2 no phase B, 4 no qkv copy, 8 no LayerNorm pass, 16 no DMA
  unsigned drop_seed, drop_thr;   // thr = keep * 2^24, 0: dropout off
  float drop_inv_keep;
};

template <int OFF> __device__ __forceinline__ void mh_read128(bf16x8_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void mh_write64(unsigned addr, u32x2_t v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void mh_wait5(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d, bf16x8_t& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "i"(N));
}
__device__ __forceinline__ void mh_swap(unsigned& a, unsigned& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}

__global__ __launch_bounds__(MH_NT, 2) void mhsa_fwd_kernel(const MhsaArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[MH_LDS];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave & 3, half = wave >> 2;      // row block (32 rows), which tile of a stage
  const int ml = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((lds_vp)smem);
  float* bias_lds = reinterpret_cast<float*>(smem + MH_BIAS_OFF);

  // bias in head-major order (zero for the padding columns)
  for (int i = tid; i < 4 * 256; i += MH_NT) {
    const int col = mh_std_col(i >> 8, i & 255);
    bias_lds[i] = col >= 0 ? g.bias[col] : 0.f;
  }
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.image), 0, (int)MH_IMAGE_BYTES, 0x00020000);
  // A stage is 40 pieces of 1 KB.  Normally every wavefront issues five.  The stage that OPENS a head is issued (at the last
  // iteration of the previous head) by wavefronts 4-7 alone, ten each: wavefronts 0-3 store their s pieces at the very end of phase B,
  // and a wait for "my DMA has landed" (vmcnt counts stores too) would make them sit out the latency of those stores.
  auto issue = [&](int buf, int st, bool head_opener) {
    if (g.dbg & 16) return;
    if (!head_opener) {
      unsigned char* sb = smem + buf * MH_STAGE + wave * 1024;
#pragma unroll
      for (int p = 0; p < MH_PPW; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + p * 8192), 16, lane * 16, st * MH_STAGE + wave * 1024 + p * 8192, 0, 0);
    } else if (half == 1) {
      unsigned char* sb = smem + buf * MH_STAGE + (wave - 4) * 1024;
#pragma unroll
      for (int p = 0; p < 2 * MH_PPW; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rimg, (lds_vp)(sb + p * 4096), 16, lane * 16, st * MH_STAGE + (wave - 4) * 1024 + p * 4096, 0, 0);
    }
  };
  // weight-fragment addresses: row ml of the wavefront's tile, logical slot 2 c + hi -> physical (.. ^ sw); c = 4 a + k
  const int sw = (ml >> 1) & 7;
  unsigned a_off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a_off[k] = lds0 + half * MH_TILE + ml * 640 + (((2 * k + hi) ^ sw) * 16);

  const int Tp = g.Tp, T = g.T, lg = g.lgTp;
  const int epw = 128 >> lg;                     // examples per workgroup
  const float scale = 0.11180339887498948f;      // 1 / sqrt(80)
  int gs = 0;                                    // stages consumed so far by this workgroup
  issue(0, 0, true);

  for (int tile = (int)blockIdx.x; tile < g.tiles; tile += (int)gridDim.x) {
    // ---- this lane's row: local row r -> (example, position)
    const int r_loc = 32 * rb + ml;
    const int e_loc = r_loc >> lg, t_pos = r_loc & (Tp - 1);
    const int ex = tile * epw + e_loc;
    const bool rvalid = (ex < g.B) && (t_pos < T);
    const long long grow = (long long)ex * T + t_pos;          // row in the [B*T, *] tensors
    const long long growc = rvalid ? grow : 0;
    const int len = (ex < g.B) ? g.lens[ex] : 0;
    // input rows as B fragments (k permuted: lower lane 0-3 | 8-11, upper 4-7 | 12-15 of every 16-chunk)
    bf16x8_t X[MH_KC];
    {
      const bf16_t* xr = g.x + growc * MH_D + 8 * hi;
#pragma unroll
      for (int c = 0; c < MH_KC; ++c) {
        const u32x4_t vl = *reinterpret_cast<const u32x4_t*>(xr + 16 * c);
        uint4 v = make_uint4(vl[0], vl[1], vl[2], vl[3]);
        if (!rvalid) v = make_uint4(0u, 0u, 0u, 0u);
        mh_swap(v.x, v.z);
        mh_swap(v.y, v.w);
        X[c] = __builtin_bit_cast(bf16x8_t, v);
      }
    }
    float rsum = 0.f, rsq = 0.f;   // row statistics of s (attention wavefronts)

#pragma unroll 1
    for (int h = 0; h < MH_H; ++h) {
      // ================= phase A: (Q_h | K_h | V_h) = x W_h + b, tiles j = 2 it + half =================
#pragma unroll 1
      for (int it = 0; it < MH_TILES_PER_HEAD / 2; ++it, ++gs) {
        const int buf = gs & 1;
        // stage gs has landed (vmcnt also counts this wavefront's stores, all at least an iteration old here; wavefronts 0-3 issued
        // nothing of a head-opening stage and must not wait for their fresh s stores)
        if (it != 0 || half == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // ... for everybody; everybody has left the other buffer
        const int j = 2 * it + half;                         // tile of this wavefront
        const bool vtile = j >= 5;   // wave-uniform: V tiles are computed un-transposed (rows = x rows), Q / K tiles transposed
        // bias = initial accumulator (read BEFORE the DMA below is issued: hipcc drains every DMA in flight ahead of a visible LDS read):
        // transposed form -> per register row c' = 8 q + 4 hi + i; V form -> per lane column c' = ml
        f32x16_t acc;
        {
          const float* bl = bias_lds + h * 256 + 32 * j;
          if (vtile) {
            const float bv = bl[ml];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bv;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bl + 8 * q + 4 * hi);
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[4 * q + i] = b4[i];
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(buf ^ 1, (gs + 1) % MH_NSTAGE, it == MH_TILES_PER_HEAD / 2 - 1);   // (past the last tile: a harmless re-fetch)
        const unsigned so = (unsigned)buf * MH_STAGE;
        auto do_tile = [&](auto vc) {
          constexpr bool VT = decltype(vc)::value;
          bf16x8_t R0[5], R1[5];
          auto rd = [&](bf16x8_t (&R)[5], auto bic) {
            constexpr int b = decltype(bic)::value;
            mfor<5>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              constexpr int c = b * 5 + i;
              mh_read128<(c >> 2) * 128>(R[i], a_off[c & 3] + so);
            });
          };
          rd(R0, std::integral_constant<int, 0>{});
          mfor<4>([&](auto bic) {
            constexpr int b = decltype(bic)::value;
            if constexpr (b + 1 < 4) {
              if constexpr ((b & 1) == 0) rd(R1, std::integral_constant<int, b + 1>{});
              else rd(R0, std::integral_constant<int, b + 1>{});
            }
            auto mm = [&](bf16x8_t (&R)[5]) {
#pragma unroll
              for (int i = 0; i < 5; ++i) {
                if constexpr (VT) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[b * 5 + i], R[i], acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R[i], X[b * 5 + i], acc, 0, 0, 0);
              }
            };
            if constexpr (b + 1 < 4) {
              if constexpr ((b & 1) == 0) { mh_wait5<5>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0); }
              else { mh_wait5<5>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1); }
            } else {
              if constexpr ((b & 1) == 0) { mh_wait5<0>(R0[0], R0[1], R0[2], R0[3], R0[4]); mm(R0); }
              else { mh_wait5<0>(R1[0], R1[1], R1[2], R1[3], R1[4]); mm(R1); }
            }
          });
          unsigned pk[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) pk[p] = dmt_pack_bf16(acc[2 * p], acc[2 * p + 1]);
          if constexpr (!VT) {
            // lane (row m = 32 rb + ml, hi): register group q = head-major columns 32 j + 8 q + 4 hi .. +3
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cp = 32 * j + 8 * q + 4 * hi;          // < 160 here
              const unsigned base = cp < 80 ? (lds0 + MH_Q_OFF + cp * 2) : (lds0 + MH_K_OFF + (cp - 80) * 2);
              mh_write64(base + (32 * rb + ml) * MH_QK_STRIDE, u32x2_t{pk[2 * q], pk[2 * q + 1]});
            }
          } else {
            // lane (column d = 32 (j - 5) + ml of V_h, hi): register group q = rows 32 rb + 8 q + 4 hi .. +3 (keys).
            // V^T[d][key] with the keys of every 16-chunk in MFMA k order: the 4-group g = 2 (q & 1) + hi goes to position (g & 1) * 2 + (g >> 1)
            const int d = 32 * (j - 5) + ml;
            if (d < MH_DH) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int grp = 2 * (q & 1) + hi;
                const int key = 32 * rb + 16 * (q >> 1) + 4 * (((grp & 1) << 1) | (grp >> 1));
                mh_write64(lds0 + MH_VT_OFF + d * MH_VT_STRIDE + key * 2, u32x2_t{pk[2 * q], pk[2 * q + 1]});
              }
            }
          }
        };
        if (g.dbg & 1) { }
        else if (vtile) do_tile(std::true_type{});
        else do_tile(std::false_type{});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // Q_h, K_h, V_h^T complete in LDS

      // ================= phase B: attention of head h for the 32 queries of this row block (wavefronts 0..3) =================
      if (half == 0 && !(g.dbg & 2)) {
        auto phase_b = [&](auto dropc, auto tpc) {
          constexpr bool DROP = decltype(dropc)::value;
          constexpr int TPK = decltype(tpc)::value;          // 64, 32 or 16
          constexpr int NKT = TPK == 64 ? 2 : 1;             // key tiles of 32 rows
          const int kwin = (TPK == 64) ? 64 * (rb >> 1) : 32 * rb;   // first local row of the key window
          // S^T[key, query] = K Q^T: A = K rows, B = Q rows (both row-major, 16-byte chunks, natural k order)
          f32x16_t S[NKT];
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
          {
            const unsigned qa = lds0 + MH_Q_OFF + (32 * rb + ml) * MH_QK_STRIDE + 16 * hi;
            const unsigned ka = lds0 + MH_K_OFF + (kwin + ml) * MH_QK_STRIDE + 16 * hi;
            bf16x8_t qf[5], k0[5], k1[5];
            mfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mh_read128<c * 32>(qf[c], qa); });
            mfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mh_read128<c * 32>(k0[c], ka); });
            if constexpr (NKT == 2) mfor<5>([&](auto cc) { constexpr int c = decltype(cc)::value; mh_read128<c * 32 + 32 * MH_QK_STRIDE>(k1[c], ka); });
            mh_wait5<(NKT == 2 ? 10 : 5)>(qf[0], qf[1], qf[2], qf[3], qf[4]);
            mh_wait5<(NKT == 2 ? 5 : 0)>(k0[0], k0[1], k0[2], k0[3], k0[4]);
            if constexpr (NKT == 2) mh_wait5<0>(k1[0], k1[1], k1[2], k1[3], k1[4]);
#pragma unroll
            for (int c = 0; c < 5; ++c) {
              S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0[c], qf[c], S[0], 0, 0, 0);
              if constexpr (NKT == 2) S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1[c], qf[c], S[1], 0, 0, 0);
            }
          }
          // lane = query (position t_pos of its example, length len); register (kt, r) = key 32 kt + (r & 3) + 8 (r >> 2) + 4 hi of the
          // window.  For Tp >= 32 the window IS the query's example, so "the key exists" is key < T and "is valid" is key < len: both
          // compare a compile-time constant with T - 4 hi / len - 4 hi.  For Tp = 16 a window holds two examples.
          const bool q_live = rvalid && (t_pos < len);
          int Tm = T - 4 * hi, Lm = len - 4 * hi;            // key < T  <=>  const(r) < Tm
          const int ehalf = ml >> 4;                         // (Tp = 16) which example of the 32-row window this query belongs to
          auto kexists = [&](int cr, int tm) -> bool {
            if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr & 15) + 4 * hi - 16 * (((cr & 15) + 4 * hi) >> 4)) < T);
            else return cr < tm;
          };
          auto kvalid_f = [&](int cr, int lm) -> bool {
            if constexpr (TPK == 16) return (((cr + 4 * hi) >> 4) == ehalf) && (((cr + 4 * hi) & 15) < len);
            else return cr < lm;
          };
          float sv[16 * NKT];
          float mx = -3.0e38f;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cr = 32 * kt + (r & 3) + 8 * (r >> 2);
              float v = S[kt][r] * scale;
              v = kvalid_f(cr, Lm) ? v : MH_PAD;
              v = kexists(cr, Tm) ? v : -3.0e38f;
              sv[16 * kt + r] = v;
              mx = fmaxf(mx, v);
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
          const unsigned qbase = ((unsigned)(ex * MH_H + h) * (unsigned)T + (unsigned)t_pos) * (unsigned)T + (unsigned)(4 * hi);
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
          // O^T[d, query] = V^T P^T: A = V^T rows (d), keys in MFMA k order; B = P fragments from the registers above
          f32x16_t O[3];
#pragma unroll
          for (int td = 0; td < 3; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[td][r] = 0.f;
          {
            bf16x8_t vf[3][4];
#pragma unroll
            for (int td = 0; td < 3; ++td) {
              const int drow = (32 * td + ml) < MH_DH ? (32 * td + ml) : (MH_DH - 1);
              const unsigned va = lds0 + MH_VT_OFF + drow * MH_VT_STRIDE + kwin * 2 + 16 * hi;
              mh_read128<0>(vf[td][0], va);
              mh_read128<32>(vf[td][1], va);
              if constexpr (NKT == 2) { mh_read128<64>(vf[td][2], va); mh_read128<96>(vf[td][3], va); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]));
            if constexpr (NKT == 2) asm volatile("" : "+v"(vf[0][2]), "+v"(vf[0][3]), "+v"(vf[1][2]), "+v"(vf[1][3]), "+v"(vf[2][2]), "+v"(vf[2][3]));
#pragma unroll
            for (int ck = 0; ck < 2 * NKT; ++ck) {
              const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, make_uint4(pf[4 * ck], pf[4 * ck + 1], pf[4 * ck + 2], pf[4 * ck + 3]));
#pragma unroll
              for (int td = 0; td < 3; ++td) O[td] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[td][ck], pb, O[td], 0, 0, 0);
            }
          }
          // s = O + x on this head's 80 columns: register (td, 4 q + i) of lane (query, hi) = column 80 h + 32 td + 8 q + 4 hi + i
          {
            const bf16_t* xr = g.x + growc * MH_D + 80 * h;
            bf16_t* sr = g.s_out + growc * MH_D + 80 * h;
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) {           // pairs of register groups (q even, q odd): d = 16 pr .. 16 pr + 15
              const int td = pr >> 1, q0 = 2 * (pr & 1);
              unsigned a[2], b[2];
              float va[4], vb[4];
              // residual: 16 bytes per lane, undo the pairing to get this lane's two 4-groups
              u32x4_t xv = {0u, 0u, 0u, 0u};
              if (rvalid) xv = *reinterpret_cast<const u32x4_t*>(xr + 16 * pr + 8 * hi);
              unsigned x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
              mh_swap(x0, x2); mh_swap(x1, x3);
              va[0] = O[td][4 * q0 + 0] + __uint_as_float(x0 << 16); va[1] = O[td][4 * q0 + 1] + __uint_as_float(x0 & 0xFFFF0000u);
              va[2] = O[td][4 * q0 + 2] + __uint_as_float(x1 << 16); va[3] = O[td][4 * q0 + 3] + __uint_as_float(x1 & 0xFFFF0000u);
              vb[0] = O[td][4 * q0 + 4] + __uint_as_float(x2 << 16); vb[1] = O[td][4 * q0 + 5] + __uint_as_float(x2 & 0xFFFF0000u);
              vb[2] = O[td][4 * q0 + 6] + __uint_as_float(x3 << 16); vb[3] = O[td][4 * q0 + 7] + __uint_as_float(x3 & 0xFFFF0000u);
              a[0] = dmt_pack_bf16(va[0], va[1]); a[1] = dmt_pack_bf16(va[2], va[3]);
              b[0] = dmt_pack_bf16(vb[0], vb[1]); b[1] = dmt_pack_bf16(vb[2], vb[3]);
              // statistics of the ROUNDED values (what the LayerNorm gradient will read back)
#pragma unroll
              for (int z = 0; z < 2; ++z) {
                const float f0 = __uint_as_float(a[z] << 16), f1 = __uint_as_float(a[z] & 0xFFFF0000u);
                const float f2 = __uint_as_float(b[z] << 16), f3 = __uint_as_float(b[z] & 0xFFFF0000u);
                rsum += (f0 + f1) + (f2 + f3);
                rsq += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
              }
              mh_swap(a[0], b[0]); mh_swap(a[1], b[1]);
              if (rvalid) *reinterpret_cast<u32x4_t*>(sr + 16 * pr + 8 * hi) = u32x4_t{a[0], a[1], b[0], b[1]};
            }
          }
        };
        const bool drop = g.drop_thr != 0u;
        if (Tp == 64) { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 64>{}); else phase_b(std::false_type{}, std::integral_constant<int, 64>{}); }
        else if (Tp == 32) { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 32>{}); else phase_b(std::false_type{}, std::integral_constant<int, 32>{}); }
        else { if (drop) phase_b(std::true_type{}, std::integral_constant<int, 16>{}); else phase_b(std::false_type{}, std::integral_constant<int, 16>{}); }
      }
      else if (half != 0 && g.qkv != nullptr && !(g.dbg & 4)) {
        // wavefronts 4-7 copy (Q_h | K_h | V_h) from LDS to the packed qkv tensor for the backward pass while 0-3 do the attention:
        // 16-byte pieces, ten consecutive lanes per 160-byte row segment; all reads of a batch first, then the stores
        const int t4 = tid - 256;
        const unsigned char* sm = smem;
#pragma unroll 1
        for (int ch = t4; ch < 128 * 30; ch += 256) {
          const int m = (ch * 2185) >> 16, cc = ch - m * 30;   // row (ch / 30), 8-column piece: 0-9 Q, 10-19 K, 20-29 V
          const int el = m >> lg, tl = m & (Tp - 1);
          const int exr = tile * epw + el;
          if (exr < g.B && tl < T) {
            u32x4_t v;
            if (cc < 20) {
              const int off = (cc < 10 ? MH_Q_OFF + cc * 16 : MH_K_OFF + (cc - 10) * 16) + m * MH_QK_STRIDE;
              v = *reinterpret_cast<const u32x4_t*>(sm + off);
            } else {
              // V^T[d][key position]: key m sits at 16 (m / 16) + 4 * swap(g) + (m & 3), g = (m % 16) / 4
              const int gk = (m >> 2) & 3;
              const int pos = (m & ~15) + 4 * (((gk & 1) << 1) | (gk >> 1)) + (m & 3);
              const unsigned short* vt = reinterpret_cast<const unsigned short*>(sm + MH_VT_OFF) + pos + ((cc - 20) * 8) * (MH_VT_STRIDE / 2);
              unsigned w4[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) w4[e] = (unsigned)vt[(2 * e) * (MH_VT_STRIDE / 2)] | ((unsigned)vt[(2 * e + 1) * (MH_VT_STRIDE / 2)] << 16);
              v = u32x4_t{w4[0], w4[1], w4[2], w4[3]};
            }
            const int col = (cc < 10 ? 0 : (cc < 20 ? 320 - 80 : 640 - 160)) + 80 * h + cc * 8;
            *reinterpret_cast<u32x4_t*>(g.qkv + ((long long)exr * T + tl) * 960 + col) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // the next head may overwrite Q / K / V^T
    }

    // ================= LayerNorm over the row (attention wavefronts): re-read this lane's own pieces of s =================
    if (half == 0 && !(g.dbg & 8)) {
      rsum += __shfl_xor(rsum, 32, 64);
      rsq += __shfl_xor(rsq, 32, 64);
      const float mean = rsum / (float)MH_D;
      float var = rsq / (float)MH_D - mean * mean;
      var = var > 0.f ? var : 0.f;
      const float rstd = 1.f / sqrtf(var + g.eps);
      if (rvalid) {
        if (g.stats != nullptr && hi == 0) { g.stats[2 * grow] = mean; g.stats[2 * grow + 1] = rstd; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's s stores are in the L2
        const bf16_t* sr = g.s_out + grow * MH_D;
        bf16_t* yr = g.y_out + grow * MH_D;
#pragma unroll 5
        for (int c = 0; c < 20; ++c) {
          const int col = 16 * c + 8 * hi;
          const u32x4_t sv1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(sr + col));
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
          *reinterpret_cast<u32x4_t*>(yr + col) = u32x4_t{dmt_pack_bf16(o[0], o[1]), dmt_pack_bf16(o[2], o[3]), dmt_pack_bf16(o[4], o[5]), dmt_pack_bf16(o[6], o[7])};
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring runs one stage ahead: let it land before the LDS goes away
}

}  // namespace

extern "C" int dmt_mhsa_image_bytes(int64_t* bytes) {
  DMT_CHECK_ARG(bytes != nullptr, "dmt_mhsa_image_bytes: null output");
  *bytes = MH_IMAGE_BYTES;
  return DMT_OK;
}

extern "C" int dmt_mhsa_image_build(const float* wqkv, int64_t ldw, void* image, void* stream) {
  DMT_CHECK_ARG(wqkv && image && ldw >= 960, "dmt_mhsa_image_build: bad argument");
  hipLaunchKernelGGL(mhsa_image_kernel, dim3(160), dim3(256), 0, (hipStream_t)stream, wqkv, (long long)ldw, (unsigned char*)image);
  DMT_CHECK_LAUNCH("dmt_mhsa_image_build");
  return DMT_OK;
}

extern "C" int dmt_mhsa_block_fwd(const dmt_mhsa_desc* d, void* stream) {
  DMT_CHECK_ARG(d != nullptr, "dmt_mhsa_block_fwd: null descriptor");
  DMT_CHECK_ARG(d->d_model == MH_D && d->num_heads == MH_H, "dmt_mhsa_block_fwd: built for d_model %d, %d heads (got %d, %d)", MH_D, MH_H, d->d_model, d->num_heads);
  DMT_CHECK_ARG(d->B > 0 && d->T > 0 && d->T <= 64, "dmt_mhsa_block_fwd: 1 <= T <= 64 (got %d)", d->T);
  DMT_CHECK_ARG(d->x && d->lens && d->image && d->bias && d->gamma && d->beta && d->s_out && d->y_out, "dmt_mhsa_block_fwd: null pointer");
  DMT_CHECK_ARG((((uintptr_t)d->x | (uintptr_t)d->s_out | (uintptr_t)d->y_out | (uintptr_t)d->qkv) & 15) == 0, "dmt_mhsa_block_fwd: tensors must be 16-byte aligned");
  DMT_CHECK_ARG((long long)d->B * d->T * d->num_heads * d->T < (1ll << 32), "dmt_mhsa_block_fwd: dropout counter range");
  MhsaArgs a;
  a.x = (const bf16_t*)d->x; a.lens = d->lens; a.image = (const unsigned char*)d->image;
  a.bias = d->bias; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
  a.qkv = (bf16_t*)d->qkv; a.s_out = (bf16_t*)d->s_out; a.y_out = (bf16_t*)d->y_out; a.stats = d->stats;
  a.B = d->B; a.T = d->T;
  a.Tp = d->T > 32 ? 64 : (d->T > 16 ? 32 : 16);
  a.lgTp = a.Tp == 64 ? 6 : (a.Tp == 32 ? 5 : 4);
  a.tiles = (d->B + (128 / a.Tp) - 1) / (128 / a.Tp);
  a.drop_seed = d->drop_seed;
  const bool drop = d->drop_keep > 0.f && d->drop_keep < 1.f;
  a.drop_thr = drop ? (unsigned)(d->drop_keep * 16777216.0f) : 0u;
  a.drop_inv_keep = drop ? 1.0f / d->drop_keep : 1.0f;
  a.dbg = 0;
#ifdef DMT_TIMING_EXPERIMENTS
  { const char* e = getenv("DMT_MHSA_DEBUG"); a.dbg = e ? atoi(e) : 0; }
#endif
  const int grid = a.tiles < 256 ? a.tiles : 256;
  hipLaunchKernelGGL(mhsa_fwd_kernel, dim3(grid), dim3(MH_NT), 0, (hipStream_t)stream, a);
  DMT_CHECK_LAUNCH("dmt_mhsa_block_fwd");
  return DMT_OK;
}
