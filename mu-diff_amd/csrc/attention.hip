// Fused single-head self-attention for AttnBlockpp (reference backbones/layerspp.py:118-122):
//     out[b,i,:] = sum_j softmax_j(q_i . k_j * scale) v_j        N = H*W positions, head dim C <= 256
// Flash-style: the N x N score matrix never exists in memory (the unfused path wrote, re-read twice and read
// again 4*N*N bytes per sample).  Same split arithmetic as the convolutions (mud_common.h): every fp32 operand is
// hi + lo fp16 and every product is lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 (fp32 accumulate).
//
// One workgroup = 4 waves = 128 queries (32 per wave), one wave per SIMD (the kernel owns the register file:
// Q as 2*C/16 B-fragments and the C x 32 output accumulator stay in registers for the whole key loop).
// Everything is computed TRANSPOSED so that a query is a LANE (MFMA column) from start to finish:
//     S^T[key, query] = K . Q^T         A = K tile (LDS, shared by the 4 waves), B = Q^T (registers)
//     online softmax over keys          per lane: 16 accumulator registers + 1 shuffle with lane^32
//     O^T[ch, query] += V^T . P^T       A = V^T tile (LDS, transposed + key-permuted on the way in),
//                                       B = P^T = the S^T accumulator registers themselves (cdna guide:
//                                       "an accumulator tile as the next MFMA's operand"), no LDS round trip
// K / V tiles of 32 keys are double-buffered in LDS; the next tile's global loads are in flight (registers)
// during the MFMAs of the current one; one barrier per tile.
#include "mud_common.h"
#include <stdlib.h>

template <int C16>
struct AtGeo {
  static constexpr int C = 16 * C16;
  static constexpr int CT = (C16 + 1) / 2;            // 32-channel output tiles
  static constexpr int CP = CT * 32;                  // channels padded to a multiple of 32
  static constexpr int KROW = C * 4 + 16;             // LDS bytes per key row: [hi C x fp16 | lo C x fp16 | pad]
  static constexpr int VROW = 144;                    // LDS bytes per channel row: [hi 32 keys | lo 32 keys | pad]
  static constexpr int KT = 32 * KROW, VT = CP * VROW;
  static constexpr int BUF = KT + VT;
  static constexpr int LDS_BYTES = 2 * BUF;
  static constexpr int KITEMS = 32 * (C / 4), KN = (KITEMS + 255) / 256;   // float4 loads per thread for a K tile
  static constexpr int VITEMS = CP * 8, VN = (VITEMS + 255) / 256;         // 4-key groups per thread for a V tile
};

template <int C16>
__global__ __launch_bounds__(256, 1) void k_attention(const float* __restrict__ qkv, int N, int ld, float scale_log2,
                                                       float* __restrict__ out, int ldo, int tiles_per_split, float* __restrict__ part) {
  using G = AtGeo<C16>;
  constexpr int C = G::C, CT = G::CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 128 + wave * 32 + r;              // this lane's query (both half-waves hold the same 32 queries)
  const float* base = qkv + (int64_t)b * N * ld;
  // key range of this workgroup: all of it, or - when there are too few (batch x query-block) workgroups to fill the chip -
  // one of gridDim.z slices; the partial (unnormalised O, running max, running sum) results are merged by k_attention_combine
  const int kt0 = blockIdx.z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, kt0 + tiles_per_split);

  // ---- Q^T B-fragments, split once: lane (query r, k half hh) holds channels 16s + 8hh .. +7 of its query
  mud_h16x8 qh[C16], ql[C16];
#pragma unroll
  for (int s = 0; s < C16; ++s) {
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (qi < N) {
      const float* qp = base + (int64_t)qi * ld + 16 * s + 8 * hh;
      v0 = *(const f32x4*)qp;
      v1 = *(const f32x4*)(qp + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mud_h16 h0, h1, l0, l1;
      mud_split1(v0[e], h0, l0);
      mud_split1(v1[e], h1, l1);
      qh[s][e] = h0; qh[s][4 + e] = h1;
      ql[s][e] = l0; ql[s][4 + e] = l1;
    }
  }

  // ---- staging loads through a buffer descriptor: ONE per-lane byte offset per stream (K, V) + scalar (SGPR) tile /
  // row offsets, and keys >= N (or padded channels) fall outside the descriptor's range and read as 0 - no
  // per-load address registers, no predication.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((int64_t)N * ld * 4 < 0x7fffffffLL ? (int64_t)N * ld * 4 : 0x7fffffffLL), 0x00020000);
  constexpr int KJ = 256 / (C / 4);                    // key rows covered by one pass of the 256 threads (K stream)
  constexpr int VG = 256 / G::CP;                      // 4-key groups covered by one pass (V stream)
  const unsigned OOR = 0x7ffffff0u;                    // out-of-range offset -> reads 0
  const unsigned kvoff = (tid < G::KITEMS) ? (unsigned)(((tid / (C / 4)) * ld + C + 4 * (tid % (C / 4))) * 4) : OOR;
  const unsigned vvoff = ((tid % G::CP) < C && tid < G::VITEMS) ? (unsigned)((2 * C + (tid % G::CP)) * 4) : OOR;
  f32x4 kraw[G::KN];
  float vraw[G::VN][4];
  auto fetch_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < G::KN; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff, (kt * 32 + KJ * i) * ld * 4, 0);
      kraw[i] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto fetch_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < G::VN; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gkey = 4 * (tid / G::CP + VG * i) + u;            // key inside the tile (the per-lane part is 0 when CP == 256)
        vraw[i][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vvoff + (unsigned)((4 * (tid / G::CP)) * ld * 4),
                                                                                 (kt * 32 + 4 * VG * i + u) * ld * 4, 0));
        (void)gkey;
      }
    }
  };
  auto stash_k = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < G::KN; ++i) {
      const int item = tid + 256 * i, j = item / (C / 4), c4 = item % (C / 4);
      if (item < G::KITEMS) {
        mud_h16x4 hi, lo;
        mud_split4(kraw[i], hi, lo);
        *(mud_h16x4*)(buf + j * G::KROW + c4 * 8) = hi;
        *(mud_h16x4*)(buf + j * G::KROW + C * 2 + c4 * 8) = lo;
      }
    }
  };
  auto stash_v = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < G::VN; ++i) {
      const int item = tid + 256 * i, c = item % G::CP, g = item / G::CP;
      if (item < G::VITEMS) {
        // keys 4g..4g+3 of the tile -> k-step s, lane half h, element block: the order in which the S^T accumulator
        // registers enumerate keys (element e of half h of step s is key 16s + 8(e>>2) + 4h + (e&3))
        const int jj = (4 * g) & 15, s = (4 * g) >> 4, h = (jj >> 2) & 1, blk = (jj >> 3) & 1;
        mud_h16x4 hi, lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mud_h16 th, tl;
          mud_split1(vraw[i][u], th, tl);
          hi[u] = th;
          lo[u] = tl;
        }
        char* p = buf + G::KT + c * G::VROW + s * 32 + h * 16 + blk * 8;
        *(mud_h16x4*)p = hi;
        *(mud_h16x4*)(p + 64) = lo;
      }
    }
  };

  f32x16 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[ct][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  fetch_k(kt0);
  stash_k(smem);
  fetch_v(kt0);
  stash_v(smem);
  __syncthreads();

  for (int kt = kt0; kt < ntiles; ++kt) {
    const char* cur = smem + ((kt - kt0) & 1) * G::BUF;
    char* nxt = smem + ((kt - kt0 + 1) & 1) * G::BUF;
    const bool more = kt + 1 < ntiles;
    if (more) fetch_k(kt + 1);          // next K tile: in flight during S^T, parked in LDS right after it

    // ---- S^T = K . Q^T
    f32x16 st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int s = 0; s < C16; ++s) {
      const char* kp = cur + r * G::KROW + (16 * s + 8 * hh) * 2;
      const mud_h16x8 kh = *(const mud_h16x8*)kp;
      const mud_h16x8 kl = *(const mud_h16x8*)(kp + C * 2);
      st = mud_mfma16(kl, qh[s], st);
      st = mud_mfma16(kh, ql[s], st);
      st = mud_mfma16(kh, qh[s], st);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads from piling up in registers
    }
    if (more) {
      stash_k(nxt);
      fetch_v(kt + 1);                  // next V tile: in flight during softmax + PV (same registers' time slot)
    }

    // ---- online softmax over the 32 keys of the tile (log2 domain); register i of half hh is key (i&3)+8(i>>2)+4hh
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      st[i] = key < N ? st[i] * scale_log2 : -INFINITY;
      mt = fmaxf(mt, st[i]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);                    // finite: every tile holds at least one real key
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st[i] = __builtin_amdgcn_exp2f(st[i] - m_new);
      psum += st[i];
    }
    l_run = l_run * alpha + psum;                            // per-lane partial (its 16 keys); halves are added at the end
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[ct][i] *= alpha;
    }

    // ---- P^T B-fragments straight from the accumulator registers: step s2 = registers 8*s2 .. 8*s2+7
    mud_h16x8 ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = st[8 * s2 + e];
        const mud_h16 t = (mud_h16)p;      // p in [0, 1]: no saturation needed; tiny probabilities are fp16 subnormals, which the MFMA keeps
        ph[s2][e] = t;
        pl[s2][e] = (mud_h16)(p - (float)t);
      }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char* vp = cur + G::KT + (ct * 32 + r) * G::VROW + s2 * 32 + hh * 16;
        const mud_h16x8 vh = *(const mud_h16x8*)vp;
        const mud_h16x8 vl = *(const mud_h16x8*)(vp + 64);
        o[ct] = mud_mfma16(vl, ph[s2], o[ct]);
        o[ct] = mud_mfma16(vh, pl[s2], o[ct]);
        o[ct] = mud_mfma16(vh, ph[s2], o[ct]);
      }
      if ((ct & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }

    if (more) stash_v(nxt);
    __syncthreads();
  }

  // ---- normalise and store: lane = query, register i of tile ct is channel ct*32 + (i&3) + 8(i>>2) + 4hh
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (part) {             // split keys: part[b][split][query][C + 2] = unnormalised O^T column, running max (log2 domain), running sum
    if (qi < N) {
      float* pp = part + (((int64_t)b * gridDim.z + blockIdx.z) * N + qi) * (C + 4);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c0 = ct * 32 + 8 * q4 + 4 * hh;
          if (c0 < C) *(f32x4*)(pp + c0) = f32x4{o[ct][4 * q4], o[ct][4 * q4 + 1], o[ct][4 * q4 + 2], o[ct][4 * q4 + 3]};
        }
      if (hh == 0) { pp[C] = m_run; pp[C + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (qi < N) {
    float* op = out + ((int64_t)b * N + qi) * ldo;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = ct * 32 + 8 * q4 + 4 * hh;
        if (c0 < C) {
          f32x4 v = {o[ct][4 * q4] * inv, o[ct][4 * q4 + 1] * inv, o[ct][4 * q4 + 2] * inv, o[ct][4 * q4 + 3] * inv};
          *(f32x4*)(op + c0) = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Channel-split pair variant for C >= 128 (two waves per SIMD).  k_attention keeps Q (2*C/16 fragments) and the C x 32 output
// accumulator of a 32-query column in ONE wave: 511 VGPRs, one wave per SIMD, so the softmax / staging arithmetic of a wave
// runs with its SIMD's matrix pipe idle (the key loop spent about two thirds of its cycles outside MFMAs).  Here a 32-query
// column is owned by a PAIR of waves (w, w + 4), each with half of the channels:
//     partial S^T = K[:, my half of C] . Q^T[my half]      24 MFMAs of the 48 (C = 256)
//     exchange the two partial tiles through LDS (4 KiB per wave), one barrier, add            -> the full S^T in both waves
//     online softmax (both waves, identical values), P^T fragments from the accumulator registers as before
//     O^T[my half of C] += V^T[my half] . P^T              24 MFMAs of the 48
// Registers per wave halve (Q 64 + O 64), the workgroup has 8 waves = 2 per SIMD, and the K / V staging work is spread over
// 512 threads.  K tiles stay double-buffered; the V tile is single-buffered (a second barrier per key tile orders its refill
// behind the last P.V read) to leave LDS room for the exchange slots.
template <int C16>
struct At2Geo {
  static_assert(C16 % 4 == 0 && C16 >= 8, "pair variant: C in {128, 256}");
  static constexpr int C = 16 * C16;
  static constexpr int H16 = C16 / 2;                 // k16 steps of a wave's channel half
  static constexpr int CTH = C16 / 4;                 // 32-channel output tiles of a wave's half
  static constexpr int KROW = C * 4 + 16;             // LDS bytes per key row: [hi C x fp16 | lo C x fp16 | pad]
  static constexpr int VROW = 144;                    // LDS bytes per channel row: [hi 32 keys | lo 32 keys | pad]
  static constexpr int KT = 32 * KROW, VT = C * VROW;
  static constexpr int V_OFF = 2 * KT, X_OFF = V_OFF + VT;
  static constexpr int LDS_BYTES = X_OFF + 8 * 4096;
  static constexpr int KN = (32 * (C / 4)) / 512;     // float4 loads per thread for a K tile
  static constexpr int VN = (C * 8) / 512;            // 4-key groups per thread for a V tile
  static_assert(KN * 512 == 32 * (C / 4) && VN * 512 == C * 8, "staging split");
};

template <int C16>
__global__ __launch_bounds__(512, 2) void k_attention_pair(const float* __restrict__ qkv, int N, int ld, float scale_log2,
                                                            float* __restrict__ out, int ldo, int tiles_per_split, float* __restrict__ part) {
  using G = At2Geo<C16>;
  constexpr int C = G::C, H16 = G::H16, CTH = G::CTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int pair = wave & 3, half = wave >> 2;                  // query column of the pair, channel half of this wave
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 128 + pair * 32 + r;
  const float* base = qkv + (int64_t)b * N * ld;
  const int kt0 = blockIdx.z * tiles_per_split;
  const int ntiles = min((N + 31) / 32, kt0 + tiles_per_split);

  // ---- Q^T B-fragments of this wave's channel half
  mud_h16x8 qh[H16], ql[H16];
#pragma unroll
  for (int s = 0; s < H16; ++s) {
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (qi < N) {
      const float* qp = base + (int64_t)qi * ld + 16 * (half * H16 + s) + 8 * hh;
      v0 = *(const f32x4*)qp;
      v1 = *(const f32x4*)(qp + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mud_h16 h0, h1, l0, l1;
      mud_split1(v0[e], h0, l0);
      mud_split1(v1[e], h1, l1);
      qh[s][e] = h0; qh[s][4 + e] = h1;
      ql[s][e] = l0; ql[s][4 + e] = l1;
    }
  }

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((int64_t)N * ld * 4 < 0x7fffffffLL ? (int64_t)N * ld * 4 : 0x7fffffffLL), 0x00020000);
  constexpr int KJ = 512 / (C / 4);                    // key rows covered by one pass of the 512 threads (K stream)
  constexpr int VG = 512 / C;                          // 4-key groups covered by one pass (V stream)
  const unsigned kvoff = (unsigned)(((tid / (C / 4)) * ld + C + 4 * (tid % (C / 4))) * 4);
  const unsigned vvoff = (unsigned)((2 * C + (tid % C)) * 4 + (4 * (tid / C)) * ld * 4);
  f32x4 kraw[G::KN];
  float vraw[G::VN][4];
  auto fetch_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < G::KN; ++i)
      kraw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, kvoff, (kt * 32 + KJ * i) * ld * 4, 0));   // keys >= N read 0
  };
  auto fetch_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < G::VN; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        vraw[i][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vvoff, (kt * 32 + 4 * VG * i + u) * ld * 4, 0));
  };
  auto stash_k = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < G::KN; ++i) {
      const int item = tid + 512 * i, j = item / (C / 4), c4 = item % (C / 4);
      mud_h16x4 hi, lo;
      mud_split4(kraw[i], hi, lo);
      *(mud_h16x4*)(buf + j * G::KROW + c4 * 8) = hi;
      *(mud_h16x4*)(buf + j * G::KROW + C * 2 + c4 * 8) = lo;
    }
  };
  auto stash_v = [&]() {
#pragma unroll
    for (int i = 0; i < G::VN; ++i) {
      const int item = tid + 512 * i, c = item % C, g = item / C;
      // keys 4g..4g+3 of the tile -> k-step s, lane half h, element block: the order in which the S^T accumulator
      // registers enumerate keys (element e of half h of step s is key 16s + 8(e>>2) + 4h + (e&3))
      const int jj = (4 * g) & 15, s = (4 * g) >> 4, h = (jj >> 2) & 1, blk = (jj >> 3) & 1;
      mud_h16x4 hi, lo;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mud_h16 th, tl;
        mud_split1(vraw[i][u], th, tl);
        hi[u] = th;
        lo[u] = tl;
      }
      char* p = smem + G::V_OFF + c * G::VROW + s * 32 + h * 16 + blk * 8;
      *(mud_h16x4*)p = hi;
      *(mud_h16x4*)(p + 64) = lo;
    }
  };

  f32x16 o[CTH];
#pragma unroll
  for (int ct = 0; ct < CTH; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[ct][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4* const xmine = (f32x4*)(smem + G::X_OFF + wave * 4096) + lane;          // [4 register quads][64 lanes] of 16 B
  const f32x4* const xpeer = (const f32x4*)(smem + G::X_OFF + (wave ^ 4) * 4096) + lane;

  fetch_k(kt0);
  stash_k(smem);
  fetch_v(kt0);
  stash_v();
  __syncthreads();

  for (int kt = kt0; kt < ntiles; ++kt) {
    const char* cur = smem + ((kt - kt0) & 1) * G::KT;
    char* nxt = smem + ((kt - kt0 + 1) & 1) * G::KT;
    const bool more = kt + 1 < ntiles;
    if (more) fetch_k(kt + 1);

    // ---- partial S^T over this wave's channel half
    f32x16 st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int s = 0; s < H16; ++s) {
      const char* kp = cur + r * G::KROW + (16 * (half * H16 + s) + 8 * hh) * 2;
      const mud_h16x8 kh = *(const mud_h16x8*)kp;
      const mud_h16x8 kl = *(const mud_h16x8*)(kp + C * 2);
      st = mud_mfma16(kl, qh[s], st);
      st = mud_mfma16(kh, ql[s], st);
      st = mud_mfma16(kh, qh[s], st);
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) xmine[q4 * 64] = f32x4{st[4 * q4], st[4 * q4 + 1], st[4 * q4 + 2], st[4 * q4 + 3]};
    if (more) {
      stash_k(nxt);                     // (its last readers passed the previous tile's barriers)
      fetch_v(kt + 1);                  // in flight during softmax + P.V, parked in LDS behind the second barrier
    }
    __syncthreads();                    // the partner's partial tile is visible
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 p = xpeer[q4 * 64];
      // the two halves are always added in the same order (low channels + high channels) so that both waves of the
      // pair hold bit-identical scores
      if (half == 0) { st[4 * q4] += p[0]; st[4 * q4 + 1] += p[1]; st[4 * q4 + 2] += p[2]; st[4 * q4 + 3] += p[3]; }
      else { st[4 * q4] = p[0] + st[4 * q4]; st[4 * q4 + 1] = p[1] + st[4 * q4 + 1]; st[4 * q4 + 2] = p[2] + st[4 * q4 + 2]; st[4 * q4 + 3] = p[3] + st[4 * q4 + 3]; }
    }

    // ---- online softmax over the 32 keys of the tile (log2 domain); register i of half hh is key (i&3)+8(i>>2)+4hh
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      st[i] = key < N ? st[i] * scale_log2 : -INFINITY;
      mt = fmaxf(mt, st[i]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st[i] = __builtin_amdgcn_exp2f(st[i] - m_new);
      psum += st[i];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int ct = 0; ct < CTH; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[ct][i] *= alpha;
    }
    mud_h16x8 ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = st[8 * s2 + e];
        const mud_h16 t = (mud_h16)p;      // p in [0, 1]: no saturation needed; tiny probabilities are fp16 subnormals, which the MFMA keeps
        ph[s2][e] = t;
        pl[s2][e] = (mud_h16)(p - (float)t);
      }

    // ---- O^T[my channel half] += V^T . P^T
#pragma unroll
    for (int ct = 0; ct < CTH; ++ct) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char* vp = smem + G::V_OFF + ((half * CTH + ct) * 32 + r) * G::VROW + s2 * 32 + hh * 16;
        const mud_h16x8 vh = *(const mud_h16x8*)vp;
        const mud_h16x8 vl = *(const mud_h16x8*)(vp + 64);
        o[ct] = mud_mfma16(vl, ph[s2], o[ct]);
        o[ct] = mud_mfma16(vh, pl[s2], o[ct]);
        o[ct] = mud_mfma16(vh, ph[s2], o[ct]);
      }
    }
    __syncthreads();                    // every wave is done with the V tile (and with its partner's exchange slot)
    if (more) stash_v();                // ordered before the next P.V by the next tile's first barrier
  }

  // ---- normalise and store this wave's channel half
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (part) {
    if (qi < N) {
      float* pp = part + (((int64_t)b * gridDim.z + blockIdx.z) * N + qi) * (C + 4);
#pragma unroll
      for (int ct = 0; ct < CTH; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c0 = (half * CTH + ct) * 32 + 8 * q4 + 4 * hh;
          *(f32x4*)(pp + c0) = f32x4{o[ct][4 * q4], o[ct][4 * q4 + 1], o[ct][4 * q4 + 2], o[ct][4 * q4 + 3]};
        }
      if (hh == 0 && half == 0) { pp[C] = m_run; pp[C + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  if (qi < N) {
    float* op = out + ((int64_t)b * N + qi) * ldo;
#pragma unroll
    for (int ct = 0; ct < CTH; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = (half * CTH + ct) * 32 + 8 * q4 + 4 * hh;
        *(f32x4*)(op + c0) = f32x4{o[ct][4 * q4] * inv, o[ct][4 * q4 + 1] * inv, o[ct][4 * q4 + 2] * inv, o[ct][4 * q4 + 3] * inv};
      }
  }
}

// out[b, q, :] = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s,  m = max_s m_s   (one thread per 4 channels)
__global__ __launch_bounds__(256) void k_attention_combine(const float* __restrict__ part, int N, int C, int nsplit, float* __restrict__ out, int ldo) {
  const int C4 = C / 4;
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= (unsigned)(N * C4)) return;
  const int q = (int)(t / (unsigned)C4), c = (int)(t - (unsigned)q * (unsigned)C4) * 4, b = blockIdx.y;
  const int64_t row = C + 4;
  const float* p0 = part + ((int64_t)b * nsplit * N + q) * row;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, p0[(int64_t)s * N * row + C]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* ps = p0 + (int64_t)s * N * row;
    const float w = __builtin_amdgcn_exp2f(ps[C] - m);
    acc += w * *(const f32x4*)(ps + c);
    l += w * ps[C + 1];
  }
  *(f32x4*)(out + ((int64_t)b * N + q) * ldo + c) = acc * (1.0f / l);
}

// Key splits used for (B, N).  A workgroup owns 128 queries and the whole key range; the chip runs 256 workgroups at a time (one
// per CU), so the launch takes ceil(wgs / 256) rounds of full-length workgroups.  Splitting the keys ns ways gives ns times the
// workgroups of 1/ns the length (+ a small merge): chosen to minimise rounds / ns, e.g. 128 workgroups (4 slices of 64x64):
// 1 round -> 2 splits, half the time; 384 (12 slices): 2 rounds -> 2 splits, 3 rounds of half length; 512: no split.
static int at_splits(int B, int N) {
  const int64_t wgs = (int64_t)B * mud_cdiv(N, 128);
  const int ntiles = (int)mud_cdiv(N, 32);
  if (ntiles < 8) return 1;
  int best = 1;
  double best_cost = (double)mud_cdiv(wgs, 256);
  for (int ns = 2; ns <= 16 && ns <= ntiles / 4; ++ns) {        // at least 4 key tiles (128 keys) per split
    const double cost = (double)mud_cdiv(wgs * ns, 256) / ns + 0.01 * ns;      // + merge pass and partial traffic
    if (cost < best_cost - 1e-9) best_cost = cost, best = ns;
  }
  const int tps = (int)mud_cdiv(ntiles, best);
  return (int)mud_cdiv(ntiles, tps);                  // no empty split
}

template <int C16>
static int at_launch_pair(const float* qkv, int B, int N, int ld, float scale, float* out, int ldo, float* ws, hipStream_t s) {
  using G = At2Geo<C16>;
  static mud_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)k_attention_pair<C16>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      mud_set_error("mud_attention: cannot reserve %d B of LDS: %s", G::LDS_BYTES, hipGetErrorString(e));
      return MUD_ERR_LAUNCH;
    }
    attr_once.ok();
  }
  const int ntiles = (int)mud_cdiv(N, 32);
  const int ns = ws ? at_splits(B, N) : 1;
  const int tps = (int)mud_cdiv(ntiles, ns);
  hipLaunchKernelGGL((k_attention_pair<C16>), dim3((unsigned)mud_cdiv(N, 128), B, ns), dim3(512), G::LDS_BYTES, s, qkv, N, ld,
                     scale * 1.44269504088896340736f, out, ldo, tps, ns > 1 ? ws : (float*)nullptr);
  MUD_CHECK_LAUNCH("mud_attention(pair)");
  if (ns > 1) {
    hipLaunchKernelGGL(k_attention_combine, dim3((unsigned)mud_cdiv((int64_t)N * (G::C / 4), 256), B), dim3(256), 0, s, ws, N, G::C, ns, out, ldo);
    MUD_CHECK_LAUNCH("mud_attention(combine)");
  }
  return MUD_OK;
}

template <int C16>
static int at_launch(const float* qkv, int B, int N, int ld, float scale, float* out, int ldo, float* ws, hipStream_t s) {
  using G = AtGeo<C16>;
  static mud_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)k_attention<C16>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      mud_set_error("mud_attention: cannot reserve %d B of LDS: %s", G::LDS_BYTES, hipGetErrorString(e));
      return MUD_ERR_LAUNCH;
    }
    attr_once.ok();
  }
  const int ntiles = (int)mud_cdiv(N, 32);
  const int ns = ws ? at_splits(B, N) : 1;
  const int tps = (int)mud_cdiv(ntiles, ns);
  hipLaunchKernelGGL((k_attention<C16>), dim3((unsigned)mud_cdiv(N, 128), B, ns), dim3(256), G::LDS_BYTES, s, qkv, N, ld,
                     scale * 1.44269504088896340736f, out, ldo, tps, ns > 1 ? ws : (float*)nullptr);
  MUD_CHECK_LAUNCH("mud_attention");
  if (ns > 1) {
    hipLaunchKernelGGL(k_attention_combine, dim3((unsigned)mud_cdiv((int64_t)N * (G::C / 4), 256), B), dim3(256), 0, s, ws, N, G::C, ns, out, ldo);
    MUD_CHECK_LAUNCH("mud_attention(combine)");
  }
  return MUD_OK;
}

extern "C" int64_t mud_attention_ws_bytes(int B, int N, int C) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  const int ns = at_splits(B, N);
  return ns > 1 ? (int64_t)B * ns * N * (C + 4) * 4 : 0;
}

extern "C" int mud_attention_supported(int C) { return C == 16 || C == 32 || C == 64 || C == 128 || C == 256; }

extern "C" int mud_attention(const float* qkv, int B, int N, int C, int ld, float scale, float* out, int ldo, void* ws, void* stream) {
  MUD_REQUIRE(qkv && out, "mud_attention: null pointer");
  MUD_REQUIRE(B >= 0 && B <= 65535 && N > 0 && ld >= 3 * C && ldo >= C, "mud_attention: bad sizes");
  MUD_REQUIRE(mud_attention_supported(C), "mud_attention: head dim %d not in {16,32,64,128,256}", C);
  MUD_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && mud_aligned16(qkv) && mud_aligned16(out) && mud_aligned16(ws), "mud_attention: needs ld %% 4 == 0 and 16-byte aligned buffers");
  if (B == 0) return MUD_OK;
  hipStream_t s = (hipStream_t)stream;
  static const bool at_pair = !(getenv("MUD_ATT_SINGLE") && atoi(getenv("MUD_ATT_SINGLE")));   // A/B knob: 1 = the one-wave-per-column kernel for every C
  switch (C) {
    case 16: return at_launch<1>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s);
    case 32: return at_launch<2>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s);
    case 64: return at_launch<4>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s);
    case 128: return at_pair ? at_launch_pair<8>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s) : at_launch<8>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s);
    default: return at_pair ? at_launch_pair<16>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s) : at_launch<16>(qkv, B, N, ld, scale, out, ldo, (float*)ws, s);
  }
}
