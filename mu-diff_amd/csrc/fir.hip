// Up-FIR-down resampling (StyleGAN2 upfirdn2d) for gfx950.
//
// Maths (utils/op/upfirdn2d.py:201-242 of the reference): zero-stuff by `up`, pad by (pad0, pad1)
// (negative = crop), TRUE convolution with the kh x kw kernel, keep every `down`-th sample:
//   out[oy,ox] = sum_{m,n} k[kh-1-m][kw-1-n] * U[oy*down + m - pad0][ox*down + n - pad0],
//   U[Y][X] = in[Y/up][X/up] if Y%up==0 and X%up==0 and inside, else 0.
// Only the taps that land on real samples are visited (polyphase): for the 4x4 [1,3,3,1] filter that
// is 4 of 16 taps per output when up == 2.
//
// Two forms: planes [P,H,W] (the reference's pybind op boundary; W is the coalesced axis) and NHWC
// views (inside the generators; C is the coalesced axis, 16 B per lane, optional AdaGN+SiLU prologue
// and a second, un-normalised output produced from the same loads).
#include "mud_common.h"
#include <stdlib.h>

#define FIR_MAX_TAPS 64
struct FirKernel { float k[FIR_MAX_TAPS]; };

__global__ __launch_bounds__(256) void k_upfirdn2d_planes(const float* __restrict__ in, int64_t planes, int H, int W,
                                                          const float* __restrict__ kern, int kh, int kw, int up_x, int up_y,
                                                          int down_x, int down_y, int pad_x0, int pad_y0, int Ho, int Wo,
                                                          float* __restrict__ out) {
  __shared__ float sk[FIR_MAX_TAPS];
  for (int i = threadIdx.x; i < kh * kw; i += 256) sk[i] = kern[i];
  __syncthreads();
  const int64_t total = planes * Ho * Wo;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const int64_t r = idx / Wo;
    const int oy = (int)(r % Ho);
    const int64_t pl = r / Ho;
    const float* ip = in + pl * H * W;
    float acc = 0.f;
    for (int m = 0; m < kh; ++m) {
      const int Y = oy * down_y + m - pad_y0;
      if (Y < 0 || Y % up_y) continue;
      const int iy = Y / up_y;
      if (iy >= H) continue;
      for (int n = 0; n < kw; ++n) {
        const int X = ox * down_x + n - pad_x0;
        if (X < 0 || X % up_x) continue;
        const int ix = X / up_x;
        if (ix >= W) continue;
        acc = fmaf(sk[(kh - 1 - m) * kw + (kw - 1 - n)], ip[(int64_t)iy * W + ix], acc);
      }
    }
    out[idx] = acc;
  }
}

extern "C" int mud_upfirdn2d(const float* in, int64_t planes, int H, int W, const float* kernel, int kh, int kw, int up_x,
                             int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float* out,
                             void* stream) {
  MUD_REQUIRE(in && kernel && out, "mud_upfirdn2d: null pointer");
  MUD_REQUIRE(planes >= 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && kh * kw <= FIR_MAX_TAPS, "mud_upfirdn2d: bad sizes (kernel up to %d taps)", FIR_MAX_TAPS);
  MUD_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "mud_upfirdn2d: up/down must be >= 1");
  const int Ho = (H * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int Wo = (W * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  MUD_REQUIRE(Ho > 0 && Wo > 0 && (H * up_y + pad_y0 + pad_y1 - kh) >= 0 && (W * up_x + pad_x0 + pad_x1 - kw) >= 0, "mud_upfirdn2d: empty output");
  if (planes == 0) return MUD_OK;
  int64_t blocks = mud_cdiv(planes * Ho * Wo, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_upfirdn2d_planes, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, kernel, kh, kw,
                     up_x, up_y, down_x, down_y, pad_x0, pad_y0, Ho, Wo, out);
  MUD_CHECK_LAUNCH("mud_upfirdn2d");
  return MUD_OK;
}

// UP/DOWN = 0: runtime up/down and kernel size (generic).  UP/DOWN > 0: compile-time factors with a 4x4 kernel:
// the tap loops are unrolled and, for UP == 2, only the 2x2 polyphase taps that land on real samples are visited.
template <bool WITH_H, bool WITH_X, int UP, int DOWN>
__global__ __launch_bounds__(256) void k_fir_nhwc(const float* __restrict__ x, int B, int H, int W, int C4, int ldx, FirKernel fk,
                                                  int kh, int kw, int up, int down, int pad0, int Ho, int Wo,
                                                  const float* __restrict__ psc, const float* __restrict__ psh, int pro_ld,
                                                  int pro_mode, float* __restrict__ out_h, int ldh, float* __restrict__ out_x,
                                                  int ldxo) {
  // grid = (ceil(Wo*C4 / 256), Ho, B): one 32-bit division per output instead of five 64-bit ones (each of those is a
  // ~100-instruction software routine - more work than the filter itself)
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t < (unsigned)(Wo * C4)) {
    const int ox = (int)(t / (unsigned)C4);
    const int c = (int)(t - (unsigned)ox * (unsigned)C4) * 4;
    const int oy = blockIdx.y;
    const int b = blockIdx.z;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (WITH_H && pro_mode != MUD_PRO_NONE) {
      sc = *(const f32x4*)(psc + (int64_t)b * pro_ld + c);
      sh = *(const f32x4*)(psh + (int64_t)b * pro_ld + c);
    }
    f32x4 ah = {0.f, 0.f, 0.f, 0.f}, ax = {0.f, 0.f, 0.f, 0.f};
    constexpr bool SPEC = UP > 0;
    const int upv = SPEC ? UP : up, downv = SPEC ? DOWN : down, khv = SPEC ? 4 : kh, kwv = SPEC ? 4 : kw;
    // first tap whose (o*down + m - pad0) is a multiple of `up`, then every up-th tap
    const int my0 = SPEC ? (((pad0 - oy * downv) % upv) + upv) % upv : 0;
    const int mx0 = SPEC ? (((pad0 - ox * downv) % upv) + upv) % upv : 0;
#pragma unroll
    for (int mi = 0; mi < (SPEC ? 4 / UP : 64); ++mi) {
      const int m = SPEC ? my0 + mi * upv : mi;
      if (!SPEC && m >= khv) break;
      const int Y = oy * downv + m - pad0;
      if (Y < 0 || (!SPEC && Y % upv)) continue;
      const int iy = Y / upv;
      if (iy >= H) continue;
#pragma unroll
      for (int ni = 0; ni < (SPEC ? 4 / UP : 64); ++ni) {
        const int n = SPEC ? mx0 + ni * upv : ni;
        if (!SPEC && n >= kwv) break;
        const int X = ox * downv + n - pad0;
        if (X < 0 || (!SPEC && X % upv)) continue;
        const int ix = X / upv;
        if (ix >= W) continue;
        const float kv = fk.k[(khv - 1 - m) * kwv + (kwv - 1 - n)];
        const f32x4 v = *(const f32x4*)(x + (((int64_t)b * H + iy) * W + ix) * ldx + c);
        if (WITH_X) ax += kv * v;
        if (WITH_H) {
          f32x4 t;
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = mud_prologue_fast(v[j], sc[j], sh[j], pro_mode);
          ah += kv * t;
        }
      }
    }
    const int64_t op = ((int64_t)b * Ho + oy) * Wo + ox;
    if (WITH_H) *(f32x4*)(out_h + op * ldh + c) = ah;
    if (WITH_X) *(f32x4*)(out_x + op * ldxo + c) = ax;
  }
}

// ---- 2x2-output-per-thread form of the x2 up-sampler (4x4 filter, pad (2,1)).  One output per thread makes very short
// threads (4 loads, 1 store, exit): that kernel is bound by the latency of the chain times the number of workgroups a CU
// can hold (2.3 TB/s), not by HBM.  Here a thread owns a 2x2 output quad: its 3x3 input loads are independent and in
// flight together, every input is activated once per quad instead of once per output, and 4 stores leave per thread
// (4.2 TB/s).  (The same trick on the /2 down-sampler - 6x6 inputs, two outputs - costs more registers than it saves.)
__device__ __forceinline__ f32x4 fir_ld(const float* __restrict__ x, int64_t base, int iy, int ix, int H, int W, int ldx, int c, float& ok) {
  const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
  ok = in ? 1.f : 0.f;
  const int cy = in ? iy : 0, cx = in ? ix : 0;                 // clamped address, value masked: branch-free
  return *(const f32x4*)(x + (base + (int64_t)cy * W + cx) * ldx + c);
}

template <bool WITH_H, bool WITH_X>
__global__ __launch_bounds__(256) void k_fir_up2_quad(const float* __restrict__ x, int H, int W, int C4, int ldx, FirKernel fk,
                                                      const float* __restrict__ psc, const float* __restrict__ psh, int pro_ld, int pro_mode,
                                                      float* __restrict__ out_h, int ldh, float* __restrict__ out_x, int ldxo) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;           // grid = (ceil(W*C4/256), H, B): quad (i, j) <-> input pixel
  if (t >= (unsigned)(W * C4)) return;
  const int j = (int)(t / (unsigned)C4), c = (int)(t - (unsigned)j * (unsigned)C4) * 4, i = blockIdx.y, b = blockIdx.z;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (WITH_H && pro_mode != MUD_PRO_NONE) {
    sc = *(const f32x4*)(psc + (int64_t)b * pro_ld + c);
    sh = *(const f32x4*)(psh + (int64_t)b * pro_ld + c);
  }
  const int64_t base = (int64_t)b * H * W;
  f32x4 v[3][3];
  float ok[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) v[dy][dx] = fir_ld(x, base, i + dy - 1, j + dx - 1, H, W, ldx, c, ok[dy][dx]);
  f32x4 ah[2][2], ax[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) ah[a][bb] = ax[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // out[2i+a] = sum_m K[3-m] U[2i+a+m-2], U[2q] = in[q]: a = 0 takes rows (i-1: K row 3, i: K row 1), a = 1 takes (i: 2, i+1: 0)
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      f32x4 tt = v[dy][dx];
      if (WITH_H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tt[e] = mud_prologue_fast(tt[e], sc[e], sh[e], pro_mode) * ok[dy][dx];
      }
      const f32x4 vx = v[dy][dx] * ok[dy][dx];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ry = a == 0 ? (dy == 0 ? 3 : (dy == 1 ? 1 : -1)) : (dy == 1 ? 2 : (dy == 2 ? 0 : -1));
        if (ry < 0) continue;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int rx = bb == 0 ? (dx == 0 ? 3 : (dx == 1 ? 1 : -1)) : (dx == 1 ? 2 : (dx == 2 ? 0 : -1));
          if (rx < 0) continue;
          const float kv = fk.k[ry * 4 + rx];
          if (WITH_H) ah[a][bb] += kv * tt;
          if (WITH_X) ax[a][bb] += kv * vx;
        }
      }
    }
  const int Wo = 2 * W;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int64_t op = ((int64_t)b * 2 * H + 2 * i + a) * Wo + 2 * j + bb;
      if (WITH_H) *(f32x4*)(out_h + op * ldh + c) = ah[a][bb];
      if (WITH_X) *(f32x4*)(out_x + op * ldxo + c) = ax[a][bb];
    }
}

extern "C" int mud_fir_nhwc(const float* x, int B, int H, int W, int C, int ldx, const float* kernel_host, int kh, int kw, int up,
                            int down, int pad0, int pad1, const float* pro_scale, const float* pro_shift, int pro_ld,
                            int pro_mode, float* out_h, int ldh, float* out_x, int ldxo, void* stream) {
  MUD_REQUIRE(x && kernel_host && (out_h || out_x), "mud_fir_nhwc: null pointer");
  MUD_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C && mud_aligned16(x), "mud_fir_nhwc: needs C%%4==0, ld%%4==0, aligned x");
  MUD_REQUIRE(kh > 0 && kw > 0 && kh * kw <= FIR_MAX_TAPS && up >= 1 && down >= 1, "mud_fir_nhwc: bad filter");
  MUD_REQUIRE(pro_mode == MUD_PRO_NONE || (pro_scale && pro_shift && pro_ld >= C && out_h), "mud_fir_nhwc: prologue arrays missing");
  MUD_REQUIRE(!out_h || (ldh >= C && ldh % 4 == 0 && mud_aligned16(out_h)), "mud_fir_nhwc: bad out_h view");
  MUD_REQUIRE(!out_x || (ldxo >= C && ldxo % 4 == 0 && mud_aligned16(out_x)), "mud_fir_nhwc: bad out_x view");
  const int Ho = (H * up + pad0 + pad1 - kh) / down + 1, Wo = (W * up + pad0 + pad1 - kw) / down + 1;
  MUD_REQUIRE(Ho > 0 && Wo > 0, "mud_fir_nhwc: empty output");
  if (B == 0) return MUD_OK;
  FirKernel fk;
  for (int i = 0; i < FIR_MAX_TAPS; ++i) fk.k[i] = i < kh * kw ? kernel_host[i] : 0.f;
  MUD_REQUIRE(Ho <= 65535 && B <= 65535 && (int64_t)Wo * (C / 4) < (1ll << 31), "mud_fir_nhwc: output too large for the launch grid");
  dim3 grid((unsigned)mud_cdiv((int64_t)Wo * (C / 4), 256), (unsigned)Ho, (unsigned)B), block(256);
  hipStream_t s = (hipStream_t)stream;
#define FIR_LAUNCH(HH, XX, UU, DD) hipLaunchKernelGGL((k_fir_nhwc<HH, XX, UU, DD>), grid, block, 0, s, x, B, H, W, C / 4, ldx, fk, kh, kw, up, down, pad0, Ho, Wo, pro_scale, pro_shift, pro_ld, pro_mode, out_h, ldh, out_x, ldxo)
#define FIR_PICK(UU, DD) do { if (out_h && out_x) FIR_LAUNCH(true, true, UU, DD); else if (out_h) FIR_LAUNCH(true, false, UU, DD); else FIR_LAUNCH(false, true, UU, DD); } while (0)
  #define FIR_QUAD(KERN, GX, GY, ...) do { dim3 qg((unsigned)mud_cdiv((int64_t)(GX), 256), (unsigned)(GY), (unsigned)B); \
    if (out_h && out_x) hipLaunchKernelGGL((KERN<true, true>), qg, block, 0, s, __VA_ARGS__); \
    else if (out_h) hipLaunchKernelGGL((KERN<true, false>), qg, block, 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERN<false, true>), qg, block, 0, s, __VA_ARGS__); } while (0)
  if (kh == 4 && kw == 4 && up == 2 && down == 1 && pad0 == 2 && pad1 == 1)
    FIR_QUAD(k_fir_up2_quad, (int64_t)W * (C / 4), H, x, H, W, C / 4, ldx, fk, pro_scale, pro_shift, pro_ld, pro_mode, out_h, ldh, out_x, ldxo);
  else if (kh == 4 && kw == 4 && up == 2 && down == 1) FIR_PICK(2, 1);
  else if (kh == 4 && kw == 4 && up == 1 && down == 2) FIR_PICK(1, 2);
  else if (kh == 4 && kw == 4 && up == 1 && down == 1) FIR_PICK(1, 1);
  else FIR_PICK(0, 0);
#undef FIR_QUAD
#undef FIR_PICK
#undef FIR_LAUNCH
  MUD_CHECK_LAUNCH("mud_fir_nhwc");
  return MUD_OK;
}
