// MX-fp8 convolution path (BASELINE.json configs[4]: "fp8 MFMA conv path with bf16 accumulate") for gfx950: forward 3x3 / 1x1
// stride 1 / 2, data gradient stride 1 and the 4-phase x2 transposed convolution -- the launches of conv_halo_kernel -- on
// v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3 operands and one E8M0 scale per 32 reduction elements (the only fp8 form that
// is faster than bf16 on this chip: twice the bf16 FLOPs per cycle).  Feature maps stay bf16 in HBM; accumulation is fp32.
//
// Replaces the same reference calls as conv_igemm.hip (F.conv2d custom_layers.py:41,43,83; F.conv_transpose2d :78; modulation
// :62-72; convolution_backward), at reduced operand precision: every staged 32-channel run of a pixel (one MX block) is scaled by
// a power of two so that its largest magnitude lands in (224, 448] and rounded to e4m3 (3 mantissa bits) WHILE it is staged into
// LDS; weights are quantised once per optimiser step by lcgan_conv_weight_prep_fp8.  Products are exact in the fp32 accumulator.
//
// Operand layout, measured with scripts/probes/mx_probe.hip (exact integer data): lane l of the 32x32x64 form holds row (l & 31) and
// 32 bytes; byte j is k = 32 (j >> 4) + 16 (l >> 5) + (j & 15); the scale in lanes 0-31 multiplies k = 0..31 (bytes 0-15 of BOTH
// lane halves), the scale in lanes 32-63 multiplies k = 32..63 (bytes 16-31).  A pixel's 64-channel step is therefore stored in LDS
// as [c 0-15 | c 32-47 | c 16-31 | c 48-63] so that lane half h reads 32 contiguous bytes at 32 h, followed by its two scale bytes.
#include "common.h"

#include <algorithm>
#include <cstdio>

namespace {

constexpr int BN8 = 128;                      // output channels per workgroup
constexpr int HT8 = 16;                       // 16 x 16 output positions of one sample
constexpr int PROW = 80;                      // bytes per staged pixel / weight row: 64 data + 2 scales + pad (80 B pitch: conflict-free)
constexpr int BTILE8 = BN8 * PROW;            // bytes per staged weight tile

struct TapTable8 { int n; int dy[9]; int dx[9]; int wt[9]; };

struct Fp8Args {
  const __bf16* x; const unsigned char* w; const unsigned char* wsc; __bf16* y;    // w: [taps][N][K64][64] e4m3 (interleaved), wsc: [taps][N][K64][2] E8M0
  const float* pre; const float* post; const float* bias; const __bf16* residual; int res_half;
  int B, Hin, Win, Cin, Hout, Wout, Cout, Hm, Wm, N, K64;
  int out_mul, tiles_x, tiles_y;
  float bias_scale, gain; int act;
  TapTable8 taps[4];
  int hy0[4], hx0[4], hh[4], hw[4];
  int halo_bytes;
};

__device__ __forceinline__ bf16x8 zero_bf16x8_() {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)0.f;
  return r;
}

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// E8M0 exponent byte e (value 2^(e - 127)) such that m * 2^-(e - 127) lies in (224, 448]: the block's largest magnitude just fits e4m3
__device__ __forceinline__ int mx_scale_byte(float m) {
  if (!(m > 0.f)) return 127;
  int ex;
  const float f = frexpf(m * (1.f / 448.f), &ex);      // m / 448 = f * 2^ex, f in [0.5, 1)
  (void)f;
  return min(max(ex + 127, 1), 254);
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __builtin_bit_cast(float, (unsigned)(254 - e) << 23); }   // 2^(127 - e)

// 8 floats (already divided by the block scale) -> 8 e4m3 bytes
__device__ __forceinline__ uint2 pack_fp8x8(const float* v) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}
// byte position of channels [8 s, 8 s + 8) of MX block q inside a pixel's interleaved 64-byte step (see the header)
__device__ __host__ __forceinline__ int mx_pos(int q, int s) { return 32 * (s >> 1) + 16 * q + 8 * (s & 1); }

// ---- weight preparation: w [A][Bc][kk] f32 -> e4m3 [kk][N][K64][64] (interleaved) + E8M0 [kk][N][K64][2] ----------------------
__global__ void prep_weight_fp8_kernel(const float* __restrict__ w, int A, int Bc, int kk, float scale, int transpose,
                                       unsigned char* __restrict__ out, unsigned char* __restrict__ osc, int N, int Kc, int K64) {
  const size_t nblk = (size_t)kk * N * K64 * 2;                         // one thread per MX block (32 reduction channels)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  const int q = (int)(i & 1);
  const int k64 = (int)((i >> 1) % K64);
  const int n = (int)((i >> 1) / K64 % N);
  const int t = (int)((i >> 1) / ((size_t)K64 * N));
  float v[32], m = 0.f;
  for (int j = 0; j < 32; ++j) {
    const int c = k64 * 64 + q * 32 + j;
    float x = 0.f;
    if (c < Kc) {
      const int aa = transpose ? c : n, bb = transpose ? n : c;
      x = w[((size_t)aa * Bc + bb) * kk + t] * scale;
    }
    v[j] = x; m = fmaxf(m, fabsf(x));
  }
  const int e = mx_scale_byte(m);
  const float inv = mx_inv_scale(e);
  for (int j = 0; j < 32; ++j) v[j] *= inv;
  unsigned char* row = out + ((size_t)(t * N + n) * K64 + k64) * 64;
  for (int s = 0; s < 4; ++s) *(uint2*)(row + mx_pos(q, s)) = pack_fp8x8(v + 8 * s);
  osc[((size_t)(t * N + n) * K64 + k64) * 2 + q] = (unsigned char)e;
}

// =========================================================================================================
// halo-tile kernel, MX-fp8 operands.  Same tile / wave decomposition as conv_halo_kernel (16 x 16 positions x 128 channels,
// 8 waves = 4 x 2 of 64 x 64, one barrier per tap), K step = 64 channels: 4 MFMAs of 32x32x64 per wave and step.
// =========================================================================================================
template <int EPI>
__global__ __launch_bounds__(512, 4) void conv_halo_fp8_kernel(Fp8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI = 3;                         // halo (pixel, 8-channel vector) items per thread and 32-channel MX block: ceil(18*18*4 / 512)
  unsigned char* halo = (unsigned char*)smem;   // 2 x [halo pixels][PROW]: the image of the next 64-channel step is built while this one is read
  unsigned char* Bt = halo + 2 * a.halo_bytes;  // 2 x [128][PROW]
  float* psc = (float*)(Bt + 2 * BTILE8);       // [K64 * 64] style scales of this sample (modulated convs)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int phase = gridDim.z - 1 - blockIdx.z, n0 = blockIdx.y * BN8;
  int tile = blockIdx.x;
  if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
  const TapTable8& tt = a.taps[phase];
  const int hy0 = a.hy0[phase], hx0 = a.hx0[phase], hh = a.hh[phase], hw = a.hw[phase];
  const int gy0 = ty * HT8 + hy0, gx0 = tx * HT8 + hx0;

  // ---- halo items: (pixel, run s = tid & 3 of 8 channels) of ONE 32-channel MX block at a time --------------------------------
  const int hs = tid & 3;
  int goff[NI], loff[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int hp = (tid >> 2) + k * 128;
    loff[k] = -1; goff[k] = -1;
    if (hp < hh * hw) {
      const int hy = hp / hw, hx = hp - hy * hw;
      const int gy = gy0 + hy, gx = gx0 + hx;
      loff[k] = hp * PROW;
      if ((unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win) goff[k] = ((b * a.Hin + gy) * a.Win + gx) * a.Cin + hs * 8;
    }
  }
  bf16x8 hreg[NI];
  int h_c0 = 0;                                  // first channel of the block in flight
  auto halo_load = [&](int c0) {
    h_c0 = c0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const bool ok = goff[k] >= 0 && c0 + hs * 8 < a.Cin;
      hreg[k] = ok ? *(const bf16x8*)(a.x + (size_t)goff[k] + c0) : zero_bf16x8_();
    }
  };
  // quantise while staging: style multiply, block maximum over the 4 lanes that share the 32-channel run, power-of-two scale, e4m3
  auto halo_store = [&](int buf, int q) {        // q: MX block (0 / 1) inside the 64-channel step
    unsigned char* hb = halo + buf * a.halo_bytes;
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0;
    if (a.pre) { s0 = *(const f32x4*)(psc + h_c0 + hs * 8); s1 = *(const f32x4*)(psc + h_c0 + hs * 8 + 4); }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      float v[8], m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = (float)hreg[k][j] * (j < 4 ? s0[j] : s1[j - 4]); m = fmaxf(m, fabsf(v[j])); }
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const int e = mx_scale_byte(m);
      const float inv = mx_inv_scale(e);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= inv;
      if (loff[k] >= 0) {
        *(uint2*)(hb + loff[k] + mx_pos(q, hs)) = pack_fp8x8(v);
        if (hs == 0) hb[loff[k] + 64 + q] = (unsigned char)e;
      }
    }
  };

  // ---- weight tile: 128 rows x 64 bytes + 2 scale bytes; one 16-byte vector per thread ---------------------------------------
  const int brow = tid >> 2, bvec = tid & 3;
  const int ntaps = tt.n, nchunks = a.K64, total = ntaps * nchunks;
  const bool bvalid = n0 + brow < a.N;
  struct BReg { i32x4 d; unsigned short sc; };
  auto b_load = [&](int c, int t) -> BReg {
    BReg r; r.d = i32x4{0, 0, 0, 0}; r.sc = 0x7f7f;
    if (bvalid) {
      const size_t row = ((size_t)tt.wt[t] * a.N + n0 + brow) * a.K64 + c;
      r.d = *(const i32x4*)(a.w + row * 64 + bvec * 16);
      if (bvec == 0) r.sc = *(const unsigned short*)(a.wsc + row * 2);
    }
    return r;
  };
  auto b_store = [&](int buf, const BReg& r) {
    unsigned char* p = Bt + buf * BTILE8 + brow * PROW;
    *(i32x4*)(p + bvec * 16) = r.d;
    if (bvec == 0) *(unsigned short*)(p + 64) = r.sc;
  };

  const int lrow = lane & 31, lh = lane >> 5;
  int abase[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int r = wm * 64 + mi * 32 + lrow;
    abase[mi] = ((r >> 4) * hw + (r & 15)) * PROW;
  }
  const int bbase = (wn * 64 + lrow) * PROW;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int lc = 0, lt = 0;
  auto advance = [&]() { if (++lt == ntaps) { lt = 0; ++lc; } };

  if (a.pre) {
    for (int i = tid; i < a.K64 * 64; i += 512) psc[i] = i < a.Cin ? a.pre[(size_t)b * a.Cin + i] : 0.f;
    __syncthreads();
  }
  halo_load(0);  halo_store(0, 0);
  halo_load(32); halo_store(0, 1);
  b_store(0, b_load(0, 0));
  advance();
  BReg r0 = (1 < total) ? b_load(lc, lt) : BReg{i32x4{0, 0, 0, 0}, 0x7f7f};
  advance();
  BReg r1 = BReg{i32x4{0, 0, 0, 0}, 0x7f7f};
  __syncthreads();

  // the next step's image is built in the OTHER halo buffer in two sub-stagings (one MX block each, so only 3 vectors per thread are
  // ever in flight): block 0 loaded at tap 0 and stored at tap t_mid, block 1 loaded at t_mid and stored at the last tap
  const int t_mid = ntaps >> 1;
  int c = 0, t = 0;
  auto step = [&](int q, BReg& rs, BReg& rl) {
    const bool more = c + 1 < nchunks;
    if (more && t == 0) halo_load((c + 1) * 64);
    if (q + 2 < total) { rl = b_load(lc, lt); advance(); }
    const int toff = ((tt.dy[t] - hy0) * hw + (tt.dx[t] - hx0)) * PROW;
    const unsigned char* Bc = Bt + (q & 1) * BTILE8;
    const unsigned char* Hc = halo + (c & 1) * a.halo_bytes;
    i32x8 af[2], bf[2];
    int sa[2], sb[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const unsigned char* p = Hc + abase[mi] + toff;
      const i32x4 lo = *(const i32x4*)(p + 32 * lh), hi = *(const i32x4*)(p + 32 * lh + 16);
      af[mi] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      sa[mi] = p[64 + lh];
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const unsigned char* p = Bc + bbase + ni * 32 * PROW;
      const i32x4 lo = *(const i32x4*)(p + 32 * lh), hi = *(const i32x4*)(p + 32 * lh + 16);
      bf[ni] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      sb[ni] = p[64 + lh];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[mi], bf[ni], acc[mi][ni], 0, 0, 0, sa[mi], 0, sb[ni]);
    // keep the MFMAs HERE: hipcc otherwise sinks them below the staging branches and carries the 32 fragment registers across them (spills)
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    if (more && t == t_mid) { halo_store((c + 1) & 1, 0); halo_load((c + 1) * 64 + 32); }
    if (more && t == ntaps - 1) halo_store((c + 1) & 1, 1);
    if (q + 1 < total) b_store((q + 1) & 1, rs);
    __syncthreads();
    if (++t == ntaps) { t = 0; ++c; }
  };
  for (int q = 0; q < total; q += 2) {
    step(q, r0, r1);
    if (q + 1 < total) step(q + 1, r1, r0);
  }

  // ---- epilogue (as conv_halo_kernel): demod / bias / act in registers -> bf16 tile in LDS -> 16-byte coalesced stores --------
  constexpr int OROW = BN8 + 8;
  __bf16* ot = (__bf16*)smem;
  if (EPI != 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + k * 512;
      const int row = idx >> 4, vv = idx & 15;
      const int py = ty * HT8 + (row >> 4), px = tx * HT8 + (row & 15);
      const int n = n0 + vv * 8;
      bf16x8 rr = zero_bf16x8_();
      if (py < a.Hm && px < a.Wm && n < a.Cout) {
        const int oy = py * a.out_mul + (phase >> 1), ox = px * a.out_mul + (phase & 1);
        rr = (EPI == 2) ? *(const bf16x8*)(a.residual + ((size_t)(b * (a.Hout >> 1) + (oy >> 1)) * (a.Wout >> 1) + (ox >> 1)) * a.Cout + n)
                        : *(const bf16x8*)(a.residual + ((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + n);
      }
      *(bf16x8*)(ot + row * OROW + vv * 8) = rr;
    }
    __syncthreads();
  }
  constexpr float res_scale = EPI == 2 ? 0.25f : 1.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int nl = wn * 64 + ni * 32 + (lane & 31);
    const int n = n0 + nl;
    const float bv = (a.bias && n < a.N) ? a.bias[n] * a.bias_scale : 0.f;
    const float pv = (a.post && n < a.Cout) ? a.post[(size_t)b * a.Cout + n] : 1.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[mi][ni][r] * pv + bv;
        v = (a.act == ACT_LRELU ? (v > 0.f ? v : v * LRELU_SLOPE) : v) * a.gain;
        if (EPI != 0) v += res_scale * (float)ot[row * OROW + nl];
        ot[row * OROW + nl] = (__bf16)v;
      }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = tid + k * 512;
    const int row = idx >> 4, vv = idx & 15;
    const int py = ty * HT8 + (row >> 4), px = tx * HT8 + (row & 15);
    const int n = n0 + vv * 8;
    if (py >= a.Hm || px >= a.Wm || n >= a.Cout) continue;
    const int oy = py * a.out_mul + (phase >> 1), ox = px * a.out_mul + (phase & 1);
    *(bf16x8*)(a.y + ((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + n) = *(const bf16x8*)(ot + row * OROW + vv * 8);
  }
}

int launch_fp8(Fp8Args& a, int nphase, hipStream_t s) {
  const int in_mul = 1;
  int max_halo = 0;
  for (int p = 0; p < nphase; ++p) {
    int ymin = 99, ymax = -99, xmin = 99, xmax = -99;
    for (int t = 0; t < a.taps[p].n; ++t) {
      ymin = std::min(ymin, a.taps[p].dy[t]); ymax = std::max(ymax, a.taps[p].dy[t]);
      xmin = std::min(xmin, a.taps[p].dx[t]); xmax = std::max(xmax, a.taps[p].dx[t]);
    }
    a.hy0[p] = ymin; a.hx0[p] = xmin;
    a.hh[p] = (HT8 - 1) * in_mul + (ymax - ymin) + 1; a.hw[p] = (HT8 - 1) * in_mul + (xmax - xmin) + 1;
    max_halo = std::max(max_halo, a.hh[p] * a.hw[p]);
  }
  if (max_halo * 4 > 3 * 512) return LCGAN_EINVAL;
  a.tiles_x = cdiv(a.Wm, HT8); a.tiles_y = cdiv(a.Hm, HT8);
  a.halo_bytes = max_halo * PROW;
  const size_t smem = std::max((size_t)2 * a.halo_bytes + 2 * BTILE8 + (size_t)a.K64 * 64 * sizeof(float), (size_t)256 * (BN8 + 8) * sizeof(__bf16));
  dim3 grid(a.B * a.tiles_x * a.tiles_y, cdiv(a.Cout, BN8), nphase);
#define LAUNCH8(EP)                                                                                                     \
  {                                                                                                                     \
    static bool set = false;                                                                                            \
    if (!set) { hipFuncSetAttribute((const void*)conv_halo_fp8_kernel<EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
    hipLaunchKernelGGL((conv_halo_fp8_kernel<EP>), grid, dim3(512), smem, s, a);                                        \
  }
  if (a.residual && a.res_half) LAUNCH8(2) else if (a.residual) LAUNCH8(1) else LAUNCH8(0)
#undef LAUNCH8
  return launch_status();
}

}  // namespace

extern "C" {

// w [A][Bc][k][k] f32 (reference layout) * scale -> e4m3 wp [k*k][N][K64][64] + E8M0 wsc [k*k][N][K64][2]; N = transpose ? Bc : A,
// K64 = ceil((transpose ? A : Bc) / 64); layout of a 64-channel step: see the file header.
int lcgan_conv_weight_prep_fp8(const float* w, int A, int Bc, int k, float scale, int transpose, void* wp, void* wsc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (k != 1 && k != 3) return LCGAN_EINVAL;
  const int kk = k * k, N = transpose ? Bc : A, Kc = transpose ? A : Bc, K64 = (Kc + 63) / 64;
  ProfScope p(KID_WEIGHT_PREP, 0, (double)A * Bc * kk * 5, s);
  const size_t nblk = (size_t)kk * N * K64 * 2;
  hipLaunchKernelGGL(prep_weight_fp8_kernel, dim3((unsigned)((nblk + 127) / 128)), dim3(128), 0, s, w, A, Bc, kk, scale, transpose,
                     (unsigned char*)wp, (unsigned char*)wsc, N, Kc, K64);
  return launch_status();
}

// lcgan_conv_fwd with MX-fp8 operands (bf16 feature maps only; stride 1; Cin a multiple of 8; grids of at least 16 x 16 positions)
int lcgan_conv_fwd_fp8(const void* x, const void* wp, const void* wsc, void* y,
                       int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                       const float* pre, const float* post, const float* bias, float bias_scale,
                       int act, float gain, const void* residual, int residual_half, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((k != 1 && k != 3) || stride != 1 || (Cin & 7) || (Cout & 7) || N > Cout || act == ACT_TANH) return LCGAN_EINVAL;   // (stride 2 stays bf16)
  Fp8Args a = {};
  a.x = (const __bf16*)x; a.w = (const unsigned char*)wp; a.wsc = (const unsigned char*)wsc; a.y = (__bf16*)y;
  a.pre = pre; a.post = post; a.bias = bias; a.residual = (const __bf16*)residual; a.res_half = residual ? residual_half : 0;
  a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin;
  a.Hout = (Hin + stride - 1) / stride; a.Wout = (Win + stride - 1) / stride; a.Cout = Cout; a.Hm = a.Hout; a.Wm = a.Wout;
  if (a.Hm < HT8 || a.Wm < HT8 || (a.res_half && ((a.Hout | a.Wout) & 1))) return LCGAN_EINVAL;
  if ((long long)B * Hin * Win * Cin >= (1ll << 31) || (long long)B * a.Hout * a.Wout * Cout >= (1ll << 31)) return LCGAN_EINVAL;
  a.N = N; a.K64 = (Cin + 63) / 64; a.out_mul = 1; a.bias_scale = bias_scale; a.gain = gain; a.act = act;
  const int pad = k / 2;
  TapTable8& t = a.taps[0];
  t.n = k * k;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) { const int i = ky * k + kx; t.dy[i] = ky - pad; t.dx[i] = kx - pad; t.wt[i] = i; }
  char tag[96] = "";
  if (lcgan_prof_active()) snprintf(tag, sizeof(tag), "fp8 fwd B%d %dx%d C%d->%d k%d s%d%s", B, Hin, Win, Cin, N, k, stride, pre ? " mod" : "");
  ProfScope p(KID_CONV_IGEMM, 2.0 * B * a.Hm * a.Wm * N * (double)Cin * k * k, 0, s, tag);
  return launch_fp8(a, 1, s);
}

// lcgan_conv_bwd_data with MX-fp8 operands (stride 2: the 4-phase x2 transposed convolution)
int lcgan_conv_bwd_data_fp8(const void* g, const void* wpT, const void* wscT, void* gx,
                            int B, int Hg, int Wg, int Cg, int Cout, int N, int k, int stride,
                            const float* pre, const float* post, const float* bias, float bias_scale,
                            int act, float gain, const void* residual, int residual_half, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || (Cg & 7) || (Cout & 7) || N > Cout || act == ACT_TANH) return LCGAN_EINVAL;
  if (stride == 2 && k != 3) return LCGAN_EINVAL;
  Fp8Args a = {};
  a.x = (const __bf16*)g; a.w = (const unsigned char*)wpT; a.wsc = (const unsigned char*)wscT; a.y = (__bf16*)gx;
  a.pre = pre; a.post = post; a.bias = bias; a.residual = (const __bf16*)residual; a.res_half = residual ? residual_half : 0;
  a.B = B; a.Hin = Hg; a.Win = Wg; a.Cin = Cg; a.Hout = Hg * stride; a.Wout = Wg * stride; a.Cout = Cout; a.Hm = Hg; a.Wm = Wg;
  if (a.Hm < HT8 || a.Wm < HT8 || (a.res_half && ((a.Hout | a.Wout) & 1))) return LCGAN_EINVAL;
  if ((long long)B * Hg * Wg * Cg >= (1ll << 31) || (long long)B * a.Hout * a.Wout * Cout >= (1ll << 31)) return LCGAN_EINVAL;
  a.N = N; a.K64 = (Cg + 63) / 64; a.out_mul = stride; a.bias_scale = bias_scale; a.gain = gain; a.act = act;
  const int pad = k / 2;
  int nphase = 1;
  double taps_total = k * k;
  if (stride == 1) {
    TapTable8& t = a.taps[0];
    t.n = k * k;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) { const int i = ky * k + kx; t.dy[i] = pad - ky; t.dx[i] = pad - kx; t.wt[i] = i; }
  } else {
    nphase = 4;
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        TapTable8& t = a.taps[ph * 2 + pw];
        t.n = 0;
        for (int ky = 0; ky < 3; ++ky) {
          if ((ph + 1 - ky) & 1) continue;
          for (int kx = 0; kx < 3; ++kx) {
            if ((pw + 1 - kx) & 1) continue;
            t.dy[t.n] = (ph + 1 - ky) / 2; t.dx[t.n] = (pw + 1 - kx) / 2; t.wt[t.n] = ky * 3 + kx; ++t.n;
          }
        }
      }
    taps_total = 9.0 / 4.0;
  }
  char tag[96] = "";
  if (lcgan_prof_active()) snprintf(tag, sizeof(tag), "fp8 dgrad B%d %dx%d C%d->%d k%d s%d%s", B, Hg, Wg, Cg, N, k, stride, pre ? " mod" : "");
  ProfScope p(KID_CONV_IGEMM, 2.0 * (double)B * a.Hout * a.Wout * N * Cg * taps_total, 0, s, tag);
  return launch_fp8(a, nphase, s);
}

}  // extern "C"
