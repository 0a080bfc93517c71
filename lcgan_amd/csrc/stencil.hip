// NHWC stencil / elementwise / gather kernels of the LC-GAN step for gfx950.  All are HBM-bound: every thread
// moves one 8-channel vector (16 B of bf16 / 32 B of f32), consecutive lanes walk consecutive channel vectors of
// a pixel, so every wave access is a run of whole 16-byte lanes.
//
// Reference ops replaced (file:line in /root/reference):
//   box3_act        F.avg_pool2d(3,1,1) [+ leaky_relu*gain | tanh]     custom_layers.py:136-138,150-155,196-206
//   up2box          F.interpolate(x2 nearest) + box filter             custom_layers.py:146-147
//   avgpool2        F.avg_pool2d(2,2)                                  custom_layers.py:202
//   act_bwd_reduce  leaky_relu backward + bias / demod-statistic sums  (autograd of :85, :155, :158, :205, :208)
//   scale_reduce    style gradient  sum_p x * u                        (autograd of :62-64)
//   warp            F.grid_sample(bicubic, zeros, align_corners=False) custom_layers.py:127-134,162-165 (+ grid_sampler backward)
//   mbstd           MinibatchStdLayer                                  custom_layers.py:243-256
//   rgb_*           1x1 convs touching the 3-channel NCHW image        cnn.py:20 ; custom_layers.py:175,181
#include "common.h"
#include <cstdio>

namespace {

constexpr int TPB = 256;

static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + TPB - 1) / TPB)); }

// per-launch tag of the profiling table (bench.py --launch-table): kernel name + shape; formatted only while profiling is on
struct Tag {
  char s[96];
  Tag(const char* name, int B, int H, int W, int C) {
    s[0] = 0;
    if (lcgan_prof_active()) snprintf(s, sizeof(s), "%s B%d %dx%d C%d", name, B, H, W, C);
  }
};

// ------------------------------------------------------------------------------------------------------------
// box filter (3x3 mean, zero padding, always /9) fused with activation
// ------------------------------------------------------------------------------------------------------------
// Separable sliding window: a thread owns one (column w, 8-channel vector) and walks BOX_RH consecutive rows, keeping the
// last three horizontal 3-sums in registers: 3 vector loads per output instead of 9 (the 9-tap version was L1-bound, 2.3x
// off the HBM roofline).
// Rows per thread: 16 on the big maps (the two halo rows of a strip are re-read from L2: 25 % extra loads at 8 rows, 12 % at 16;
// -3..-8 % at >= 64 x 64 x 512, batch 32), 8 below 4M channel vectors, where 16 leaves too few threads (+30 % at 32 x 32 x 512).
// Thread index -> (vector v, column x, row y, sample b) of a [B][Hh][Ww][nvec] grid, gid = ((b Hh + y) Ww + x) nvec + v.  As four 64-bit
// divisions this was the larger part of the instructions of the one-output-vector-per-thread kernels (up2box, the pooling kernels); where
// the three extents are powers of two and the grid has fewer than 2^31 vectors -- every launch of the networks -- it is shifts and masks.
__device__ __forceinline__ void decode_vxyb(long long gid, long long total, int nvec, int Ww, int Hh, int& v, int& x, int& y, int& b) {
  const bool p2 = total < (1ll << 31) && !(nvec & (nvec - 1)) && !(Ww & (Ww - 1)) && !(Hh & (Hh - 1));      // (kernel-uniform)
  if (p2) {
    unsigned g = (unsigned)gid;
    v = (int)(g & (unsigned)(nvec - 1)); g >>= __builtin_ctz((unsigned)nvec);
    x = (int)(g & (unsigned)(Ww - 1)); g >>= __builtin_ctz((unsigned)Ww);
    y = (int)(g & (unsigned)(Hh - 1)); b = (int)(g >> __builtin_ctz((unsigned)Hh));
  } else {
    v = (int)(gid % nvec);
    long long t = gid / nvec;
    x = (int)(t % Ww); t /= Ww;
    y = (int)(t % Hh); b = (int)(t / Hh);
  }
}

int box_rh(long long nvectors) { return nvectors >= (1ll << 22) ? 16 : 8; }
template <typename T>
__global__ void box3_act_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int act, float gain,
                                int BOX_RH) {
  const int nvec = C >> 3;
  const int strips = (H + BOX_RH - 1) / BOX_RH;
  const long long total = (long long)B * strips * W * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  int v, w, strip, b;
  decode_vxyb(gid, total, nvec, W, strips, v, w, strip, b);
  const int h0 = strip * BOX_RH, h1 = min(h0 + BOX_RH, H);
  const T* xb = x + (size_t)b * H * W * C + v * 8;
  // branch-free taps: the column neighbours are clamped and masked once per thread, a row outside the image contributes zero
  const int xl = max(w - 1, 0) * C, xc = w * C, xr = min(w + 1, W - 1) * C;
  const float ml = w > 0 ? 1.f : 0.f, mr = w + 1 < W ? 1.f : 0.f;
  auto rowsum = [&](int hh) {                          // (rows too: a row outside the image is read clamped and scaled by 0 -- a branch
    F8 s;                                              //  around the loads kept the next row's loads from issuing under this row's)
    const float mh = (unsigned)hh < (unsigned)H ? 1.f : 0.f;
    const T* row = xb + (size_t)min(max(hh, 0), H - 1) * W * C;
    const F8 a = Feat<T>::load(row + xl), c = Feat<T>::load(row + xc), d = Feat<T>::load(row + xr);
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = (a.v[j] * ml + c.v[j] + d.v[j] * mr) * mh;
    return s;
  };
  F8 r0 = rowsum(h0 - 1), r1 = rowsum(h0);
  for (int hh = h0; hh < h1; ++hh) {
    const F8 r2 = rowsum(hh + 1);
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = act_fwd((r0.v[j] + r1.v[j] + r2.v[j]) * (1.f / 9.f), act) * gain;
    Feat<T>::store(y + (((size_t)b * H + hh) * W + w) * C + v * 8, o);
    r0 = r1; r1 = r2;
  }
}

// gz = box3(gy) * act'(y) ; gbias[c] += sum_{b,p} gz     -- the backward of  y = act(conv + bias) -> box3(y)  up to the conv's
// pre-activation (DiscriminatorBlock conv0 -> blur, custom_layers.py:204-206): one pass instead of a box-filter pass plus an
// activation-backward pass.  Same sliding window as box3_act_kernel, over taller strips (BOXB_RH rows) so that the per-channel
// bias reduction ends in few global atomics (LDS float atomics inside the block).
// Strip height: see lcgan_box3_actbwd_reduce (taller strips = fewer same-address atomics, until too few threads are left).
// mask (optional, instead of y): the activation's sign bits as a convolution epilogue leaves them (lcgan_conv_fwd_m): byte v of a pixel's
// C / 8 mask bytes holds channels 8 v .. 8 v + 7 -- 1/16 of the bytes of y
__device__ __forceinline__ float lrelu_grad_bit(unsigned bits, int j, float gain) { return ((bits >> j) & 1u) ? gain : gain * LRELU_SLOPE; }

template <typename T>
__global__ void box3_actbwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ y, const unsigned char* __restrict__ mask, T* __restrict__ gz,
                                          float* __restrict__ gbias, int B, int H, int W, int C, int Clog, int act, float gain,
                                          int BOXB_RH) {
  extern __shared__ float red[];                       // [C]
  const int nvec = C >> 3;
  const int strips = (H + BOXB_RH - 1) / BOXB_RH;
  const long long total = (long long)B * strips * W * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gbias) {
    for (int c = threadIdx.x; c < C; c += TPB) red[c] = 0.f;
    __syncthreads();
  }
  float sb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) sb[j] = 0.f;
  int v = 0;
  if (gid < total) {
    int w, strip, b;
    decode_vxyb(gid, total, nvec, W, strips, v, w, strip, b);
    const int h0 = strip * BOXB_RH, h1 = min(h0 + BOXB_RH, H);
    const T* gb = gy + (size_t)b * H * W * C + v * 8;
    const int xl = max(w - 1, 0) * C, xc = w * C, xr = min(w + 1, W - 1) * C;     // branch-free column taps (see box3_act_kernel)
    const float ml = w > 0 ? 1.f : 0.f, mr = w + 1 < W ? 1.f : 0.f;
    auto rowsum = [&](int hh) {
      F8 s;
      const float mh = (unsigned)hh < (unsigned)H ? 1.f : 0.f;
      const T* row = gb + (size_t)min(max(hh, 0), H - 1) * W * C;
      const F8 a = Feat<T>::load(row + xl), c = Feat<T>::load(row + xc), d = Feat<T>::load(row + xr);
#pragma unroll
      for (int j = 0; j < 8; ++j) s.v[j] = (a.v[j] * ml + c.v[j] + d.v[j] * mr) * mh;
      return s;
    };
    F8 r0 = rowsum(h0 - 1), r1 = rowsum(h0);
    for (int hh = h0; hh < h1; ++hh) {
      const F8 r2 = rowsum(hh + 1);
      const size_t off = (((size_t)b * H + hh) * W + w) * C + v * 8;
      F8 o;
      if (mask) {
        const unsigned bits = mask[off >> 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o.v[j] = (r0.v[j] + r1.v[j] + r2.v[j]) * (1.f / 9.f) * lrelu_grad_bit(bits, j, gain);
          sb[j] += o.v[j];
        }
      } else {
        const F8 yo = Feat<T>::load(y + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o.v[j] = (r0.v[j] + r1.v[j] + r2.v[j]) * (1.f / 9.f) * act_grad_from_out(yo.v[j], act, gain);
          sb[j] += o.v[j];
        }
      }
      Feat<T>::store(gz + off, o);
      r0 = r1; r1 = r2;
    }
  }
  if (!gbias) return;
  if (gid < total) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&red[v * 8 + j], sb[j]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Clog; c += TPB) atomicAdd(gbias + c, red[c]);
}

// gx = box3(gy * act'(y))   (box3 is self-adjoint).  Same separable sliding window as box3_act_kernel: a thread owns one
// (column, 8-channel vector), walks BOX_RH rows and keeps the last three horizontal 3-sums of gy*act'(y) in registers
// (6 vector loads per output instead of 18: the 9-tap version ran 2.2x off the HBM roofline).
// A wave holds 64 / nvec whole columns of one row strip (lanes [k nvec, (k+1) nvec) = column w0 + k), so for C <= 256 a thread
// forms gy * act'(y) for its OWN column only and takes the neighbours' from the lanes nvec away; only the wave's first / last
// column still load their outer neighbour: 2 + 2 * (2 nvec / 64) vector loads per output row instead of 6 (the kernel is bound by
// load issue, not by HBM: 3.7 TB/s against the 5 TB/s of the 2-load activation backward).
template <typename T, int ACT>                                     // ACT: compile-time activation, -1 = the run-time argument (see rgb_expand_kernel)
__global__ void box3_act_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ gx,
                                    int B, int H, int W, int C, int act_rt, float gain, int BOX_RH) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int nvec = C >> 3;
  const int strips = (H + BOX_RH - 1) / BOX_RH;
  const long long total = (long long)B * strips * W * nvec;
  const long long gid0 = (long long)blockIdx.x * TPB + threadIdx.x;
  const bool live = gid0 < total;
  const long long gid = live ? gid0 : total - 1;                     // (dead lanes of the last wave stay in step for the shuffles)
  int v, w, strip, b;
  decode_vxyb(gid, total, nvec, W, strips, v, w, strip, b);
  const int h0 = strip * BOX_RH, h1 = min(h0 + BOX_RH, H);
  const size_t base = (size_t)b * H * W * C + v * 8;
  const int xo[3] = {max(w - 1, 0), w, min(w + 1, W - 1)};          // branch-free column taps (see box3_act_kernel)
  const float mk[3] = {w > 0 ? 1.f : 0.f, 1.f, w + 1 < W ? 1.f : 0.f};
  const bool share = nvec <= 32 && (64 % nvec) == 0;                // uniform
  const int lane = threadIdx.x & 63;
  const bool first_col = lane < nvec, last_col = lane >= 64 - nvec;
  auto term = [&](int hc, int col, float m) {                       // gy * act'(y) * m at (hc, col)
    const size_t off = base + ((size_t)hc * W + col) * C;
    const F8 g = Feat<T>::load(gy + off);
    F8 r;
    if (act == ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = g.v[j] * (gain * m);
    } else {
      const F8 yo = Feat<T>::load(y + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = g.v[j] * (act_grad_from_out(yo.v[j], act, gain) * m);
    }
    return r;
  };
  auto rowsum = [&](int hh) {
    const float mh = (unsigned)hh < (unsigned)H ? 1.f : 0.f;        // branch-free rows as well (see box3_act_kernel)
    const int hc = min(max(hh, 0), H - 1);
    const F8 c = term(hc, xo[1], mh);
    F8 l, r, s;
    if (share) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { l.v[j] = __shfl_up(c.v[j], nvec, 64); r.v[j] = __shfl_down(c.v[j], nvec, 64); }
      if (first_col) l = term(hc, xo[0], mh);
      if (last_col) r = term(hc, xo[2], mh);
#pragma unroll
      for (int j = 0; j < 8; ++j) s.v[j] = (mk[0] != 0.f ? l.v[j] : 0.f) + c.v[j] + (mk[2] != 0.f ? r.v[j] : 0.f);
    } else {
      l = term(hc, xo[0], mk[0] * mh); r = term(hc, xo[2], mk[2] * mh);
#pragma unroll
      for (int j = 0; j < 8; ++j) s.v[j] = l.v[j] + c.v[j] + r.v[j];
    }
    return s;
  };
  F8 r0 = rowsum(h0 - 1), r1 = rowsum(h0);
  for (int hh = h0; hh < h1; ++hh) {
    const F8 r2 = rowsum(hh + 1);
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = (r0.v[j] + r1.v[j] + r2.v[j]) * (1.f / 9.f);
    if (live) Feat<T>::store(gx + base + ((size_t)hh * W + w) * C, o);
    r0 = r1; r1 = r2;
  }
}

// ------------------------------------------------------------------------------------------------------------
// nearest x2 upsample followed by the box filter == separable taps {1,2}/3 ; fused residual add
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void up2box_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                              int B, int H, int W, int C) {   // x: [B,H,W,C] -> y: [B,2H,2W,C]
  const int nvec = C >> 3, H2 = 2 * H, W2 = 2 * W;
  const long long total = (long long)B * H2 * W2 * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  int v, X, Y, b;
  decode_vxyb(gid, total, nvec, W2, H2, v, X, Y, b);
  // output row Y=2i: (x[i-1] + 2 x[i]) / 3 ; Y=2i+1: (2 x[i] + x[i+1]) / 3
  const int i = Y >> 1, j = X >> 1;
  const int i2 = (Y & 1) ? i + 1 : i - 1, j2 = (X & 1) ? j + 1 : j - 1;
  // branch-free taps: the neighbour row / column is clamped and weighted 0 outside the image
  const float wy2 = ((unsigned)i2 < (unsigned)H) ? (1.f / 3.f) : 0.f, wx2 = ((unsigned)j2 < (unsigned)W) ? (1.f / 3.f) : 0.f;
  const int i2c = min(max(i2, 0), H - 1), j2c = min(max(j2, 0), W - 1);
  const T* xb = x + (size_t)b * H * W * C + v * 8;
  const F8 t00 = Feat<T>::load(xb + ((size_t)i * W + j) * C), t01 = Feat<T>::load(xb + ((size_t)i * W + j2c) * C);
  const F8 t10 = Feat<T>::load(xb + ((size_t)i2c * W + j) * C), t11 = Feat<T>::load(xb + ((size_t)i2c * W + j2c) * C);
  F8 s;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    s.v[q] = (2.f / 3.f) * ((2.f / 3.f) * t00.v[q] + wx2 * t01.v[q]) + wy2 * ((2.f / 3.f) * t10.v[q] + wx2 * t11.v[q]);
  const size_t off = (size_t)gid * 8;
  if (res) {
    const F8 r = Feat<T>::load(res + off);
#pragma unroll
    for (int q = 0; q < 8; ++q) s.v[q] += r.v[q];
  }
  Feat<T>::store(y + off, s);
}

template <typename T>
__global__ void up2box_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int C) {
  // gy: [B,2H,2W,C] -> gx: [B,H,W,C];  x[i] feeds rows 2i-1 (1/3), 2i (2/3), 2i+1 (2/3), 2i+2 (1/3)
  const int nvec = C >> 3, H2 = 2 * H, W2 = 2 * W;
  const long long total = (long long)B * H * W * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  int v, j, i, b;
  decode_vxyb(gid, total, nvec, W, H, v, j, i, b);
  F8 s = f8_zero();
#pragma unroll
  for (int a = -1; a <= 2; ++a) {
    const int Y = 2 * i + a;
    if ((unsigned)Y >= (unsigned)H2) continue;
    const float wy = (a == 0 || a == 1) ? (2.f / 3.f) : (1.f / 3.f);
#pragma unroll
    for (int c = -1; c <= 2; ++c) {
      const int X = 2 * j + c;
      if ((unsigned)X >= (unsigned)W2) continue;
      const float wgt = wy * ((c == 0 || c == 1) ? (2.f / 3.f) : (1.f / 3.f));
      const F8 t = Feat<T>::load(gy + (((size_t)b * H2 + Y) * W2 + X) * C + v * 8);
#pragma unroll
      for (int q = 0; q < 8; ++q) s.v[q] += t.v[q] * wgt;
    }
  }
  Feat<T>::store(gx + (size_t)gid * 8, s);
}

// ------------------------------------------------------------------------------------------------------------
// 2x2 average pool and its adjoint
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void avgpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {  // y: [B,H/2,W/2,C]
  const int nvec = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * Ho * Wo * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  int v, j, i, b;
  decode_vxyb(gid, total, nvec, Wo, Ho, v, j, i, b);
  F8 s = f8_zero();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const F8 t = Feat<T>::load(x + (((size_t)b * H + 2 * i + a) * W + 2 * j + c) * C + v * 8);
#pragma unroll
      for (int q = 0; q < 8; ++q) s.v[q] += t.v[q];
    }
#pragma unroll
  for (int q = 0; q < 8; ++q) s.v[q] *= 0.25f;
  Feat<T>::store(y + (size_t)gid * 8, s);
}

template <typename T>
__global__ void avgpool2_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int C) {  // gx: [B,H,W,C]
  const int nvec = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * H * W * nvec;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  int v, X, Y, b;
  decode_vxyb(gid, total, nvec, W, H, v, X, Y, b);
  F8 t = Feat<T>::load(gy + (((size_t)b * Ho + (Y >> 1)) * Wo + (X >> 1)) * C + v * 8);
#pragma unroll
  for (int q = 0; q < 8; ++q) t.v[q] *= 0.25f;
  Feat<T>::store(gx + (size_t)gid * 8, t);
}

// ------------------------------------------------------------------------------------------------------------
// activation backward fused with the per-channel reductions that consume the same bytes:
//   gz        = gy * act'(y)                                          (written when gz != nullptr)
//   gbias[c] += sum_{b,p} gz                                           (optional)
//   gdq[b,c] += sum_p gz * (ypre - bias[c]*bias_scale),  ypre = act^-1(y / gain)   (optional: demod gradient)
// One block = P consecutive pixels of ONE sample; a thread keeps one channel vector in registers.
// ------------------------------------------------------------------------------------------------------------
// GDQ: the demodulation statistic is wanted as well (modulated convolutions only); the plain case carries neither its second
// accumulator set nor its arithmetic (128 -> fewer registers, one more wave per SIMD).
template <typename T, bool GDQ>
__global__ void act_bwd_reduce_kernel(const T* __restrict__ gy, const T* __restrict__ y, const unsigned char* __restrict__ mask, T* __restrict__ gz,
                                      const float* __restrict__ oscale, const float* __restrict__ bias, float bias_scale,
                                      float* __restrict__ gbias, float* __restrict__ gdq,
                                      int HW, int C, int Clog, int act, float gain, int P) {
  __shared__ float red[GDQ ? 2 : 1][TPB * 8];
  const int nvec = C >> 3;
  const int groups = TPB / nvec;                       // pixel groups running in parallel (>= 1; nvec <= 256)
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * P, p1 = min(p0 + P, HW);
  const bool active = grp < groups;
  float sb[8], sq[8], bv[8], ov[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sb[j] = 0.f; sq[j] = 0.f; bv[j] = 0.f; ov[j] = 1.f; }
  if (GDQ && active && bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (v * 8 + j < Clog) ? bias[v * 8 + j] * bias_scale : 0.f;
  }
  // oscale [B][C] (GDQ instantiation): the STORED gradient is gz * oscale[b, c] -- a modulated convolution's data- and weight-gradient
  // launches both consume d[b, o] * gz, which they otherwise form per sample while staging -- the reductions see the unscaled gz
  if (GDQ && active && oscale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = oscale[(size_t)b * C + v * 8 + j];
  }
  if (active) {
    const bool need_y = act != ACT_NONE || GDQ;
    const float inv_gain = 1.f / gain;
    auto one = [&](size_t off, const F8& g, const F8& yo) {
      F8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        z.v[j] = g.v[j] * act_grad_from_out(yo.v[j], act, gain);
        sb[j] += z.v[j];
        if (GDQ) {
          float t = yo.v[j] * inv_gain;
          if (act == ACT_LRELU && t < 0.f) t *= (1.f / LRELU_SLOPE);
          sq[j] += z.v[j] * (t - bv[j]);
          z.v[j] *= ov[j];
        }
      }
      if (gz) Feat<T>::store(gz + off, z);
    };
    // the same with the activation's sign bits instead of y (leaky ReLU, no demodulation statistic): 1 mask byte per 16 bytes of gy
    auto one_m = [&](size_t off, const F8& g, unsigned bits) {
      F8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) { z.v[j] = g.v[j] * lrelu_grad_bit(bits, j, gain); sb[j] += z.v[j]; }
      if (gz) Feat<T>::store(gz + off, z);
    };
    // four pixels per trip: all eight loads are issued before the first is used (a thread otherwise has one pixel in flight)
    int p = p0 + grp;
    if (!GDQ && mask) {
      for (; p + 3 * groups < p1; p += 4 * groups) {
        size_t off[4]; F8 g[4]; unsigned bits[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          off[u] = ((size_t)b * HW + p + u * groups) * C + v * 8;
          g[u] = Feat<T>::load(gy + off[u]);
          bits[u] = mask[off[u] >> 3];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one_m(off[u], g[u], bits[u]);
      }
      for (; p < p1; p += groups) {
        const size_t off = ((size_t)b * HW + p) * C + v * 8;
        one_m(off, Feat<T>::load(gy + off), mask[off >> 3]);
      }
    } else {
    for (; p + 3 * groups < p1; p += 4 * groups) {
      size_t off[4]; F8 g[4], yo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        off[u] = ((size_t)b * HW + p + u * groups) * C + v * 8;
        g[u] = Feat<T>::load(gy + off[u]);
        yo[u] = need_y ? Feat<T>::load(y + off[u]) : f8_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(off[u], g[u], yo[u]);
    }
    for (; p < p1; p += groups) {
      const size_t off = ((size_t)b * HW + p) * C + v * 8;
      const F8 g = Feat<T>::load(gy + off);
      const F8 yo = need_y ? Feat<T>::load(y + off) : f8_zero();
      one(off, g, yo);
    }
    }
  }
  if (!gbias && !GDQ) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][threadIdx.x * 8 + j] = sb[j]; if (GDQ) red[1][threadIdx.x * 8 + j] = sq[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TPB) {
    const int vv = c >> 3, jj = c & 7;
    float tb = 0.f, tq = 0.f;
    for (int gI = 0; gI < groups; ++gI) { const int t = gI * nvec + vv; tb += red[0][t * 8 + jj]; if (GDQ) tq += red[1][t * 8 + jj]; }
    if (gbias && c < Clog) atomicAdd(gbias + c, tb);
    if (GDQ) atomicAdd(gdq + (size_t)b * C + c, tq);
  }
}

// ------------------------------------------------------------------------------------------------------------
// style-gradient reduction fused with the modulation scale of the data gradient:
//   gs[b,c] += sum_p x[b,p,c] * u[b,p,c] ;   u[b,p,c] <- s[b,c] * u[b,p,c]   (in place)
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void scale_reduce_kernel(T* __restrict__ u, const T* __restrict__ x, const float* __restrict__ s,
                                    float* __restrict__ gs, const T* __restrict__ res, int HW, int C, int P) {
  __shared__ float red[TPB * 8];
  const int nvec = C >> 3;
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * P, p1 = min(p0 + P, HW);
  const bool active = grp < groups;
  float acc[8], sv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = 0.f; sv[j] = 0.f; }
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sv[j] = s[(size_t)b * C + v * 8 + j];
    for (int p = p0 + grp; p < p1; p += groups) {
      const size_t off = ((size_t)b * HW + p) * C + v * 8;
      F8 uu = Feat<T>::load(u + off);
      const F8 xx = Feat<T>::load(x + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[j] += uu.v[j] * xx.v[j]; uu.v[j] *= sv[j]; }
      if (res) {                                                   // u <- s * u + res (the gradient of another consumer of the same tensor)
        const F8 rr = Feat<T>::load(res + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) uu.v[j] = Feat<T>::rnd(uu.v[j]) + rr.v[j];
      }
      Feat<T>::store(u + off, uu);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TPB) {
    const int vv = c >> 3, jj = c & 7;
    float t = 0.f;
    for (int gI = 0; gI < groups; ++gI) t += red[(gI * nvec + vv) * 8 + jj];
    atomicAdd(gs + (size_t)b * C + c, t);
  }
}

// ------------------------------------------------------------------------------------------------------------
// bicubic feature warp (grid_sample: bicubic, zero padding, align_corners=False sampling of a grid whose base
// coordinates use the align_corners=True formula -- custom_layers.py:127-134,162-165)
// ------------------------------------------------------------------------------------------------------------
constexpr float CUBIC_A = -0.75f;
__device__ __forceinline__ float cc1(float x) { return ((CUBIC_A + 2.f) * x - (CUBIC_A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((CUBIC_A * x - 5.f * CUBIC_A) * x + 8.f * CUBIC_A) * x - 4.f * CUBIC_A; }
__device__ __forceinline__ float dcc1(float x) { return (3.f * (CUBIC_A + 2.f) * x - 2.f * (CUBIC_A + 3.f)) * x; }
__device__ __forceinline__ float dcc2(float x) { return (3.f * CUBIC_A * x - 10.f * CUBIC_A) * x + 8.f * CUBIC_A; }
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  c[0] = cc2(t + 1.f); c[1] = cc1(t); c[2] = cc1(1.f - t); c[3] = cc2(2.f - t);
}
__device__ __forceinline__ void cubic_dcoeffs(float t, float d[4]) {   // d c[k] / d t
  d[0] = dcc2(t + 1.f); d[1] = dcc1(t); d[2] = -dcc1(1.f - t); d[3] = -dcc2(2.f - t);
}
template <typename T>
__device__ __forceinline__ void warp_coords(const T* flow, size_t pix, int h, int w, int H, int W, float scale,
                                            float& ix, float& iy) {
  const float fx = Feat<T>::ld1(flow + pix * 8), fy = Feat<T>::ld1(flow + pix * 8 + 1);
  const float gx = 2.f * (float)w / (float)(W - 1) - 1.f + fx * scale;
  const float gy = 2.f * (float)h / (float)(H - 1) - 1.f + fy * scale;
  ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
}

// Branch-free taps: out-of-range taps are clamped to a valid address and given weight 0, so the 16 loads issue back to back
// (the version with a bounds branch per tap spent ~70 % of its 1300 instructions on address arithmetic and exec-mask branches).
// Round 3: the kernel was vector-ALU-bound, not memory-bound (1 150 instructions per output vector, 760 of them VALU, of which the
// 16 x (unpack + 8 FMAs) are 200): taps are raw buffer loads (scalar resource + one 32-bit byte offset = row part + column part),
// index decompositions are shifts for the power-of-two sizes, products are 24-bit multiplies.
// (Walking the pixels of a block as 4 x 4 tiles instead of a run of one image row -- a 7 x 7 instead of a 4 x 19 bicubic footprint per 16
// output pixels -- was measured: 0 ... +12 % SLOWER on the forward kernel, -4 % on the backward; not kept.)
// (Two ways of doing the per-pixel part -- flow, sample point, coefficients, clamped offsets: ~200 of the 475 vector instructions -- once
// per pixel instead of once per lane were measured too and not kept.  A thread owning four of the pixel's vectors: 1.66 -> 2.98 ms per
// iteration for the forward launches, every load instruction then touching 16 half cache lines instead of 8 whole ones.  A wave doing
// the pixel work for 64 pixels, parking it in LDS and walking the pixels with the lane layout of this kernel: 1.70 -> 2.29 ms, the
// 16 dependent tap loads of a step no longer overlapping with another pixel's.  The kernel is bound by its tap traffic through the L1 /
// texture path at this occupancy, not by its instruction count.)
struct WarpDims {
  FastDiv nvec, W, H;
  __device__ __forceinline__ void pixel(unsigned pix, unsigned& b, unsigned& h, unsigned& w) const {
    w = W.mod(pix); const unsigned t1 = W.div(pix); h = H.mod(t1); b = H.div(t1);
  }
};
static inline WarpDims warp_dims(int C, int H, int W) {
  WarpDims dm; dm.nvec = FastDiv((unsigned)(C / 8)); dm.W = FastDiv((unsigned)W); dm.H = FastDiv((unsigned)H);
  return dm;
}
template <typename T>
__global__ __launch_bounds__(256) void warp_fwd_kernel(const T* __restrict__ x, const T* __restrict__ flow, T* __restrict__ y,
                                                       int B, int H, int W, int C, float scale, WarpDims dm) {
  const unsigned total = (unsigned)B * H * W * (C >> 3);
  const unsigned gid = blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const unsigned v = dm.nvec.mod(gid), tp = dm.nvec.div(gid);
  unsigned b, h, w;
  dm.pixel(tp, b, h, w);
  const unsigned pix = (b * (unsigned)H + h) * (unsigned)W + w;
  float ix, iy;
  warp_coords<T>(flow, (size_t)pix, (int)h, (int)w, H, W, scale, ix, iy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  float cx[4], cy[4];
  cubic_coeffs(ix - fx0, cx);
  cubic_coeffs(iy - fy0, cy);
  const int x0 = (int)fx0 - 1, y0 = (int)fy0 - 1;
  constexpr unsigned ES = sizeof(T);
  const unsigned pitch = (unsigned)C * ES;                       // bytes per pixel
  const __amdgpu_buffer_rsrc_t xr = buf_rsrc(x, (unsigned)B * H * W * pitch);
  unsigned xo[4], yo[4];                                       // byte offsets of the 4 tap columns / rows (clamped), weights zeroed outside
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int xx = x0 + j;
    if ((unsigned)xx >= (unsigned)W) cx[j] = 0.f;
    xo[j] = (unsigned)__umul24((unsigned)min(max(xx, 0), W - 1), pitch) + v * (8 * ES);
  }
  const unsigned rowpitch = (unsigned)W * pitch, b0 = b * (unsigned)H;          // (W * C * element size < 2^24, checked by the launcher)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = y0 + i;
    if ((unsigned)yy >= (unsigned)H) cy[i] = 0.f;
    yo[i] = (unsigned)__umul24(b0 + (unsigned)min(max(yy, 0), H - 1), rowpitch);
  }
  F8 s = f8_zero();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float wgt = cy[i] * cx[j];
      const F8 t = buf_load8<T>(xr, yo[i] + xo[j]);
#pragma unroll
      for (int q = 0; q < 8; ++q) s.v[q] += t.v[q] * wgt;
    }
  }
  Feat<T>::store(y + ((size_t)pix * (C >> 3) + v) * 8, s);
}

// ---- backward of the warp ------------------------------------------------------------------------------------
// The scatter  gx[q] += wgt * gy[p]  (16 taps per output pixel p) has the same (p -> q, wgt) pattern for all C channels,
// so it is transposed ONCE per launch at pixel level into an exact CSR structure and then executed as a gather:
//   warp_bwd_grid_kernel   gflow[p] = d/d(grid): forward-like gather of x around the sample point; its v == 0 lanes also
//                          COUNT the taps that land on every input pixel (integer atomics)
//   warp_scan_*            exclusive prefix sum of the counts -> list offsets (three small kernels)
//   warp_fill_kernel       one thread per output pixel: entries[offs[q] + slot] = (p, wgt), slot from counting cnt[q] down
//   warp_gather_kernel     one thread per (input pixel q, 8-channel vector): gx[q] = sum over its list of wgt * gy[p]
// This replaces 16 float atomics per element (bounded by the ~1.3 TB/s chip-wide float-atomic rate) by 32 integer
// atomics per PIXEL plus coalesced 16-byte gathers, and it is exact for ANY flow field: the first version used fixed
// 32-entry lists with a global overflow list, which silently dropped entries beyond its capacity and went quadratic
// (0.85 s per launch) on flows that compress an area by more than 2x.
struct WarpEntry { int p; float w; };
constexpr int SCAN_TILE = 1024;                              // elements per scan block (256 threads x 4)

template <typename T>
__global__ __launch_bounds__(256) void warp_bwd_grid_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ flow,
                                                            T* __restrict__ gflow, int* __restrict__ cnt, int B, int H, int W, int C, float scale,
                                                            WarpDims dm) {
  const int nvec = C >> 3;                                   // power of two, <= 64 (checked by the launcher)
  const unsigned total = (unsigned)B * H * W * nvec;
  const unsigned gid0 = blockIdx.x * TPB + threadIdx.x;
  const bool live = gid0 < total;
  const unsigned gg = live ? gid0 : total - 1;
  const unsigned v = dm.nvec.mod(gg), tp = dm.nvec.div(gg);
  unsigned b, h, w;
  dm.pixel(tp, b, h, w);
  const unsigned pix = (b * (unsigned)H + h) * (unsigned)W + w;
  float ix, iy;
  warp_coords<T>(flow, (size_t)pix, (int)h, (int)w, H, W, scale, ix, iy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  float cx[4], cy[4], dx[4], dy[4];
  cubic_coeffs(ix - fx0, cx); cubic_coeffs(iy - fy0, cy);
  cubic_dcoeffs(ix - fx0, dx); cubic_dcoeffs(iy - fy0, dy);
  const int x0 = (int)fx0 - 1, y0 = (int)fy0 - 1;
  constexpr unsigned ES = sizeof(T);
  const unsigned pitch = (unsigned)C * ES;
  const __amdgpu_buffer_rsrc_t xr = buf_rsrc(x, (unsigned)B * H * W * pitch);
  const typename Vec8<T>::Raw g = Vec8<T>::load(gy + ((size_t)pix * nvec + v) * 8);
  float gix = 0.f, giy = 0.f;
  // Branch-free taps as in the forward kernel (clamped address, zero weights outside), raw buffer loads with 32-bit offsets
  unsigned xo[4], yo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int xx = x0 + j;
    if ((unsigned)xx >= (unsigned)W) { cx[j] = 0.f; dx[j] = 0.f; }
    xo[j] = (unsigned)__umul24((unsigned)min(max(xx, 0), W - 1), pitch) + v * (8 * ES);
  }
  const unsigned rowpitch = (unsigned)W * pitch, b0 = b * (unsigned)H;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = y0 + i;
    if ((unsigned)yy >= (unsigned)H) { cy[i] = 0.f; dy[i] = 0.f; }
    yo[i] = (unsigned)__umul24(b0 + (unsigned)min(max(yy, 0), H - 1), rowpitch);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float rx = 0.f, rd = 0.f;                                // sum_j dot_ij * dx[j] , sum_j dot_ij * cx[j]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dot = Vec8<T>::dot(Vec8<T>::buf(xr, yo[i] + xo[j]), g);      // bf16: 4 x v_dot2c_f32_bf16 on the packed vectors
      rx += dot * dx[j];
      rd += dot * cx[j];
    }
    gix += rx * cy[i];
    giy += rd * dy[i];
  }
  if (live && v == 0) {                                       // count the taps that land on every input pixel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = y0 + i;
      if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = x0 + j;
        if ((unsigned)xx < (unsigned)W) atomicAdd(cnt + ((int)b * H + yy) * W + xx, 1);
      }
    }
  }
  // reduce over the nvec lanes that share this pixel (adjacent lanes of one wave)
  for (int o = nvec >> 1; o > 0; o >>= 1) { gix += __shfl_xor(gix, o, 64); giy += __shfl_xor(giy, o, 64); }
  if (live && v == 0) {
    F8 o = f8_zero();
    o.v[0] = gix * (0.5f * (float)W) * scale;
    o.v[1] = giy * (0.5f * (float)H) * scale;
    Feat<T>::store(gflow + (size_t)pix * 8, o);
  }
}

// exclusive prefix sum of cnt[0..n) -> offs[0..n): per-tile scan, scan of the tile sums (one block), add-back
__global__ __launch_bounds__(256) void warp_scan_tiles_kernel(const int* __restrict__ cnt, int* __restrict__ offs,
                                                              int* __restrict__ tile_sums, long long n) {
  __shared__ int sh[256];
  const long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (base + k < n) ? cnt[base + k] : 0;
  const int mine = v[0] + v[1] + v[2] + v[3];
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                          // Hillis-Steele inclusive scan of the 256 thread sums
    const int t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = sh[threadIdx.x] - mine;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (base + k < n) offs[base + k] = run; run += v[k]; }
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(1024) void warp_scan_sums_kernel(int* __restrict__ tile_sums, int ntiles) {
  __shared__ int sh[1024];
  int carry = 0;
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int mine = i < ntiles ? tile_sums[i] : 0;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < ntiles) tile_sums[i] = carry + sh[threadIdx.x] - mine;
    carry += sh[1023];
    __syncthreads();
  }
}
__global__ void warp_scan_add_kernel(int* __restrict__ offs, const int* __restrict__ tile_sums, long long n) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) offs[i] += tile_sums[i / SCAN_TILE];
}

// entries[offs[q] + slot] = (p, wgt); slot counts cnt[q] down to 0, so no second counter array is needed
template <typename T>
__global__ void warp_fill_kernel(const T* __restrict__ flow, int* __restrict__ cnt, const int* __restrict__ offs,
                                 WarpEntry* __restrict__ entries, int B, int H, int W, float scale) {
  const long long total = (long long)B * H * W;
  const long long pix = (long long)blockIdx.x * TPB + threadIdx.x;
  if (pix >= total) return;
  const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  float ix, iy;
  warp_coords<T>(flow, (size_t)pix, h, w, H, W, scale, ix, iy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  float cx[4], cy[4];
  cubic_coeffs(ix - fx0, cx); cubic_coeffs(iy - fy0, cy);
  const int x0 = (int)fx0 - 1, y0 = (int)fy0 - 1;
  // three passes over the 16 taps so that the returning atomics, then the offset loads, are all in flight together (one tap at a
  // time the chain  atomic -> offset -> store  was 16 round trips per thread)
  int q[16], slot[16], base[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = y0 + i, xx = x0 + j;
      q[4 * i + j] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (b * H + yy) * W + xx : -1;
    }
#pragma unroll
  for (int t = 0; t < 16; ++t) slot[t] = q[t] >= 0 ? atomicSub(cnt + q[t], 1) - 1 : 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) base[t] = offs[max(q[t], 0)];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (q[t] < 0) continue;
    WarpEntry e; e.p = (int)pix; e.w = cy[t >> 2] * cx[t & 3];
    entries[(size_t)base[t] + slot[t]] = e;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void warp_gather_kernel(const T* __restrict__ gy, const int* __restrict__ offs, const WarpEntry* __restrict__ entries,
                                                          T* __restrict__ gx, unsigned npix, int C, int H, int W, WarpDims dm) {
  const unsigned total = npix * (unsigned)(C >> 3);
  const unsigned gid = blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const unsigned v = dm.nvec.mod(gid), tq = dm.nvec.div(gid);
  unsigned qb, qh, qw;
  dm.pixel(tq, qb, qh, qw);
  const unsigned q = (qb * (unsigned)H + qh) * (unsigned)W + qw;
  constexpr unsigned ES = sizeof(T);
  const unsigned pitch = (unsigned)C * ES, voff = v * (8 * ES);
  const __amdgpu_buffer_rsrc_t gr = buf_rsrc(gy, npix * pitch);
  const int beg = offs[q], end = offs[q + 1];                  // offs has npix + 1 entries
  F8 s = f8_zero();
  int k = beg;
  for (; k + 4 <= end; k += 4) {                               // four list entries in flight (the list -> address -> load chain is
    WarpEntry en[4];                                           // all latency); summed in list order, as the one-by-one tail does
    F8 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) en[u] = entries[k + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = buf_load8<T>(gr, (unsigned)en[u].p * pitch + voff);   // (a full 32-bit product: pixel indices pass 2^24 at 1024 x 1024, batch 32)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) s.v[j] += en[u].w * t[u].v[j];
  }
  for (; k < end; ++k) {
    const WarpEntry en = entries[k];
    const F8 t = buf_load8<T>(gr, (unsigned)en.p * pitch + voff);
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] += en.w * t.v[j];
  }
  Feat<T>::store(gx + ((size_t)q * (C >> 3) + v) * 8, s);
}

template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long long nvec8) {
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= nvec8) return;
  Feat<T>::store(dst + gid * 8, Feat<float>::load(src + gid * 8));
}

// ------------------------------------------------------------------------------------------------------------
// minibatch standard deviation (group G = min(8, N), STRIDED grouping: sample n = g*M + m shares group m)
// x: [N,H,W,C] -> y: [N,H,W,Cy], channel C = stat[m], channels > C zero.   One block per m.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// (C and Cy multiples of 8: a thread owns 8 consecutive channels of a pixel -- one 16-byte load per sample of the group instead of
// eight 2-byte ones.  The tensors are tiny (B x 4 x 4 x 512), the kernels are pure latency: 45 -> ~10 us per launch, 8 launches
// per iteration.  Per-element arithmetic is unchanged; only the order of the block-wide sums differs.)
template <typename T>
__global__ void mbstd_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int G, int HW, int C, int Cy) {
  __shared__ float sh[16];
  const int M = N / G, m = blockIdx.x, I = HW * C;
  const bool vec = ((C | Cy) & 7) == 0;
  float part = 0.f;
  if (vec) {
    for (int i = threadIdx.x * 8; i < I; i += blockDim.x * 8) {
      F8 mu = f8_zero(), var = f8_zero();
      for (int g = 0; g < G; ++g) {
        const F8 t = Feat<T>::load(x + (size_t)(g * M + m) * I + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) mu.v[j] += t.v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) mu.v[j] /= (float)G;
      for (int g = 0; g < G; ++g) {
        const F8 t = Feat<T>::load(x + (size_t)(g * M + m) * I + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float e = t.v[j] - mu.v[j]; var.v[j] += e * e; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) part += sqrtf(var.v[j] / (float)G + 1e-8f);
    }
  } else {
    for (int i = threadIdx.x; i < I; i += blockDim.x) {
      float mu = 0.f;
      for (int g = 0; g < G; ++g) mu += Feat<T>::ld1(x + (size_t)(g * M + m) * I + i);
      mu /= (float)G;
      float var = 0.f;
      for (int g = 0; g < G; ++g) { const float e = Feat<T>::ld1(x + (size_t)(g * M + m) * I + i) - mu; var += e * e; }
      part += sqrtf(var / (float)G + 1e-8f);
    }
  }
  const float stat = block_sum(part, sh) / (float)I;
  if (vec) {
    const int cv = Cy >> 3, per = HW * cv;
    for (int i = threadIdx.x; i < G * per; i += blockDim.x) {
      const int g = i / per, r = i - g * per, p = r / cv, c = (r - p * cv) * 8;
      const size_t n = (size_t)(g * M + m);
      F8 t = f8_zero();
      if (c < C) t = Feat<T>::load(x + n * I + (size_t)p * C + c);
      else if (c == C) t.v[0] = stat;
      Feat<T>::store(y + (n * HW + p) * Cy + c, t);
    }
    return;
  }
  for (int g = 0; g < G; ++g) {
    const size_t n = (size_t)(g * M + m);
    for (int i = threadIdx.x; i < HW * Cy; i += blockDim.x) {
      const int p = i / Cy, c = i - p * Cy;
      const float v = c < C ? Feat<T>::ld1(x + n * I + (size_t)p * C + c) : (c == C ? stat : 0.f);
      Feat<T>::st1(y + n * HW * Cy + i, v);
    }
  }
}

// gx[g,m,i] = gy[g,m,i] + k * gstat[m] * (x - mu) / sigma ,  k = 1 / (I * G),  gstat[m] = sum_{g,p} gy[g,m,p,C]
template <typename T>
__global__ void mbstd_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x, T* __restrict__ gx,
                                 int N, int G, int HW, int C, int Cy) {
  __shared__ float sh[16];
  const int M = N / G, m = blockIdx.x, I = HW * C;
  float part = 0.f;
  for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {
    const int g = i / HW, p = i - g * HW;
    part += Feat<T>::ld1(gy + ((size_t)(g * M + m) * HW + p) * Cy + C);
  }
  const float kg = block_sum(part, sh) / ((float)I * (float)G);
  if (((C | Cy) & 7) == 0) {                                  // 8 channels of a pixel per thread (see mbstd_fwd_kernel)
    for (int i = threadIdx.x * 8; i < I; i += blockDim.x * 8) {
      const int p = i / C, c = i - p * C;
      F8 mu = f8_zero(), var = f8_zero(), inv_sigma;
      for (int g = 0; g < G; ++g) {
        const F8 t = Feat<T>::load(x + (size_t)(g * M + m) * I + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) mu.v[j] += t.v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) mu.v[j] /= (float)G;
      for (int g = 0; g < G; ++g) {
        const F8 t = Feat<T>::load(x + (size_t)(g * M + m) * I + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float e = t.v[j] - mu.v[j]; var.v[j] += e * e; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) inv_sigma.v[j] = rsqrtf(var.v[j] / (float)G + 1e-8f);
      for (int g = 0; g < G; ++g) {
        const size_t n = (size_t)(g * M + m);
        const F8 t = Feat<T>::load(x + n * I + i), gyv = Feat<T>::load(gy + (n * HW + p) * Cy + c);
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = gyv.v[j] + kg * (t.v[j] - mu.v[j]) * inv_sigma.v[j];
        Feat<T>::store(gx + n * I + i, o);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < I; i += blockDim.x) {
    const int p = i / C, c = i - p * C;
    float mu = 0.f;
    for (int g = 0; g < G; ++g) mu += Feat<T>::ld1(x + (size_t)(g * M + m) * I + i);
    mu /= (float)G;
    float var = 0.f;
    for (int g = 0; g < G; ++g) { const float e = Feat<T>::ld1(x + (size_t)(g * M + m) * I + i) - mu; var += e * e; }
    const float inv_sigma = rsqrtf(var / (float)G + 1e-8f);
    for (int g = 0; g < G; ++g) {
      const size_t n = (size_t)(g * M + m);
      const float e = Feat<T>::ld1(x + n * I + i) - mu;
      const float gyi = Feat<T>::ld1(gy + (n * HW + p) * Cy + c);
      Feat<T>::st1(gx + n * I + i, gyi + kg * e * inv_sigma);
    }
  }
}

// Backward of mbstd_bwd for a cotangent v on gx:
//   ggy[..., :C] = v ; ggy[..., C] = k * sum_{g,i} v e / sigma ; ggy[..., >C] = 0
//   gx2[g',m,i] = k * gstat[m] * ( (v[g'] - mean_g v) / sigma - e[g'] * (sum_g v[g] e[g]) / (G sigma^3) )
template <typename T>
__global__ void mbstd_bwd2_kernel(const T* __restrict__ v, const T* __restrict__ gy, const T* __restrict__ x,
                                  T* __restrict__ ggy, T* __restrict__ gx2, int N, int G, int HW, int C, int Cy) {
  __shared__ float sh[16];
  const int M = N / G, m = blockIdx.x, I = HW * C;
  float part = 0.f;
  for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {
    const int g = i / HW, p = i - g * HW;
    part += Feat<T>::ld1(gy + ((size_t)(g * M + m) * HW + p) * Cy + C);
  }
  const float k = 1.f / ((float)I * (float)G);
  const float kg = block_sum(part, sh) * k;
  float dstat = 0.f;
  const bool vec = ((C | Cy) & 7) == 0;                       // 8 channels of a pixel per thread (see mbstd_fwd_kernel)
  if (vec) {
    for (int i = threadIdx.x * 8; i < I; i += blockDim.x * 8) {
      F8 mu = f8_zero(), vm = f8_zero(), var = f8_zero(), ve = f8_zero(), inv_sigma, c3;
      for (int g = 0; g < G; ++g) {
        const size_t o = (size_t)(g * M + m) * I + i;
        const F8 tx = Feat<T>::load(x + o), tv = Feat<T>::load(v + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu.v[j] += tx.v[j]; vm.v[j] += tv.v[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { mu.v[j] /= (float)G; vm.v[j] /= (float)G; }
      for (int g = 0; g < G; ++g) {
        const size_t o = (size_t)(g * M + m) * I + i;
        const F8 tx = Feat<T>::load(x + o), tv = Feat<T>::load(v + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float e = tx.v[j] - mu.v[j]; var.v[j] += e * e; ve.v[j] += tv.v[j] * e; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        inv_sigma.v[j] = rsqrtf(var.v[j] / (float)G + 1e-8f);
        dstat += ve.v[j] * inv_sigma.v[j];
        c3.v[j] = ve.v[j] * inv_sigma.v[j] * inv_sigma.v[j] * inv_sigma.v[j] / (float)G;
      }
      for (int g = 0; g < G; ++g) {
        const size_t o = (size_t)(g * M + m) * I + i;
        const F8 tx = Feat<T>::load(x + o), tv = Feat<T>::load(v + o);
        F8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = kg * ((tv.v[j] - vm.v[j]) * inv_sigma.v[j] - (tx.v[j] - mu.v[j]) * c3.v[j]);
        Feat<T>::store(gx2 + o, r);
      }
    }
  }
  for (int i = threadIdx.x; i < (vec ? 0 : I); i += blockDim.x) {
    float mu = 0.f, vm = 0.f;
    for (int g = 0; g < G; ++g) {
      mu += Feat<T>::ld1(x + (size_t)(g * M + m) * I + i);
      vm += Feat<T>::ld1(v + (size_t)(g * M + m) * I + i);
    }
    mu /= (float)G; vm /= (float)G;
    float var = 0.f, ve = 0.f;
    for (int g = 0; g < G; ++g) {
      const size_t o = (size_t)(g * M + m) * I + i;
      const float e = Feat<T>::ld1(x + o) - mu;
      var += e * e; ve += Feat<T>::ld1(v + o) * e;
    }
    const float inv_sigma = rsqrtf(var / (float)G + 1e-8f);
    dstat += ve * inv_sigma;
    const float c3 = ve * inv_sigma * inv_sigma * inv_sigma / (float)G;
    for (int g = 0; g < G; ++g) {
      const size_t o = (size_t)(g * M + m) * I + i;
      const float e = Feat<T>::ld1(x + o) - mu;
      Feat<T>::st1(gx2 + o, kg * ((Feat<T>::ld1(v + o) - vm) * inv_sigma - e * c3));
    }
  }
  const float ds = block_sum(dstat, sh) * k;
  if (vec) {
    const int cv = Cy >> 3, per = HW * cv;
    for (int i = threadIdx.x; i < G * per; i += blockDim.x) {
      const int g = i / per, r = i - g * per, p = r / cv, c = (r - p * cv) * 8;
      const size_t n = (size_t)(g * M + m);
      F8 t = f8_zero();
      if (c < C) t = Feat<T>::load(v + n * I + (size_t)p * C + c);
      else if (c == C) t.v[0] = ds;
      Feat<T>::store(ggy + (n * HW + p) * Cy + c, t);
    }
    return;
  }
  for (int g = 0; g < G; ++g) {
    const size_t n = (size_t)(g * M + m);
    for (int i = threadIdx.x; i < HW * Cy; i += blockDim.x) {
      const int p = i / Cy, c = i - p * Cy;
      const float val = c < C ? Feat<T>::ld1(v + n * I + (size_t)p * C + c) : (c == C ? ds : 0.f);
      Feat<T>::st1(ggy + n * HW * Cy + i, val);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// 1x1 convolutions touching the 3-channel fp32 NCHW image.  w: [Bw][3][C] fp32 (Bw = 1 shared, or B per-sample)
// ------------------------------------------------------------------------------------------------------------
// One block = PB consecutive pixels of ONE sample; a thread owns one channel vector and keeps its 24 weights + 8 biases in
// registers while it walks the pixels (the first version reloaded them per pixel: 18x off the HBM roofline).
// ACT: the activation as a compile-time constant (leaky ReLU / none: the two the step uses), -1 = take the run-time argument.  With a
// run-time activation every element carried a chain of uniform branches (and the tanh path): a pure write kernel ran at 2.8 TB/s.
template <typename T, int ACT>
__global__ void rgb_expand_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                  float bias_scale, T* __restrict__ y, int HW, int C, int Clog, int per_sample,
                                  int act_rt, float gain, int PB) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int nvec = C >> 3;
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  if (grp >= groups) return;
  const int b = blockIdx.y;
  const float* wb = w + (per_sample ? (size_t)b * 3 * C : 0);
  float w0[8], w1[8], w2[8], bv[8];
  {                                                          // 8 consecutive floats each: float4 pairs (32 dword loads per thread were a third of the kernel's memory instructions)
    const f32x4 a0 = *(const f32x4*)(wb + v * 8), a1 = *(const f32x4*)(wb + v * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(wb + C + v * 8), b1 = *(const f32x4*)(wb + C + v * 8 + 4);
    const f32x4 c0 = *(const f32x4*)(wb + 2 * C + v * 8), c1 = *(const f32x4*)(wb + 2 * C + v * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { w0[j] = a0[j]; w0[4 + j] = a1[j]; w1[j] = b0[j]; w1[4 + j] = b1[j]; w2[j] = c0[j]; w2[4 + j] = c1[j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = v * 8 + j; bv[j] = (bias && c < Clog) ? bias[c] * bias_scale : 0.f; }
  }
  const int p0 = blockIdx.x * PB, p1 = min(p0 + PB, HW);
  const float* ib = img + (size_t)b * 3 * HW;
  for (int p = p0 + grp; p < p1; p += groups) {
    const float i0 = ib[p], i1 = ib[HW + p], i2 = ib[2 * HW + p];
    F8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = i0 * w0[j] + i1 * w1[j] + i2 * w2[j] + bv[j];
      s.v[j] = (v * 8 + j < Clog) ? act_fwd(t, act) * gain : 0.f;
    }
    Feat<T>::store(y + ((size_t)b * HW + p) * C + v * 8, s);
  }
}

// rgb_expand that also leaves pooled = avg_pool2d(y, 2) (the first DiscriminatorBlock's skip input, custom_layers.py:202): a thread walks
// POOLED pixels and produces the four outputs under each, so the pooled tensor costs a quarter-size store instead of the pooling
// kernel's re-read of the whole 128-channel map.  Same per-element arithmetic as rgb_expand_kernel; pooled sums the values as STORED
// (rounded to T), in the pooling kernel's order.
template <typename T, int ACT>
__global__ void rgb_expand_pool_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                       float bias_scale, T* __restrict__ y, T* __restrict__ pooled, int H, int W, int C, int Clog,
                                       int per_sample, int act_rt, float gain, int PB) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int nvec = C >> 3;
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  if (grp >= groups) return;
  const int b = blockIdx.y, HW = H * W, Wq = W >> 1, HWq = (H >> 1) * Wq;
  const float* wb = w + (per_sample ? (size_t)b * 3 * C : 0);
  float w0[8], w1[8], w2[8], bv[8];
  {
    const f32x4 a0 = *(const f32x4*)(wb + v * 8), a1 = *(const f32x4*)(wb + v * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(wb + C + v * 8), b1 = *(const f32x4*)(wb + C + v * 8 + 4);
    const f32x4 c0 = *(const f32x4*)(wb + 2 * C + v * 8), c1 = *(const f32x4*)(wb + 2 * C + v * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { w0[j] = a0[j]; w0[4 + j] = a1[j]; w1[j] = b0[j]; w1[4 + j] = b1[j]; w2[j] = c0[j]; w2[4 + j] = c1[j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = v * 8 + j; bv[j] = (bias && c < Clog) ? bias[c] * bias_scale : 0.f; }
  }
  const int q0 = blockIdx.x * PB, q1 = min(q0 + PB, HWq);
  const float* ib = img + (size_t)b * 3 * HW;
  for (int q = q0 + grp; q < q1; q += groups) {
    const int qy = q / Wq, qx = q - qy * Wq;
    F8 acc = f8_zero();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int p = (2 * qy + a) * W + 2 * qx + c2;
        const float i0 = ib[p], i1 = ib[HW + p], i2 = ib[2 * HW + p];
        F8 s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = i0 * w0[j] + i1 * w1[j] + i2 * w2[j] + bv[j];
          s.v[j] = (v * 8 + j < Clog) ? act_fwd(t, act) * gain : 0.f;
          acc.v[j] += Feat<T>::rnd(s.v[j]);
        }
        Feat<T>::store(y + ((size_t)b * HW + p) * C + v * 8, s);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] *= 0.25f;
    Feat<T>::store(pooled + ((size_t)b * HWq + q) * C + v * 8, acc);
  }
}

// One block = PB consecutive pixels of ONE sample; a thread keeps its 3 x 8 weights in registers and walks the pixels (the first
// version re-read the 24 weights from L1 for every 16-byte feature vector: 2.9 TB/s).
template <typename T>
__global__ void rgb_reduce_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float bias_scale, float* __restrict__ img, int HW, int C, int per_sample, int PB) {
  const int nvec = C >> 3;                                   // power of two <= 64: a pixel's vectors sit in one wave
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const float* wb = w + (per_sample ? (size_t)b * 3 * C : 0);
  float w0[8], w1[8], w2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int c = v * 8 + j; w0[j] = wb[c]; w1[j] = wb[C + c]; w2[j] = wb[2 * C + c]; }
  const float b0 = bias ? bias[0] * bias_scale : 0.f, b1 = bias ? bias[1] * bias_scale : 0.f, b2 = bias ? bias[2] * bias_scale : 0.f;
  const int p0 = blockIdx.x * PB, p1 = min(p0 + PB, HW);
  float* ib = img + (size_t)b * 3 * HW;
  for (int pb = p0; pb < p1; pb += groups) {                 // every lane runs every trip (the shuffles need the whole wave)
    const int p = pb + grp;
    const bool live = p < p1;
    const F8 t = live ? Feat<T>::load(x + ((size_t)b * HW + p) * C + v * 8) : f8_zero();
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0 += t.v[j] * w0[j]; o1 += t.v[j] * w1[j]; o2 += t.v[j] * w2[j]; }
    for (int o = nvec >> 1; o > 0; o >>= 1) { o0 += __shfl_xor(o0, o, 64); o1 += __shfl_xor(o1, o, 64); o2 += __shfl_xor(o2, o, 64); }
    if (live && v == 0) { ib[p] = o0 + b0; ib[HW + p] = o1 + b1; ib[2 * HW + p] = o2 + b2; }
  }
}

// gw[bw][o][c] += sum_p img[b,o,p] * feat[b,p,c]
template <typename T>
__global__ void rgb_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ feat, float* __restrict__ gw,
                                 int HW, int C, int per_sample, int P) {
  __shared__ float red[3][TPB * 8];
  const int nvec = C >> 3;
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * P, p1 = min(p0 + P, HW);
  float a0[8], a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; }
  if (grp < groups) {
    for (int p = p0 + grp; p < p1; p += groups) {
      const F8 t = Feat<T>::load(feat + ((size_t)b * HW + p) * C + v * 8);
      const float i0 = img[((size_t)b * 3 + 0) * HW + p], i1 = img[((size_t)b * 3 + 1) * HW + p], i2 = img[((size_t)b * 3 + 2) * HW + p];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a0[j] += i0 * t.v[j]; a1[j] += i1 * t.v[j]; a2[j] += i2 * t.v[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][threadIdx.x * 8 + j] = a0[j]; red[1][threadIdx.x * 8 + j] = a1[j]; red[2][threadIdx.x * 8 + j] = a2[j]; }
  __syncthreads();
  float* gwb = gw + (per_sample ? (size_t)b * 3 * C : 0);
  for (int c = threadIdx.x; c < C; c += TPB) {
    const int vv = c >> 3, jj = c & 7;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int gI = 0; gI < groups; ++gI) {
      const int t = (gI * nvec + vv) * 8 + jj;
      t0 += red[0][t]; t1 += red[1][t]; t2 += red[2][t];
    }
    atomicAdd(gwb + c, t0); atomicAdd(gwb + C + c, t1); atomicAdd(gwb + 2 * C + c, t2);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The 2-channel flow layer of a SynthesisBlock (custom_layers.py:123,149-151: ModulatedConv2d(Cin -> 2, k 3, up 2)) as a GEMM + a
// scatter: the x2 transposed convolution out[2i-1+ky, 2j-1+kx, o] += x[i,j,:] . w[o,:,ky,kx] (custom_layers.py:73-80) is
//   t[b,i,j,(ky*3+kx)*2+o] = sum_c s[b,c] x[b,i,j,c] w[o,c,ky,kx]      a 1x1 convolution Cin -> 18 on the LOW-resolution grid (lcgan_conv_fwd), then
//   u[b,Y,X,o] = d[b,o] * sum over the (<= 4) taps that land on (Y, X) + bias[o]                                (flow_col2im_kernel)
// so the input is read once at the HBM rate instead of by a 128-output-channel MFMA tile per phase that keeps 2 of its 128 columns
// (257 us -> the time of one pass over x at 128 x 128 x 256, batch 32).  flow_im2col_kernel is the adjoint gather for the backward.
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void flow_col2im_kernel(const T* __restrict__ t, const float* __restrict__ d, const float* __restrict__ bias,
                                   T* __restrict__ u, int B, int H, int W, int Ct, int dstride) {
  const long long total = (long long)B * 4 * H * W;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int W2 = 2 * W, H2 = 2 * H;
  const int X = (int)(gid % W2), Y = (int)((gid / W2) % H2), b = (int)(gid / ((long long)W2 * H2));
  // rows: Y even -> (ky 1, i Y/2); Y odd -> (ky 0, i (Y+1)/2 if inside) and (ky 2, i (Y-1)/2); columns alike
  int kys[2], is[2], ny = 0, kxs[2], js[2], nx = 0;
  if (Y & 1) { if ((Y + 1) / 2 < H) { kys[ny] = 0; is[ny++] = (Y + 1) / 2; } kys[ny] = 2; is[ny++] = (Y - 1) / 2; }
  else { kys[ny] = 1; is[ny++] = Y / 2; }
  if (X & 1) { if ((X + 1) / 2 < W) { kxs[nx] = 0; js[nx++] = (X + 1) / 2; } kxs[nx] = 2; js[nx++] = (X - 1) / 2; }
  else { kxs[nx] = 1; js[nx++] = X / 2; }
  float a0 = 0.f, a1 = 0.f;
  for (int p = 0; p < ny; ++p)
    for (int q = 0; q < nx; ++q) {
      const T* src = t + (((size_t)b * H + is[p]) * W + js[q]) * Ct + (kys[p] * 3 + kxs[q]) * 2;
      a0 += Feat<T>::ld1(src); a1 += Feat<T>::ld1(src + 1);
    }
  F8 o = f8_zero();
  o.v[0] = a0 * d[(size_t)b * dstride] + (bias ? bias[0] : 0.f);
  o.v[1] = a1 * d[(size_t)b * dstride + 1] + (bias ? bias[1] : 0.f);
  Feat<T>::store(u + (size_t)gid * 8, o);
}

// gt[b,i,j,(ky*3+kx)*2+o] = d[b,o] * gu[b,2i-1+ky,2j-1+kx,o]   (zero outside the image; columns 18 .. Ct-1 zero)
template <typename T>
__global__ void flow_im2col_kernel(const T* __restrict__ gu, const float* __restrict__ d, T* __restrict__ gt,
                                   int B, int H, int W, int Ct, int dstride) {
  const long long total = (long long)B * H * W;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int j = (int)(gid % W), i = (int)((gid / W) % H), b = (int)(gid / ((long long)W * H));
  const float d0 = d[(size_t)b * dstride], d1 = d[(size_t)b * dstride + 1];
  F8 o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) o[k] = f8_zero();
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int Y = 2 * i - 1 + ky, X = 2 * j - 1 + kx, r = (ky * 3 + kx) * 2;
      float g0 = 0.f, g1 = 0.f;
      if ((unsigned)Y < (unsigned)(2 * H) && (unsigned)X < (unsigned)(2 * W)) {
        const T* src = gu + (((size_t)b * 2 * H + Y) * 2 * W + X) * 8;
        g0 = Feat<T>::ld1(src) * d0; g1 = Feat<T>::ld1(src + 1) * d1;
      }
      o[r >> 3].v[r & 7] = g0; o[(r + 1) >> 3].v[(r + 1) & 7] = g1;
    }
  T* dst = gt + (size_t)gid * Ct;
#pragma unroll
  for (int k = 0; k < 3; ++k) Feat<T>::store(dst + 8 * k, o[k]);
}

// ------------------------------------------------------------------------------------------------------------
// Fused backward passes of the two layers that touch the 3-channel image.  Unfused, each is three passes over a full-resolution
// 128-channel tensor (activation backward -> gz, then rgb_reduce / rgb_wgrad re-reading gz): 2.7 GB of traffic per call at
// 256 x 256, batch 32, against 1.1 GB here -- gz of the fromRGB layer never reaches memory at all.
// ------------------------------------------------------------------------------------------------------------
// Backward of rgb_expand (fromRGB: y = act(1x1 conv(img) + bias) * gain, cnn.py:20-21):  gz = gy * act'(y) in registers,
//   gimg[b,o,p]  = sum_c gz[b,p,c] w[bw,o,c]            (optional: the image gradient of the R1 / generator paths)
//   gw[bw,o,c]  += sum_p img[b,o,p] gz[b,p,c]           (optional)
//   gbias[c]    += sum_{b,p} gz[b,p,c]                   (optional)
// One block = P consecutive pixels of one sample; a pixel's channel vectors sit in adjacent lanes of one wave.
// RECOMP (leaky ReLU, img given): the sign of the pre-activation is RECOMPUTED from the image -- t = i0 w0 + i1 w1 + i2 w2 + bias, the
// forward kernel's own expression on operands this kernel holds anyway -- instead of read back from y: half of the kernel's bytes.
template <typename T, int ACT, bool RECOMP = false>
__global__ void rgb_expand_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, const float* __restrict__ img,
                                      const float* __restrict__ w, const float* __restrict__ fbias, float fbias_scale,
                                      float* __restrict__ gimg, float* __restrict__ gw,
                                      float* __restrict__ gbias, int HW, int C, int Clog, int per_sample, int act_rt, float gain, int P) {
  const int act = ACT >= 0 ? ACT : act_rt;
  __shared__ float red[4][TPB * 8];
  const int nvec = C >> 3;                                   // power of two <= 64
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const float* wb = w + (per_sample ? (size_t)b * 3 * C : 0);
  float w0[8], w1[8], w2[8], a0[8], a1[8], a2[8], sb[8], cm[8], fb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = v * 8 + j;
    w0[j] = wb[c]; w1[j] = wb[C + c]; w2[j] = wb[2 * C + c];
    a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; sb[j] = 0.f;
    cm[j] = c < Clog ? 1.f : 0.f;                            // padding channels carry no gradient
    fb[j] = (RECOMP && fbias && c < Clog) ? fbias[c] * fbias_scale : 0.f;
  }
  const int p0 = blockIdx.x * P, p1 = min(p0 + P, HW);
  const float* ib = img + (size_t)b * 3 * HW;
  float* gib = gimg ? gimg + (size_t)b * 3 * HW : nullptr;
  const bool need_img = RECOMP || gw != nullptr;
  for (int pb = p0; pb < p1; pb += 2 * groups) {             // two pixels per trip: their loads are issued together
    F8 g[2], yo[2];
    float im[2][3];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = pb + u * groups + grp;
      live[u] = p < p1;
      const size_t off = ((size_t)b * HW + (live[u] ? p : p0)) * C + v * 8;
      g[u] = Feat<T>::load(gy + off);
      yo[u] = (act != ACT_NONE && !RECOMP) ? Feat<T>::load(y + off) : f8_zero();
      const int pc = live[u] ? p : p0;
      im[u][0] = need_img ? ib[pc] : 0.f; im[u][1] = need_img ? ib[HW + pc] : 0.f; im[u][2] = need_img ? ib[2 * HW + pc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float lv = live[u] ? 1.f : 0.f;
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float ag;
        if (RECOMP) {
          const float t = im[u][0] * w0[j] + im[u][1] * w1[j] + im[u][2] * w2[j] + fb[j];      // (rgb_expand_kernel's expression)
          ag = t > 0.f ? gain : gain * LRELU_SLOPE;
        } else {
          ag = act_grad_from_out(yo[u].v[j], act, gain);
        }
        const float z = g[u].v[j] * ag * cm[j] * lv;
        o0 += z * w0[j]; o1 += z * w1[j]; o2 += z * w2[j];
        a0[j] += im[u][0] * z; a1[j] += im[u][1] * z; a2[j] += im[u][2] * z;
        sb[j] += z;
      }
      if (gib) {                                             // (every lane runs the shuffles; dead pixels contribute zeros)
        for (int o = nvec >> 1; o > 0; o >>= 1) { o0 += __shfl_xor(o0, o, 64); o1 += __shfl_xor(o1, o, 64); o2 += __shfl_xor(o2, o, 64); }
        const int p = pb + u * groups + grp;
        if (live[u] && v == 0) { gib[p] = o0; gib[HW + p] = o1; gib[2 * HW + p] = o2; }
      }
    }
  }
  if (!gw && !gbias) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][threadIdx.x * 8 + j] = a0[j]; red[1][threadIdx.x * 8 + j] = a1[j]; red[2][threadIdx.x * 8 + j] = a2[j];
    red[3][threadIdx.x * 8 + j] = sb[j];
  }
  __syncthreads();
  float* gwb = gw ? gw + (per_sample ? (size_t)b * 3 * C : 0) : nullptr;
  for (int c = threadIdx.x; c < C; c += TPB) {
    const int vv = c >> 3, jj = c & 7;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int gI = 0; gI < groups; ++gI) {
      const int t = (gI * nvec + vv) * 8 + jj;
      t0 += red[0][t]; t1 += red[1][t]; t2 += red[2][t]; t3 += red[3][t];
    }
    if (gwb) { atomicAdd(gwb + c, t0); atomicAdd(gwb + C + c, t1); atomicAdd(gwb + 2 * C + c, t2); }
    if (gbias && c < Clog) atomicAdd(gbias + c, t3);
  }
}

// Backward of  y = act(modulated conv + bias) * gain  ->  img = rgb_reduce(y, wm) + rgb bias  (ToRGBBlock, custom_layers.py:177-182),
// from the image gradient down to the conv's pre-activation gradient in ONE pass over y:
//   gfeat[b,p,c] = sum_o gimg[b,o,p] wm[bw,o,c]                        (never stored)
//   gz           = gfeat * act'(y)                                      (written: the conv's data / weight gradients read it)
//   gbias[c]    += sum_{b,p} gz ;  gdq[b,c] += sum_p gz * (ypre - bias[c] * bias_scale),  ypre = act^-1(y / gain)
//   gwm[bw,o,c] += sum_p gimg[b,o,p] y[b,p,c]
template <typename T>
__global__ void rgb_reduce_bwd_act_kernel(const float* __restrict__ gimg, const T* __restrict__ y, const float* __restrict__ wm,
                                          const float* __restrict__ bias, float bias_scale, T* __restrict__ gz, const float* __restrict__ oscale,
                                          float* __restrict__ gbias, float* __restrict__ gdq, float* __restrict__ gwm,
                                          int HW, int C, int Clog, int per_sample, int act, float gain, int P) {
  __shared__ float red[5][TPB * 8];
  const int nvec = C >> 3;
  const int groups = TPB / nvec;
  const int grp = threadIdx.x / nvec, v = threadIdx.x - grp * nvec;
  const int b = blockIdx.y;
  const bool active = grp < groups;
  const float* wb = wm + (per_sample ? (size_t)b * 3 * C : 0);
  float w0[8], w1[8], w2[8], a0[8], a1[8], a2[8], sb[8], sq[8], bv[8], ov[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = v * 8 + j;
    const bool ok = active && c < Clog;
    w0[j] = ok ? wb[c] : 0.f; w1[j] = ok ? wb[C + c] : 0.f; w2[j] = ok ? wb[2 * C + c] : 0.f;
    bv[j] = (ok && bias) ? bias[c] * bias_scale : 0.f;
    ov[j] = (active && oscale) ? oscale[(size_t)b * C + c] : 1.f;           // the stored gz carries it, the reductions do not (see act_bwd_reduce_kernel)
    a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; sb[j] = 0.f; sq[j] = 0.f;
  }
  const int p0 = blockIdx.x * P, p1 = min(p0 + P, HW);
  const float* ib = gimg + (size_t)b * 3 * HW;
  const float inv_gain = 1.f / gain;
  if (active) {
    for (int pb = p0 + grp; pb < p1; pb += 2 * groups) {
      F8 yo[2];
      float im[2][3];
      bool live[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = pb + u * groups;
        live[u] = p < p1;
        const int pc = live[u] ? p : p0;
        yo[u] = Feat<T>::load(y + ((size_t)b * HW + pc) * C + v * 8);
        im[u][0] = ib[pc]; im[u][1] = ib[HW + pc]; im[u][2] = ib[2 * HW + pc];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!live[u]) continue;
        F8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gf = im[u][0] * w0[j] + im[u][1] * w1[j] + im[u][2] * w2[j];
          z.v[j] = gf * act_grad_from_out(yo[u].v[j], act, gain);
          sb[j] += z.v[j];
          float t = yo[u].v[j] * inv_gain;
          if (act == ACT_LRELU && t < 0.f) t *= (1.f / LRELU_SLOPE);
          sq[j] += z.v[j] * (t - bv[j]);
          a0[j] += im[u][0] * yo[u].v[j]; a1[j] += im[u][1] * yo[u].v[j]; a2[j] += im[u][2] * yo[u].v[j];
          z.v[j] *= ov[j];
        }
        Feat<T>::store(gz + ((size_t)b * HW + pb + u * groups) * C + v * 8, z);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][threadIdx.x * 8 + j] = a0[j]; red[1][threadIdx.x * 8 + j] = a1[j]; red[2][threadIdx.x * 8 + j] = a2[j];
    red[3][threadIdx.x * 8 + j] = sb[j]; red[4][threadIdx.x * 8 + j] = sq[j];
  }
  __syncthreads();
  float* gwb = gwm + (per_sample ? (size_t)b * 3 * C : 0);
  for (int c = threadIdx.x; c < C; c += TPB) {
    const int vv = c >> 3, jj = c & 7;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f;
    for (int gI = 0; gI < groups; ++gI) {
      const int t = (gI * nvec + vv) * 8 + jj;
      t0 += red[0][t]; t1 += red[1][t]; t2 += red[2][t]; t3 += red[3][t]; t4 += red[4][t];
    }
    if (c < Clog) { atomicAdd(gwb + c, t0); atomicAdd(gwb + C + c, t1); atomicAdd(gwb + 2 * C + c, t2); }
    if (gbias && c < Clog) atomicAdd(gbias + c, t3);
    if (gdq) atomicAdd(gdq + (size_t)b * C + c, t4);
  }
}

// ------------------------------------------------------------------------------------------------------------
// layout converts for the (tiny) tensors that cross the NCHW fp32 module boundary
// ------------------------------------------------------------------------------------------------------------
// src fp32 [Bs][Clog][HW] (Bs = B, or 1 broadcast over the batch) -> dst T [B][HW][C] (channels >= Clog zero)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int HW, int C, int Clog, int bcast) {
  const long long total = (long long)B * HW * C;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  const int p = (int)((gid / C) % HW);
  const int b = (int)(gid / ((long long)C * HW));
  const float v = c < Clog ? src[((size_t)(bcast ? 0 : b) * Clog + c) * HW + p] : 0.f;
  Feat<T>::st1(dst + gid, v);
}
// src T [B][HW][C] -> dst fp32 [Bd][Clog][HW]; reduce != 0 sums over the batch into Bd = 1
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int HW, int C, int Clog, int reduce) {
  const int Bd = reduce ? 1 : B;
  const long long total = (long long)Bd * Clog * HW;
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int p = (int)(gid % HW);
  const int c = (int)((gid / HW) % Clog);
  const int b = (int)(gid / ((long long)HW * Clog));
  float v = 0.f;
  if (reduce) { for (int bb = 0; bb < B; ++bb) v += Feat<T>::ld1(src + ((size_t)bb * HW + p) * C + c); }
  else v = Feat<T>::ld1(src + ((size_t)b * HW + p) * C + c);
  dst[gid] = v;
}

bool pow2_le64(int n) { return n >= 1 && n <= 64 && (n & (n - 1)) == 0; }
int reduce_P(int HW, int B) {           // pixels per block for the reduction kernels: ~1024 blocks (one round of 4 per CU), >= 64 pixels
  // Every block ends with one atomic per channel on the SAME C addresses, so the block count is a trade between memory-level
  // parallelism and contention: 4096 blocks cost 5-10 % on the big activation-backward launches (332 / 187 / 112 us at
  // 256^2 x 128 / 128^2 x 256 / 64^2 x 512, batch 32, against 322 / 176 / 100 us), 512 blocks are 30 % slower again.
  int P = (int)(((long long)HW * B + 1023) / 1024);
  return P < 64 ? 64 : P;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                   \
  if ((dtype) == DT_BF16) { typedef __bf16 T; CALL; }             \
  else if ((dtype) == DT_F32) { typedef float T; CALL; }          \
  else return LCGAN_EINVAL;

extern "C" {
int lcgan_rgb_expand_bwd_r(const void* gy, const void* y, const float* img, const float* w, const float* fbias, float fbias_scale, int recompute,
                           float* gimg, float* gw, float* gbias, int B, int HW, int C, int Clog, int per_sample, int act, float gain, int dtype, void* stream);
int lcgan_act_bwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream);
int lcgan_act_bwd_reduce_s(const void* gy, const void* y, const void* mask, void* gz, const float* oscale, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream);
int lcgan_rgb_reduce_bwd_act_s(const float* gimg, const void* y, const float* wm, const float* bias, float bias_scale, void* gz, const float* oscale,
                               float* gbias, float* gdq, float* gwm, int B, int HW, int C, int Clog, int per_sample, int act, float gain,
                               int dtype, void* stream);
int lcgan_box3_actbwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, float* gbias, int B, int H, int W, int C, int Clog,
                               int act, float gain, int dtype, void* stream);

int lcgan_box3_act(const void* x, void* y, int B, int H, int W, int C, int act, float gain, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C & 7) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  Tag tg("box3_act", B, H, W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * 2 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  const int BOX_RH = box_rh(n);
  const long long nthr = (long long)B * ((H + BOX_RH - 1) / BOX_RH) * W * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL(box3_act_kernel<T>, grid1d(nthr), dim3(TPB), 0, s, (const T*)x, (T*)y, B, H, W, C, act, gain, BOX_RH));
  return launch_status();
}

int lcgan_box3_act_bwd(const void* gy, const void* y, void* gx, int B, int H, int W, int C, int act, float gain, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C & 7) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  Tag tg("box3_act_bwd", B, H, W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * 3 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  const int BOX_RH = box_rh(n);
  const long long nthr = (long long)B * ((H + BOX_RH - 1) / BOX_RH) * W * (C / 8);
#define BOXB(A) DISPATCH_T(dtype, hipLaunchKernelGGL((box3_act_bwd_kernel<T, A>), grid1d(nthr), dim3(TPB), 0, s, (const T*)gy, (const T*)y, (T*)gx, B, H, W, C, act, gain, BOX_RH))
  if (act == ACT_LRELU) { BOXB(ACT_LRELU); } else if (act == ACT_NONE) { BOXB(ACT_NONE); } else { BOXB(-1); }
#undef BOXB
  return launch_status();
}

// gz = box3(gy) * act'(y) ; gbias[c] += sum gz (gbias may be NULL; accumulated, must be zeroed by the caller)
int lcgan_box3_actbwd_reduce(const void* gy, const void* y, void* gz, float* gbias, int B, int H, int W, int C, int Clog,
                             int act, float gain, int dtype, void* stream) {
  return lcgan_box3_actbwd_reduce_m(gy, y, nullptr, gz, gbias, B, H, W, C, Clog, act, gain, dtype, stream);
}
int lcgan_box3_actbwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, float* gbias, int B, int H, int W, int C, int Clog,
                               int act, float gain, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || Clog > C) return LCGAN_EINVAL;
  if (mask && act != ACT_LRELU) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  Tag tg(mask ? "box3_actbwd_reduce(mask)" : "box3_actbwd_reduce", B, H, W, C);
  const double esz = dtype == DT_BF16 ? 2 : 4;
  ProfScope p(KID_ACT_BWD, 0, (double)n * 8 * (2 * esz + (mask ? 0.125 : esz)), s, tg.s);
  // strip height: the tallest that still leaves ~128K (32 rows) / ~64K (16 rows) threads (scripts/ab_boxb.py: 77 -> 48 us at
  // 4 x 256 x 256 x 128, 43 -> 35 us at 32 x 32 x 32 x 512; below that 8 rows win)
  const int rh = n / 32 >= (1 << 17) ? 32 : (n / 16 >= (1 << 16) ? 16 : 8);
  const long long nthr = (long long)B * ((H + rh - 1) / rh) * W * (C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL(box3_actbwd_reduce_kernel<T>, grid1d(nthr), dim3(TPB), C * sizeof(float), s, (const T*)gy,
                                       (const T*)y, (const unsigned char*)mask, (T*)gz, gbias, B, H, W, C, Clog, act, gain, rh));
  return launch_status();
}

int lcgan_up2box(const void* x, const void* residual, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C & 7) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * 4 * (C / 8);
  Tag tg("up2box", B, 2 * H, 2 * W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * (residual ? 2.25 : 1.25) * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(up2box_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)x, (const T*)residual, (T*)y, B, H, W, C));
  return launch_status();
}

int lcgan_up2box_bwd(const void* gy, void* gx, int B, int H, int W, int C, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C & 7) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  Tag tg("up2box_bwd", B, 2 * H, 2 * W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * 5 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(up2box_bwd_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)gy, (T*)gx, B, H, W, C));
  return launch_status();
}

int lcgan_avgpool2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || (H & 1) || (W & 1)) return LCGAN_EINVAL;
  const long long n = (long long)B * (H / 2) * (W / 2) * (C / 8);
  Tag tg("avgpool2", B, H, W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * 5 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(avgpool2_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)x, (T*)y, B, H, W, C));
  return launch_status();
}

int lcgan_avgpool2_bwd(const void* gy, void* gx, int B, int H, int W, int C, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || (H & 1) || (W & 1)) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  Tag tg("avgpool2_bwd", B, H, W, C);
  ProfScope p(KID_STENCIL, 0, (double)n * 8 * 1.25 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(avgpool2_bwd_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)gy, (T*)gx, B, H, W, C));
  return launch_status();
}

// gz may be NULL (reductions only); gbias / gdq may be NULL.  gbias: [Clog], gdq: [B][C] -- both accumulated (+=).
int lcgan_act_bwd_reduce(const void* gy, const void* y, void* gz, const float* bias, float bias_scale,
                         float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream) {
  return lcgan_act_bwd_reduce_m(gy, y, nullptr, gz, bias, bias_scale, gbias, gdq, B, HW, C, Clog, act, gain, dtype, stream);
}
// ... with the activation's sign mask (lcgan_conv_fwd_m) instead of y: mask != NULL needs act == leaky ReLU and no gdq; y may then be NULL
int lcgan_act_bwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream) {
  return lcgan_act_bwd_reduce_s(gy, y, mask, gz, nullptr, bias, bias_scale, gbias, gdq, B, HW, C, Clog, act, gain, dtype, stream);
}
// ... storing gz * oscale[b][c] (oscale: [B][C] fp32, needs gz and gdq) while gbias / gdq reduce the unscaled gz: the gradient a modulated
// convolution's backward launches consume already carries the demodulation factor (custom_layers.py:72-76 read backwards)
int lcgan_act_bwd_reduce_s(const void* gy, const void* y, const void* mask, void* gz, const float* oscale, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || C / 8 > TPB || Clog > C) return LCGAN_EINVAL;
  if (mask && (act != ACT_LRELU || gdq)) return LCGAN_EINVAL;
  if (oscale && (!gz || !gdq)) return LCGAN_EINVAL;
  const int P = reduce_P(HW, B);
  dim3 grid(cdiv(HW, P), B);
  const unsigned char* mk = (const unsigned char*)mask;
  Tag tg(gdq ? (gz ? "act_bwd_reduce+gdq" : "act_reduce+gdq") : (gz ? (mask ? "act_bwd_reduce(mask)" : "act_bwd_reduce") : "act_reduce"), B, HW, 1, C);
  const double esz = dtype == DT_BF16 ? 2 : 4;
  ProfScope p(KID_ACT_BWD, 0, (double)B * HW * C * ((gz ? 2 : 1) * esz + (mask ? 0.125 : esz)), s, tg.s);
  if (gdq) { DISPATCH_T(dtype, hipLaunchKernelGGL((act_bwd_reduce_kernel<T, true>), grid, dim3(TPB), 0, s, (const T*)gy, (const T*)y, mk, (T*)gz,
                                                 oscale, bias, bias_scale, gbias, gdq, HW, C, Clog, act, gain, P)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((act_bwd_reduce_kernel<T, false>), grid, dim3(TPB), 0, s, (const T*)gy, (const T*)y, mk, (T*)gz,
                                              oscale, bias, bias_scale, gbias, gdq, HW, C, Clog, act, gain, P)); }
  return launch_status();
}

// u <- s * u (+ res) in place; gs[b][c] += sum_p x * u(old)
int lcgan_scale_reduce_res(void* u, const void* x, const float* sc, float* gs, const void* res, int B, int HW, int C, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || C / 8 > TPB) return LCGAN_EINVAL;
  const int P = reduce_P(HW, B);
  dim3 grid(cdiv(HW, P), B);
  ProfScope p(KID_SCALE_REDUCE, 0, (double)B * HW * C * (res ? 4 : 3) * (dtype == DT_BF16 ? 2 : 4), s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(scale_reduce_kernel<T>, grid, dim3(TPB), 0, s, (T*)u, (const T*)x, sc, gs, (const T*)res, HW, C, P));
  return launch_status();
}
int lcgan_scale_reduce(void* u, const void* x, const float* sc, float* gs, int B, int HW, int C, int dtype, void* stream) {
  return lcgan_scale_reduce_res(u, x, sc, gs, nullptr, B, HW, C, dtype, stream);
}

// flow: [B,H,W,8] (channel 0 = x, 1 = y displacement in normalised units before * scale)
int lcgan_warp_fwd(const void* x, const void* flow, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || H < 2 || W < 2) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W * (C / 8);
  const size_t esz = dtype == DT_BF16 ? 2 : 4;
  if ((long long)W * C * 4 >= (1 << 24)) return LCGAN_EINVAL;    // 24-bit factors of the tap addressing
  if ((double)n * 8 * esz >= 4294967296.0 || (long long)B * H >= (1 << 24)) {
    // 32-bit byte offsets: a batch of 4 GB or more goes in two halves (samples are independent), recursively
    if (B < 2) return LCGAN_EINVAL;
    const int B1 = B / 2;
    const size_t o = (size_t)B1 * H * W;
    const int rc = lcgan_warp_fwd(x, flow, y, B1, H, W, C, scale, dtype, stream);
    if (rc != LCGAN_OK) return rc;
    return lcgan_warp_fwd((const char*)x + o * C * esz, (const char*)flow + o * 8 * esz, (char*)y + o * C * esz, B - B1, H, W, C, scale, dtype, stream);
  }
  Tag tg("warp_fwd", B, H, W, C);
  ProfScope p(KID_WARP_FWD, 0, (double)n * 8 * 2 * esz, s, tg.s);
  const WarpDims dm = warp_dims(C, H, W);
  DISPATCH_T(dtype, hipLaunchKernelGGL(warp_fwd_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)x, (const T*)flow, (T*)y, B, H, W, C, scale, dm));
  return launch_status();
}

// Backward of lcgan_warp_fwd.  gx: [B,H,W,C], gflow: [B,H,W,8].  Workspace (caller-allocated, any contents), npix = B*H*W:
//   ws_cnt   int [npix + 1]                       tap counts per input pixel (zeroed here)
//   ws_off   int [npix + 1]                       list offsets (exclusive scan of the counts; the last one = total entries)
//   ws_tiles int [ceil((npix + 1) / 1024)]        scan scratch
//   ws_ent   8 B x [16 * npix]                    the lists: (output pixel, weight)
int lcgan_warp_bwd(const void* gy, const void* x, const void* flow, void* gx, void* gflow,
                   int* ws_cnt, int* ws_off, int* ws_tiles, void* ws_ent,
                   int B, int H, int W, int C, float scale, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || !pow2_le64(C / 8) || H < 2 || W < 2 || (long long)W * C * 4 >= (1 << 24)) return LCGAN_EINVAL;
  const long long npix = (long long)B * H * W, n = npix * (C / 8);
  const double eb = dtype == DT_BF16 ? 2 : 4;
  // 32-bit byte offsets and list indices: a batch beyond them goes in two halves (samples are independent; the workspaces are reused,
  // the launches are stream-ordered), recursively
  if ((double)n * 8 * eb >= 4294967296.0 || (long long)B * H >= (1 << 24) || npix * 16 >= (1ll << 31)) {
    if (B < 2) return LCGAN_EINVAL;
    const int B1 = B / 2;
    const size_t o = (size_t)B1 * H * W, es = (size_t)eb;
    const int rc = lcgan_warp_bwd(gy, x, flow, gx, gflow, ws_cnt, ws_off, ws_tiles, ws_ent, B1, H, W, C, scale, dtype, stream);
    if (rc != LCGAN_OK) return rc;
    return lcgan_warp_bwd((const char*)gy + o * C * es, (const char*)x + o * C * es, (const char*)flow + o * 8 * es, (char*)gx + o * C * es,
                          (char*)gflow + o * 8 * es, ws_cnt, ws_off, ws_tiles, ws_ent, B - B1, H, W, C, scale, dtype, stream);
  }
  const WarpDims dm = warp_dims(C, H, W);
  Tag tg("warp_bwd", B, H, W, C);
  ProfScope p(KID_WARP_BWD, 0, (double)n * 8 * 4 * eb + (double)npix * 16 * 8 * 2, s, tg.s);
  hipMemsetAsync(ws_cnt, 0, (size_t)(npix + 1) * sizeof(int), s);
  const int ntiles = cdiv(npix + 1, SCAN_TILE);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(warp_bwd_grid_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)gy, (const T*)x, (const T*)flow, (T*)gflow, ws_cnt,
                       B, H, W, C, scale, dm);
    hipLaunchKernelGGL(warp_scan_tiles_kernel, dim3(ntiles), dim3(256), 0, s, ws_cnt, ws_off, ws_tiles, npix + 1);
    hipLaunchKernelGGL(warp_scan_sums_kernel, dim3(1), dim3(1024), 0, s, ws_tiles, ntiles);
    hipLaunchKernelGGL(warp_scan_add_kernel, grid1d(npix + 1), dim3(TPB), 0, s, ws_off, ws_tiles, npix + 1);
    hipLaunchKernelGGL(warp_fill_kernel<T>, grid1d(npix), dim3(TPB), 0, s, (const T*)flow, ws_cnt, ws_off, (WarpEntry*)ws_ent, B, H, W, scale);
    hipLaunchKernelGGL(warp_gather_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)gy, ws_off, (const WarpEntry*)ws_ent, (T*)gx, (unsigned)npix, C, H, W, dm);
  });
  return launch_status();
}

int lcgan_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n & 7) return LCGAN_EINVAL;
  ProfScope p(KID_LAYOUT, 0, (double)n * (4 + (dtype == DT_BF16 ? 2 : 4)), s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_f32_kernel<T>, grid1d(n / 8), dim3(TPB), 0, s, src, (T*)dst, n / 8));
  return launch_status();
}

int lcgan_mbstd_fwd(const void* x, void* y, int N, int G, int HW, int C, int Cy, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (G < 1 || N % G || Cy <= C) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 0, s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(mbstd_fwd_kernel<T>, dim3(N / G), dim3(1024), 0, s, (const T*)x, (T*)y, N, G, HW, C, Cy));
  return launch_status();
}
int lcgan_mbstd_bwd(const void* gy, const void* x, void* gx, int N, int G, int HW, int C, int Cy, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (G < 1 || N % G || Cy <= C) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 0, s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(mbstd_bwd_kernel<T>, dim3(N / G), dim3(1024), 0, s, (const T*)gy, (const T*)x, (T*)gx, N, G, HW, C, Cy));
  return launch_status();
}
int lcgan_mbstd_bwd2(const void* v, const void* gy, const void* x, void* ggy, void* gx2,
                     int N, int G, int HW, int C, int Cy, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (G < 1 || N % G || Cy <= C) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 0, s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(mbstd_bwd2_kernel<T>, dim3(N / G), dim3(1024), 0, s, (const T*)v, (const T*)gy, (const T*)x,
                                       (T*)ggy, (T*)gx2, N, G, HW, C, Cy));
  return launch_status();
}

// y[b,p,c] = act(sum_o img[b,o,p] w[bw,o,c] + bias[c]*bias_scale) * gain   (channels >= Clog are written as zero)
// pooled (may be NULL; then W is unused): also writes avg_pool2d(y, 2) as [B][HW/4][C]; needs the image width W (even, HW / W even)
int lcgan_rgb_expand(const float* img, const float* w, const float* bias, float bias_scale, void* y,
                     int B, int HW, int C, int Clog, int per_sample, int act, float gain, void* pooled, int W, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || C / 8 > TPB) return LCGAN_EINVAL;
  const long long n = (long long)B * HW * (C / 8);
  Tag tg(pooled ? "rgb_expand+pool" : "rgb_expand", B, HW, 1, C);
  ProfScope p(KID_RGB, 0, (double)n * 8 * (pooled ? 1.25 : 1.0) * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  if (pooled) {
    if (W <= 0 || (W & 1) || HW % W || ((HW / W) & 1)) return LCGAN_EINVAL;
    const int PBq = HW / 4 >= 16384 ? 256 : 64;            // pooled pixels per block: the per-thread constants are amortised over PBq / groups of them
#define RGB_POOL(A) DISPATCH_T(dtype, hipLaunchKernelGGL((rgb_expand_pool_kernel<T, A>), dim3(cdiv(HW / 4, PBq), B), dim3(TPB), 0, s, img, w, bias, \
                                                    bias_scale, (T*)y, (T*)pooled, HW / W, W, C, Clog, per_sample, act, gain, PBq))
    if (act == ACT_LRELU) { RGB_POOL(ACT_LRELU); } else if (act == ACT_NONE) { RGB_POOL(ACT_NONE); } else { RGB_POOL(-1); }
#undef RGB_POOL
    return launch_status();
  }
  const int PB = HW >= 65536 ? 1024 : 256;
#define RGB_EXP(A) DISPATCH_T(dtype, hipLaunchKernelGGL((rgb_expand_kernel<T, A>), dim3(cdiv(HW, PB), B), dim3(TPB), 0, s, img, w, bias, bias_scale, \
                                                   (T*)y, HW, C, Clog, per_sample, act, gain, PB))
  if (act == ACT_LRELU) { RGB_EXP(ACT_LRELU); } else if (act == ACT_NONE) { RGB_EXP(ACT_NONE); } else { RGB_EXP(-1); }
#undef RGB_EXP
  return launch_status();
}
// img[b,o,p] = sum_c x[b,p,c] w[bw,o,c] + bias[o]*bias_scale
int lcgan_rgb_reduce(const void* x, const float* w, const float* bias, float bias_scale, float* img,
                     int B, int HW, int C, int per_sample, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || !pow2_le64(C / 8)) return LCGAN_EINVAL;
  const long long n = (long long)B * HW * (C / 8);
  Tag tg("rgb_reduce", B, HW, 1, C);
  ProfScope p(KID_RGB, 0, (double)n * 8 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  const int PB = HW >= 65536 ? 1024 : 256;                 // (the 24 per-thread weights are amortised over PB / groups pixels)
  DISPATCH_T(dtype, hipLaunchKernelGGL(rgb_reduce_kernel<T>, dim3(cdiv(HW, PB), B), dim3(TPB), 0, s, (const T*)x, w, bias, bias_scale, img,
                                       HW, C, per_sample, PB));
  return launch_status();
}
// gw[bw][o][c] += sum_p img[b,o,p] feat[b,p,c]   (gw must be zeroed by the caller)
int lcgan_rgb_wgrad(const float* img, const void* feat, float* gw, int B, int HW, int C, int per_sample, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || C / 8 > TPB) return LCGAN_EINVAL;
  const int P = reduce_P(HW, B);
  dim3 grid(cdiv(HW, P), B);
  Tag tg("rgb_wgrad", B, HW, 1, C);
  ProfScope p(KID_RGB, 0, (double)B * HW * C * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(rgb_wgrad_kernel<T>, grid, dim3(TPB), 0, s, img, (const T*)feat, gw, HW, C, per_sample, P));
  return launch_status();
}

// u [B,2H,2W,8] = d[b,o] * col2im(t) + bias[o]: the scatter half of the flow layer's x2 transposed convolution (see the kernels).
// t: [B,H,W,24] (18 used: (ky*3+kx)*2+o), d: f32 [B][dstride] (demodulation), bias f32 [2] or NULL
int lcgan_flow_col2im(const void* t, const float* d, const float* bias, void* u, int B, int H, int W, int dstride, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B <= 0 || H <= 0 || W <= 0 || dstride < 2) return LCGAN_EINVAL;
  const long long n = (long long)B * 4 * H * W;
  ProfScope p(KID_STENCIL, 0, (double)n * (8 + 6) * (dtype == DT_BF16 ? 2 : 4), s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(flow_col2im_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)t, d, bias, (T*)u, B, H, W, 24, dstride));
  return launch_status();
}
// gt [B,H,W,24] = d[b,o] * im2col(gu): the adjoint gather (backward of lcgan_flow_col2im up to the bias / demodulation reductions)
int lcgan_flow_im2col(const void* gu, const float* d, void* gt, int B, int H, int W, int dstride, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B <= 0 || H <= 0 || W <= 0 || dstride < 2) return LCGAN_EINVAL;
  const long long n = (long long)B * H * W;
  ProfScope p(KID_STENCIL, 0, (double)n * (24 + 32) * (dtype == DT_BF16 ? 2 : 4), s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(flow_im2col_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)gu, d, (T*)gt, B, H, W, 24, dstride));
  return launch_status();
}

// Backward of lcgan_rgb_expand in one pass (gz = gy * act'(y) stays in registers).  gimg [B][3][HW] (written), gw [Bw][3][C] and
// gbias [Clog] (accumulated: zeroed by the caller); each may be NULL.  y: the saved OUTPUT of lcgan_rgb_expand (ignored for act 0).
int lcgan_rgb_expand_bwd(const void* gy, const void* y, const float* img, const float* w, float* gimg, float* gw, float* gbias,
                         int B, int HW, int C, int Clog, int per_sample, int act, float gain, int dtype, void* stream) {
  return lcgan_rgb_expand_bwd_r(gy, y, img, w, nullptr, 0.f, 0, gimg, gw, gbias, B, HW, C, Clog, per_sample, act, gain, dtype, stream);
}
// ... recompute = 1 (leaky ReLU, img != NULL): the activation's sign comes from the image and the forward layer's bias (fbias, may be
// NULL) instead of y, which is then not read (may be NULL): act'(y) needs only sign(w . img + fbias * fbias_scale)
int lcgan_rgb_expand_bwd_r(const void* gy, const void* y, const float* img, const float* w, const float* fbias, float fbias_scale, int recompute,
                           float* gimg, float* gw, float* gbias,
                           int B, int HW, int C, int Clog, int per_sample, int act, float gain, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || !pow2_le64(C / 8) || Clog > C || (gw && !img)) return LCGAN_EINVAL;
  if (recompute && (act != ACT_LRELU || !img)) return LCGAN_EINVAL;
  if (!recompute && act != ACT_NONE && !y) return LCGAN_EINVAL;
  const int P = reduce_P(HW, B);
  dim3 grid(cdiv(HW, P), B);
  Tag tg(recompute ? "rgb_expand_bwd(recompute)" : "rgb_expand_bwd", B, HW, 1, C);
  ProfScope p(KID_RGB, 0, (double)B * HW * C * ((act != ACT_NONE && !recompute) ? 2 : 1) * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
#define RGBB(A, R) DISPATCH_T(dtype, hipLaunchKernelGGL((rgb_expand_bwd_kernel<T, A, R>), grid, dim3(TPB), 0, s, (const T*)gy, (const T*)y, img, w, fbias, fbias_scale, gimg, gw, gbias, \
                                                HW, C, Clog, per_sample, act, gain, P))
  if (recompute) { RGBB(ACT_LRELU, true); } else if (act == ACT_LRELU) { RGBB(ACT_LRELU, false); } else if (act == ACT_NONE) { RGBB(ACT_NONE, false); } else { RGBB(-1, false); }
#undef RGBB
  return launch_status();
}
// Backward of  act(conv) -> lcgan_rgb_reduce  down to the conv's pre-activation gradient (see rgb_reduce_bwd_act_kernel).
// gimg f32 [B][3][HW]; y: the conv's saved activation OUTPUT [B][HW][C]; wm f32 [Bw][3][C]; bias: the CONV's bias (may be NULL).
// gz [B][HW][C] written; gbias [Clog] / gdq [B][C] (either may be NULL) and gwm [Bw][3][C] accumulated (zeroed by the caller).
int lcgan_rgb_reduce_bwd_act(const float* gimg, const void* y, const float* wm, const float* bias, float bias_scale, void* gz,
                             float* gbias, float* gdq, float* gwm, int B, int HW, int C, int Clog, int per_sample, int act, float gain,
                             int dtype, void* stream) {
  return lcgan_rgb_reduce_bwd_act_s(gimg, y, wm, bias, bias_scale, gz, nullptr, gbias, gdq, gwm, B, HW, C, Clog, per_sample, act, gain, dtype, stream);
}
// ... storing gz * oscale[b][c] (oscale [B][C] fp32 or NULL; see lcgan_act_bwd_reduce_s)
int lcgan_rgb_reduce_bwd_act_s(const float* gimg, const void* y, const float* wm, const float* bias, float bias_scale, void* gz, const float* oscale,
                               float* gbias, float* gdq, float* gwm, int B, int HW, int C, int Clog, int per_sample, int act, float gain,
                               int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 7) || C / 8 > TPB || Clog > C || !gwm || !gz) return LCGAN_EINVAL;
  const int P = reduce_P(HW, B);
  dim3 grid(cdiv(HW, P), B);
  Tag tg("rgb_reduce_bwd_act", B, HW, 1, C);
  ProfScope p(KID_RGB, 0, (double)B * HW * C * 2 * (dtype == DT_BF16 ? 2 : 4), s, tg.s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(rgb_reduce_bwd_act_kernel<T>, grid, dim3(TPB), 0, s, gimg, (const T*)y, wm, bias, bias_scale, (T*)gz, oscale,
                                       gbias, gdq, gwm, HW, C, Clog, per_sample, act, gain, P));
  return launch_status();
}

int lcgan_nchw_to_nhwc(const float* src, void* dst, int B, int HW, int C, int Clog, int bcast, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * HW * C;
  ProfScope p(KID_LAYOUT, 0, 0, s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, grid1d(n), dim3(TPB), 0, s, src, (T*)dst, B, HW, C, Clog, bcast));
  return launch_status();
}
int lcgan_nhwc_to_nchw(const void* src, float* dst, int B, int HW, int C, int Clog, int reduce, int dtype, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)(reduce ? 1 : B) * Clog * HW;
  ProfScope p(KID_LAYOUT, 0, 0, s);
  DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, grid1d(n), dim3(TPB), 0, s, (const T*)src, dst, B, HW, C, Clog, reduce));
  return launch_status();
}

}  // extern "C"
