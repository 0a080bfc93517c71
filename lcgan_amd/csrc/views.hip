// Device-side generation of the three training views (image, geometry_change, appearance_change) from one resized image batch:
// the step either side of the hot path.  Replaces, per sample, custom_dataset.py:59-88 of the reference --
// RandomHorizontalFlip (:68), albumentations Perspective (:22-23, :27-33: a homography with bilinear sampling and black
// borders), CoarseDropout (:24, one black rectangle) or ColorJitter (:19-21: brightness / contrast / saturation / hue in a random
// order) and the [-1, 1] normalisation (:81-86) -- which the reference runs on the HOST in 4 PIL / OpenCV worker processes per
// GPU (worker.py:37, 62-69).  The random draws stay on the host (lcgan_amd/data.py builds one 32-float row per sample);
// this kernel is the pixel work: one thread per output pixel, 3 channels, f32 NCHW in [-1, 1] in and out.
//
// params[b][32]:  0 flip | 1..9 Hinv (row-major: output pixel (x, y, 1) -> source pixel of the FLIPPED image) | 10 appearance
// mode (0 = dropout, 1 = colour jitter) | 11..14 hole x0, y0, x1, y1 (pixels, half-open) | 15 brightness | 16 contrast |
// 17 saturation | 18 hue shift (fraction of a turn) | 19..22 order of the four jitter ops (0 brightness, 1 contrast, 2 saturation,
// 3 hue) | 23 contrast pivot in [0, 1], or < 0: computed HERE as the mean luma of the image after the jitter ops that precede the
// contrast op (what albumentations' adjust_contrast_torchvision takes: the mean of its input image), accumulated into slot 24 by
// views_pivot_kernel | 24 pivot accumulator (zero on entry) | 25 != 0: the geometry / appearance views are rounded to the uint8
// grid k / 255 (custom_dataset.py:76-79: albumentations returns uint8 arrays, Image.fromarray + ToTensor turn them back into k / 255)
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), d = mx - mn;
  float h = 0.f;
  if (d > 0.f) {
    if (mx == r) h = (g - b) / d;
    else if (mx == g) h = 2.f + (b - r) / d;
    else h = 4.f + (r - g) / d;
    h *= (1.f / 6.f);
    h -= floorf(h);
  }
  const float s = mx > 0.f ? d / mx : 0.f, v = mx;
  h += shift;
  h -= floorf(h);
  const float h6 = h * 6.f, fi = floorf(h6), f = h6 - fi;
  const int i = (int)fi % 6;
  const float p = v * (1.f - s), q = v * (1.f - s * f), t = v * (1.f - s * (1.f - f));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// one colour-jitter op on a pixel in [0, 1] (m: the contrast pivot)
__device__ __forceinline__ void jitter_op(int op, const float* P, float m, float& r, float& g, float& bl) {
  if (op == 0) { r = clamp01(r * P[15]); g = clamp01(g * P[15]); bl = clamp01(bl * P[15]); }
  else if (op == 1) { const float f = P[16]; r = clamp01((r - m) * f + m); g = clamp01((g - m) * f + m); bl = clamp01((bl - m) * f + m); }
  else if (op == 2) { const float gr = 0.299f * r + 0.587f * g + 0.114f * bl, f = P[17]; r = clamp01(gr + (r - gr) * f); g = clamp01(gr + (g - gr) * f); bl = clamp01(gr + (bl - gr) * f); }
  else { hue_shift(r, g, bl, P[18]); }
}

// params[b][24] += sum over pixels of the luma of (image after the jitter ops in front of the contrast op); only for samples in
// colour-jitter mode whose pivot slot asks for it (params[b][23] < 0)
__global__ __launch_bounds__(256) void views_pivot_kernel(const float* __restrict__ src, float* __restrict__ params, int R) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  float* P = params + (size_t)b * 32;
  if (P[10] == 0.f || P[23] >= 0.f) return;                    // uniform per block
  const size_t plane = (size_t)R * R;
  const float* s0 = src + (size_t)b * 3 * plane;
  float acc = 0.f;
  for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < plane; pix += (size_t)gridDim.x * 256) {
    float r = (s0[pix] + 1.f) * 0.5f, g = (s0[plane + pix] + 1.f) * 0.5f, bl = (s0[2 * plane + pix] + 1.f) * 0.5f;
    for (int k = 0; k < 4; ++k) {
      const int op = (int)P[19 + k];
      if (op == 1) break;
      jitter_op(op, P, 0.f, r, g, bl);
    }
    acc += 0.299f * r + 0.587f * g + 0.114f * bl;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(P + 24, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void views_kernel(const float* __restrict__ src, const float* __restrict__ params,
                                                    float* __restrict__ out_img, float* __restrict__ out_geo,
                                                    float* __restrict__ out_app, int B, int R) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= R * R) return;
  const int y = pix / R, x = pix - y * R;
  const float* P = params + (size_t)b * 32;
  const bool flip = P[0] != 0.f;
  const size_t plane = (size_t)R * R;
  const float* s0 = src + (size_t)b * 3 * plane;
  auto fetch = [&](int c, int yy, int xx) -> float {           // source pixel of the flipped image, in [0, 1]; black outside
    if ((unsigned)yy >= (unsigned)R || (unsigned)xx >= (unsigned)R) return 0.f;
    return (s0[c * plane + (size_t)yy * R + (flip ? R - 1 - xx : xx)] + 1.f) * 0.5f;
  };
  const size_t o = (size_t)b * 3 * plane + (size_t)y * R + x;
  float c3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    c3[c] = fetch(c, y, x);
    out_img[o + c * plane] = fminf(fmaxf(c3[c] * 2.f - 1.f, -1.f), 1.f);
  }
  // ---- geometry view: homography + bilinear, constant (black) border -------------------------------------------------
  {
    const float w = P[7] * x + P[8] * y + P[9];
    const float iw = fabsf(w) > 1e-12f ? 1.f / w : 0.f;
    const float sx = (P[1] * x + P[2] * y + P[3]) * iw, sy = (P[4] * x + P[5] * y + P[6]) * iw;
    const float fx = floorf(sx), fy = floorf(sy);
    const float ax = sx - fx, ay = sy - fy;
    // far outside the image: every tap is black (also keeps the int conversion in range)
    const bool far = !(sx > -2.f && sy > -2.f && sx < R + 1.f && sy < R + 1.f);
    const int x0 = far ? -4 : (int)fx, y0 = far ? -4 : (int)fy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (1.f - ay) * ((1.f - ax) * fetch(c, y0, x0) + ax * fetch(c, y0, x0 + 1)) +
                      ay * ((1.f - ax) * fetch(c, y0 + 1, x0) + ax * fetch(c, y0 + 1, x0 + 1));
      const float vq = P[25] != 0.f ? rintf(clamp01(v) * 255.f) * (1.f / 255.f) : v;
      out_geo[o + c * plane] = fminf(fmaxf(vq * 2.f - 1.f, -1.f), 1.f);
    }
  }
  // ---- appearance view: one black rectangle, or colour jitter --------------------------------------------------------
  {
    float r = c3[0], g = c3[1], bl = c3[2];
    if (P[10] == 0.f) {
      if (x >= (int)P[11] && x < (int)P[13] && y >= (int)P[12] && y < (int)P[14]) { r = 0.f; g = 0.f; bl = 0.f; }
    } else {
      const float m = P[23] >= 0.f ? P[23] : P[24] / (float)((size_t)R * R);
#pragma unroll
      for (int k = 0; k < 4; ++k) jitter_op((int)P[19 + k], P, m, r, g, bl);
    }
    if (P[25] != 0.f) { r = rintf(r * 255.f) * (1.f / 255.f); g = rintf(g * 255.f) * (1.f / 255.f); bl = rintf(bl * 255.f) * (1.f / 255.f); }
    out_app[o] = r * 2.f - 1.f; out_app[o + plane] = g * 2.f - 1.f; out_app[o + 2 * plane] = bl * 2.f - 1.f;
  }
}

}  // namespace

// params: [B][32] floats (layout above); slot 24 of every row is WRITTEN (pivot accumulator), everything else is read only
extern "C" int lcgan_make_views(const float* src, float* params, float* out_img, float* out_geo, float* out_app,
                                int B, int R, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B <= 0 || R <= 0 || (long long)B * 3 * R * R >= (1ll << 40)) return LCGAN_EINVAL;
  ProfScope p(KID_LAYOUT, 0, (double)B * 3 * R * R * 4 * 5, s);
  hipLaunchKernelGGL(views_pivot_kernel, dim3(std::min(cdiv((long long)R * R, 256), 64), B), dim3(256), 0, s, src, params, R);
  hipLaunchKernelGGL(views_kernel, dim3(cdiv((long long)R * R, 256), B), dim3(256), 0, s, src, params, out_img, out_geo, out_app, B, R);
  return launch_status();
}
