// Latency-class kernels of the LC-GAN step for gfx950: small-M fp32 linear layers (style affines, mapping MLPs,
// discriminator epilogue and projection heads), demodulation statistics, loss reductions, and the multi-tensor
// Adam / EMA / gradient-pack kernels.
//
// Reference ops replaced (file:line in /root/reference):
//   linear_*      F.linear in EqualizedLinear                         custom_layers.py:24-25 (+ autograd)
//   demod_*       rsqrt(sum (w*s)^2 + eps)                            custom_layers.py:67
//   bce / contrastive / l2norm / abs_sum / sumsq                      worker.py:156-157,191,207-209 ; loss.py:9-24 ; cnn.py:40-41
//   adam          torch.optim.Adam(betas=(0,0.99), eps=1e-8)          worker.py:98-110
//   ema           Ema.update                                          ema.py:19-32
//   avg_latent    truncation-trick running mean                       cnn.py:95-97
#include "common.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;
constexpr int MT = 32;     // rows of the small-M GEMMs handled per block
constexpr int LF_MT = 8;   // ... of linear_fwd_kernel

// A group of L linear layers that share their input x [M,I] (L = 1: a plain linear layer).  The table travels BY VALUE in the
// kernel arguments, so a grouped launch costs no host->device copy: the 20 style affines of a generator pass are 2 launches.
constexpr int LIN_MAXL = 24;
struct LinGroup {
  const float* w[LIN_MAXL];      // [O_l, I]
  const float* aux[LIN_MAXL];    // forward: bias_l [O_l] (or null) ; backward: gy_l [M, O_l]
  float* out[LIN_MAXL];          // forward: y_l [M, O_l] ; wgrad: gw_l [O_l, I] ; colsum: gb_l [O_l]
  int O[LIN_MAXL];
  float scale[LIN_MAXL];         // equalized-lr weight scale c_l
  float bscale[LIN_MAXL];        // bias scale (lr_mul)
  const float* xin[LIN_MAXL];    // per-layer input x_l [M, I_l] (or null: the shared x of the launch) -- layers at equal depth of DIFFERENT chains (the two mapping networks)
  int Iv[LIN_MAXL];              // ... and its width I_l (0: the launch's I)
  float* out2[LIN_MAXL];         // wgrad: gb_l [O_l] = bscale_l * colsum(gy_l), written by the i == 0 blocks from their staged gy slice (or null)
};

// y_l[m,o] = act(scale_l * sum_i x[m,i] w_l[o,i] + bias_l[o]*bscale_l) * gain
// Block = 8 output columns x 32 rows of x.  Per 256-wide i-chunk BOTH operands are staged in LDS with all loads issued at once
// (x transposed [i][m] padded, the 8 weight rows [o][i]); the inner loop then runs from LDS only (weight reads are broadcasts).
// isplit > 1 (long rows: the discriminator epilogue's 8192-wide linear) spreads the i range over isplit blocks that add their
// raw partial sums into a zeroed y; linear_finalize_kernel then applies bias / activation.
constexpr int LIN_IC = 256;
__global__ __launch_bounds__(TPB) void linear_fwd_generic_kernel(const float* x, const LinGroup g, int M, int I, int isplit,
                                                          int act, float gain) {
  __shared__ float xs[LIN_IC * 33];
  __shared__ float ws[8 * LIN_IC];
  const int l = blockIdx.z, O = g.O[l];
  const int o0 = blockIdx.x * 8;
  if (o0 >= O) return;
  const float* __restrict__ w = g.w[l];
  if (g.xin[l]) { x = g.xin[l]; I = g.Iv[l]; }
  const int m = threadIdx.x & 31, ol = threadIdx.x >> 5;
  const int mt = blockIdx.y / isplit, sp = blockIdx.y - mt * isplit;
  const int m0 = mt * MT;
  const int ilen = cdiv(cdiv(I, isplit), LIN_IC) * LIN_IC;
  const int ibeg = sp * ilen, iend = min(I, ibeg + ilen);
  float acc = 0.f;
  for (int i0 = ibeg; i0 < iend; i0 += LIN_IC) {
    const int ic = min(LIN_IC, iend - i0);
    __syncthreads();
    // staging: thread t walks column ii = t (+256 ...) of every row: no integer division, coalesced along i
    for (int ii = threadIdx.x; ii < ic; ii += TPB) {
#pragma unroll 8
      for (int mm = 0; mm < MT; ++mm) xs[ii * 33 + mm] = (m0 + mm < M) ? x[(size_t)(m0 + mm) * I + i0 + ii] : 0.f;
#pragma unroll
      for (int oo = 0; oo < 8; ++oo) ws[oo * LIN_IC + ii] = (o0 + oo < O) ? w[(size_t)(o0 + oo) * I + i0 + ii] : 0.f;
    }
    __syncthreads();
    const float* wr = ws + ol * LIN_IC;
#pragma unroll 8
    for (int ii = 0; ii < ic; ++ii) acc += wr[ii] * xs[ii * 33 + m];
  }
  const int o = o0 + ol, mrow = m0 + m;
  if (o < O && mrow < M) {
    float* y = g.out[l] + (size_t)mrow * O + o;
    if (isplit > 1) { atomicAdd(y, acc * g.scale[l]); return; }
    const float bv = g.aux[l] ? g.aux[l][o] * g.bscale[l] : 0.f;
    *y = act_fwd(acc * g.scale[l] + bv, act) * gain;
  }
}

// I % 8 == 0 (every layer of the step): block = 8 output columns (2 per wave) x 32 rows of x.  A lane owns 8 consecutive i of each
// 512-wide sweep: its two weight rows arrive as four 16-byte loads, the 32 rows of x (64 KB for I = 512, L1 / L2 hits shared by
// every block) as coalesced 16-byte loads, 16 FMAs per row; the 64 lanes' partial sums meet through LDS (padded rows, conflict-free
// both ways).  The LDS-staged kernel above took 24 us for a 512 x 512 layer (a serial 512-step inner loop behind 80 scalar loads per
// thread); 63 such launches per iteration made the linear layers 2.6 ms of the step.
__global__ __launch_bounds__(TPB) void linear_fwd_kernel(const float* x, const LinGroup g, int M, int I, int isplit,
                                                          int act, float gain) {
  // LF_MT rows per block (not the 32 of the other small-M kernels): with 32 rows a lane had 64 sixteen-byte loads of x to wait for
  // in batches and a 512 x 512 layer occupied 64 CUs for 18 us; 8 rows put all 16 loads in flight at once on 256 blocks
  __shared__ float red[4][2 * LF_MT][65];
  const int l = blockIdx.z, O = g.O[l];
  const int o0 = blockIdx.x * 8;
  if (o0 >= O) return;
  const float* __restrict__ w = g.w[l];
  if (g.xin[l]) { x = g.xin[l]; I = g.Iv[l]; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int mt = blockIdx.y / isplit, sp = blockIdx.y - mt * isplit;
  const int m0 = mt * LF_MT;
  const int ilen = cdiv(cdiv(I, isplit), 512) * 512;
  const int ibeg = sp * ilen, iend = min(I, ibeg + ilen);
  const int oa = o0 + 2 * wv, ob = oa + 1;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 z4 = {0.f, 0.f, 0.f, 0.f};
  float acc[LF_MT][2];
#pragma unroll
  for (int m = 0; m < LF_MT; ++m) { acc[m][0] = 0.f; acc[m][1] = 0.f; }
  for (int i0 = ibeg + lane * 8; i0 < iend; i0 += 512) {
    const f4 wa0 = oa < O ? *(const f4*)(w + (size_t)oa * I + i0) : z4, wa1 = oa < O ? *(const f4*)(w + (size_t)oa * I + i0 + 4) : z4;
    const f4 wb0 = ob < O ? *(const f4*)(w + (size_t)ob * I + i0) : z4, wb1 = ob < O ? *(const f4*)(w + (size_t)ob * I + i0 + 4) : z4;
    f4 x0[LF_MT], x1[LF_MT];
#pragma unroll
    for (int m = 0; m < LF_MT; ++m) {                          // rows beyond M re-read row M - 1 (discarded below): no branch around the loads
      const int mr = min(m0 + m, M - 1);
      x0[m] = *(const f4*)(x + (size_t)mr * I + i0); x1[m] = *(const f4*)(x + (size_t)mr * I + i0 + 4);
    }
#pragma unroll
    for (int m = 0; m < LF_MT; ++m) {
      acc[m][0] += wa0[0] * x0[m][0] + wa0[1] * x0[m][1] + wa0[2] * x0[m][2] + wa0[3] * x0[m][3] + wa1[0] * x1[m][0] + wa1[1] * x1[m][1] + wa1[2] * x1[m][2] + wa1[3] * x1[m][3];
      acc[m][1] += wb0[0] * x0[m][0] + wb0[1] * x0[m][1] + wb0[2] * x0[m][2] + wb0[3] * x0[m][3] + wb1[0] * x1[m][0] + wb1[1] * x1[m][1] + wb1[2] * x1[m][2] + wb1[3] * x1[m][3];
    }
  }
#pragma unroll
  for (int m = 0; m < LF_MT; ++m) {
    red[wv][m * 2][lane] = acc[m][0];
    red[wv][m * 2 + 1][lane] = acc[m][1];
  }
  __syncthreads();
  if (lane < 2 * LF_MT) {
    float v = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) v += red[wv][lane][k];
    const int o = oa + (lane & 1), mrow = m0 + (lane >> 1);
    if (o < O && mrow < M) {
      float* y = g.out[l] + (size_t)mrow * O + o;
      if (isplit > 1) atomicAdd(y, v * g.scale[l]);
      else {
        const float bv = g.aux[l] ? g.aux[l][o] * g.bscale[l] : 0.f;
        *y = act_fwd(v * g.scale[l] + bv, act) * gain;
      }
    }
  }
}

__global__ void linear_finalize_kernel(float* __restrict__ y, const float* __restrict__ bias, float bscale, int M, int O, int act,
                                       float gain) {
  const int idx = blockIdx.x * TPB + threadIdx.x;
  if (idx >= M * O) return;
  const float bv = bias ? bias[idx % O] * bscale : 0.f;
  y[idx] = act_fwd(y[idx] + bv, act) * gain;
}

// gx[m,i] (+)= sum_l scale_l * sum_o gy_l[m,o] w_l[o,i]
// Block = 64 columns i x 32 rows m (4 waves x 8 rows each) x one 64-wide o-chunk of one layer: the chunk of gy is staged
// transposed in LDS ([o][m]) so each weight element (coalesced along i) meets 8 broadcast LDS values.  All (layer, chunk)
// blocks add into a zeroed gx; a single-chunk launch stores directly.  (The first version looped over all of O in 8 blocks:
// 53 us for a 512x512 layer.)
constexpr int LIN_OC = 64;
__global__ __launch_bounds__(TPB) void linear_bwd_data_kernel(const LinGroup g, float* gx, int M, int I, int nchunk,
                                                               int accumulate) {
  __shared__ float gs[LIN_OC * 32];
  const int l = blockIdx.y / nchunk, o0 = (blockIdx.y - l * nchunk) * LIN_OC, O = g.O[l];
  if (o0 >= O) return;
  if (g.Iv[l]) I = g.Iv[l];
  const int oc = min(LIN_OC, O - o0);
  const float* __restrict__ gy = g.aux[l];
  const int il = threadIdx.x & 63, mg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il, m0 = blockIdx.z * MT;
  for (int idx = threadIdx.x; idx < oc * MT; idx += TPB) {             // lanes walk m: conflict-free LDS writes
    const int oo = idx >> 5, mm = idx & 31;
    gs[idx] = (m0 + mm < M) ? gy[(size_t)(m0 + mm) * O + o0 + oo] : 0.f;
  }
  __syncthreads();
  if (i >= I) return;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float* wc = g.w[l] + (size_t)o0 * I + i;
#pragma unroll 8
  for (int oo = 0; oo < oc; ++oo) {
    const float wv = wc[(size_t)oo * I];
    const f32x4 g0 = *(const f32x4*)(gs + oo * 32 + mg * 8), g1 = *(const f32x4*)(gs + oo * 32 + mg * 8 + 4);
    acc[0] += g0[0] * wv; acc[1] += g0[1] * wv; acc[2] += g0[2] * wv; acc[3] += g0[3] * wv;
    acc[4] += g1[0] * wv; acc[5] += g1[1] * wv; acc[6] += g1[2] * wv; acc[7] += g1[3] * wv;
  }
  const float sc = g.scale[l];
  if (g.out[l]) gx = g.out[l];                                 // per-layer data gradient (layers with their own inputs)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int mrow = m0 + mg * 8 + j;
    if (mrow < M) {
      if (accumulate) atomicAdd(gx + (size_t)mrow * I + i, acc[j] * sc);
      else gx[(size_t)mrow * I + i] = acc[j] * sc;
    }
  }
}

// gw_l[o,i] = scale_l * sum_m gy_l[m,o] x[m,i]
// Block = 16 rows o x 256 columns i: the 16 x M slice of gy is staged in LDS once (broadcast reads), x is read coalesced and
// reused for the 16 rows from registers.  (One block per row re-read x 512 times and was latency-bound: 18 us for 512 x 512.)
constexpr int LW_OT = 16, LW_MC = 64;
__global__ __launch_bounds__(TPB) void linear_wgrad_kernel(const LinGroup g, const float* x, int M, int I) {
  __shared__ float gs[LW_MC][LW_OT];
  const int l = blockIdx.z, O = g.O[l];
  const int o0 = blockIdx.y * LW_OT;
  if (o0 >= O) return;
  const int i = blockIdx.x * TPB + threadIdx.x;
  const float* __restrict__ gy = g.aux[l];
  if (g.xin[l]) { x = g.xin[l]; I = g.Iv[l]; }
  float acc[LW_OT];
#pragma unroll
  for (int r = 0; r < LW_OT; ++r) acc[r] = 0.f;
  // the bias gradient (column sums of gy) leaves the same launch: the first block of every row group has the 16 columns staged anyway
  const bool bias_blk = blockIdx.x == 0 && g.out2[l] != nullptr && threadIdx.x < LW_OT;
  float gbacc = 0.f;
  for (int m0 = 0; m0 < M; m0 += LW_MC) {
    const int mc = min(LW_MC, M - m0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < mc * LW_OT; idx += TPB) {
      const int mm = idx / LW_OT, r = idx - mm * LW_OT;
      gs[mm][r] = (o0 + r < O) ? gy[(size_t)(m0 + mm) * O + o0 + r] : 0.f;
    }
    __syncthreads();
    if (bias_blk)
      for (int mm = 0; mm < mc; ++mm) gbacc += gs[mm][threadIdx.x];
    if (i < I) {
      for (int mm = 0; mm < mc; ++mm) {
        const float xv = x[(size_t)(m0 + mm) * I + i];
#pragma unroll
        for (int r = 0; r < LW_OT; ++r) acc[r] += gs[mm][r] * xv;
      }
    }
  }
  if (i < I) {
#pragma unroll
    for (int r = 0; r < LW_OT; ++r)
      if (o0 + r < O) g.out[l][(size_t)(o0 + r) * I + i] = acc[r] * g.scale[l];
  }
  if (bias_blk && o0 + (int)threadIdx.x < O) g.out2[l][o0 + threadIdx.x] = gbacc * g.bscale[l];
}

// gb_l[o] = bscale_l * sum_m gy_l[m,o]
__global__ void colsum_kernel(const LinGroup g, int M) {
  const int l = blockIdx.y, O = g.O[l];
  const int o = blockIdx.x * TPB + threadIdx.x;
  if (o >= O) return;
  const float* __restrict__ gy = g.aux[l];
  float acc = 0.f;
  for (int m = 0; m < M; ++m) acc += gy[(size_t)m * O + o];
  g.out[l][o] = acc * g.bscale[l];
}

// gz = gy * act'(y)   on fp32 vectors
__global__ void act_bwd_f32_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gz,
                                   long long n, int act, float gain) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) gz[i] = gy[i] * act_grad_from_out(y[i], act, gain);
}

// d[b,o] = rsqrt(sum_c s[b,c]^2 wsq[o,c] + eps)       one wave per (b,o); d is [B][Os] (Os >= O alloc stride)
__global__ void demod_fwd_kernel(const float* __restrict__ s, const float* __restrict__ wsq, float* __restrict__ d,
                                 int B, int C, int O, int Os, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wid, b = blockIdx.y;
  if (o >= O) return;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) { const float sv = s[(size_t)b * C + c]; acc += sv * sv * wsq[(size_t)o * C + c]; }
  acc = wave_sum(acc);
  if (lane == 0) d[(size_t)b * Os + o] = rsqrtf(acc + eps);
}
// the same for L <= 24 layers in one launch (table by value): the generator knows every layer's style before its first convolution
// (the grouped affines, cnn.py:103-104), so the 19 demodulation launches of a forward pass are one
constexpr int DM_MAXL = 24;
struct DemodGroup { const float* s[DM_MAXL]; const float* wsq[DM_MAXL]; float* d[DM_MAXL]; int C[DM_MAXL], O[DM_MAXL], Os[DM_MAXL]; };
__global__ void demod_group_kernel(const DemodGroup g, int B, float eps) {
  const int l = blockIdx.z, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wid, b = blockIdx.y, C = g.C[l];
  if (o >= g.O[l]) return;
  const float* __restrict__ s = g.s[l];
  const float* __restrict__ wsq = g.wsq[l];
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) { const float sv = s[(size_t)b * C + c]; acc += sv * sv * wsq[(size_t)o * C + c]; }
  acc = wave_sum(acc);
  if (lane == 0) g.d[l][(size_t)b * g.Os[l] + o] = rsqrtf(acc + eps);
}
// gq[b,o] = -0.5 * gdq[b,o] * d[b,o]^2 ;  gs[b,c] += 2 s[b,c] sum_o gq[b,o] wsq[o,c]
// block = (256 channels c, sample b, chunk of 64 outputs o): gq of the chunk sits in LDS, partial sums meet through atomics
__global__ void demod_bwd_s_kernel(const float* __restrict__ gdq, const float* __restrict__ d, const float* __restrict__ s,
                                   const float* __restrict__ wsq, float* __restrict__ gs, int B, int C, int O, int Os) {
  __shared__ float gq[64];
  const int c = blockIdx.x * TPB + threadIdx.x, b = blockIdx.y, o0 = blockIdx.z * 64;
  const int oc = min(64, O - o0);
  if (threadIdx.x < oc) {
    const float dv = d[(size_t)b * Os + o0 + threadIdx.x];
    gq[threadIdx.x] = -0.5f * gdq[(size_t)b * Os + o0 + threadIdx.x] * dv * dv;
  }
  __syncthreads();
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll 16
  for (int o = 0; o < oc; ++o) acc += gq[o] * wsq[(size_t)(o0 + o) * C + c];
  atomicAdd(gs + (size_t)b * C + c, 2.f * s[(size_t)b * C + c] * acc);
}
// gwsq[o,c] = sum_b gq[b,o] s[b,c]^2
__global__ void demod_bwd_w_kernel(const float* __restrict__ gdq, const float* __restrict__ d, const float* __restrict__ s,
                                   float* __restrict__ gwsq, int B, int C, int O, int Os) {
  const int c = blockIdx.x * TPB + threadIdx.x, o = blockIdx.y;
  if (c >= C) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const float dv = d[(size_t)b * Os + o], sv = s[(size_t)b * C + c];
    acc += -0.5f * gdq[(size_t)b * Os + o] * dv * dv * sv * sv;
  }
  gwsq[(size_t)o * C + c] = acc;
}

// ---- single-block loss kernels ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float softplusf(float t) { return t > 20.f ? t : log1pf(expf(t)); }
__device__ __forceinline__ float sigmoidf(float t) { return 1.f / (1.f + expf(-t)); }

// out = mean_i softplus(sign * logit_i), sign = -1 for target 1, +1 for target 0   (BCE with logits)
__global__ void bce_fwd_kernel(const float* __restrict__ logit, int n, float sign, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += TPB) acc += softplusf(sign * logit[i]);
  const float t = block_sum256(acc, sh);
  if (threadIdx.x == 0) out[0] = t / (float)n;
}
__global__ void bce_bwd_kernel(const float* __restrict__ logit, int n, float sign, const float* __restrict__ gout,
                               float* __restrict__ g) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) g[i] = gout[0] * sign * sigmoidf(sign * logit[i]) / (float)n;
}

// loss = mean_b softplus((a.n - a.p) / tau)   ; t[b] saved for backward
__global__ void contrastive_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n,
                                       int B, int D, float tau, float* __restrict__ tsave, float* __restrict__ out) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = wid; b < B; b += 4) {
    float dp = 0.f, dn = 0.f;
    for (int j = lane; j < D; j += 64) { const float av = a[(size_t)b * D + j]; dp += av * p[(size_t)b * D + j]; dn += av * n[(size_t)b * D + j]; }
    dp = wave_sum(dp); dn = wave_sum(dn);
    const float t = (dn - dp) / tau;
    if (lane == 0) { tsave[b] = t; acc += softplusf(t); }
  }
  const float tot = block_sum256(acc, sh);
  if (threadIdx.x == 0) out[0] = tot / (float)B;
}
__global__ void contrastive_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n,
                                       const float* __restrict__ tsave, const float* __restrict__ gout, int B, int D, float tau,
                                       float* __restrict__ ga, float* __restrict__ gp, float* __restrict__ gn) {
  const int idx = blockIdx.x * TPB + threadIdx.x;
  if (idx >= B * D) return;
  const int b = idx / D;
  const float k = gout[0] * sigmoidf(tsave[b]) / ((float)B * tau);
  ga[idx] = k * (n[idx] - p[idx]);
  gp[idx] = -k * a[idx];
  gn[idx] = k * a[idx];
}

// y = x / max(||x||_2, eps) per row ; norm saved
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nsave, int B, int D, float eps) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc = 0.f;
  for (int j = lane; j < D; j += 64) { const float v = x[(size_t)b * D + j]; acc += v * v; }
  const float nr = fmaxf(sqrtf(wave_sum(acc)), eps);
  for (int j = lane; j < D; j += 64) y[(size_t)b * D + j] = x[(size_t)b * D + j] / nr;
  if (lane == 0) nsave[b] = nr;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ nsave,
                                  float* __restrict__ gx, int B, int D) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc = 0.f;
  for (int j = lane; j < D; j += 64) acc += gy[(size_t)b * D + j] * y[(size_t)b * D + j];
  acc = wave_sum(acc);
  const float inv = 1.f / nsave[b];
  for (int j = lane; j < D; j += 64) gx[(size_t)b * D + j] = (gy[(size_t)b * D + j] - y[(size_t)b * D + j] * acc) * inv;
}

// out += coef * sum |x|^pw (pw = 1: abs, 2: square), multi-block with one atomic per block (out zeroed by the caller)
__global__ void powsum_kernel(const float* __restrict__ x, long long n, int pw, float coef, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float v = x[i];
    acc += pw == 1 ? fabsf(v) : v * v;
  }
  const float t = block_sum256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, t * coef);
}
// g = gout * coef * (pw == 1 ? sign(x) : 2 x)
__global__ void powsum_bwd_kernel(const float* __restrict__ x, long long n, int pw, float coef, const float* __restrict__ gout,
                                  float* __restrict__ g) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  g[i] = gout[0] * coef * (pw == 1 ? (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) : 2.f * v);
}

// avg[d] <- mean_b w[b,d] + beta * (avg[d] - mean_b w[b,d])
__global__ void avg_latent_kernel(const float* __restrict__ w, float* __restrict__ avg, int B, int D, float beta) {
  const int d = blockIdx.x * TPB + threadIdx.x;
  if (d >= D) return;
  float m = 0.f;
  for (int b = 0; b < B; ++b) m += w[(size_t)b * D + d];
  m /= (float)B;
  avg[d] = m + beta * (avg[d] - m);
}

// ---- Householder QR of an n x n matrix (n <= 64) in ONE workgroup, LAPACK convention (geqr2: R_jj = -sign(alpha)*norm),
// i.e. what torch.qr / torch.linalg.qr(mode='reduced') return for the 64 x 64 basis of MappingNetwork (custom_layers.py:274-276).
// Replaces ~200 rocSOLVER micro-launches per call.  The reflectors are applied to the augmented matrix [A | I]:
// H_{n-1}..H_0 [A | I] = [R | Q^T], so R and Q come out of the SAME n steps (two waves: thread c < 64 owns column c of A,
// thread 64+r owns column r of the identity block = row r of Q) instead of a second, serial backward accumulation.
// Columns live in registers (the n steps are unrolled so every register index is static); only the current reflector travels
// through LDS, double-buffered so one barrier per step suffices, and is read back as 16 broadcast ds_read_b128.
constexpr int QR_MAX = 64;
// One Householder step with the column index J a compile-time constant: every row predicate (i == J, i > J) folds away, so the
// thread that forms the reflector runs ~3 (64 - J) instructions instead of the ~1000 of a predicated 64-row sweep (that one
// thread was 2/3 of the kernel's 220 us), and the update skips the rows above J, where the reflector is zero.  Rows and
// columns >= n hold zeros (padded at load), so they need no predicate either.  Summation order is that of the predicated
// loops (the skipped terms were exact zeros).
template <int J>
__device__ __forceinline__ void qr_step(float (&a)[QR_MAX], float (*vrow)[QR_MAX], float* taus, int c, bool isq, int n) {
  constexpr int buf = J & 1;
  if (!isq && c == J) {                                     // dlarfg on column J, rows J..n-1
    const float alpha = a[J];
    float xn2 = 0.f;
#pragma unroll
    for (int i = J + 1; i < QR_MAX; ++i) xn2 += a[i] * a[i];
    float tau = 0.f, beta = alpha, sc = 0.f;
    if (xn2 != 0.f) {
      beta = -copysignf(sqrtf(alpha * alpha + xn2), alpha);
      tau = (beta - alpha) / beta;
      sc = 1.f / (alpha - beta);
    }
    vrow[buf][J] = 1.f;
#pragma unroll
    for (int i = J + 1; i < QR_MAX; ++i) vrow[buf][i] = a[i] * sc;
    a[J] = beta;                                            // R_JJ ; rows below the diagonal are not part of R
    taus[buf] = tau;
  }
  __syncthreads();
  if (c < n && (isq || c > J)) {                            // column -= tau v (v^T column), rows J..63
    constexpr int I0 = J & ~3;
    float v[QR_MAX];
#pragma unroll
    for (int k = I0 / 4; k < QR_MAX / 4; ++k) {
      const f32x4 t = *(const f32x4*)(&vrow[buf][4 * k]);
      v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
    }
    float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = J; i < QR_MAX; ++i) w[i & 3] += v[i] * a[i];
    const float wv = ((w[0] + w[1]) + (w[2] + w[3])) * taus[buf];
#pragma unroll
    for (int i = J; i < QR_MAX; ++i) a[i] -= v[i] * wv;
  }
}
template <int J>
__device__ __forceinline__ void qr_steps(float (&a)[QR_MAX], float (*vrow)[QR_MAX], float* taus, int c, bool isq, int n) {
  if constexpr (J < QR_MAX) {
    if (J < n) {                                            // n is uniform: every wave reaches the same barriers
      qr_step<J>(a, vrow, taus, c, isq, n);
      qr_steps<J + 1>(a, vrow, taus, c, isq, n);
    }
  }
}
__global__ __launch_bounds__(2 * QR_MAX) void qr_householder_kernel(const float* __restrict__ A, float* __restrict__ Q,
                                                                     float* __restrict__ R, int n) {
  A += (size_t)blockIdx.x * n * n; Q += (size_t)blockIdx.x * n * n; R += (size_t)blockIdx.x * n * n;   // one matrix per workgroup
  __shared__ __attribute__((aligned(16))) float vrow[2][QR_MAX];
  __shared__ float taus[2];
  const int c = threadIdx.x & (QR_MAX - 1);
  const bool isq = threadIdx.x >= QR_MAX;
  float a[QR_MAX];
#pragma unroll
  for (int i = 0; i < QR_MAX; ++i) a[i] = isq ? ((i == c) ? 1.f : 0.f) : ((i < n && c < n) ? A[i * n + c] : 0.f);
  qr_steps<0>(a, vrow, taus, c, isq, n);
  if (c < n) {
#pragma unroll
    for (int i = 0; i < QR_MAX; ++i) {
      if (i < n) {
        if (isq) Q[c * n + i] = a[i];                       // thread 64+r holds column r of Q^T = row r of Q
        else R[i * n + c] = (i <= c) ? a[i] : 0.f;
      }
    }
  }
}

// ---- multi-tensor kernels: one launch walks a device table of tensors in 64K-element chunks ----------------
struct MTDesc { void* p0; void* p1; void* p2; void* p3; long long n; float f0; float f1; };
constexpr int MT_CHUNK = 65536;
#define MT_ADAM 0
#define MT_EMA 1
#define MT_PACK 2

__global__ __launch_bounds__(TPB) void multi_tensor_kernel(const MTDesc* __restrict__ descs, const int* __restrict__ chunk_tensor,
                                                            const int* __restrict__ chunk_index, int op,
                                                            float a0, float a1, float a2) {
  const int t = chunk_tensor[blockIdx.x];
  const MTDesc d = descs[t];
  const long long base = (long long)chunk_index[blockIdx.x] * MT_CHUNK;
  const long long end = min(base + MT_CHUNK, d.n);
  typedef float f4 __attribute__((ext_vector_type(4)));
  // 16-byte path: every pointer of the tensor 16-byte aligned (chunks start at multiples of 65 536 elements); the tail and unaligned
  // tensors (views into a gradient bucket at odd offsets) take the scalar loops below.  4 bytes per lane ran these streams at 2.5 TB/s.
  const bool al16 = ((((size_t)d.p0) | ((size_t)d.p1) | ((size_t)d.p2) | ((size_t)d.p3)) & 15) == 0;
  const long long vend = al16 ? base + ((end - base) & ~3ll) : base;      // elements [base, vend) go as float4
  if (op == MT_ADAM) {         // a0 = beta1, a1 = beta2, a2 = eps ; f0 = lr / bias_correction1, f1 = 1 / sqrt(bias_correction2)
    float* p = (float*)d.p0; const float* g = (const float*)d.p1; float* m = (float*)d.p2; float* v = (float*)d.p3;
    for (long long i = base + 4 * threadIdx.x; i < vend; i += 4 * TPB) {
      const f4 g4 = *(const f4*)(g + i), m4 = *(const f4*)(m + i), v4 = *(const f4*)(v + i);
      f4 p4 = *(const f4*)(p + i), mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mo[j] = a0 * m4[j] + (1.f - a0) * g4[j];
        vo[j] = a1 * v4[j] + (1.f - a1) * g4[j] * g4[j];
        p4[j] -= d.f0 * mo[j] / (sqrtf(vo[j]) * d.f1 + a2);
      }
      *(f4*)(m + i) = mo; *(f4*)(v + i) = vo; *(f4*)(p + i) = p4;
    }
    for (long long i = vend + threadIdx.x; i < end; i += TPB) {
      const float gi = g[i];
      const float mi = a0 * m[i] + (1.f - a0) * gi;
      const float vi = a1 * v[i] + (1.f - a1) * gi * gi;
      m[i] = mi; v[i] = vi;
      p[i] -= d.f0 * mi / (sqrtf(vi) * d.f1 + a2);
    }
  } else if (op == MT_EMA) {   // p0 = ema (target), p1 = source ; a0 = decay
    float* pe = (float*)d.p0; const float* ps = (const float*)d.p1;
    for (long long i = base + 4 * threadIdx.x; i < vend; i += 4 * TPB) {
      const f4 s4 = *(const f4*)(ps + i);
      f4 e4 = *(const f4*)(pe + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) e4[j] = s4[j] + a0 * (e4[j] - s4[j]);
      *(f4*)(pe + i) = e4;
    }
    for (long long i = vend + threadIdx.x; i < end; i += TPB) { const float sv = ps[i]; pe[i] = sv + a0 * (pe[i] - sv); }
  } else {                     // MT_PACK: p0 = dst, p1 = src ; a0 = scale
    float* dst = (float*)d.p0; const float* src = (const float*)d.p1;
    for (long long i = base + 4 * threadIdx.x; i < vend; i += 4 * TPB) {
      f4 s4 = *(const f4*)(src + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) s4[j] *= a0;
      *(f4*)(dst + i) = s4;
    }
    for (long long i = vend + threadIdx.x; i < end; i += TPB) dst[i] = src[i] * a0;
  }
}

}  // namespace

extern "C" {

// ---- linear layers (single and grouped).  Grouped tables are host arrays of L device pointers / sizes.
static int linear_group_fwd(const float* x, const LinGroup& g, int L, int maxO, int M, int I, int act, float gain, hipStream_t s) {
  const int isplit = (L == 1 && I >= 2048) ? std::min(16, I / 1024) : 1;
  if (isplit > 1) hipMemsetAsync(g.out[0], 0, (size_t)M * g.O[0] * sizeof(float), s);
  if (I % 8 == 0) hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(maxO, 8), cdiv(M, LF_MT) * isplit, L), dim3(TPB), 0, s, x, g, M, I, isplit, act, gain);
  else hipLaunchKernelGGL(linear_fwd_generic_kernel, dim3(cdiv(maxO, 8), cdiv(M, MT) * isplit, L), dim3(TPB), 0, s, x, g, M, I, isplit, act, gain);
  if (isplit > 1)
    hipLaunchKernelGGL(linear_finalize_kernel, dim3(cdiv(M * g.O[0], TPB)), dim3(TPB), 0, s, g.out[0], g.aux[0], g.bscale[0], M, g.O[0],
                       act, gain);
  return launch_status();
}
static int linear_group_bwd_data(const LinGroup& g, int L, int maxO, float* gx, int M, int I, hipStream_t s) {
  const int nchunk = cdiv(maxO, LIN_OC);
  const int acc = (L * nchunk > 1) ? 1 : 0;
  if (acc) hipMemsetAsync(gx, 0, (size_t)M * I * sizeof(float), s);
  hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(cdiv(I, 64), L * nchunk, cdiv(M, MT)), dim3(TPB), 0, s, g, gx, M, I, nchunk, acc);
  return launch_status();
}

int lcgan_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int I, int O,
                     float scale, float bias_scale, int act, float gain, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0) return LCGAN_EINVAL;
  ProfScope p(KID_LINEAR, 2.0 * M * I * O, 4.0 * I * O, s);
  LinGroup g = {};
  g.w[0] = w; g.aux[0] = bias; g.out[0] = y; g.O[0] = O; g.scale[0] = scale; g.bscale[0] = bias_scale;
  return linear_group_fwd(x, g, 1, O, M, I, act, gain, s);
}
int lcgan_linear_bwd_data(const float* gy, const float* w, float* gx, int M, int I, int O, float scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0) return LCGAN_EINVAL;
  ProfScope p(KID_LINEAR, 2.0 * M * I * O, 4.0 * I * O, s);
  LinGroup g = {};
  g.w[0] = w; g.aux[0] = gy; g.O[0] = O; g.scale[0] = scale;
  return linear_group_bwd_data(g, 1, O, gx, M, I, s);
}
int lcgan_linear_wgrad(const float* gy, const float* x, float* gw, int M, int I, int O, float scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_LINEAR, 2.0 * M * I * O, 4.0 * I * O, s);
  LinGroup g = {};
  g.aux[0] = gy; g.out[0] = gw; g.O[0] = O; g.scale[0] = scale;
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(I, TPB), cdiv(O, LW_OT), 1), dim3(TPB), 0, s, g, x, M, I);
  return launch_status();
}
// lcgan_linear_wgrad + lcgan_colsum as ONE launch (gb = bias_scale * column sums of gy): the backward of an EqualizedLinear with bias
int lcgan_linear_wgrad_bias(const float* gy, const float* x, float* gw, float* gb, int M, int I, int O, float scale, float bias_scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!gb) return LCGAN_EINVAL;
  ProfScope p(KID_LINEAR, 2.0 * M * I * O, 4.0 * I * O, s);
  LinGroup g = {};
  g.aux[0] = gy; g.out[0] = gw; g.out2[0] = gb; g.O[0] = O; g.scale[0] = scale; g.bscale[0] = bias_scale;
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(I, TPB), cdiv(O, LW_OT), 1), dim3(TPB), 0, s, g, x, M, I);
  return launch_status();
}
int lcgan_colsum(const float* gy, float* gb, int M, int O, float scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  LinGroup g = {};
  g.aux[0] = gy; g.out[0] = gb; g.O[0] = O; g.bscale[0] = scale;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(O, TPB), 1), dim3(TPB), 0, s, g, M);
  return launch_status();
}
int lcgan_linear_group_fwd(const float* x, const float* const* w, const float* const* bias, float* const* y, const int* O,
                           const float* scale, const float* bias_scale, int L, int M, int I, int act, float gain, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0 || L <= 0 || L > LIN_MAXL) return LCGAN_EINVAL;
  LinGroup g = {};
  int maxO = 0; double sumO = 0;
  for (int l = 0; l < L; ++l) {
    g.w[l] = w[l]; g.aux[l] = bias ? bias[l] : nullptr; g.out[l] = y[l]; g.O[l] = O[l]; g.scale[l] = scale[l];
    g.bscale[l] = bias_scale[l]; maxO = std::max(maxO, O[l]); sumO += O[l];
  }
  ProfScope p(KID_LINEAR, 2.0 * M * I * sumO, 4.0 * I * sumO, s);
  return linear_group_fwd(x, g, L, maxO, M, I, act, gain, s);
}
int lcgan_linear_group_bwd(const float* const* gy, const float* x, const float* const* w, const int* O, const float* scale,
                           const float* bias_scale, int L, int M, int I, float* gx, float* const* gw, float* const* gb, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0 || L <= 0 || L > LIN_MAXL) return LCGAN_EINVAL;
  LinGroup g = {};
  int maxO = 0; double sumO = 0;
  for (int l = 0; l < L; ++l) {
    g.w[l] = w[l]; g.aux[l] = gy[l]; g.O[l] = O[l]; g.scale[l] = scale[l]; g.bscale[l] = bias_scale[l];
    maxO = std::max(maxO, O[l]); sumO += O[l];
  }
  ProfScope p(KID_LINEAR, (gx ? 2.0 : 0.0) * M * I * sumO + (gw ? 2.0 : 0.0) * M * I * sumO, 4.0 * I * sumO, s);
  if (gx) { const int rc = linear_group_bwd_data(g, L, maxO, gx, M, I, s); if (rc) return rc; }
  if (gw) {
    for (int l = 0; l < L; ++l) { g.out[l] = gw[l]; g.out2[l] = gb ? gb[l] : nullptr; }
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(I, TPB), cdiv(maxO, LW_OT), L), dim3(TPB), 0, s, g, x, M, I);
    if (gb) return launch_status();                             // (the bias gradients left the weight-gradient launch)
  }
  if (gb) {
    for (int l = 0; l < L; ++l) g.out[l] = gb[l];
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(maxO, TPB), L), dim3(TPB), 0, s, g, M);
  }
  return launch_status();
}
// L <= 24 linear layers with their OWN inputs x_l [M, I_l] and shapes [O_l, I_l] -- the layers at equal depth of the two mapping networks
// (custom_layers.py:259-287; cnn.py:66-72), which the reference runs as two sequential chains -- in one launch forward and two backward
// (all gx_l; all gw_l + gb_l).  Host arrays of L device pointers / sizes.
int lcgan_linear_multi_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y, const int* I, const int* O,
                           const float* scale, const float* bias_scale, int L, int M, int act, float gain, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0 || L <= 0 || L > LIN_MAXL) return LCGAN_EINVAL;
  LinGroup g = {};
  int maxO = 0, all8 = 1; double fl = 0;
  for (int l = 0; l < L; ++l) {
    if (I[l] <= 0 || I[l] >= 2048) return LCGAN_EINVAL;
    g.xin[l] = x[l]; g.Iv[l] = I[l]; g.w[l] = w[l]; g.aux[l] = bias ? bias[l] : nullptr; g.out[l] = y[l]; g.O[l] = O[l]; g.scale[l] = scale[l];
    g.bscale[l] = bias_scale[l]; maxO = std::max(maxO, O[l]); all8 &= (I[l] % 8 == 0); fl += (double)I[l] * O[l];
  }
  ProfScope p(KID_LINEAR, 2.0 * M * fl, 4.0 * fl, s);
  return linear_group_fwd(x[0], g, L, maxO, M, all8 ? 8 : 7, act, gain, s);     // (the launch's I only selects the kernel: every layer carries its own)
}
int lcgan_linear_multi_bwd(const float* const* gy, const float* const* x, const float* const* w, const int* I, const int* O, const float* scale,
                           const float* bias_scale, int L, int M, float* const* gx, float* const* gw, float* const* gb, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0 || L <= 0 || L > LIN_MAXL) return LCGAN_EINVAL;
  LinGroup g = {};
  int maxO = 0, maxI = 0; double fl = 0;
  for (int l = 0; l < L; ++l) {
    g.w[l] = w[l]; g.aux[l] = gy[l]; g.xin[l] = x[l]; g.Iv[l] = I[l]; g.O[l] = O[l]; g.scale[l] = scale[l]; g.bscale[l] = bias_scale[l];
    maxO = std::max(maxO, O[l]); maxI = std::max(maxI, I[l]); fl += (double)I[l] * O[l];
  }
  ProfScope p(KID_LINEAR, ((gx ? 2.0 : 0.0) + (gw ? 2.0 : 0.0)) * M * fl, 4.0 * fl, s);
  if (gx) {
    const int nchunk = cdiv(maxO, LIN_OC), acc = nchunk > 1 ? 1 : 0;
    for (int l = 0; l < L; ++l) {
      g.out[l] = gx[l];
      if (acc) hipMemsetAsync(gx[l], 0, (size_t)M * I[l] * sizeof(float), s);
    }
    hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(cdiv(maxI, 64), L * nchunk, cdiv(M, MT)), dim3(TPB), 0, s, g, gx[0], M, maxI, nchunk, acc);
  }
  if (gw) {
    for (int l = 0; l < L; ++l) { g.out[l] = gw[l]; g.out2[l] = gb ? gb[l] : nullptr; }
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(maxI, TPB), cdiv(maxO, LW_OT), L), dim3(TPB), 0, s, g, x[0], M, maxI);
  }
  return launch_status();
}
int lcgan_act_bwd_f32(const float* gy, const float* y, float* gz, long long n, int act, float gain, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(act_bwd_f32_kernel, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, gy, y, gz, n, act, gain);
  return launch_status();
}

int lcgan_demod_fwd(const float* sv, const float* wsq, float* d, int B, int C, int O, int Os, float eps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(demod_fwd_kernel, dim3(cdiv(O, 4), B), dim3(TPB), 0, s, sv, wsq, d, B, C, O, Os, eps);
  return launch_status();
}
int lcgan_demod_group(const float* const* sv, const float* const* wsq, float* const* d, const int* C, const int* O, const int* Os,
                      int L, int B, float eps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (L <= 0 || L > DM_MAXL || B <= 0) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 0, s);
  DemodGroup g = {};
  int maxO = 0;
  for (int l = 0; l < L; ++l) { g.s[l] = sv[l]; g.wsq[l] = wsq[l]; g.d[l] = d[l]; g.C[l] = C[l]; g.O[l] = O[l]; g.Os[l] = Os[l]; maxO = std::max(maxO, O[l]); }
  hipLaunchKernelGGL(demod_group_kernel, dim3(cdiv(maxO, 4), B, L), dim3(TPB), 0, s, g, B, eps);
  return launch_status();
}
// gs[b,c] += (demod path) ; gwsq[o,c] = (demod path).  gdq is the raw reduction of lcgan_act_bwd_reduce.
int lcgan_demod_bwd(const float* gdq, const float* d, const float* sv, const float* wsq, float* gs, float* gwsq,
                    int B, int C, int O, int Os, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(demod_bwd_s_kernel, dim3(cdiv(C, TPB), B, cdiv(O, 64)), dim3(TPB), 0, s, gdq, d, sv, wsq, gs, B, C, O, Os);
  hipLaunchKernelGGL(demod_bwd_w_kernel, dim3(cdiv(C, TPB), O), dim3(TPB), 0, s, gdq, d, sv, gwsq, B, C, O, Os);
  return launch_status();
}

int lcgan_bce_fwd(const float* logit, int n, int target_one, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(TPB), 0, s, logit, n, target_one ? -1.f : 1.f, out);
  return launch_status();
}
int lcgan_bce_bwd(const float* logit, int n, int target_one, const float* gout, float* g, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, logit, n, target_one ? -1.f : 1.f, gout, g);
  return launch_status();
}
int lcgan_contrastive_fwd(const float* a, const float* pp, const float* n, int B, int D, float tau, float* tsave, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(contrastive_fwd_kernel, dim3(1), dim3(TPB), 0, s, a, pp, n, B, D, tau, tsave, out);
  return launch_status();
}
int lcgan_contrastive_bwd(const float* a, const float* pp, const float* n, const float* tsave, const float* gout,
                          int B, int D, float tau, float* ga, float* gp, float* gn, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(contrastive_bwd_kernel, dim3(cdiv((long long)B * D, TPB)), dim3(TPB), 0, s, a, pp, n, tsave, gout, B, D, tau, ga, gp, gn);
  return launch_status();
}
int lcgan_l2norm_fwd(const float* x, float* y, float* nsave, int B, int D, float eps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(B, 4)), dim3(TPB), 0, s, x, y, nsave, B, D, eps);
  return launch_status();
}
int lcgan_l2norm_bwd(const float* gy, const float* y, const float* nsave, float* gx, int B, int D, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(B, 4)), dim3(TPB), 0, s, gy, y, nsave, gx, B, D);
  return launch_status();
}
// out[0] += coef * sum |x|^pw   (out zeroed by the caller)
int lcgan_powsum(const float* x, long long n, int pw, float coef, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (pw != 1 && pw != 2) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 4.0 * n, s);
  const int blocks = (int)min((long long)1024, (n + TPB - 1) / TPB);
  hipLaunchKernelGGL(powsum_kernel, dim3(blocks), dim3(TPB), 0, s, x, n, pw, coef, out);
  return launch_status();
}
int lcgan_powsum_bwd(const float* x, long long n, int pw, float coef, const float* gout, float* g, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 8.0 * n, s);
  hipLaunchKernelGGL(powsum_bwd_kernel, dim3(cdiv(n, TPB)), dim3(TPB), 0, s, x, n, pw, coef, gout, g);
  return launch_status();
}
// Q, R of the reduced QR of nb row-major n x n matrices A [nb][n][n] (n <= 64), LAPACK Householder convention; one workgroup each
int lcgan_qr_householder(const float* A, float* Q, float* R, int nb, int n, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n < 1 || n > QR_MAX || nb < 1) return LCGAN_EINVAL;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(qr_householder_kernel, dim3(nb), dim3(2 * QR_MAX), 0, s, A, Q, R, n);
  return launch_status();
}
int lcgan_avg_latent(const float* w, float* avg, int B, int D, float beta, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(KID_SMALL, 0, 0, s);
  hipLaunchKernelGGL(avg_latent_kernel, dim3(cdiv(D, TPB)), dim3(TPB), 0, s, w, avg, B, D, beta);
  return launch_status();
}

// descs: device array of {p0,p1,p2,p3 (8 B each), n (8 B), f0, f1 (4 B each)} = 48 B per tensor;
// chunk_tensor / chunk_index: device int arrays, one entry per 65536-element chunk.
//   op 0 (Adam): p0 param, p1 grad, p2 exp_avg, p3 exp_avg_sq, f0 = lr/bias_corr1, f1 = 1/sqrt(bias_corr2); a0,a1,a2 = beta1,beta2,eps
//   op 1 (EMA):  p0 target, p1 source; a0 = decay           op 2 (pack): p0 dst, p1 src; a0 = scale
int lcgan_multi_tensor(const void* descs, const int* chunk_tensor, const int* chunk_index, int n_chunks, int op,
                       float a0, float a1, float a2, double total_elems, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_chunks <= 0) return LCGAN_OK;
  if (op < 0 || op > 2) return LCGAN_EINVAL;
  const double bpe = op == MT_ADAM ? 28.0 : (op == MT_EMA ? 12.0 : 8.0);
  ProfScope p(KID_OPTIM, 0, bpe * total_elems, s);
  hipLaunchKernelGGL(multi_tensor_kernel, dim3(n_chunks), dim3(TPB), 0, s, (const MTDesc*)descs, chunk_tensor, chunk_index, op, a0, a1, a2);
  return launch_status();
}

}  // extern "C"
