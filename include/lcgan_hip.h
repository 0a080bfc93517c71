/* lcgan_hip.h -- C ABI of liblcgan_hip.so: the MI355X (gfx950) kernels behind the LC-GAN G+D training step.
 *
 * The reference (rakutentech/lcgan) has no FFI: its hot path is Python calling PyTorch ATen.  This header is
 * therefore the boundary a maintainer would bind from Python (ctypes stubs: INTEGRATION.md; the in-tree
 * binding is lcgan_amd/_lib.py).  Every entry point names the reference call it replaces
 * (file:line in the reference checkout).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers; the caller owns every buffer
 *     (outputs and workspaces are allocated by the caller); `stream` is a hipStream_t passed as void*.
 *   - return value: 0 = launched, -1 = invalid argument, -2 = HIP launch error.  Nothing synchronises.
 *   - feature maps are NHWC ([B][H][W][C], C a multiple of 8; logical channels <= C, padding channels are 0),
 *     element type `dtype`: 0 = f32 ("parity mode": 3-way bf16 split, 6 MFMAs per product, fp32-grade), 1 = bf16.
 *   - per-sample channel vectors (styles, demodulation) are f32 [B][C] with the alloc width of the tensor
 *     they scale.  Images crossing the module boundary are f32 NCHW [B][3][H][W].
 *   - act: 0 none, 1 leaky_relu(0.2), 2 tanh;  out = act(v) * gain.
 */
#ifndef LCGAN_HIP_H
#define LCGAN_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- convolution family (bf16 MFMA implicit GEMM) --------------------------------------------------------
 * replaces F.conv2d custom_layers.py:41,43,83 ; F.conv_transpose2d custom_layers.py:78 ; EqualizedWeight
 * scaling custom_layers.py:10,14 ; modulation/demodulation custom_layers.py:62-68 ; and their autograd. */

/* w [A][Bc][k][k] f32 (reference layout) -> wp bf16 [parts][k*k][N][Kpad], N = transpose ? Bc : A,
 * Kpad = roundup32(transpose ? A : Bc); part p holds the p-th bf16 term of the split scale*w = p0 + p1 + p2
 * (parts = 1 for dtype bf16, 3 for dtype f32); wsq [A][Bc] (= sum_taps (scale*w)^2, the demodulation statistic) may be NULL. */
int lcgan_conv_weight_prep(const float* w, int A, int Bc, int k, float scale, int transpose,
                           void* wp, int parts, float* wsq, void* stream);
/* every weight preparation of a network in one launch (EqualizedWeight.forward custom_layers.py:14 for all its convs, both
 * GEMM layouts, + the demodulation statistics).  descs: DEVICE array of 64-byte jobs {const float* w; int64 out_off (bf16
 * elements into out_base); int64 wsq_off (floats into wsq_base); int A, Bc, kk, transpose, parts, N, Kc, Kpad; float scale;
 * int pad}; chunk_entry / chunk_index: device int arrays, one entry per chunk of a job (index >= 0: tile `index` of the
 * prepared weight, 16 rows n x 64 channels c x all taps, row-major over ceil(N/16) x ceil(Kpad/64); index < 0: the
 * 16384-element chunk -index-1 of wsq; a job with kk <= 9 and pad != 0 gets wsq written by its tiles instead and needs no
 * such chunks).  Output layout per job as lcgan_conv_weight_prep. */
int lcgan_conv_weight_prep_group(const void* descs, const int* chunk_entry, const int* chunk_index, int n_chunks,
                                 void* out_base, float* wsq_base, double total_elems, void* stream);
/* gw[a][b][t] = scale*gwp[t][a][b] + 2 scale^2 w[a][b][t] gwsq[a][b]  (w, gwsq may be NULL);
 * transposed != 0 reads gwp as [t][Bc][A] (weight gradient of the transposed convolution) */
int lcgan_conv_wgrad_unprep(const float* gwp, int A, int Bc, int k, float scale, int transposed, const float* w,
                            const float* gwsq, float* gw, void* stream);
/* lcgan_conv_wgrad + lcgan_conv_wgrad_unprep in one call (the slab reduction writes the weight-layout gradient directly);
 * gwp: scratch of k*k*A*Bc floats, contents irrelevant on entry (unlike lcgan_conv_wgrad, which ACCUMULATES into gwp);
 * replaces the same autograd step as the two calls above: torch conv weight gradient, custom_layers.py:41,43,78,83 */
int lcgan_conv_wgrad_fused(const void* x, const void* g, float* gwp,
                           int B, int Hx, int Wx, int Cx, int Hg, int Wg, int Cg, int A, int Bc, int k, int stride,
                           const float* pre_x, const float* pre_g, int dtype,
                           float scale, int transposed, const float* w, const float* gwsq, float* gw, void* stream);
/* y = act(post[b,n] * conv_{k,stride,pad=k/2}(pre[b,c] * x, wp) + bias[n]*bias_scale) * gain + residual
   residual_half = 1: residual is [B,Hout/2,Wout/2,Cout] and enters as 0.25 * residual[ho/2][wo/2], the adjoint of
   F.avg_pool2d(x, 2) (custom_layers.py:202) -- the gradient of a DiscriminatorBlock's pooled skip branch lands in the
   epilogue of conv0's data gradient instead of a separate up-sample + add.
   xs / gs (both or neither; need post, no bias / act / residual): the style-gradient reduction of a modulated conv's backward
   (autograd of custom_layers.py:62-64) fused into its data-gradient launch: with u = the unscaled result, y = post[b,n] * u and
   gs[b,n] += sum_pixels xs[b,p,n] * u[b,p,n]   (xs: [B,Hout,Wout,Cout] = the conv's forward input; gs: [B][Cout] f32, accumulated).
   pool_out (may be NULL; lcgan_conv_fwd only; needs even Hout / Wout): also writes avg_pool2d(y, 2) [B,Hout/2,Wout/2,Cout] -- what the
   NEXT DiscriminatorBlock's skip branch reads (custom_layers.py:202) -- from the output tile while it is still on chip (the closing
   1x1 + residual convolution of a block, custom_layers.py:203,209), or by the pooling kernel where the launch path has no such epilogue. */
int lcgan_conv_fwd(const void* x, const void* wp, void* y,
                   int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                   const float* pre, const float* post, const float* bias, float bias_scale,
                   int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, void* pool_out,
                   int dtype, void* stream);
/* lcgan_conv_fwd with one more optional by-product: mask_out [B*Hout*Wout][Cout/32] 32-bit words, bit c % 32 of word c / 32 =
 * (pre-activation of channel c > 0) -- everything the backward of F.leaky_relu (custom_layers.py:205,208) needs of y, in 1/16 of its bytes
 * (lcgan_act_bwd_reduce_m, lcgan_box3_actbwd_reduce_m).  *mask_written = 1 where the launch path wrote it (bf16 halo-tile kernels, leaky
 * ReLU, no residual / xs, Cout % 32 == 0), else 0 and the caller keeps using y. */
int lcgan_conv_fwd_m(const void* x, const void* wp, void* y,
                     int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                     const float* pre, const float* post, const float* bias, float bias_scale,
                     int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, void* pool_out,
                     void* mask_out, int* mask_written, int dtype, void* stream);
/* adjoint of lcgan_conv_fwd w.r.t. x (weights from weight_prep(transpose=1)); stride 2 == the x2 transposed
 * convolution of ModulatedConv2d(up=2): output [B][Hg*stride][Wg*stride][Cout]. */
int lcgan_conv_bwd_data(const void* g, const void* wpT, void* gx,
                        int B, int Hg, int Wg, int Cg, int Cout, int N, int k, int stride,
                        const float* pre, const float* post, const float* bias, float bias_scale,
                        int act, float gain, const void* residual, int residual_half, const void* xs, float* gs, int dtype, void* stream);
/* gwp[t][a][c] += sum_{b,i,j} (pre_g g)[b,i,j,a] (pre_x x)[b,i*stride+ky-pad,j*stride+kx-pad,c]; gwp f32, zeroed by caller */
int lcgan_conv_wgrad(const void* x, const void* g, float* gwp,
                     int B, int Hx, int Wx, int Cx, int Hg, int Wg, int Cg, int A, int Bc, int k, int stride,
                     const float* pre_x, const float* pre_g, int dtype, void* stream);

/* ---- stencils / elementwise (HBM-bound) -------------------------------------------------------------------
 * box filter F.avg_pool2d(3,1,1) custom_layers.py:136-138,196-198 fused with leaky_relu*gain / tanh (:150-155,:205-206) */
int lcgan_box3_act(const void* x, void* y, int B, int H, int W, int C, int act, float gain, int dtype, void* stream);
int lcgan_box3_act_bwd(const void* gy, const void* y, void* gx, int B, int H, int W, int C, int act, float gain, int dtype, void* stream);
/* backward of  act(conv + bias) -> 3x3 box filter  (DiscriminatorBlock conv0 -> blur, custom_layers.py:204-206) in one pass:
   gz = box3(gy) * act'(y), gbias[c] += sum gz (gbias optional, accumulated) */
int lcgan_box3_actbwd_reduce(const void* gy, const void* y, void* gz, float* gbias, int B, int H, int W, int C, int Clog,
                             int act, float gain, int dtype, void* stream);
/* ... with the sign mask instead of y (see lcgan_act_bwd_reduce_m) */
int lcgan_box3_actbwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, float* gbias, int B, int H, int W, int C, int Clog,
                               int act, float gain, int dtype, void* stream);
/* F.interpolate(x2, nearest) + box filter (+ residual), custom_layers.py:146-147,159 ; x [B,H,W,C] -> y [B,2H,2W,C] */
int lcgan_up2box(const void* x, const void* residual, void* y, int B, int H, int W, int C, int dtype, void* stream);
int lcgan_up2box_bwd(const void* gy, void* gx, int B, int H, int W, int C, int dtype, void* stream);
/* F.avg_pool2d(2,2) custom_layers.py:202 ; H, W are the LARGE (input-side) extents in both calls */
int lcgan_avgpool2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int lcgan_avgpool2_bwd(const void* gy, void* gx, int B, int H, int W, int C, int dtype, void* stream);
/* gz = gy*act'(y) (y = saved OUTPUT); gbias[c] += sum gz; gdq[b,c] += sum_p gz*(act^-1(y/gain) - bias[c]*bias_scale) */
int lcgan_act_bwd_reduce(const void* gy, const void* y, void* gz, const float* bias, float bias_scale,
                         float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream);
/* ... reading the activation's SIGN MASK (lcgan_conv_fwd_m: one bit per element, 1/16 of y's bytes) instead of y: the backward of
 * F.leaky_relu (custom_layers.py:205,208; cnn.py:21) only needs the sign of the pre-activation.  mask: [B*HW][C/8] bytes, bit j of byte v =
 * channel 8 v + j; needs act == leaky ReLU and gdq == NULL; y may then be NULL.  mask == NULL: lcgan_act_bwd_reduce. */
int lcgan_act_bwd_reduce_m(const void* gy, const void* y, const void* mask, void* gz, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream);
/* ... storing gz * oscale[b][c] (oscale: [B][C] fp32; needs gz and gdq) while gbias / gdq reduce the unscaled gz: a ModulatedConv2d's
 * data- and weight-gradient launches both consume d[b,o] * gz (the demodulation of custom_layers.py:72-76 read backwards) and then take
 * their operand without a per-sample scale.  oscale == NULL: lcgan_act_bwd_reduce_m. */
int lcgan_act_bwd_reduce_s(const void* gy, const void* y, const void* mask, void* gz, const float* oscale, const float* bias, float bias_scale,
                           float* gbias, float* gdq, int B, int HW, int C, int Clog, int act, float gain, int dtype, void* stream);
/* style gradient: gs[b,c] += sum_p x*u ; u <- s[b,c]*u in place   (autograd of custom_layers.py:62-64) */
int lcgan_scale_reduce(void* u, const void* x, const float* s, float* gs, int B, int HW, int C, int dtype, void* stream);
/* ... with u <- s*u + res (res may be NULL): the data gradient another consumer of the same tensor already produced joins here
 * (autograd's accumulation of custom_layers.py:145-153: a SynthesisBlock's input feeds skip_layer, flow_layer and modulated_conv0) */
int lcgan_scale_reduce_res(void* u, const void* x, const float* s, float* gs, const void* res, int B, int HW, int C, int dtype, void* stream);
/* bicubic feature warp: get_coordinates + grid_sample(bicubic, zeros, align_corners=False), custom_layers.py:127-134,162-165
 * flow [B,H,W,8] (ch 0 = x, ch 1 = y).  Backward = exact pixel-level CSR transpose (count, scan, fill) + gather, no float
 * atomics, any flow field; workspaces (contents irrelevant on entry), npix = B*H*W (16*npix < 2^31):
 * ws_cnt int[npix+1], ws_off int[npix+1], ws_tiles int[ceil((npix+1)/1024)], ws_ent 8 B x [16*npix]. */
int lcgan_warp_fwd(const void* x, const void* flow, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream);
int lcgan_warp_bwd(const void* gy, const void* x, const void* flow, void* gx, void* gflow,
                   int* ws_cnt, int* ws_off, int* ws_tiles, void* ws_ent,
                   int B, int H, int W, int C, float scale, int dtype, void* stream);
int lcgan_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream);
/* MinibatchStdLayer custom_layers.py:243-256 (G = min(8,N), strided groups); x [N][HW][C] -> y [N][HW][Cy], Cy > C */
int lcgan_mbstd_fwd(const void* x, void* y, int N, int G, int HW, int C, int Cy, int dtype, void* stream);
int lcgan_mbstd_bwd(const void* gy, const void* x, void* gx, int N, int G, int HW, int C, int Cy, int dtype, void* stream);
int lcgan_mbstd_bwd2(const void* v, const void* gy, const void* x, void* ggy, void* gx2,
                     int N, int G, int HW, int C, int Cy, int dtype, void* stream);
/* 1x1 convs touching the f32 NCHW image: fromRGB cnn.py:20-21, toRGB custom_layers.py:175,181; w f32 [Bw][3][C] */
/* (lcgan_rgb_expand: pooled != NULL also writes avg_pool2d(y, 2) [B][HW/4][C], the first DiscriminatorBlock's skip input
 * custom_layers.py:202, in the same pass; W = image width, used only then) */
int lcgan_rgb_expand(const float* img, const float* w, const float* bias, float bias_scale, void* y,
                     int B, int HW, int C, int Clog, int per_sample, int act, float gain, void* pooled, int W, int dtype, void* stream);
int lcgan_rgb_reduce(const void* x, const float* w, const float* bias, float bias_scale, float* img,
                     int B, int HW, int C, int per_sample, int dtype, void* stream);
int lcgan_rgb_wgrad(const float* img, const void* feat, float* gw, int B, int HW, int C, int per_sample, int dtype, void* stream);
/* fused backward passes of the same two layers (autograd of cnn.py:20-21 and of custom_layers.py:177-182):
 * lcgan_rgb_expand_bwd: gz = gy * act'(y) in registers -> gimg = sum_c gz w (written; may be NULL), gw += sum_p img gz (may be NULL),
 *   gbias += sum gz (may be NULL): one pass instead of activation backward + rgb_reduce + rgb_wgrad;
 * lcgan_rgb_reduce_bwd_act: image gradient -> the preceding conv's pre-activation gradient: gz = (sum_o gimg wm) * act'(y) (written),
 *   gbias += sum gz, gdq[b,c] += sum_p gz (ypre - bias) (the demodulation statistic of ModulatedConv2d), gwm += sum_p gimg y. */
int lcgan_rgb_expand_bwd(const void* gy, const void* y, const float* img, const float* w, float* gimg, float* gw, float* gbias,
                         int B, int HW, int C, int Clog, int per_sample, int act, float gain, int dtype, void* stream);
/* ... recompute = 1 (leaky ReLU, img != NULL): the sign the activation backward needs is recomputed from the image, the weights and the
 * forward layer's bias (fbias * fbias_scale; may be NULL) -- cnn.py:20-21 is a 3-term dot product per channel -- and y is not read. */
int lcgan_rgb_expand_bwd_r(const void* gy, const void* y, const float* img, const float* w, const float* fbias, float fbias_scale, int recompute,
                           float* gimg, float* gw, float* gbias,
                           int B, int HW, int C, int Clog, int per_sample, int act, float gain, int dtype, void* stream);
int lcgan_rgb_reduce_bwd_act(const float* gimg, const void* y, const float* wm, const float* bias, float bias_scale, void* gz,
                             float* gbias, float* gdq, float* gwm, int B, int HW, int C, int Clog, int per_sample, int act, float gain,
                             int dtype, void* stream);
/* ... storing gz * oscale[b][c] (oscale [B][C] fp32 or NULL), the reductions unscaled: see lcgan_act_bwd_reduce_s */
int lcgan_rgb_reduce_bwd_act_s(const float* gimg, const void* y, const float* wm, const float* bias, float bias_scale, void* gz, const float* oscale,
                               float* gbias, float* gdq, float* gwm, int B, int HW, int C, int Clog, int per_sample, int act, float gain,
                               int dtype, void* stream);
/* the 2-channel flow layer of a SynthesisBlock (ModulatedConv2d(Cin -> 2, k 3, up 2), custom_layers.py:123,149-151; F.conv_transpose2d
 * :73-80) as a 1x1 convolution Cin -> 18 on the low-resolution grid (lcgan_conv_fwd with the [18][Cin] weight (ky*3+kx)*2+o) followed by
 * lcgan_flow_col2im: u[b,2i-1+ky,2j-1+kx,o] += d[b,o] * t[b,i,j,(ky*3+kx)*2+o] (+ bias[o]); lcgan_flow_im2col is its adjoint
 * (gt = d * gather(gu)) for the backward.  t / gt: [B,H,W,24]; u / gu: [B,2H,2W,8]; d: f32 [B][dstride]. */
int lcgan_flow_col2im(const void* t, const float* d, const float* bias, void* u, int B, int H, int W, int dstride, int dtype, void* stream);
int lcgan_flow_im2col(const void* gu, const float* d, void* gt, int B, int H, int W, int dstride, int dtype, void* stream);
/* layout converts at the NCHW f32 boundary (const input cnn.py:106, flatten custom_layers.py:232 / cnn.py:39) */
int lcgan_nchw_to_nhwc(const float* src, void* dst, int B, int HW, int C, int Clog, int bcast, int dtype, void* stream);
int lcgan_nhwc_to_nchw(const void* src, float* dst, int B, int HW, int C, int Clog, int reduce, int dtype, void* stream);

/* ---- small f32 linears: EqualizedLinear custom_layers.py:24-25 (+ autograd) ------------------------------- */
int lcgan_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int I, int O,
                     float scale, float bias_scale, int act, float gain, void* stream);
int lcgan_linear_bwd_data(const float* gy, const float* w, float* gx, int M, int I, int O, float scale, void* stream);
int lcgan_linear_wgrad(const float* gy, const float* x, float* gw, int M, int I, int O, float scale, void* stream);
int lcgan_colsum(const float* gy, float* gb, int M, int O, float scale, void* stream);
/* the two calls above as one launch: gw = scale * gy^T x, gb = bias_scale * colsum(gy) (backward of F.linear with bias, custom_layers.py:25) */
int lcgan_linear_wgrad_bias(const float* gy, const float* x, float* gw, float* gb, int M, int I, int O, float scale, float bias_scale, void* stream);
/* L <= 24 linear layers sharing their input x [M,I] -- the style affines of every synthesis layer (SynthesisLayer.linear,
   custom_layers.py:100,108; all blocks receive the same latent, cnn.py:103-104) in ONE launch.  w/bias/y/gy/gw/gb are HOST
   arrays of L device pointers, O/scale/bias_scale host arrays of L values; y_l = x w_l^T scale_l + bias_l*bias_scale_l.
   The backward entry writes gx = sum_l gy_l w_l scale_l (if gx), gw_l = scale_l gy_l^T x (if gw), gb_l = bias_scale_l colsum(gy_l) (if gb). */
int lcgan_linear_group_fwd(const float* x, const float* const* w, const float* const* bias, float* const* y, const int* O,
                           const float* scale, const float* bias_scale, int L, int M, int I, int act, float gain, void* stream);
int lcgan_linear_group_bwd(const float* const* gy, const float* x, const float* const* w, const int* O, const float* scale,
                           const float* bias_scale, int L, int M, int I, float* gx, float* const* gw, float* const* gb, void* stream);
/* L <= 24 linear layers with their OWN inputs x_l [M, I_l] and shapes [O_l, I_l]: the layers at equal depth of the geometry and the appearance
   mapping network (MappingNetwork.forward, custom_layers.py:283-287; cnn.py:66-72), which the reference evaluates as two sequential chains.
   x/w/bias/y/gy/gx/gw/gb are HOST arrays of L device pointers, I/O/scale/bias_scale host arrays of L values. */
int lcgan_linear_multi_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y, const int* I, const int* O,
                           const float* scale, const float* bias_scale, int L, int M, int act, float gain, void* stream);
int lcgan_linear_multi_bwd(const float* const* gy, const float* const* x, const float* const* w, const int* I, const int* O, const float* scale,
                           const float* bias_scale, int L, int M, float* const* gx, float* const* gw, float* const* gb, void* stream);
int lcgan_act_bwd_f32(const float* gy, const float* y, float* gz, long long n, int act, float gain, void* stream);
/* demodulation statistic custom_layers.py:67 : d[b,o] = rsqrt(sum_c s^2 wsq[o,c] + eps), d is [B][Os] */
int lcgan_demod_fwd(const float* s, const float* wsq, float* d, int B, int C, int O, int Os, float eps, void* stream);
/* L <= 24 demodulation vectors in one launch (host arrays of L device pointers / sizes): every modulated layer of a generator pass, whose
   styles are all known before the first convolution (cnn.py:103-104) */
int lcgan_demod_group(const float* const* s, const float* const* wsq, float* const* d, const int* C, const int* O, const int* Os,
                      int L, int B, float eps, void* stream);
int lcgan_demod_bwd(const float* gdq, const float* d, const float* s, const float* wsq, float* gs, float* gwsq,
                    int B, int C, int O, int Os, void* stream);

/* ---- losses: worker.py:156-157,191 (BCE with logits), loss.py:9-15 (contrastive), cnn.py:40-41 (normalize),
 *      worker.py:207-209 (L1 sparsity, pw=1), loss.py:20-23 (R1 square sum, pw=2) ---------------------------- */
int lcgan_bce_fwd(const float* logit, int n, int target_one, float* out, void* stream);
int lcgan_bce_bwd(const float* logit, int n, int target_one, const float* gout, float* g, void* stream);
int lcgan_contrastive_fwd(const float* a, const float* p, const float* n, int B, int D, float tau, float* tsave, float* out, void* stream);
int lcgan_contrastive_bwd(const float* a, const float* p, const float* n, const float* tsave, const float* gout,
                          int B, int D, float tau, float* ga, float* gp, float* gn, void* stream);
int lcgan_l2norm_fwd(const float* x, float* y, float* nsave, int B, int D, float eps, void* stream);
int lcgan_l2norm_bwd(const float* gy, const float* y, const float* nsave, float* gx, int B, int D, void* stream);
int lcgan_powsum(const float* x, long long n, int pw, float coef, float* out, void* stream);          /* out zeroed by caller */
int lcgan_powsum_bwd(const float* x, long long n, int pw, float coef, const float* gout, float* g, void* stream);
/* torch.qr(tanh(basis))[0] custom_layers.py:274-276: Householder QR (LAPACK sign convention) of nb n x n matrices [nb][n][n],
 * n <= 64, one workgroup per matrix (the geometry and the appearance mapping network share one launch) */
int lcgan_qr_householder(const float* A, float* Q, float* R, int nb, int n, void* stream);
/* cnn.py:95-97 */
int lcgan_avg_latent(const float* w, float* avg, int B, int D, float beta, void* stream);

/* ---- multi-tensor optimiser kernels: torch.optim.Adam worker.py:98-110 (op 0), Ema.update ema.py:19-32 (op 1),
 *      gradient bucket pack for the RCCL all-reduce that replaces DDP's reducer worker.py:88-96 (op 2) -------- */
int lcgan_multi_tensor(const void* descs, const int* chunk_tensor, const int* chunk_index, int n_chunks, int op,
                       float a0, float a1, float a2, double total_elems, void* stream);

/* tuning switches (A/B tests inside one process; every one only ROUTES between kernels that compute the same result); returns the
 * previous value, LCGAN_EINVAL for an unknown option.  0 bf16 halo-tile conv kernels on/off; 1 split-K for small-M convolutions;
 * 2 explicit workgroup target of the row-segment weight gradient (0 = cost model); 3 debug bits; 4 16x16x32 MFMA shape; 5 narrow /
 * 1x1 routing of the row-segment weight gradient; 6 smallest halo grid before the split-K GEMM takes over; 7 smallest grid of the
 * narrow-layer kernel; 8 smallest split count that reduces through a slab; 9 packed channel groups; 10 LDS-DMA staging of the halo
 * kernel (0 off, 1 / 2 = taps per barrier); 11 the same record layout for convolutions with per-sample input scales; 12 LDS-DMA
 * weight-gradient kernel (0 off, 1, 2 = split for two workgroups per CU, 3 = also stride 2); 13 parity-plane stride-2 forward;
 * 14 KB of weights concurrent channel blocks of one tile may keep in an XCD's L2 (0 = one input pass per channel block);
 * 15 XCD-grouped weight-gradient workgroup order (on: operand bytes cross the fabric 1.0-1.7x instead of 3-5x); 16 LDS-DMA staging of the generic implicit-GEMM kernel (2 = modulated low-resolution layers too, through one prescale pass; 3 = its eight-wave form, 4 = with four stages);
 * 17 MB of operands up to which a weight gradient with per-sample scales applies them by one elementwise pass and reduces the whole batch as one range (0 = never);
 * 18 MB of per-sample weight copies up to which a convolution with per-sample input scales folds them into the weights (0 = never);
 * 19 split-K launches of the eight-wave generic kernel with at most this many splits exchange partials through per-split slabs and the last split to arrive finishes the tile (more, or 0: atomics + a finalize launch);
 * 20 launch plan of the small-grid (8 x 8, 16 x 16) weight gradients: splits from a measured cost model, a single split writes the gradient in weight layout from its epilogue (0 = the round-2 plan);
 * 21 forced split count of the small-grid weight gradients (tuning; 0 = automatic);
 * 22 workgroup count up to which a halo-tile launch that option 6 would turn away splits its input-channel range instead (0 = never);
 * 23 one-pass weight gradient of the flow layer's 1x1 GEMM (0 = the row-segment kernel);
 * 24 the four sub-pixel phases of a transposed-convolution tile as neighbours in one XCD's queue instead of grid.z planes (off: fabric reads -3.6x, time 0 ... +25 %);
 * 25 partial weight-gradient tiles of a bf16 launch with at least 8 splits cross memory as bf16 (fp32 accumulation within a split and across them);
 * 26 plain stride-2 forward convolutions on the kernel with two anti-phased teams per 1024-thread workgroup (conv_s2duo_kernel).
 * DESIGN.md section 2 ("Switches") has the measurements behind the defaults. */
int lcgan_set_option(int option, int value);

/* ---- per-launch HIP-event profiling (bench.py roofline) ---------------------------------------------------- */
int lcgan_prof_enable(int on);
int lcgan_prof_collect(double* out_ms, double* out_flops, double* out_bytes, long long* out_count);
int lcgan_prof_active(void);
/* one CSV row per recorded launch, in launch order: kid, ms, flops, bytes, tag (convolutions tag their geometry); clears the records */
int lcgan_prof_dump(const char* path);

/* ---- MX-fp8 convolution path (BASELINE configs[4]: fp8 MFMA operands, fp32 accumulate, bf16 feature maps) ------------------
 * the same reference calls as lcgan_conv_fwd / lcgan_conv_bwd_data (F.conv2d custom_layers.py:41,43,83; F.conv_transpose2d :78;
 * modulation :62-72) with OCP e4m3 operands and one E8M0 power-of-two scale per 32 reduction channels (v_mfma_scale_f32_32x32x64_f8f6f4):
 * activations are quantised while they are staged, weights once per optimiser step by lcgan_conv_weight_prep_fp8.
 * wp: e4m3 [k*k][N][K64][64], wsc: E8M0 [k*k][N][K64][2], K64 = ceil(reduction channels / 64) (layout: csrc/conv_fp8.hip).
 * Restrictions: bf16 feature maps, output grids of at least 16 x 16 positions, no tanh, no fused style-gradient reduction. */
int lcgan_conv_weight_prep_fp8(const float* w, int A, int Bc, int k, float scale, int transpose, void* wp, void* wsc, void* stream);
int lcgan_conv_fwd_fp8(const void* x, const void* wp, const void* wsc, void* y,
                       int B, int Hin, int Win, int Cin, int Cout, int N, int k, int stride,
                       const float* pre, const float* post, const float* bias, float bias_scale,
                       int act, float gain, const void* residual, int residual_half, void* stream);
int lcgan_conv_bwd_data_fp8(const void* g, const void* wpT, const void* wscT, void* gx,
                            int B, int Hg, int Wg, int Cg, int Cout, int N, int k, int stride,
                            const float* pre, const float* post, const float* bias, float bias_scale,
                            int act, float gain, const void* residual, int residual_half, void* stream);

/* ---- device-side training views (the data step in front of the hot path) ---------------------------------------------
 * replaces custom_dataset.py:59-88 (h-flip :68, albumentations Perspective :22-23,27-33, CoarseDropout :24 / ColorJitter :19-21,
 * normalisation :81-86), which the reference runs on the host in DataLoader workers (worker.py:37,62-69).
 * src: f32 [B][3][R][R] in [-1,1] (the resized image); params: f32 [B][32], one row of host-drawn randomness per sample
 * (layout: lcgan_amd/csrc/views.hip); outputs f32 [B][3][R][R]: the flipped image, its perspective view (bilinear, black
 * border) and its appearance view (one black rectangle, or brightness / contrast / saturation / hue jitter).
 * params slot 23 < 0: the contrast pivot is computed on the device (mean luma after the jitter ops that precede the contrast op,
 * accumulated into slot 24, which must be 0 on entry and is WRITTEN); slot 25 != 0: the two augmented views are rounded to the
 * uint8 grid k/255 (custom_dataset.py:76-79: Image.fromarray + ToTensor of albumentations' uint8 result). */
int lcgan_make_views(const float* src, float* params, float* out_img, float* out_geo, float* out_app,
                     int B, int R, void* stream);

#ifdef __cplusplus
}
#endif
#endif
