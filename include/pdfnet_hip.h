/* libpdfnet_hip.so -- C ABI of the MI355X-native (gfx950) PDFNet RGB-D two-hand hot path.
 *
 * Plain pointers and sizes only: every pointer is DEVICE memory (hipMalloc / a torch tensor's
 * data_ptr()), `stream` is a hipStream_t passed as void*, nothing here allocates, synchronises or
 * touches the default stream, so every call is hipGraph-capturable.  Return value: 0 on success,
 * PDF_E_* (<0) for a rejected argument, or a positive hipError_t from the launch.
 *
 * Layout convention: activations are row-major [rows][channels] with an explicit row stride `ld*`
 * in floats (NHWC for images: rows = n*H*W pixels).  Conv weights are [Cout][KH][KW][Cin] -- the
 * channels_last storage of the reference's OIHW nn.Conv2d weight; ConvTranspose weights are
 * [Cin][KH][KW][Cout] (channels_last storage of [Cin,Cout,KH,KW]).
 *
 * Each entry point names the reference interface (file:line under zijinxuxu/PDFNet @2024_08_07) it
 * replaces.  The reference has no native op on this path (its one .cu, lib/utils/roi_align, is dead
 * code), so the "FFI" a maintainer binds is these symbols via ctypes -- see INTEGRATION.md.
 */
#ifndef PDFNET_HIP_H
#define PDFNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define PDF_E_BADARG (-1)
#define PDF_E_WORKSPACE (-2)
#define PDF_ACT_NONE 0
#define PDF_ACT_RELU 1
#define PDF_ACT_LRELU01 2

/* ---- dense contractions: fp32 MFMA implicit GEMM (csrc/gemm.hip) ------------------------------ */

/* y[M][N] = act(x[M][K] w[N][K]^T + bias).  nn.Linear / 1x1 nn.Conv2d call sites:
 * model_attn/gcn.py:66 (cl(x)), self_attn.py:66-68,79, intaghand_encoder.py:48-103 (netR_*), :213-218 (SFT convs). */
int pdf_linear_fwd(const float* x, const float* w, const float* bias, float* y,
                   int M, int N, int K, int ldx, int ldw, int ldy, int act, void* stream);
/* dW[N][K] (+)= dy[M][N]^T x[M][K]; ws >= pdf_wgrad_workspace_floats(M,N,K) floats. (autograd of the above)
 * db != NULL: also db[N] (+)= column sums of dy (the bias gradient rides along in the same launches). */
int pdf_linear_bwd_weight(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                          int M, int N, int K, int ldx, int lddy, int accumulate, void* stream);
long pdf_wgrad_workspace_floats(int M, int NI, int NJ);
/* measurement aids (bench.py): BM*1000+BN of the calling thread's last implicit-GEMM launch (0 = streaming small-K
 * kernel), and the number of implicit-GEMM kernels it has launched so far (one entry point may launch several) */
/* Allocates the library's ticket-counter ring (csrc/common.h pdf_last_block_arrives: in-launch finalisation of BatchNorm
 * statistics and weight-gradient slabs).  Called once per process after the device is selected, outside any stream capture;
 * the entry points call it lazily otherwise. */
int pdf_init(void);
/* The device the rings were created on (-1 before pdf_init).  One process drives one GPU: pdf_init from another current
 * device returns PDF_E_WORKSPACE. */
int pdf_debug_init_device(void);
/* Per-kernel timing for bench.py's roofline object: while on, every GEMM-family kernel launch is bracketed by two events on
 * its launch stream and recorded under its symbol name (as rocprofv3 prints it, without the `void` / argument list) together
 * with the algorithmic FLOPs and bytes (every operand once) of that launch.  on != 0 clears the records and starts, 0 stops.
 * pdf_debug_kernel_record reads record i once the device is idle. */
int pdf_debug_kernel_timing(int on);
/* BatchNorm statistics out of the producing GEMM's epilogue (reference: nn.Conv2d -> nn.BatchNorm2d pairs, resnet.py:102-122,
 * intaghand_encoder.py:48-103,742-743): pdf_set_stats_output hands the NEXT pdf_conv2d_fwd / pdf_linear_fwd call of this thread
 * a buffer for per-row-block (mean, sum of squared deviations) pairs of its output columns, part[(t * N + c) * 2 + {0,1}];
 * afterwards pdf_stats_result_tiles() / pdf_stats_result_rows() give the number of row blocks and rows per block it wrote
 * (0 tiles: the kernel it chose has no statistics epilogue, or cap_floats < tiles * N * 2).  pdf_set_bn_tile_stats hands
 * such partials to the NEXT pdf_bn_train_fwd / pdf_bn_relu_maxk_fwd call, which then skips its own statistics pass. */
int pdf_set_stats_output(float* part, long cap_floats);      /* compat: arms PdfCallOpts::stats_out for the next plain call (see `_x` forms below) */
long pdf_stats_result_tiles(void);
long pdf_stats_result_rows(void);
int pdf_set_bn_tile_stats(const float* part, long tiles, long rows_per_tile);
int pdf_debug_kernel_record_count(void);
int pdf_debug_kernel_record(int i, char* name, int cap, double* flops, double* bytes, float* ms);
/* Stream fork / join for the host layer: everything issued on `signaler` so far completes before anything issued on
 * `waiter` afterwards (hipEventRecord + hipStreamWaitEvent on a ring of timing-disabled events created by pdf_init; valid
 * inside a stream capture).  The reference has no counterpart -- it runs on one stream. */
int pdf_stream_wait(void* waiter, void* signaler);
/* Operand precision of every GEMM-family entry point below: 0 (default) = fp32-input MFMA, exact fp32; 1 = operands rounded
 * to bf16 (RNE) while staged into LDS, bf16 MFMA with fp32 accumulation (BASELINE configs 4 / 5: bf16 compute, fp32 master
 * weights and fp32 normalisation / loss statistics).  Process-wide; set before the first step. */
int pdf_set_gemm_precision(int bf16);
int pdf_debug_gemm_precision(void);
int pdf_debug_last_tile(void);
int pdf_debug_igemm_launches(void);
/* dx[M][K] = dy[M][N] w[N][K] (autograd of nn.Linear wrt its input); w is read in its forward storage */
int pdf_linear_bwd_data(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, int ldw, int lddx, void* stream);
/* Paired forms: two same-shaped layers with their own parameters in one launch -- the left / right hand branches of the
 * mesh decoder (model_attn/DualGraph.py:83-84 graph_left/graph_right, inter_attn.py:66-67 L_/R_self_attn_layer, ffL/ffR).
 * Rows [0, M) of x / y / dy / dx belong to (w0, b0), rows [M, 2M) to (w1, b1).
 * pdf_linear_bwd_weight_pair: ws >= 2 * pdf_wgrad_workspace_floats(M, N, K). */
int pdf_linear_fwd_pair(const float* x, const float* w0, const float* w1, const float* b0, const float* b1, float* y,
                        int M, int N, int K, int ldx, int ldw, int ldy, int act, void* stream);
int pdf_linear_bwd_data_pair(const float* dy, const float* w0, const float* w1, float* dx, int M, int N, int K,
                             int lddy, int ldw, int lddx, void* stream);
int pdf_linear_bwd_weight_pair(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1,
                               float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, void* stream);

/* nn.Conv2d forward on NHWC: lib/models/networks/resnet.py:202-218 (trunk), intaghand_encoder.py:602 (p2),
 * :617 (feat), :621 (e_conv1), :627-628 (center_feat_up0/1), :675-693 (heads), :270-316 (decoders). */
int pdf_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                   int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                   int stride, int pad, int OH, int OW, int ldy, int act, void* stream);
/* dx from dy and the forward weight w = [Cout][KH][KW][Cin] (no transposed copy); caller zero-fills dx when stride > kernel */
int pdf_conv2d_bwd_data(const float* dy, const float* w, float* dx,
                        int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                        int stride, int pad, int OH, int OW, int lddy, void* stream);
/* dx += ... (stride 1): accumulates into an existing gradient -- the autograd add of a two-consumer tensor (a ResNet block input:
 * resnet.py:100-122, `out += identity`) done by the second producer's epilogue */
int pdf_conv2d_bwd_data_add(const float* dy, const float* w, float* dx,
                        int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                        int stride, int pad, int OH, int OW, int lddy, void* stream);
int pdf_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                          int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                          int stride, int pad, int OH, int OW, int lddy, int accumulate, void* stream);

/* nn.ConvTranspose2d (pyramid laterals p3/p4/p5, intaghand_encoder.py:603-605,721-729).
 * w = the weight in its natural [Cin][KH][KW][Cout] (channels_last) storage in all three entry points. */
int pdf_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                     int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                     int stride, int pad, int OH, int OW, int ldy, void* stream);
int pdf_deconv2d_bwd_data(const float* dy, const float* w, float* dx,
                          int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW,
                          int stride, int pad, int OH, int OW, int lddy, void* stream);
int pdf_deconv2d_bwd_weight(const float* x, const float* dy, float* dw, float* ws, long ws_floats,
                            int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW,
                            int stride, int pad, int OH, int OW, int lddy, int accumulate, void* stream);
/* C[b][m][n] (+)= sum_{r<RB} sum_k A[b,r][m][k] B[b,r][k][n], element strides (0 = broadcast).
 * Small matmuls: avg_head / unsample_layer (intaghand_decoder.py:205,224), full_regressor (Mano_model.py:309-323). */
int pdf_bmm_strided(const float* A, const float* B, float* C, int batch, int M, int N, int K, int RB,
                    long sab, long sar, long sam, long sak, long sbb, long sbr, long sbk, long sbn,
                    long scb, long scm, long scn, int beta, void* stream);

/* ---- PointNet++ set abstraction (csrc/pointops.hip) ------------------------------------------ */

/* group_points / group_points_2 (lib/utils/utils.py:134-163, :165-188): centroids = first S points,
 * kNN(K) on squared distances + ball mask (d2 > r2 -> centroid's own index) + gather + centre-subtract
 * of channels 0:3, one launch.  pts [Bc][N][ldp] (xyz = channels 0:3, C channels), idx [Bc][S][K] int32,
 * grouped [Bc][S][K][ldg] (channels C..ldg-1 zero-filled) or NULL. */
int pdf_knn_ball_group(const float* pts, int ldp, int C, int Bc, int N, int S, int K, float r2,
                       int* idx, float* grouped, int ldg, void* stream);
/* autograd of the gather (values only; indices carry no gradient): dpts must be zero-filled */
int pdf_group_bwd(const float* dg, int ldg, const int* idx, float* dpts, int ldd, int C,
                  int Bc, int N, int S, int K, void* stream);
/* First 1x1 convolution of a set-abstraction MLP applied BEFORE the grouping (group_points + netR_*[0], lib/utils/utils.py:
 * 134-188 + intaghand_encoder.py:48-52): with u = conv(points) [Bc][N][ldu] (bias included) and v = W[:, :3] centre_xyz
 * [Bc][S][ldv], y[b,s,k,:] = u[b, idx[b,s,k], :] - v[b,s,:] equals the convolution of the centre-subtracted grouped block,
 * which is never materialised.  bwd: du (caller zero-fills) += scatter of dy, dv = -sum_k dy.  C % 4 == 0, C <= 256. */
int pdf_gather_sub_fwd(const float* u, int ldu, const float* v, int ldv, const int* idx, int Bc, int N, int S, int K, int C,
                       float* y, int ldy, void* stream);
int pdf_gather_sub_bwd(const float* dy, int lddy, const int* idx, float* du, int ldu, float* dv, int ldv,
                       int Bc, int N, int S, int K, int C, void* stream);
/* Deterministic form of the same backward (no float atomics; du needs no zero fill): pdf_invert_index turns idx [Bc][E = S*K]
 * (values in [0, N)) into per-point slot lists -- start [Bc][N + 1], list [Bc][E] ascending within a point's segment, tmp [Bc][E]
 * unused (kept for ABI stability) -- once per forward; pdf_gather_sub_bwd_sorted sums each point's rows of dy in list order.
 * A stable counting sort in LDS, one block per cloud: (2 N + 1) * 4 <= 160 KiB, i.e. N <= 20,479; any E. */
int pdf_invert_index(const int* idx, int Bc, int N, int E, int* start, int* list, int* tmp, void* stream);
int pdf_gather_sub_bwd_sorted(const float* dy, int lddy, const int* start, const int* list, float* du, int ldu, float* dv, int ldv,
                              int Bc, int N, int S, int K, int C, void* stream);
/* _tranpose_and_gather_feat (lib/models/utils.py:22-26) on an NHWC map: out[b][m][:] = feat[b][ind'[b][m]][:],
 * ind' = pyramid index of intaghand_encoder.py:125-126 when shift > 0.  ind is int64 [B][>=M] with batch stride. */
int pdf_gather_rows(const float* feat, int ldf, int C, long HW, const long* ind, long ind_bstride,
                    int B, int M, int R, int shift, float* out, int ldo, void* stream);
int pdf_scatter_rows_add(const float* dout, int ldo, int C, long HW, const long* ind, long ind_bstride,
                         int B, int M, int R, int shift, float* dfeat, int ldf, void* stream);
/* 5x5 / 3x3 windows around the centre pixels for the exact sparse evaluation of center_feat_up0 -> center_feat_up1 ->
 * _tranpose_and_gather_feat (intaghand_encoder.py:790-792).  buf is [B*M][win][win][C], win = 2r+1.
 * mode 0: buf = windows of feat (zero outside); 1: feat += buf (backward); 2: buf = inside-image ? src : 0. */
int pdf_window_op(float* feat, int ldf, int C, int H, int W, const long* ind, long ind_bstride,
                  int B, int M, int r, float* buf, const float* src, int mode, void* stream);
/* centre decode of the test path: _nms(5x5) + _topk(K=1) per (sample, channel) map (intaghand_encoder.py:349-367,750-758;
 * lib/trains/simplified.py:378-384).  hm [BC][H][W] plain, ind int64 [BC], score [BC] or NULL. */
int pdf_nms_top1(const float* hm, int BC, int H, int W, long* ind, float* score, void* stream);
/* nn.MaxPool2d((1,K)) / ((S2,1)) of netR_1/2/3 (intaghand_encoder.py:62,82,100): x [R][K][ldx] -> y [R][ldy], arg [R][C] */
int pdf_maxk_fwd(const float* x, int ldx, int C, long R, int K, float* y, int ldy, int* arg, void* stream);
int pdf_maxk_bwd(const float* dy, int ldy, const int* arg, int C, long R, int K, float* dx, int ldx, void* stream);
/* BatchNorm2d -> ReLU -> MaxPool2d over the K neighbours, the tail of every set-abstraction MLP (intaghand_encoder.py:59-62,
 * 79-82,97-100), on the convolution output y [R][K][ldy] without writing the normalised tensor: out [R][ldo], arg int32 [R][C].
 * training: batch statistics over the R*K rows (+ running-statistics update); else running statistics.  save_mean / save_rstd /
 * scale / shift [C] feed the backward, which rebuilds d y [R][K][lddy] from (dout [R][lddo], arg, y) -- the two BatchNorm sums
 * only involve the R*C selected elements -- and (accumulates) dgamma / dbeta.  C % 4 == 0.
 * ws: pdf_bn_workspace_floats(C, R*K) floats (fwd), pdf_bn_workspace_floats(C, R) + 3*C (bwd). */
int pdf_bn_relu_maxk_fwd(const float* y, int ldy, int C, long R, int K, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int training,
                         float* out, int ldo, int* arg, float* save_mean, float* save_rstd, float* scale, float* shift,
                         float* ws, void* stream);
int pdf_bn_relu_maxk_bwd(const float* dout, int lddo, const int* arg, const float* y, int ldy, const float* save_mean, const float* save_rstd,
                         const float* gamma, const float* scale, const float* shift, int C, long R, int K,
                         float* dy, int lddy, float* dgamma, float* dbeta, int accumulate, float* ws, void* stream);

/* ---- normalisation (csrc/norm.hip) ------------------------------------------------------------ */

/* nn.BatchNorm2d/1d in train mode (+ optional residual add and ReLU): resnet.py:100-122, intaghand_encoder.py:48-103,
 * :618 (feat_bn, momentum 0.01), :192-198, :304-306.  ws >= pdf_bn_workspace_floats(C,R). */
long pdf_bn_workspace_floats(int C, long R);
int pdf_bn_train_fwd(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     const float* res, int ldr, int relu, float* y, int ldy,
                     float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, void* stream);
int pdf_bn_eval_fwd(const float* x, int ldx, int C, long R, const float* gamma, const float* beta,
                    const float* running_mean, const float* running_var, float eps,
                    const float* res, int ldr, int relu, float* y, int ldy, float* scale, float* shift, void* stream);
/* ws >= pdf_bn_workspace_floats(C,R) + 3*C.  relu: 0 none; 1 ReLU, mask read from the saved output y; 2 ReLU with no
 * residual: mask recomputed from x with the forward's scale / shift (y may be NULL, dres must be NULL) */
int pdf_bn_train_bwd(const float* dy, int lddy, const float* y, int ldy, int relu, const float* x, int ldx,
                     const float* save_mean, const float* save_rstd, const float* gamma,
                     const float* scale, const float* shift, int C, long R,
                     float* dx, int lddx, float* dres, int lddr, float* dgamma, float* dbeta, int accumulate,
                     float* ws, void* stream);
/* out[c] (+)= sum_r g[r][c] (bias gradients); ws >= pdf_bn_workspace_floats(C,R) */
int pdf_colsum(const float* g, int ldg, int C, long R, float* out, int accumulate, float* ws, void* stream);
/* paired: rows [0, R) -> out0, rows [R, 2R) -> out1; ws >= 2 * pdf_bn_workspace_floats(C,R) */
int pdf_colsum_pair(const float* g, int ldg, int C, long R, float* out0, float* out1, int accumulate, float* ws, void* stream);
/* nn.LayerNorm(eps=1e-6): gcn.py:92-97, self_attn.py:58, inter_attn.py:66-67, intaghand_decoder.py:139-142 */
int pdf_layernorm_fwd(const float* x, int ldx, int F, long R, const float* gamma, const float* beta, float eps,
                      float* y, int ldy, float* mean, float* rstd, void* stream);
int pdf_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, int F, long R, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, void* stream);
/* Fused / paired LayerNorm of the mesh decoder blocks (gcn.py:100-110, self_attn.py:24-33,78-84):
 *   z = x + dropout_p(add)   (add == NULL: z = x, nothing written to z)
 *   y = act(LayerNorm(z) * gamma_g + beta_g),  act 0 none / 1 relu,  g = 0 for rows < R_split else 1 (R_split >= R: one set).
 * Backward: dz = LN-backward(dy * act'(y)) + dz_in (dz_in may be NULL) is the gradient of x; dadd = dropout mask applied to dz
 * (NULL when there was no add); dgamma / dbeta of both sets are ACCUMULATED (atomics; zero-fill or pre-load them).
 * The two halves may be launched separately: all four parameter-gradient pointers NULL = data gradient only (no atomics -- the
 * half that sits on the dependent chain); dz == NULL (then dz_in and dadd NULL too) = parameter gradients only. */
int pdf_layernorm_fused_fwd(const float* x, int ldx, const float* add, int ldadd, float p, unsigned long long seed,
                            const unsigned long long* step, int F, long R, long R_split,
                            const float* gamma0, const float* beta0, const float* gamma1, const float* beta1, float eps, int act,
                            float* z, int ldz, float* y, int ldy, float* mean, float* rstd, void* stream);
int pdf_layernorm_fused_bwd(const float* dy, int lddy, const float* y, int ldy, int act, const float* z, int ldz, int F, long R, long R_split,
                            const float* gamma0, const float* gamma1, const float* mean, const float* rstd,
                            const float* dz_in, int lddzin, float* dz, int lddz, float* dadd, int lddadd,
                            float p, unsigned long long seed, const unsigned long long* step,
                            float* dgamma0, float* dbeta0, float* dgamma1, float* dbeta1, void* stream);
/* L2Norm.forward (intaghand_encoder.py:318-334) */
int pdf_l2norm_fwd(const float* x, int ldx, int C, long R, const float* w, float eps, float* y, int ldy, float* norm, void* stream);
int pdf_l2norm_bwd(const float* dy, int lddy, const float* x, int ldx, int C, long R, const float* w, float eps,
                   const float* norm, float* dx, int lddx, float* dw, void* stream);
/* The pyramid concat of L2Norm'd maps (intaghand_encoder.py:724-739: torch.cat of four L2Norm outputs) in one launch per
 * direction: part i = rows x[i] [R][C[i]] (contiguous), written to / read from channel offset C[0]+..+C[i-1] of the
 * concatenated rows y / dy [R][ld].  x, w, norm, dx, dw: HOST arrays of nparts (<= 4) device pointers; C: host array.
 * bwd: dw[i] zero-filled by the caller (atomically accumulated); dx16: optional host array of bf16 shadow outputs of dx (bf16 mode,
 * channel counts multiples of 64), NULL or NULL entries = none. */
int pdf_l2norm_cat_fwd(int nparts, const float* const* x, const int* C, const float* const* w, float eps, long R,
                       float* y, int ldy, float* const* norm, void* stream);
int pdf_l2norm_cat_bwd(int nparts, const float* dy, int lddy, const float* const* x, const int* C, const float* const* w, float eps, long R,
                       float* const* norm, float* const* dx, float* const* dw, void* const* dx16_or_null, void* stream);

/* ---- elementwise / spatial (csrc/elementwise.hip) --------------------------------------------- */
int pdf_act_fwd(const float* x, int ldx, float* y, int ldy, int C, long R, int act, void* stream);
int pdf_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int C, long R, int act, void* stream);
/* SFTLayer.forward modulation fea*(scale+1)+shift (intaghand_encoder.py:219) */
int pdf_sft_fwd(const float* fea, int ldf, const float* scale, int lds, const float* shift, int ldh,
                float* out, int ldo, int C, long R, void* stream);
int pdf_sft_bwd(const float* g, int ldg, const float* fea, int ldf, const float* scale, int lds,
                float* dfea, int lddf, float* dscale, int ldds, int C, long R, void* stream);
/* SFTLayer(3, 3) -- the four 3->3 point convolutions, the LeakyReLU(0.1) pair and the modulation of sft0
 * (intaghand_encoder.py:120-122 calling :205-219) in one launch per direction.  Weights [3][3] row-major, biases [3].
 * bwd: ws = 48*256 floats; parameter gradients (NULL = not wanted) are written, or added to when accumulate != 0. */
int pdf_sft3_fwd(const float* fea, int ldf, const float* cond, int ldc, const float* w_scale0, const float* b_scale0,
                 const float* w_scale1, const float* b_scale1, const float* w_shift0, const float* b_shift0,
                 const float* w_shift1, const float* b_shift1, float* out, int ldo, long R, void* stream);
int pdf_sft3_bwd(const float* g, int ldg, const float* fea, int ldf, const float* cond, int ldc,
                 const float* w_scale0, const float* b_scale0, const float* w_scale1, const float* b_scale1,
                 const float* w_shift0, const float* b_shift0, const float* w_shift1, const float* b_shift1,
                 float* dfea, int lddf, float* dcond, int lddc,
                 float* dw_scale0, float* db_scale0, float* dw_scale1, float* db_scale1,
                 float* dw_shift0, float* db_shift0, float* dw_shift1, float* db_shift1, int accumulate,
                 float* ws, long R, void* stream);
/* bf16 shadows (bf16 mode of pdf_set_gemm_precision).  A producer can write the values of an fp32 tensor a second time, rounded to
 * bf16 (RNE), same layout and leading dimensions; a GEMM-family consumer given that copy stages 2-byte operands (half the L2 -> LDS
 * bytes of kernels bound by exactly those; results bit-identical to rounding while staging).  Passed through thread-local slots so
 * the signatures above stay as they are: call the setter, then the entry point, on the same thread.
 *   pdf_set_bf16_operands(op0, op1): shadows of the NEXT conv2d / deconv2d / linear call's operands -- forward: (x, w);
 *     backward-data: (dy, w); backward-weight: (x, dy); NULL = none.  Every GEMM-family entry point clears them.
 *   pdf_set_bf16_output(out): pdf_bn_train_fwd writes the shadow of y, pdf_bn_train_bwd of dx, pdf_l2norm_cat_fwd of y.
 *   pdf_cast_bf16: dst[i] = bf16(src[i]), n % 4 == 0 (weight shadows from the flat fp32 master buffer).
 * bf16 STORAGE of the conv -> BatchNorm tensors (bf16 mode, optional; reference: the nn.Conv2d -> nn.BatchNorm2d pairs of resnet.py:102-122):
 *   pdf_set_bf16_output(y16) before pdf_conv2d_fwd: the output is written to y16 (bf16, RNE, same shape / ldy) INSTEAD of y, which
 *     is not touched (whole 64- / 128-row tiles, even Cout, no fused bias / activation needed; PDF_E_BADARG if the launch cannot);
 *   pdf_set_bn_input_bf16(x16) before pdf_bn_train_fwd / pdf_bn_train_bwd: x is read from x16 instead of the fp32 pointer;
 *   pdf_bn_train_bwd with dx == NULL and pdf_set_bf16_output(dx16): only the bf16 input gradient is written -- the backward GEMMs
 *     of the producing conv take it through pdf_set_bf16_operands and never read an fp32 dy. */
int pdf_set_bf16_operands(const void* op0_bf16, const void* op1_bf16);
int pdf_set_bf16_output(void* out_bf16);
int pdf_set_bn_input_bf16(const void* x16);
/* BatchNorm + ReLU applied by the CONSUMER (reference: the conv -> bn -> relu -> conv chains of the set-abstraction MLPs,
 * intaghand_encoder.py:48-103): pdf_bn_train_fwd with y == NULL computes the statistics and (scale, shift) only; the next
 * pdf_linear_fwd / pdf_linear_bwd_weight call of this thread, given them through pdf_set_input_affine_relu, reads its x operand as
 * relu(x * scale[c] + shift[c]) -- the normalised tensor is never written.  fp32 kernels, plain rows, K % 16 == 0. */
int pdf_set_input_affine_relu(const float* scale, const float* shift);
int pdf_cast_bf16(const float* src, void* dst, long n, void* stream);
/* Transposed bf16 shadows of many weight tensors in ONE launch: for every table entry {long src (floats into `src`), long dst
 * (elements into `dst`), int R, int T, int C, int tile0} the tensor w[R][T][C] is written as wt[C][T][R] (bf16, RNE); tile0 = number of
 * 32x32 tiles of the entries before it (T * ceil(R/32) * ceil(C/32) each), total_tiles their sum.  `table` is DEVICE memory. */
int pdf_cast_bf16_transposed(const float* src, void* dst, const void* table, int nlayers, long total_tiles, void* stream);
int pdf_debug_shadow_operands(void);      /* shadow operands consumed by bf16 GEMM launches so far (tests) */
/* nn.Dropout(p) with a stateless (seed, index) mask: the same call is its own backward (gcn.py:96, self_attn.py:51-52).
 * step: optional DEVICE counter mixed into the seed so a replayed hipGraph draws a fresh mask every step. */
int pdf_dropout(const float* x, float* y, long n, float p, unsigned long long seed, const unsigned long long* step, void* stream);
/* y = res + dropout(x) (residual tails, self_attn.py:31-33,80-84); backward: d res = dy, dx = pdf_dropout(dy, same seed) */
int pdf_dropout_add(const float* x, const float* res, float* y, long n, float p, unsigned long long seed,
                    const unsigned long long* step, void* stream);
/* resnet.maxpool (resnet.py:206).  bwd: when C % 4 == 0 and the pointers are 16-byte aligned every dx element is written exactly once
 * (gather over the <= 4 windows of an input element); otherwise dx must be zero-filled by the caller (atomic scatter). */
int pdf_maxpool3s2_fwd(const float* x, int N, int H, int W, int C, float* y, unsigned char* arg, void* stream);
int pdf_maxpool3s2_bwd(const float* dy, const unsigned char* arg, int N, int H, int W, int C, float* dx, void* stream);
/* dx += (dx holds the gradient of the pooled tensor's other consumers; autograd's add pass folded into this one) */
int pdf_maxpool3s2_bwd_add(const float* dy, const unsigned char* arg, int N, int H, int W, int C, float* dx, void* stream);
/* nn.Upsample(scale_factor=2, bilinear, align_corners=True) (intaghand_encoder.py:287-302) */
int pdf_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream);
int pdf_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream);
/* dst [R2][C2] (ldd) = src [R][C] (lds) in its top-left corner, zeros elsewhere; R2 <= R / C2 <= C crops.  Replaces the reference's
 * zero-padding of the PointNet++ 1x1 layers' matrices to aligned widths (torch.nn.functional.pad = fill + strided copy per pad), forward and
 * (as the crop of the gradient) backward: `intaghand_encoder.py:48-103` layers with 3 / 131 / 259 input channels. */
int pdf_pad2d(const float* src, int lds, long R, int C, float* dst, int ldd, long R2, int C2, void* stream);
/* torch.optim.Adam step (main.py:63) over one flat buffer; corr = device [1-b1^t, 1-b2^t] */
int pdf_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                  const float* corr, float grad_scale, void* stream);

/* ---- mesh decoder (csrc/graph.hip) ------------------------------------------------------------ */
/* graph_conv_cheby, K=2 (model_attn/gcn.py:34-69): out[b][v][2f]=x, out[b][v][2f+1]=(Lx); L as ELL [V][Wd] */
int pdf_cheby2_fwd(const float* x, int ldx, int B, int V, int F, const int* col, const float* val, int Wd,
                   float* out, int ldo, void* stream);
int pdf_cheby2_bwd(const float* d, int ldd, int B, int V, int F, const int* colT, const float* valT, int Wd,
                   float* dx, int lddx, void* stream);
/* paired: samples [0, B) use Laplacian 0, samples [B, 2B) Laplacian 1 (left / right hand graphs, same ELL width) */
int pdf_cheby2_fwd_pair(const float* x, int ldx, int B, int V, int F, const int* col0, const float* val0,
                        const int* col1, const float* val1, int Wd, float* out, int ldo, void* stream);
int pdf_cheby2_bwd_pair(const float* d, int ldd, int B, int V, int F, const int* colT0, const float* valT0,
                        const int* colT1, const float* valT1, int Wd, float* dx, int lddx, void* stream);
/* multi-head softmax attention (self_attn.py:63-76, inter_attn.py:82-105); stat [B][H][V][2], dvec [B][H][V].
 * Queries of sample b attend to keys / values of sample (b + kv_shift) % B: with both hands stacked along the batch axis,
 * kv_shift = B/2 is the cross-hand attention, 0 the self attention. */
int pdf_attn_fwd(const float* q, const float* k, const float* v, int ld, int B, int V, int H, int dh, int kv_shift,
                 float pdrop, unsigned long long seed, const unsigned long long* step, float* out, int ldo, float* stat, void* stream);
int pdf_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* o, const float* dout, int ldo,
                 const float* stat, int B, int V, int H, int dh, int kv_shift, float pdrop, unsigned long long seed, const unsigned long long* step,
                 float* dq, float* dk, float* dv, int lddq, float* dvec, void* stream);

/* Farthest point sampling, `farthest_point_sampling_fast` (lib/datasets/interhand.py:147-178; `--sample_strategy FPS`,
 * opts.py:231): xyz [Bc][N][ld] (N <= 16384), start [Bc] (first pick, NULL = 0) -> idx [Bc][S] in pick order (the helper
 * returns np.unique of it).  Once every remaining distance is <= 1e-8 the picks repeat, like the helper's. */
int pdf_fps(const float* xyz, int ld, int Bc, int N, int S, const int* start, int* idx, void* stream);

/* ---- mesh loss terms (csrc/loss.hip) ---------------------------------------------------------- */
/* out[r] = mean_i f(pred[r][i] - tgt[r][i]); mode 0: |.| (the `l1` of lib/trains/simplified.py:427-436,481,506-511),
 * mode 1: (.)^2 (F.mse_loss, :425,482,499).  bwd: dpred[r][i] = gout[r]/n * f'(.) */
int pdf_rowloss_fwd(const float* pred, const float* tgt, long rows, long n, int mode, float* out, void* stream);
int pdf_rowloss_bwd(const float* pred, const float* tgt, const float* gout, long rows, long n, int mode, float* dpred, void* stream);
/* normal_loss + edge_length_loss (lib/trains/simplified.py:66-115) for G vertex sets (hands) x B samples: pred, gt [G][B][V][3],
 * faces [G][Fc][3] int64.  fwd: part[(g*B+b)*2 + {0,1}] = per-sample sums of the normal / edge terms over the faces.
 * bwd: dpred of sum_g (wn[g]*normal_sum + we[g]*edge_sum); we may be NULL (no edge term).  V <= 1024. */
int pdf_face_loss_fwd(const float* pred, const float* gt, const long long* faces, int G, int B, int V, int Fc, float* part, void* stream);
int pdf_face_loss_bwd(const float* pred, const float* gt, const long long* faces, int G, int B, int V, int Fc,
                      const float* wn, const float* we, float* dpred, void* stream);

/* Dense-map terms of CtdetLoss (lib/trains/simplified.py:368 SmoothL1 on `mask`, :374 MSE on `hms`, :376,391 focal loss
 * lib/models/losses.py:138-165 on the clamped sigmoid lib/models/utils.py:8-10 of `hm`), all three in one forward (partials +
 * finalize) and one backward launch.  Predictions NHWC [B][HW][C], targets NCHW [B][C][HW].  ws: pdf_dense_loss_workspace_floats(B).
 * out[0] = mask mean, out[1] = hms mean, out[2+b] = focal loss of sample b, out[2+B+b] = num_pos[b], out[2+2B] = 1 when the
 * batch has no positive (losses.py:161); 3 + 2B floats.  bwd: g_mask, g_hms (one float each), g_hm [B] = upstream gradients,
 * stat = the forward's out; a NULL d* or g_* pointer skips that term. */
long pdf_dense_loss_workspace_floats(int B);
int pdf_dense_loss_fwd(const float* mask, const float* mask_gt, int mask_c, int mask_hw,
                       const float* hms, const float* hms_gt, int hms_c, int hms_hw,
                       const float* hm, const float* hm_gt, int hm_c, int hm_hw, int B, float* ws, float* out, void* stream);
int pdf_dense_loss_bwd(const float* mask, const float* mask_gt, float* dmask, int mask_c, int mask_hw,
                       const float* hms, const float* hms_gt, float* dhms, int hms_c, int hms_hw,
                       const float* hm, const float* hm_gt, float* dhm, int hm_c, int hm_hw, int B,
                       const float* g_mask, const float* g_hms, const float* g_hm, const float* stat, void* stream);
/* Evaluation metric of BaseTrainer.evaluation (lib/trains/base_trainer.py:263-323): out[r] = sum_i ||pred[r][i] - gt[r][i]||_2
 * over the n points (dim 2 or 3) of row r = (sample, hand). */
int pdf_point_dist_sum(const float* pred, const float* gt, int rows, int n, int dim, float* out, void* stream);

/* ---- depth front end (csrc/frontend.hip) ------------------------------------------------------ */
/* depth2pcl (intaghand_encoder.py:369-491 + get_points_coordinate lib/utils/utils.py:251-262) batched on the GPU:
 * depth [B][H][W] metres, mask [B][2][H][W] (right, left), K [B][3][3], valid [B][2] ->
 * choose int64 [B][2][1024] (left, right), cloud [B][2][1024][3], count int32 [B][2] (window candidates) or NULL. */
int pdf_depth2pcl(const float* depth, const float* mask, const float* K, const float* valid, int B, int H, int W,
                  unsigned long long seed, long* choose, float* cloud, int* count, void* stream);

/* ---- MANO (csrc/mano.hip) --------------------------------------------------------------------- */
/* ManoLayer.forward, use_pca=False (lib/models/networks/manolayer.py:257-334): axis-angle root [B][3], pose [B][45],
 * shape [B][10], trans [B][3] or NULL -> verts [B][778][3], joints [B][21][3]; center_idx < 0 = None. */
int pdf_mano_lbs_fwd(const float* root_aa, const float* pose_aa, const float* shape, const float* trans,
                     const float* v_template, const float* shapedirs, const float* posedirs, const float* J_reg,
                     const float* weights, int B, int left_side, int center_idx, float* verts, float* joints, void* stream);
/* autograd of the above: gradients of (verts [B][778][3], joints [B][21][3]) w.r.t. root / pose axis-angles, shape and trans.
 * dverts / djoints may be NULL (no gradient from that output); dshape / dtrans may be NULL (not wanted). */
int pdf_mano_lbs_bwd(const float* root_aa, const float* pose_aa, const float* shape,
                     const float* v_template, const float* shapedirs, const float* posedirs, const float* J_reg,
                     const float* weights, int B, int left_side, int center_idx, const float* dverts, const float* djoints,
                     float* droot, float* dpose, float* dshape, float* dtrans, void* stream);
/* ManoRender.Split_coeff (lib/models/hand3d/Mano_render.py:145-194; call site lib/trains/simplified.py:730-736): decode of the
 * 122-channel `params` head at each hand's centre pixel -- per hand [orient 3 | pose 45 | shape 10 (x 0) | trans 3], trans z += 0.6,
 * x / y un-projected from the centre pixel with K.  params NHWC [B][HW][ldp], ind int64 [B][2] (left, right), K [B][3][3].
 * Outputs [2][B][3], [2][B][45], [2][B][10], [2][B][3] feed pdf_mano_lbs_fwd.  bwd scatters into dparams (caller zero-fills). */
int pdf_mano_split_coeff(const float* params, int ldp, long HW, const long* ind, const float* K, int B, int input_res, int down,
                         float* orient, float* pose, float* shape, float* trans, void* stream);
int pdf_mano_split_coeff_bwd(const float* params, float* dparams, int ldp, long HW, const long* ind, const float* K, int B, int input_res, int down,
                             const float* dorient, const float* dpose, const float* dtrans, void* stream);


/* ---- explicit per-call options (round 4) ------------------------------------------------------------------------------
 * Everything a call can take beyond its positional arguments travels in ONE structure handed to the `_x` form of the entry
 * point (NULL = no options): the convention SURVEY 8(b) derives from the reference's only native op, whose forward / backward take
 * every tensor they touch as an argument (lib/utils/roi_align/src/crop_and_resize_gpu.cpp:6-13,100-109).  Inputs are read by that
 * call only; `stats_tiles` / `stats_rows` are written back by it.  The plain entry points above are thin wrappers: they TAKE AND
 * CLEAR every slot the pdf_set_* functions armed on the calling thread before they look at a single argument, so no return path
 * -- PDF_E_BADARG included -- leaves a slot armed for an unrelated later call (pdf_debug_armed_slots() == 0 after any call).
 *
 *   op0_bf16 / op1_bf16  bf16 shadows (same elements, RNE-rounded, same layout and leading dimensions in ELEMENTS) of the call's two
 *                        operands -- forward: (x, w); backward-data: (dy, w); weight gradient: (x, dy).  bf16 mode only.
 *   out_bf16             pdf_bn_train_fwd: shadow of y; pdf_bn_train_bwd: shadow of dx (dx == NULL: the only output);
 *                        pdf_l2norm_cat_fwd: shadow of y; pdf_conv2d_fwd (bf16 mode): the output INSTEAD of y (storage mode).
 *   bn_x_bf16            pdf_bn_train_fwd / _bwd: x is read from this bf16 tensor instead of the fp32 pointer.
 *   stats_out, stats_cap pdf_conv2d_fwd / pdf_linear_fwd: per-row-block (mean, M2) pairs of the output columns,
 *                        stats_out[(t * N + c) * 2 + {0,1}]; -> stats_tiles row blocks of stats_rows rows (0: not produced).
 *   tile_stats, tile_n, tile_rows   pdf_bn_train_fwd / pdf_bn_relu_maxk_fwd: such partials; the call skips its statistics pass.
 *   in_scale, in_shift   pdf_linear_fwd / pdf_linear_bwd_weight: x is read as relu(x * in_scale[k] + in_shift[k]).
 *   op1_bf16_t           pdf_conv2d_bwd_data(_add) / pdf_linear_bwd_data (bf16 mode): the weight's transposed bf16 shadow,
 *                        wt[c][tap][r] for w[r][tap][c] (pdf_cast_bf16_transposed); lets the LDS-DMA kernel take the launch.
 *   ws, ws_floats        pdf_conv2d_fwd / pdf_conv2d_bwd_data(_add) / pdf_conv2d_bwd_weight, fp32 mode: a workspace of pdf_conv2d_winograd_workspace_floats floats
 *                        (asked with backward = 0 / 1 / 2 for the three passes; 0 floats = the layer does not qualify).  A stride-1 3x3 layer with
 *                        >= 128 input channels (PDF_WINOGRAD_MINC), 16-float aligned channel counts and planes x tiles >= 16,384 (PDF_WINOGRAD_MINPT) is
 *                        then computed in the Winograd domain (csrc/winograd.hip): F(4x4, 3x3) -- 36 planes, 4x fewer multiplications, ~4e-5 absolute
 *                        error on values of a few units -- where the map edges are multiples of 4 and PDF_WINOGRAD_F4 (bit mask, default 15 = every
 *                        launch) permits, else F(2x2, 3x3) (16 planes, 2.25x fewer, MORE accurate than the direct fp32 sum); the weight gradient is
 *                        taken in the transform domain too (backward = 2).  PDF_WINOGRAD=2: F(2x2) only; =0: the direct kernels.  *   wino_v               pdf_conv2d_bwd_weight (Winograd F(4x4) path): the transformed input V [36][tiles][Cin] that the FORWARD of the same
 *                        convolution left in its workspace (at float offset pdf_conv2d_winograd_v_offset(...) of the forward `ws`): the
 *                        weight gradient then skips its own input transform (same kernel, same values).  The caller keeps that workspace
 *                        alive and unmodified from the forward to the weight-gradient call.
 *                        pdf_conv2d_fwd (F(4x4) launches): V of the SAME input tensor from another convolution's forward workspace (several
 *                        heads reading one feature map): this forward skips its input transform.  V depends on (x, N, H, W, Cin) only.
 */
typedef struct PdfCallOpts {
    const void* op0_bf16; const void* op1_bf16;
    void* out_bf16;
    const void* bn_x_bf16;
    float* stats_out; long stats_cap;
    long stats_tiles; long stats_rows;
    const float* tile_stats; long tile_n; long tile_rows;
    const float* in_scale; const float* in_shift;
    const void* op1_bf16_t;
    float* ws; long ws_floats;
    const float* wino_v;
} PdfCallOpts;
/* float offset of V inside the forward workspace of this convolution, or -1 when its forward is not an F(4x4) launch (no V the weight gradient could take) */
long pdf_conv2d_winograd_v_offset(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int pdf_linear_fwd_x(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int ldx, int ldw, int ldy, int act, void* stream, PdfCallOpts* opts);
int pdf_linear_fwd_pair_x(const float* x, const float* w0, const float* w1, const float* b0, const float* b1, float* y, int M, int N, int K, int ldx, int ldw, int ldy, int act, void* stream, PdfCallOpts* opts);
int pdf_linear_bwd_data_x(const float* dy, const float* w, float* dx, int M, int N, int K, int lddy, int ldw, int lddx, void* stream, PdfCallOpts* opts);
int pdf_linear_bwd_data_pair_x(const float* dy, const float* w0, const float* w1, float* dx, int M, int N, int K, int lddy, int ldw, int lddx, void* stream, PdfCallOpts* opts);
int pdf_conv2d_fwd_x(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int ldy, int act, void* stream, PdfCallOpts* opts);
int pdf_conv2d_bwd_data_x(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int lddy, void* stream, PdfCallOpts* opts);
int pdf_conv2d_bwd_data_add_x(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int lddy, void* stream, PdfCallOpts* opts);
int pdf_linear_bwd_weight_x(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, void* stream, PdfCallOpts* opts);
int pdf_linear_bwd_weight_pair_x(const float* x, const float* dy, float* dw0, float* dw1, float* db0, float* db1, float* ws, long ws_floats, int M, int N, int K, int ldx, int lddy, int accumulate, void* stream, PdfCallOpts* opts);
int pdf_conv2d_bwd_weight_x(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int lddy, int accumulate, void* stream, PdfCallOpts* opts);
int pdf_deconv2d_fwd_x(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int ldy, void* stream, PdfCallOpts* opts);
int pdf_deconv2d_bwd_data_x(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int lddx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int lddy, void* stream, PdfCallOpts* opts);
int pdf_deconv2d_bwd_weight_x(const float* x, const float* dy, float* dw, float* ws, long ws_floats, int N, int H, int W, int Cin, int ldx, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, int lddy, int accumulate, void* stream, PdfCallOpts* opts);
int pdf_bn_train_fwd_x(const float* x, int ldx, int C, long R, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, const float* res, int ldr, int relu, float* y, int ldy, float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, void* stream, PdfCallOpts* opts);
int pdf_bn_train_bwd_x(const float* dy, int lddy, const float* y, int ldy, int relu, const float* x, int ldx, const float* save_mean, const float* save_rstd, const float* gamma, const float* scale, const float* shift, int C, long R, float* dx, int lddx, float* dres, int lddr, float* dgamma, float* dbeta, int accumulate, float* ws, void* stream, PdfCallOpts* opts);
int pdf_bn_relu_maxk_fwd_x(const float* y, int ldy, int C, long R, int K, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int training, float* out, int ldo, int* arg, float* save_mean, float* save_rstd, float* scale, float* shift, float* ws, void* stream, PdfCallOpts* opts);
int pdf_l2norm_cat_fwd_x(int nparts, const float* const* x, const int* C, const float* const* w, float eps, long R, float* y, int ldy, float* const* norm, void* stream, PdfCallOpts* opts);
/* number of hand-over slots currently armed on the calling thread (0 after every entry-point call), and sizeof(PdfCallOpts) as the
 * library was built (a binding checks its own layout against it) */
/* Winograd F(4x4, 3x3) / F(2x2, 3x3) for stride-1 3x3 convolutions with >= 128 input channels (nn.Conv2d sites intaghand_encoder.py:602,617,
 * 675-693, 270-316 and ResNet layers 2-3; selection rule: PdfCallOpts::ws above): floats of workspace the layer wants (PdfCallOpts::ws) for its
 * forward (backward = 0), backward-data (1) or weight-gradient (2) pass, 0 when it does not qualify.  `feat` at B = 32: ~1.5 GB per pass. */
long pdf_conv2d_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward);
int pdf_debug_armed_slots(void);
int pdf_debug_callopts_size(void);

/* ---- x3 arithmetic (round 6, csrc/gemm_x3.hip): fp32 products on the bf16 matrix pipe ------------------------------------------
 * gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 MFMA rate.  A fp32 value is exactly h + m + l with three bf16 values
 * (8 + 8 + 8 significand bits), so six bf16 MFMAs with fp32 accumulation (hh', hm', mh', mm', hl', lh') reproduce the fp32 product to
 * ~2^-24 relative -- the order of the native instruction's own rounding -- in 6/16 of the matrix-pipe time (hh' accumulates in a register set of its
 * own, the five small products in a second: measured rms error against float64 0.31-0.37x the native fp32-MFMA kernel's).  Operands are "x3 planes":
 * three bf16 tensors of the operand's shape, component c at base + c * cs elements.  Used by the Winograd-domain products of the stride-1
 * 3x3 convolutions (nn.Conv2d sites intaghand_encoder.py:602,617,675-693 and ResNet layers 2-3; PDF_X3=0 keeps the native fp32 MFMA).
 *   pdf_x3_split            x [n] fp32 -> out: 3 planes of n bf16 (n % 8 == 0, cs % 8 == 0, 16-byte aligned)
 *   pdf_x3_batched_gemm_nt  C_b [M][N] = A_b [M][K] B_b [N][K]^T, b < batch, both operands x3 planes (batch strides gsA / gsB in elements,
 *                           gsC in floats); K % 32 == 0; variant: tile choice (0 = 256x128, 1 = 128x128, 2 = 128x64); nprod: 6 | 9 | 3
 *   pdf_x3_batched_gemm_tn  slab_b [split][NI][NJ] = sum over the rows m of split of P_b [m][NI]^T Q_b [m][NJ] (weight-gradient shape)
 *   pdf_batched_gemm_nt     the native fp32-MFMA batched product on plain fp32 operands (the comparison arm of tools/x3_bench.py) */
int pdf_x3_split(const float* x, void* out, long n, long cs, void* stream);
/* x3 form of a transposed convolution with kernel == stride and a long reduction (the pyramid's p4 / p5, intaghand_encoder.py:603-611): floats of
 * workspace (PdfCallOpts::ws of pdf_deconv2d_fwd_x, backward = 0, pdf_deconv2d_bwd_data_x, backward = 1, and pdf_deconv2d_bwd_weight_x, backward = 2) with which
 * the call runs as a plain x3 GEMM (+ pixel shuffle); 0 when the layer does not qualify (PDF_X3_DECONV=0 / PDF_X3=0: never). */
long pdf_deconv2d_x3_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int backward);
/* which launches take the x3 form: bit 0 = the wide Winograd-domain products, bit 1 = the transposed convolutions, bit 2 = the fused mesh decoder's linear
 * products (pdf_mesh_level_*_x3: the caller's choice of entry point) (default 7; env PDF_X3=0 / PDF_X3_DECONV=0 / PDF_X3_MESH=0); -1 = back to the environment's choice.  Workspace sizes (pdf_conv2d_winograd_workspace_floats, _v_offset, pdf_deconv2d_x3_workspace_floats)
 * depend on it: query them again after a change, and change it BETWEEN steps only (a forward's transformed input kept for its weight gradient is in the
 * format of the mode it was written under). */
int pdf_set_x3_mode(int mode);
/* a HIP stream of the given priority (clamped to the device's range, returned in *lo (least) / *hi (greatest) when not NULL; out == NULL: only the range) */
int pdf_stream_create(void** out, int priority, int* lo, int* hi);
int pdf_debug_x3_mode(void);
int pdf_debug_x3_stamps(unsigned long long* out);       /* diagnostic builds (-DX3_STAMPS=1) only: phase clocks of block 0 of the last x3gemm_nt launch; 0 otherwise */
int pdf_x3_batched_gemm_nt(const void* A3, long csA, const void* B3, long csB, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, int variant, int nprod, void* stream);
int pdf_batched_gemm_nt(const float* A, const float* B, float* C, int batch, long gsA, long gsB, long gsC, int M, int N, int K, void* stream);
int pdf_x3_batched_gemm_tn(const void* P3, long csP, const void* Q3, long csQ, float* slab, int batch, long gsP, long gsQ, int M, int NI, int NJ, int splits, int variant, int nprod, void* stream);
int pdf_batched_gemm_tn(const float* P, const float* Q, float* slab, int batch, long gsP, long gsQ, int M, int NI, int NJ, int splits, void* stream);

/* ---- fused mesh decoder (round 5, csrc/meshdec.hip) ---------------------------------------------------------------------
 * One DualGraphLayer of the dual-hand GCN / attention decoder (lib/models/networks/model_attn/DualGraph.py:62-92: 4 x GCN_ResBlock per hand,
 * gcn.py:99-110 / 34-69; SelfAttn per hand, self_attn.py:63-85; cross-hand attention + MLP blocks, inter_attn.py:73-125) in two launches per
 * direction: one 256-thread workgroup per (hand, sample) keeps that hand's [V][C] rows in LDS through the whole chain.
 * V = 63 << level, C = 256 >> level; rows are stacked [2 (left, right)][B][V].  Index [0] / [1] of every parameter pair = left / right hand
 * (the shared cross-hand projections pass the same pointer twice).  x already carries the position embedding (DualGraph.py:76-77).
 * The forward writes what the backward and the weight-gradient GEMMs read into `tape` (pdf_mesh_tape_floats) and `qkv`
 * ([3][2][B][V][C]); both buffers are needed in eval mode too (stage hand-offs).  The backward writes dx, ACCUMULATES every parameter
 * gradient into the g* pointers (the LayerNorm ones with atomics from the data kernels on `stream`, the Linear ones with the library's
 * weight-gradient GEMMs on `side_stream`, which first waits for `stream`), and uses `gtape` (pdf_mesh_gtape_floats) and `wg_ws`
 * (>= 2 * pdf_wgrad_workspace_floats(B * V, C, 4 * C) floats) as scratch.  Dropout masks are the stateless hash of (seed, element index)
 * the unfused kernels use, so the same seeds give the same masks. */
typedef struct PdfMeshLin { const float* w[2]; const float* b[2]; } PdfMeshLin;
typedef struct PdfMeshLN { const float* g[2]; const float* b[2]; } PdfMeshLN;
typedef struct PdfMeshGcn { PdfMeshLin fc1, fc2, sc; PdfMeshLN n2, n3; unsigned long long seed; } PdfMeshGcn;
typedef struct PdfMeshAttn {
    PdfMeshLN ln; PdfMeshLin q, k, v, fc; PdfMeshLN ffln; PdfMeshLin f1, f2;
    unsigned long long seed_att, seed_z, seed_t, seed_x;
} PdfMeshAttn;
typedef struct PdfMeshLinG { float* w[2]; float* b[2]; } PdfMeshLinG;
typedef struct PdfMeshLNG { float* g[2]; float* b[2]; } PdfMeshLNG;
typedef struct PdfMeshGcnG { PdfMeshLinG fc1, fc2, sc; PdfMeshLNG n2, n3; } PdfMeshGcnG;
typedef struct PdfMeshAttnG { PdfMeshLNG ln; PdfMeshLinG q, k, v, fc; PdfMeshLNG ffln; PdfMeshLinG f1, f2; } PdfMeshAttnG;
typedef struct PdfMeshLevel {
    int level, B, training, cin0;
    float p;
    const unsigned long long* step;
    const float* x;
    float* out;
    const int* ell_col[2]; const float* ell_val[2]; const int* ell_colT[2]; const float* ell_valT[2]; int ell_w;
    PdfMeshGcn gcn[4];
    PdfMeshAttn self_, cross;
    float* tape;
    float* qkv;
    const float* dout;
    float* dx;
    float* gtape;
    PdfMeshGcnG ggcn[4]; PdfMeshAttnG gself, gcross;
    float* wg_ws; long wg_ws_floats;
} PdfMeshLevel;
long pdf_mesh_tape_floats(int level, int B);
long pdf_mesh_gtape_floats(int level, int B);
int pdf_mesh_level_fwd(const PdfMeshLevel* a, void* stream);
int pdf_mesh_level_bwd(const PdfMeshLevel* a, void* stream, void* side_stream);
/* The same level with its LINEAR products on the bf16 MFMA (operands rounded to bf16 as they are fed, fp32 accumulation; attention, LayerNorm,
 * graph product in fp32): for the library's bf16 mode (BASELINE configs 4-5).  Same argument block and tape layout. */
int pdf_mesh_level_fwd_bf16(const PdfMeshLevel* a, void* stream);
int pdf_mesh_level_bwd_bf16(const PdfMeshLevel* a, void* stream, void* side_stream);
/* ... and with its linear products as x3 arithmetic (both operands split into three bf16 values as they are fed, six bf16 MFMAs per product: fp32-grade
 * results, see "x3 arithmetic" above): what the fp32 mode runs when bit 2 of pdf_debug_x3_mode() is set.  Same argument block and tape layout. */
int pdf_mesh_level_fwd_x3(const PdfMeshLevel* a, void* stream);
int pdf_mesh_level_bwd_x3(const PdfMeshLevel* a, void* stream, void* side_stream);
int pdf_debug_mesh_level_size(void);

/* ---- fused mesh loss (round 5, csrc/loss.hip) --------------------------------------------------------------------------------
 * Every mesh term of CtdetLoss.forward's train branch (lib/trains/simplified.py:425-525; lib/models/losses.py:26-94 bone directions;
 * Mano_render.py:203-223 root un-projection and pinhole projection; Mano_model.py:309-323 joint regressor) in one forward launch pair and one
 * backward launch, one workgroup per (hand, sample).  Predictions are stacked [2 (left, right)][B][...]; `part` is [2][B][12] scratch.
 * out (4 + 9 B floats): verts2d, norm, edge, gcn_2d, then B-vectors root, verts, abs_verts, gcn, abs_joints, joints2d, joints, bone (weighted by
 * `valid` and x1000 where the reference does, :506-525), then the per-sample weighted sum of the twelve with `coef` (the reference's :610-640
 * weights, same order).  The backward writes the gradients of sum_b gmp[b] * weighted_sum[b]. */
typedef struct PdfMeshLoss {
    const float* vp; const float* v2p; const float* hd3; const float* hd2; const float* r;
    const float* vgt[2]; const float* jgt[2]; const float* v2gt[2]; const float* lmsgt[2];     /* ground truth per hand, [B][...] each */
    const long long* ind; const float* K; const float* valid;
    const float* reg[2]; const long long* faces; const long long* perm[2];
    int B, Fc, size, down;
    float* part;
    float* out;
    float coef[12];
    const float* gmp;
    int edge_grad;
    float* dvp; float* dv2p; float* dhd3; float* dhd2; float* dr;
} PdfMeshLoss;
int pdf_mesh_loss_fwd(const PdfMeshLoss* a, void* stream);
int pdf_mesh_loss_bwd(const PdfMeshLoss* a, void* stream);
int pdf_debug_mesh_loss_size(void);
int pdf_debug_mesh_stamps(unsigned long long* out);      /* diagnostic builds (-DMD_STAMPS=1) only: stage clocks of workgroup 0; 0 otherwise */

#ifdef __cplusplus
}
#endif
#endif
