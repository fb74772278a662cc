/* p2i_hip.h — C ABI of libp2i_hip.so, the MI355X (gfx950) kernel library under the P2I-GAN hot path.
 *
 * The reference (NTU-CompHydroMet-Lab/P2I-GAN-benchmark) is pure Python on PyTorch: the
 * arithmetic of its hot path is reached through ATen call sites (F.conv2d, nn.Conv3d, cdist,
 * topk, max_pool2d, Upsample, spectral_norm, kl_div, Adam).  Each entry point below replaces one
 * group of those call sites; the file:line next to it is the reference code it stands in for
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 (or int32 where stated); the caller
 *    (PyTorch) owns all memory including workspaces; the library never allocates or frees;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*), never syncs;
 *  - state kept by the library: none that a result depends on.  Per calling THREAD it remembers the tile plan of the last
 *    conv / wgrad launch (p2i_conv_last_plan / p2i_wgrad_last_plan, diagnostics) and the scratch pointers of a running
 *    p2i_conv_*_x6 / p2i_conv_wgrad_ws call.  Tuning switches (P2I_* environment variables): the engine on/off switches
 *    (P2I_CONV_X6C, P2I_CONV_X6C_FUSED, P2I_X6C_KSPLIT, P2I_CONV_V4, P2I_CONV_CK16, P2I_CONV_KG27, P2I_DGRAD_FUSED, P2I_DGRAD_PAIR,
 *    P2I_C1_FAST, P2I_O1_FWD, P2I_WGRAD_X4, P2I_WGRAD_WINDOW) are read ONCE per process; the per-launch choices that the parity
 *    tests flip inside one process (P2I_X6C_MIN_WG, P2I_X6C_MIN_WG_LOW, P2I_X6C_TILE, P2I_X6C_TPS, P2I_X6C_PC, P2I_X6P_NPW, P2I_X6P_TN1,
 *    P2I_X6C_STAGGER, P2I_X6C_FUSED_KSPLIT, P2I_WGRAD_X6, P2I_WGRAD_X6_PC, P2I_WGRAD_X6_S4) are read on EVERY call;
 *  - aliasing: an output may alias the epilogue operand of the same call that is read element-for-element at the position it is
 *    written (y == residual, dx == dx_add, dx == mask_y): every kernel reads it before it writes that element, and the split-K
 *    launches of the bf16-split kernels (which zero-fill the destination first) are not chosen for an aliased call.  No other
 *    overlap between inputs and outputs is allowed;
 *  - return value: 0 on success, negative P2I_E* on a rejected argument, positive = hipError_t
 *    of a failed launch.  Nothing throws across the ABI;
 *  - activations are NC(T)HW; a 2-D tensor is the T == 1 case of the 5-D one.
 */
#ifndef P2I_HIP_H
#define P2I_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2I_OK 0
#define P2I_EINVAL (-1)   /* bad shape / unsupported configuration */
#define P2I_ELDS (-2)     /* tile configuration does not fit LDS */

/* activation codes shared by the conv epilogue and its backward prologue */
#define P2I_ACT_NONE 0
#define P2I_ACT_RELU 1
#define P2I_ACT_LEAKY 2   /* negative slope 0.2 (p2igan.py:122-139) */
#define P2I_ACT_TANH 3

int p2i_abi_version(void);
const char* p2i_last_error(void);

/* ------------------------------------------------------------------ convolution engine
 * Replaces F.conv2d / nn.Conv3d forward + convolution_backward (deconv_pytorch.py:103-109,
 * layer.py:390,402-407, p2igan.py:120-142).  Weights are consumed in PACKED form
 * Wp[tap][k_channel][m_channel_padded] (m padded to a multiple of 32, produced by the
 * p2i_*_pack / p2i_doconv_fold entry points), so that the MFMA A-operand rows are contiguous.
 *
 * p2i_conv_fwd :  y = act(conv(x, W) + bias) + residual
 * p2i_conv_dgrad: dx = conv_transpose(dy * act'(y), W) + dx_add   (Wd = pack with roles swapped;
 *                 dx_add, dx-shaped or NULL, fuses the skip-path gradient of a residual block;
 *                 mask_y (dx-shaped or NULL): dx is finally multiplied by mask_act'(mask_y), the
 *                 derivative of the activation that produced this layer's INPUT, so the next
 *                 dgrad/wgrad up the chain consume a ready gradient and need no prologue)
 * p2i_conv_wgrad: dWp[tap][cin][cout_pad] += sum_pixels x * (dy * act'(y))   (atomic fp32 adds;
 *                 caller zeroes dWp), db[cout] += sum dy*act'(y) when db != NULL.
 * dims: x (B,Cin,Ti,Hi,Wi), y (B,Cout,To,Ho,Wo); kernel (kt,kh,kw); stride (st,sh,sw);
 * padding (pt,ph,pw).  bias / residual / y_act may be NULL.
 */
typedef struct {
  int B, Cin, Cout;
  int Ti, Hi, Wi;
  int To, Ho, Wo;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;
} p2i_conv_desc;

int p2i_conv_fwd(const p2i_conv_desc* d, const float* x, const float* wp, const float* bias,
                 const float* residual, float* y, int act, void* stream);
int p2i_conv_dgrad(const p2i_conv_desc* d, const float* dy, const float* y_act, int act,
                   const float* wp_d, const float* dx_add, const float* mask_y, int mask_act,
                   float* dx, void* stream);
int p2i_conv_wgrad(const p2i_conv_desc* d, const float* x, const float* dy, const float* y_act,
                   int act, float* dwp, float* dbias, void* stream);
/* p2i_conv_wgrad with a caller-owned scratch `ws` of ws_floats floats: when it holds every workgroup's partial tile
 * (<= 256 * kh*kw * Cin * pad32(Cout) floats, 37.7 MB for the generator's 3x3 C->C layers) the partial tiles are stored
 * and summed by a second kernel instead of being added with float atomics: faster, and bit-reproducible.  ws == NULL or
 * too small: the atomic path of p2i_conv_wgrad.
 * With a sufficient scratch, 3x3 pad-1 2-D layers and 3x3x3 pad-1 layers with t stride <= 2 (three t slices), spatial stride 1,
 * whose Cin and Cout are multiples of 64, Ho % 4 == 0, Wo % 16 == 0, called without act'(y) prologue (the generator's DO-Conv stack,
 * the discriminators' 256 -> 256 and 128 -> 128 layers) are computed on the bf16 matrix pipe with fp32 accuracy (wgrad_x6.hip: x
 * and dy split exactly into three bf16 terms each, six MFMA products per fp32 product, transposed LDS reads), dbias included
 * (summed from the staged dy tiles, added atomically); same result contract, same slices + reduce.  P2I_WGRAD_X6=0 (read per
 * call) keeps the fp32-MFMA kernel. */
int p2i_conv_wgrad_ws(const p2i_conv_desc* d, const float* x, const float* dy, const float* y_act, int act, float* dwp,
                      float* dbias, float* ws, int64_t ws_floats, void* stream);
/* Same contracts as p2i_conv_fwd / p2i_conv_dgrad (without the act'(y) prologue).  With a contraction width K % 16 == 0 and
 * enough output tiles to fill the chip, these layers are computed on the bf16 matrix pipe with fp32 accuracy (conv_x6c.hip):
 *   - 3x3 stride-1 pad-1 2-D layers, forward and data gradient;
 *   - 3x3x3 pad-1 layers with spatial stride 1 and t stride 1 or 2 (the 27 taps as three t slices of nine), forward and data
 *     gradient (one launch per t-parity class);
 *   - the data gradient of 3x3 / 3x3x3 layers with stride (1,2,2) and even input height / width: the four input-parity classes
 *     of the destination fused in one workgroup (patch_gemm_x6c_kernel<8,1,true,.>).
 * Every fp32 operand is split exactly into three bf16 terms and six v_mfma_f32_32x32x16_bf16 products are accumulated in fp32
 * (the dropped products are <= 2^-23 |a b| each, below the rounding of an fp32 fma chain).  Every other layer runs on the
 * fp32-MFMA kernels, exactly as p2i_conv_fwd / p2i_conv_dgrad would; the choice is the library's.
 * `wsplit`: caller-owned scratch of 3*ntaps*K*pad32(M) uint16 (K = contraction channels: Cin for fwd, Cout for dgrad; M the
 * other one), overwritten by the call; wsplit == NULL forces the fp32-MFMA kernels.  P2I_X6C_MIN_WG (read per call) overrides
 * the tile-count thresholds (default: 200 workgroups for a tile variant to be preferred, 64 for the smallest tile as the last resort). */
int p2i_conv_fwd_x6(const p2i_conv_desc* d, const float* x, const float* wp, uint16_t* wsplit, const float* bias,
                    const float* residual, float* y, int act, void* stream);
int p2i_conv_dgrad_x6(const p2i_conv_desc* d, const float* dy, const float* wp_d, uint16_t* wsplit, const float* dx_add,
                      const float* mask_y, int mask_act, float* dx, void* stream);
/* Pre-split variant: split a weight tensor -- or a whole stack of same-shape packed tensors [n][ntaps][K][pad32(M)], passed as
 * ntaps = n * taps -- ONCE per weight update (p2i_x6_split; wb: 3 * ntaps * K * Mpad uint16, image Wb[plane][ntaps][K/8][Mpad][8]),
 * then call p2i_conv_fwd_x6s / p2i_conv_dgrad_x6s any number of times with wb_layer = wb + layer * taps * K * Mpad (uint16 elements:
 * plane 0 of that layer) and ntaps_w = the stack's total tap count.  p2i_x6c_would_take tells (1 / 0) whether a call with this
 * geometry (and, for a forward call, this epilogue activation) would use the split at all, so that callers can skip splitting
 * stacks no layer needs.  Layers with too few output tiles for 256 CUs whose epilogue is linear (no activation) run with the channel
 * chunks split over two workgroups per tile that add their partial sums into the zeroed destination (P2I_X6C_KSPLIT=0 disables). */
int p2i_x6_split(const float* wp, uint16_t* wb, int ntaps, int K, int Mpad, void* stream);
/* the same for n <= 24 packed tensors of different shapes in one launch (wb[i]: 3 * ntaps[i] * K[i] * Mpad[i] uint16 each): the
 * discriminators' layers after p2i_weight_pack_batched */
int p2i_x6_split_batched(const float* const* wp, uint16_t* const* wb, const int* ntaps, const int* K, const int* Mpad, int n,
                         void* stream);
int p2i_x6c_would_take(const p2i_conv_desc* d, int dgrad, int act);
int p2i_conv_fwd_x6s(const p2i_conv_desc* d, const float* x, const float* wp, const uint16_t* wb_layer, int ntaps_w,
                     const float* bias, const float* residual, float* y, int act, void* stream);
int p2i_conv_dgrad_x6s(const p2i_conv_desc* d, const float* dy, const float* wp_d, const uint16_t* wb_layer, int ntaps_w,
                       const float* dx_add, const float* mask_y, int mask_act, float* dx, void* stream);
/* tile plan {MB, NPIX, WAVES_M, CK, NT, KG} of the calling thread's most recent fwd/dgrad launch: names the
 * patch_gemm_dma_kernel<MB,NPIX,WAVES_M,CK,NT,KG> instance (NT = -1: the prologue kernel
 * patch_gemm_kernel<MB,NPIX,WAVES_M,CK>; KG = 7: patch_gemm_x6c_kernel<8, MB/32, false, TPS>, fields {MB = 32 or 64, 256, ksplit,
 * 16, TPS = taps per pipeline stage (3 or 9), 7}; KG = 8: the fused strided data gradient patch_gemm_x6c_kernel<8, 1, true, TPS>,
 * fields {32, 256, ksplit, 16, TPS, 8}; KG = 3: o1_fwd_kernel (single output channel), fields {1, 64, 8, 1, 9, 3}; KG > 10: the f32
 * fused strided data gradient patch_gemm_fused_kernel with KG - 10 parity classes per workgroup)
 * so that bench.py's roofline can be matched to rocprofv3 rows */
int p2i_conv_last_plan(int* out6);
/* same for the most recent p2i_conv_wgrad: {kind (0 wgrad_kernel<64>, 1 wgrad_dma_kernel<64,NTAP,Y4,CB>, 2 c1_wgrad_kernel,
 * 3 wgrad_x6_kernel), NTAP (wgrad_x6: 9 * kt taps in one launch), Y4, CB} */
int p2i_wgrad_last_plan(int* out4);

/* ------------------------------------------------------------------ weight preparation
 * p2i_doconv_fold_fwd: DoW = einsum('ims,ois->oim', D + D_diag, W.reshape(O/g, I, 9)) with the
 *   reference's memory reinterpretation to (O, I/g, 3, 3) (deconv_pytorch.py:111-127), written
 *   densely (zeros outside the group) in both packed layouts: wp_f[9][I][Opad] for p2i_conv_fwd
 *   and wp_d[9][O][Ipad] for p2i_conv_dgrad.  ksz==1: DoW = W.reshape (no D).  `identity_rep`>0
 *   adds the `+ x.repeat_interleave(rep, dim=1)` skip of p2igan.py:79 as a centre-tap identity.
 * p2i_doconv_fold_bwd: dW, dD from the packed weight gradient produced by p2i_conv_wgrad.
 */
int p2i_doconv_fold_fwd(const float* W, const float* D, const float* D_diag, int O, int I, int groups,
                        int ksz, int identity_rep, float* wp_f, float* wp_d, void* stream);
int p2i_doconv_fold_bwd(const float* dwp_f, const float* W, const float* D, const float* D_diag,
                        int O, int I, int groups, int ksz, float* dW, float* dD, void* stream);

/* The same fold / fold backward for n <= 16 layers of ONE shape (3x3, groups 1, O and I multiples of 32: the generator's
 * residual stack has 8 per level) in one / two launches; arrays of n HOST entries holding device pointers. */
int p2i_doconv_fold_fwd_batched(const float* const* W, const float* const* D, const float* const* D_diag, int n, int O, int I,
                                float* const* wp_f, float* const* wp_d, void* stream);
int p2i_doconv_fold_bwd_batched(const float* const* dwp_f, const float* const* W, const float* const* D,
                                const float* const* D_diag, int n, int O, int I, float* const* dW, float* const* dD,
                                void* stream);

/* Plain (O, I, ntaps) weights <-> packed.  scale_ptr (device scalar, may be NULL) divides: used
 * for the spectral-norm weight = weight_orig / sigma. */
int p2i_weight_pack(const float* w, int O, int I, int ntaps, const float* inv_div_ptr,
                    float* wp_f, float* wp_d, void* stream);
/* dw[o][i][tap] = dwp_f[tap][i][o] / sigma  - (sum(dwp_f .* w)/sigma^2) * u[o] * v[i*ntaps+tap]
 * (gradient through torch.nn.utils.spectral_norm's `weight / sigma`); sigma_ptr NULL => plain unpack. */
int p2i_weight_unpack_grad(const float* dwp_f, int O, int I, int ntaps, const float* w_orig,
                           const float* sigma_ptr, const float* u, const float* v, float* scratch,
                           float* dw, void* stream);

/* p2i_weight_pack / p2i_weight_unpack_grad for n <= 16 layers of different shapes in one launch each (arrays of n HOST
 * entries).  The caller zeroes padded pack outputs (pad32(O) != O or pad32(I) != I); `dots`: n floats of device scratch. */
int p2i_weight_pack_batched(const float* const* w, const int* O, const int* I, const int* ntaps, const float* const* inv_div,
                            float* const* wp_f, float* const* wp_d, int n, void* stream);
int p2i_weight_unpack_grad_batched(const float* const* dwp_f, const int* O, const int* I, const int* ntaps,
                                   const float* const* w_orig, const float* const* sigma, const float* const* u,
                                   const float* const* v, float* dots, float* const* dw, int n, void* stream);

/* p2i_weight_unpack_grad_batched with accumulate != 0: dw += ... (the discriminator's second backward pass of a D step adds to
 * the first one's gradient: train.py:264-283 calls D twice before loss_d.backward()) */
int p2i_weight_unpack_grad_batched_acc(const float* const* dwp_f, const int* O, const int* I, const int* ntaps,
                                       const float* const* w_orig, const float* const* sigma, const float* const* u,
                                       const float* const* v, float* dots, float* const* dw, int n, int accumulate, void* stream);

/* ------------------------------------------------------------------ spectral norm
 * One power iteration of torch.nn.utils.spectral_norm (call sites layer.py:402-407,
 * p2igan.py:141): v <- normalize(W^T u), u <- normalize(W v), sigma = u^T W v, eps 1e-12.
 * training==0: sigma only from the stored u, v.  W is (O, K) row-major.  scratch >= O + K + 4 floats.
 */
int p2i_spectral_norm(const float* w, int O, int K, float* u, float* v, int training, float* sigma,
                      float* scratch, void* stream);

/* The same power iteration for n <= 16 layers at once (blockIdx.z = layer): 4 launches for the whole discriminator instead
 * of 4-5 per layer.  Arrays of n HOST entries holding device pointers / sizes; scratch[i] >= O[i] + K[i] + 4 floats. */
int p2i_spectral_norm_batched(const float* const* w, const int* O, const int* K, float* const* u, float* const* v,
                              int training, float* const* sigma, float* const* scratch, float* const* u_snap,
                              float* const* v_snap, int n, void* stream);   /* u_snap / v_snap (may be NULL): copies of the
                              updated u, v for the backward pass (training only) */

/* ------------------------------------------------------------------ generator glue
 * AttentionBlock x2 (layer.py:296-304, 318-322): per pixel relu(x + x*(Wx+b)) over the T=16 vector. */
int p2i_attn_fwd(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                 float* out, int B, int T, int HW, void* stream);
int p2i_attn_bwd(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                 const float* dout, float* dw0, float* db0, float* dw1, float* db1,
                 int B, int T, int HW, void* stream);

/* Gauge-point IDW (layer.py:324-361 + 259-293 + 246-256): nonzero(mask>0) in (t,y,x) order,
 * 4-NN by torch.cdist's fp32 formula (|a|^2+|b|^2-2ab as an fma chain), torch.topk's
 * partial-sort selection, w = 1/(d+tau)^2 normalised (+1e-12).  grid_[xyz] are the
 * torch.linspace(0,1,n) tables.  Work buffers (caller-allocated):
 *   pt_pos  int32 [B*Q]  flat (t,y,x) index of each point;  pt_count int32 [B] ; frame_count int32[B*T]
 *   row_start int32 [B*T*(H+1)]  first point of frame t in a row >= y (lets the scan skip, exactly, the
 *             frames / rows that cannot beat the current 4th-nearest distance)
 *   sel_idx int32 [B*Q*4] (indices into the point list), sel_w float [B*Q*4]  (saved for backward)
 * Empty mask => zeros (layer.py:330-332).  0 < N < 4 points in a sample is an error in the reference (torch.topk with k > N
 * raises, layer.py:282).  The library is enqueue-only and cannot raise without a device sync: it writes ZEROS for that sample
 * (out, sel_idx, sel_w; p2i_idw_bwd then gives a zero gradient), other samples of the batch are unaffected, and pt_count[b]
 * holds the count for a caller that wants the reference's error (the Python wrapper: P2I_IDW_STRICT=1, one host sync).
 * Pinned by tests/test_ops_gpu.py::test_idw_fewer_than_four_points_gives_zeros_or_raises. */
int p2i_idw_fwd(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int B, int T, int H, int W, float tau,
                void* stream);
/* The same result from a two-pass search (round 3).  The reference's scan order matters only where the 4th and 5th smallest
 * computed distances of a voxel are EQUAL (which tied point is kept depends on the heap's history); everywhere else the set of
 * selected points is the four nearest, whatever the order.  Pass 1 searches outwards from the voxel (own frame and rows first:
 * ~10 % of the distance evaluations of the index-order scan) and lists the voxels it finds tied; pass 2 replays the reference's
 * scan for those only (3-4 % with one gauge mask shared by all frames).  Ties among the four selected points permute equal
 * weights in the 4-term output sum (<= 1 ulp of the output; sel_idx / sel_w order may differ there).
 * Round 4: pass 2 first DECIDES a listed voxel from the points within its (now known) 4th distance D, met in index order -- the
 * reference's heap holds the four smallest values seen so far, so the first four points with d <= D stay unless a point with d < D
 * follows them (idw.hip, IDW_CONSIDER_FIXED) -- and only the voxels where one does (~7 %) take the full replay, from a second list.
 *   amb int32 [B*(2Q+2+ceil(Q/256))]: per sample the number of listed voxels, the lists of pass 1's workgroups (256 slots each) and
 *       their lengths / prefix sums; behind the B samples' blocks the second list, per sample [count, voxels ...] (work buffer). */
int p2i_idw_fwd_ws(const float* vals_src, const float* mask, const float* grid_x, const float* grid_y,
                   const float* grid_z, float* out, int32_t* pt_pos, int32_t* pt_count, int32_t* frame_count,
                   int32_t* row_start, float* pt_xyzn, int32_t* sel_idx, float* sel_w, int32_t* amb, int B, int T, int H, int W,
                   float tau, void* stream);
int p2i_idw_bwd(const float* dout, const int32_t* pt_pos, const int32_t* pt_count, const int32_t* sel_idx,
                const float* sel_w, float* dvals_src, int B, int T, int H, int W, void* stream);

/* DownsampleDuplicateChannels (layer.py:205-214): 2x2 max-pool then duplicate every channel. */
int p2i_pooldup_fwd(const float* x, float* y, int B, int C, int H, int W, void* stream);
int p2i_pooldup_bwd(const float* x, const float* dy, float* dx, int B, int C, int H, int W, void* stream);

/* UPPos front half (layer.py:392-396): u = bilinear_x2(x, align_corners=True) * 2*sigmoid(pos). */
int p2i_upmod_fwd(const float* x, const float* pos, float* u, int B, int C, int S, int S2w /*in W*/, void* stream);
/* The same with the rest of UPPos behind it: u = act(bilinear_x2(v) * 2*sigmoid(pos) + bias[c]).  The modulation is ONE factor per
 * pixel for all channels and the upsampling acts on every channel alike, so the 1x1 projection (layer.py:390,397) commutes with
 * both: relu(W (up(h) * s) + b) == relu(up(W h) * s + b).  The build projects at the LOW resolution (a quarter of the positions,
 * v = W h through p2i_conv_fwd) and finishes here; equal to the reference's order up to fp32 rounding of the reordered sums. */
int p2i_upmod_fwd_ba(const float* v, const float* pos, const float* bias, int act, float* u, int B, int C, int S, int S2w, void* stream);
int p2i_upmod_bwd(const float* x, const float* pos, const float* du, float* dx, float* dpos,
                  int B, int C, int S, int S2w, void* stream);

/* ------------------------------------------------------------------ discriminator tail
 * p2igan.py:165-173: mean over T' of out3d, bilinear (align_corners=False) to out2d's size,
 * fused = sigmoid(alpha2d)*out2d + that. */
int p2i_dtail_fwd(const float* out2d, const float* out3d, const float* alpha2d, float* fused,
                  int B, int H2, int W2, int T3, int H3, int W3, void* stream);
int p2i_dtail_bwd(const float* out2d, const float* alpha2d, const float* dfused, float* dout2d,
                  float* dout3d, float* dalpha2d, int B, int H2, int W2, int T3, int H3, int W3, void* stream);

/* ------------------------------------------------------------------ losses
 * ReconstructionLoss (losses.py:38-48,56-85): out[0]=pool (weighted L1), out[1]=reg (KL of
 * temporal-difference softmaxes, batchmean), out[2]=pool+k1*reg; dpred = d out[2]/d pred.
 * scratch >= B*(T-1)*HW + 4096 floats. */
int p2i_recloss(const float* pred, const float* target, float k1_alpha, float* out3, float* dpred,
                float* scratch, int B, int T, int HW, void* stream);
/* Hinge / lsgan / nsgan losses (losses.py:210-226) with gradients; loss_type 0 hinge, 1 lsgan (MSE to the label), 2 nsgan
 * (nn.BCELoss on the raw logits, losses.py:201-202: any logit outside [0,1] is an error in torch; here *loss comes back NaN
 * and the Python wrapper raises).  mode: 0 = discriminator (0.5*(L(real)+L(fake)), train.py:266-283), 1 = generator
 * (hinge: -mean * weight; otherwise L(logits, real_label) * weight, train.py:301-308). */
int p2i_gan_loss(const float* logits_a, const float* logits_b, int n, int loss_type, int mode,
                 float weight, float real_label, float fake_label, float* loss, float* dlogits_a,
                 float* dlogits_b, void* stream);

/* ------------------------------------------------------------------ optimiser
 * torch.optim.Adam step on flat fp32 buffers (train.py:125-136): no weight decay, no amsgrad. */
int p2i_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
             float beta2, float eps, int step, void* stream);

/* ------------------------------------------------------------------ batch assembly
 * Loader post-processing on the device (sti_dataset.py:209,223-224; train.py:468-473): frames = u8/255,
 * masks = (mask_u8 != 0), masked = frames*masks, all (B,T,1,H,W) fp32.  mask_u8 has mask_numel elements:
 * H*W (one gauge map for all frames and samples: `stis`/`sti`), T*H*W or B*T*H*W. */
int p2i_assemble_batch(const uint8_t* frames_u8, const uint8_t* mask_u8, int64_t mask_numel, float* frames,
                       float* masked, float* masks, int B, int T, int H, int W, void* stream);

/* ------------------------------------------------------------------ sliding-window inference (infer.py:188-262)
 * Window w of an event (L frames of HW floats) = frames w*step .. w*step+win-1, frames past the end repeat the last one
 * (infer.py:219-227).  p2i_window_gather builds windows w0 .. w0+nw-1 of up to two tensors (masked frames and masks) as
 * (nw, win, HW); p2i_window_mean turns the generator's predictions for ALL nwin windows, (nwin, win, HW), into the event:
 * out[l] = max(0, scale * mean over the windows that hold frame l) (the repeated copies are not counted, infer.py:229-245). */
int p2i_window_gather(const float* a, const float* b /*may be NULL*/, float* wa, float* wb, int L, int64_t HW, int w0, int nw, int win,
                      int step, void* stream);
int p2i_window_mean(const float* pred_windows, float* out, int L, int64_t HW, int nwin, int win, int step, float scale, void* stream);

/* ------------------------------------------------------------------ evaluation metrics
 * metrics/metric.py on the device.  p2i_metrics_pointwise (one pass over pred/target, n elements): sums2[0] += sum|d|,
 * sums2[1] += sum d^2 with d = T(pred) - T(target), T(x) = 10^(x/16)*0.036 when apply_transform (RegressionMetrics.update
 * :42-52); counts[k*4 + {0 hits, 1 misses, 2 false alarms, 3 correct negatives}] += ... for T(pred), T(target) >=
 * thresholds[k] (CategoricalMetrics.update :92-111; always transformed); bits (may be NULL): byte plane, bit k =
 * T(pred)>=thr[k], bit 4+k = T(target)>=thr[k], the input of p2i_metrics_fss.  thresholds/scales are HOST arrays (<= 4).
 * p2i_metrics_fss (FractionalSkillScoreMetric.update :152-170): num/den[k*ns + s] += sum over the avg_pool2d(kernel s,
 * stride 1, padding s/2) fraction fields of (fp-ft)^2 and fp^2+ft^2; the caller divides by N*ho*wo (ho = H+2*(s/2)-s+1). */
int p2i_metrics_pointwise(const float* pred, const float* target, int64_t n, const float* thresholds_host, int nt,
                          int apply_transform, float* sums2, unsigned long long* counts, uint8_t* bits, void* stream);
int p2i_metrics_fss(const uint8_t* bits, int N, int H, int W, int nt, const int* scales_host, int ns, float* num, float* den,
                    void* stream);

/* Same update with the step counter kept on the device: step_dev[0] is incremented by the call and the bias
 * corrections are derived from it there (coef2: 2 floats of scratch), so that the launch can be captured in a hipGraph
 * and replayed every training step. */
int p2i_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, int32_t* step_dev, float* coef2, void* stream);

/* small helpers on the same stream */
int p2i_axpy(float* y, const float* x, float a, int64_t n, void* stream);        /* y += a*x */
int p2i_add2(float* out, const float* a, const float* b, int64_t n, void* stream);   /* out = a + b (the skip x_4 + res1, p2igan.py:95) */
int p2i_zero(float* p, int64_t n, void* stream);                                     /* memset 0 on the stream */
/* out = dy * act'(y) for a saved post-activation tensor y (may alias dy): ReLU / LeakyReLU(0.2) / tanh backward */
int p2i_act_bwd(const float* dy, const float* y, int act, float* out, int64_t n, void* stream);
int p2i_bias_grad(const float* dy, const float* y_act, int act, float* db, int B, int C, int64_t inner, void* stream);
/* both in one pass: out = dy * act'(y) and db[c] += its sum over samples and positions (db zeroed by the caller; inner % 4 == 0) */
int p2i_act_bwd_bias(const float* dy, const float* y, int act, float* out, float* db, int B, int C, int64_t inner, void* stream);

/* ---- Native step sequencer: launch tapes (round 4).
 * Replaces the per-launch host work of the reference's step and inference loops -- scripts/train.py:240-326 (one G+D iteration),
 * scripts/infer.py:217-241 (one window batch) -- which on this build is ~400 kernel launches + ~400 other calls issued one by one
 * through the binding (7.1 ms of host time per step, whatever the batch).  Between p2i_tape_begin and p2i_tape_end every kernel
 * launch, memset and stream dependency this library enqueues ON THE CALLING THREAD is also appended to a tape (the calls execute as
 * usual, or are captured if the stream is capturing); p2i_tape_replay re-enqueues the recorded operations with the recorded
 * arguments in one call.  Rules:
 *   - the tape holds HOST memory only (opaque handle, freed by p2i_tape_free); every device buffer it names is the caller's and
 *     must still be alive, at the same address, when it is replayed (the Python engine records inside a private memory pool);
 *   - operations recorded on `origin_stream` are replayed on the origin stream given to p2i_tape_replay (the caller's current
 *     stream); operations on other streams (side streams) are replayed on those same streams;
 *   - cross-stream dependencies must be expressed through p2i_event_record / p2i_event_wait (slots 0..255 of a per-thread table
 *     of events without timing) to be seen by a tape; enqueue-only like everything else, no synchronisation;
 *   - host-side values baked into kernel arguments are replayed as recorded (use p2i_adam_dev, whose step counter lives on the
 *     device). */
/* ---- Pre-split source planes (round 4, EXPERIMENT: measured, not used by the engine -- DESIGN.md section 7).  p2i_x6_split_planes:
 * fp32 activations (B, C, P) (P = frames x pixels, C % 8 == 0) -> their exact 3-way bf16 split as planes [plane][b][C/8][P][8]
 * (3 * B * C * P bf16, caller-owned).  p2i_x6_next_source_planes: the next split-pipe forward / data-gradient call of the calling
 * thread (p2i_conv_fwd_x6s / _dgrad_x6s ...) reads its source from these planes instead of splitting the fp32 tensor in the kernel
 * (32-channel tiles with whole-chunk stages; any other tile ignores them); the same deconv_pytorch.py:103-109 convolution, bit-equal
 * results (tests/test_ops_gpu.py::test_presplit_source_planes_are_bit_equal). */
int p2i_x6_split_planes(const float* x, void* planes, int B, int C, int64_t P, void* stream);
int p2i_x6_next_source_planes(const void* planes);

/* ---- Run-to-run reproducibility (round 4).  The step's small cross-workgroup sums (bias / position / attention-weight / alpha
 * gradients, spectral-norm dot products, the 1 -> 32 layer's weight gradient, the IDW backward scatter) are float atomics by
 * default: their order, hence their rounding, changes from run to run (the reference on CPU is deterministic).  With a scratch
 * registered here they are summed in a fixed order instead (the workgroups store their partials, one more small launch adds them
 * in workgroup order; bias gradients ride in the weight-gradient slices; the IDW scatter runs in 64-bit fixed point).  part: >= 65 536
 * floats (the Python binding gives 128 MB: pieces are handed out round-robin and must not be reused while an earlier kernel may
 * still run); counters: >= 1 024 unsigned, zero-initialised (reserved).
 * Process-wide (one GPU per process); part == NULL unregisters. */
int p2i_det_workspace(float* part, int64_t part_floats, unsigned* counters, int n_counters);
int p2i_event_record(int slot, void* stream);
int p2i_event_wait(int slot, void* stream);
int p2i_tape_begin(void* origin_stream);
int p2i_tape_end(void** tape_out);
/* counts4: kernels, memsets, event operations, distinct streams */
int p2i_tape_info(const void* tape, int* counts4);
int p2i_tape_replay(const void* tape, void* origin_stream);
int p2i_tape_free(void* tape);

#ifdef __cplusplus
}
#endif
#endif /* P2I_HIP_H */
