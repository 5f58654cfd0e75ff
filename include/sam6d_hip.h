/*
 * libsam6d_hip.so -- C ABI of the MI355X (gfx950) implementation of SAM-6D's geometric-matching hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / ATen types.  Every pointer is a
 * DEVICE pointer unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream);
 * launches are asynchronous on that stream, never synchronise, never allocate, and are graph-capturable.
 * Tensors are dense row-major float32 / int32 exactly as the reference's call sites hand them over
 * (callers `.contiguous()` first: PEM/utils/model_utils.py:77-79, PEM/model/fine_point_matching.py:128,136).
 *
 * Return value: 0 on success; <0 for a rejected argument (SAM6D_EINVAL = -1, the analogue of the reference's
 * TORCH_CHECK in EXT/include/utils.h:20-45); >0 = hipError_t of a failed launch (the reference prints and exit(-1)s,
 * EXT/include/cuda_utils.h:42-51).  sam6d_last_error() returns the message of the last failure on this thread.
 *
 * Path shorthands: PEM = SAM-6D/Pose_Estimation_Model, EXT = PEM/model/pointnet2/_ext_src,
 *                  ISM = SAM-6D/Instance_Segmentation_Model (all under the reference repository root).
 */
#ifndef SAM6D_HIP_H
#define SAM6D_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* sam6d_last_error(void);
/* ABI version of this header: bumped on any change of a signature OR of the size / layout of a buffer an entry reads or writes.
 * A consumer compares sam6d_abi_version() of the loaded library with the SAM6D_ABI_VERSION it was compiled against and refuses a
 * mismatch (sam6d_hip/_lib.py and tests/cabi/cabi_check.c do).
 *   1  rounds 1-3
 *   2  sam6d_set_thread_matmul_mode added; round 3's layout change made visible: sam6d_linattn_kv_pack / sam6d_linattn_kv_image write 4*B floats to `inv` (one image
 *      scale per head), sam6d_linattn_layer reads kvinv as (B,4) -- a version-1 consumer allocated B floats */
#define SAM6D_ABI_VERSION 2
int sam6d_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * B1: the `pointnet2._ext` seam (EXT/src/bindings.cpp:11-24).  Semantics = the reference's CPU loops, bit-exact.
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces `at::Tensor furthest_point_sampling(at::Tensor points, const int nsamples)`
 * (EXT/src/sampling.cpp:184-212; loop :76-118).  xyz (B,N,3) f32 -> idx (B,m) i32.
 * temp: (B,N) f32 scratch, required only when N > 4096 (the reference allocates the same `tmp`, :192-194). */
int sam6d_furthest_point_sampling(const float* xyz, int B, int N, int m, float* temp, int* idx, void* stream);
/* Test hook of the multi-workgroup FPS (N > 4096; EXT/src/sampling.cpp:76-118 is the loop it replaces): the number of polls a
 * workgroup waits at the grid-wide arg-max of a round before it gives the cloud up to the one-workgroup kernel queued behind it
 * (cap < 0: the default, 2^24; 0: give up at once).  Results are identical on both paths. */
int sam6d_fps_debug_spin_cap(long cap);

/* replaces `at::Tensor gather_points(at::Tensor points, at::Tensor idx)` (EXT/src/sampling.cpp:120-150; loop :23-44).
 * points (B,C,N) f32, idx (B,M) i32 -> out (B,C,M) f32; out-of-range index -> 0. */
int sam6d_gather_points(const float* points, const int* idx, int B, int C, int N, int M, float* out, void* stream);

/* replaces `at::Tensor ball_query(at::Tensor new_xyz, at::Tensor xyz, const float radius, const int nsample)`
 * (EXT/src/ball_query.cpp:64-93; loop :16-62).  new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32,
 * every slot written (empty ball -> zeros, like the reference's zero-initialised output). */
int sam6d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample, int* idx,
                     void* stream);
/* sam6d_ball_query for two radii over the same (queries, cloud) in one pass -- PositionalEncoding queries r1 = 0.1 / 32 samples
 * and r2 = 0.2 / 64 samples around the same points (PEM/model/fine_point_matching.py:108-131, EXT/src/ball_query.cpp:16-62);
 * idx1 (B,M,nsample1), idx2 (B,M,nsample2), each identical to the single-radius call. */
int sam6d_ball_query2(const float* new_xyz, const float* xyz, int B, int N, int M, float radius1, int nsample1, int* idx1,
                      float radius2, int nsample2, int* idx2, void* stream);
/* sam6d_ball_query2 with spatial pruning (same call site, PEM/model/fine_point_matching.py:108-131; semantics EXT/src/ball_query.cpp:16-62,
 * identical outputs): the cloud is bucketed into a uniform grid first and a query only tests the points of the 27 cells around it;
 * the first-nsample-in-index-order rule is kept by collecting hits in a per-query bit mask.  ws: sam6d_ball_query2_grid_workspace_bytes(B, N)
 * bytes of scratch, 16-byte aligned.  N > 8192 falls back to the all-pairs scan. */
size_t sam6d_ball_query2_grid_workspace_bytes(int B, int N);
int sam6d_ball_query2_grid(const float* new_xyz, const float* xyz, int B, int N, int M, float radius1, int nsample1, int* idx1,
                           float radius2, int nsample2, int* idx2, void* ws, size_t ws_bytes, void* stream);

/* replaces `at::Tensor group_points(at::Tensor points, at::Tensor idx)` (EXT/src/group_points.cpp:79-108; loop :20-45).
 * points (B,C,N) f32, idx (B,M,S) i32 -> out (B,C,M,S) f32. */
int sam6d_group_points(const float* points, const int* idx, int B, int C, int N, int M, int S, float* out,
                       void* stream);

/* Row-major companion of gather_points used inside the pipeline (the reference transposes to (B,C,N), gathers and
 * transposes back: PEM/utils/model_utils.py:76-80, PEM/model/transformer.py:667-705):
 * out[b,j,:] = feats[b, idx[b,j] + idx_off, :], rows of C floats (C % 4 == 0); batch strides in floats. */
int sam6d_gather_rows(const float* feats, const int* idx, int B, int N, int M, int C, long in_stride_b,
                      long out_stride_b, int idx_off, float* out, void* stream);

/* The same with row 0 of every feats[b] supplied separately: lead[b] (rows lead_stride_b floats apart) stands for feats[b,0,:], which
 * need not be written.  out[b,0,:] = lead[b,:], out[b,1+j,:] = feats[b, idx[b,j] + idx_off, :] (an index that lands on row 0 reads
 * lead[b]) -- the sparse tokens of SparseToDenseTransformer in one launch: bg token + FPS rows of the cat [bg; dense]
 * (PEM/model/transformer.py:667-705).  `out` rows 0 .. M. */
int sam6d_gather_rows_lead(const float* feats, const int* idx, int B, int N, int M, int C, long in_stride_b, long out_stride_b,
                           int idx_off, const float* lead, long lead_stride_b, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * B2: building blocks of the PEM Python modules (CoarsePointMatching / FinePointMatching / GeometricTransformer /
 * model_utils).  The reference runs these as torch ops; here each is one or a few HIP launches.
 * ---------------------------------------------------------------------------------------------------------- */

/* C = act(((A . W^T) / divisor) * colscale + bias) + residual  on the fp32 matrix cores.
 * replaces nn.Linear / 1x1-conv call sites (PEM/model/transformer.py:127-129,186-188,390-393,548-550;
 * PEM/model/coarse_point_matching.py:35-38) and the similarity contraction (PEM/utils/model_utils.py:144-150).
 * A (M,K) lda; W (N,K) ldw; C (M,N) ldc; residual (M,N) ldr or NULL; bias/colscale (N) or NULL; act 0 none / 1 ReLU,
 * + 16 marks a geometric operand (the proj_p / Chebyshev folds of the RPE query) that keeps the fp16 x3 split in matmul mode 2;
 * `batch` independent problems with strides sA/sW/sC/sR (floats).  divisor = 1 disables the division. */
int sam6d_gemm_nt(const float* A, const float* W, const float* bias, const float* colscale, const float* residual,
                  float* C, int M, int N, int K, long lda, long ldw, long ldc, long ldr, int batch, long sA, long sW,
                  long sC, long sR, float divisor, int act, void* stream);
/* sam6d_gemm_nt with the weight operand cut into fp16 hi / lo halves ONCE, at weight-load time: Wh / Wl = the two outputs of
 * sam6d_split_f16(W, n, w_scale) -- same (N,K) layout, ldw and batch stride as W, w_scale a power of two that puts max |W| into
 * [2^13, 2^14).  Staging the weight tile is then a copy instead of a per-k-step split (the same nn.Linear call sites,
 * PEM/model/transformer.py:127-129,186-188,390-393,548-550).  W (fp32) is still required: out-of-range activation tiles are
 * recomputed with the exact fp32 loop.  In matmul mode 0 this is sam6d_gemm_nt. */
int sam6d_gemm_nt_w16(const float* A, const float* W, const void* Wh, const void* Wl, float w_scale, const float* bias,
                      const float* colscale, const float* residual, float* C, int M, int N, int K, long lda, long ldw, long ldc,
                      long ldr, int batch, long sA, long sW, long sC, long sR, float divisor, int act, void* stream);
/* sam6d_gemm_nt with a second (inner) batch level and no epilogue: problem (b1, b2) uses A + b1*sA + b2*sA2 etc.
 * (the per-cloud, per-head q.k^T and P.v products of the attention: PEM/model/transformer.py:137-149, 408-419). */
int sam6d_gemm_nt_b2(const float* A, const float* W, float* C, int M, int N, int K, long lda, long ldw, long ldc, int batch,
                     long sA, long sW, long sC, int batch2, long sA2, long sW2, long sC2, void* stream);

/* Matrix-core arithmetic of gemm_nt / geo_embed: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain),
 * 1 = fp16 x3 split (x = hi + lo in fp16, 3 MFMAs, ~1e-6 relative; default), 2 = fp16 single product (hi . hi only, fp32
 * accumulate, ~1e-3 relative: the "fp16 MFMA GeometricTransformer" arithmetic of BASELINE.json config 5; applies to sam6d_gemm_nt,
 * the fused block / cross-attention / fine-match kernels and the RPE score contraction; the geometric indices, the outlier embedding
 * rows and the PE MLP keep the split).  All replace the same F.linear call sites (PEM/model/transformer.py:127-129); the mode is
 * process-wide.
 * Operand range in mode 1: ANY finite fp32 input is accepted.  sam6d_gemm_nt keeps the split for tiles whose operands lie in
 * [2^-6, 2^15) in magnitude (by their tile maximum; what the matching path produces itself is always there) and recomputes a tile
 * whose A or W block leaves that range with the exact fp32 MFMA loop of mode 0, so |x| >= 65520 cannot become inf - inf and uniformly
 * tiny operands do not lose their fp16 lo halves; the fused block kernels (sam6d_token_block, sam6d_linattn_layer,
 * sam6d_fine_match) scale every operand row / matrix by a power of two instead (exact to undo). */
int sam6d_set_matmul_mode(int mode);
int sam6d_get_matmul_mode(void);
/* The calling THREAD's override of the process-wide mode (same F.linear call sites, PEM/model/transformer.py:127-129): 0 / 1 / 2 as
 * above, -1 = follow the process default again.  sam6d_get_matmul_mode() returns the override while one is set.  A host that serves
 * two weight sets with different arithmetic in one process brackets each call sequence with it (sam6d_hip/pem.py does, from the
 * Options object its entry points take) instead of flipping the process default under another thread's launches. */
int sam6d_set_thread_matmul_mode(int mode);
int sam6d_get_thread_matmul_mode(void);

/* nn.LayerNorm(256) over `rows` rows (PEM/model/transformer.py:158,189,436,597).  eps as in torch (1e-5). */
int sam6d_layernorm256(const float* x, const float* gamma, const float* beta, float* y, long rows, long ldx, long ldy,
                       float eps, void* stream);
/* attention.linear / output.squeeze + residual + LayerNorm of a transformer block in one launch
 * (PEM/model/transformer.py:152-158, 184-199, 436-441, 597-603): Y (M,256) = LayerNorm(A (M,K) . W (256,K)^T + bias + residual)
 * * gamma + beta, eps as nn.LayerNorm.  Split-precision mode only (rc < 0 in mode 0: use sam6d_gemm_nt + sam6d_layernorm256). */
int sam6d_gemm_ln256(const float* A, const float* W, const float* bias, const float* residual, const float* gamma,
                     const float* beta, float* Y, int M, int K, long lda, long ldw, long ldr, long ldy, float eps, void* stream);

/* Fused transformer-block kernels (csrc/block.hip): a tile of 128 token rows (256 channels) stays on chip from the block's input to
 * its second LayerNorm; split-precision (fp16 x3) MFMA arithmetic with power-of-two operand scaling (range-safe for any finite fp32
 * input: no operand can overflow fp16).
 * sam6d_token_block: the tail of every attention layer -- AttentionLayer / RPEAttentionLayer after the attention itself, then
 *   AttentionOutput (PEM/model/transformer.py:152-160, 184-199, 436-444):
 *     y = LayerNorm(hidden . Wlin^T + b + x);  out = LayerNorm(relu(y . Wexp^T + b) . Wsq^T + b + y)        all (M,256)
 * sam6d_linattn_layer: the whole LinearTransformerLayer of the dense lift (PEM/model/transformer.py:532-622) on rows row0 .. I-1 of
 *   each of the B clouds of D (B,I,256): proj_q, focused kernel function, z, (phi(q) kv) z per head, then the tail above with x = D.
 *   kvimage / kvinv: sam6d_linattn_kv_pack of the (B,4,64,64) kv^T that sam6d_linattn_kv returns -- kvinv (B,4): the inverse
 *   power-of-two image scale of every head; ksum (B,256) its key sums.
 * wimage: the layer's weights as the kernels' LDS panel image (sam6d_token_block_image_bytes(mode) bytes, mode 1 = with proj_q),
 *   written by sam6d_pack_panels: rows x K fp32 -> 32-row panels of fp16 hi/lo halves, K in `ksteps` steps of 16 from column k0,
 *   values multiplied by `scale` (a power of two).  Image order: linear (8 panels, K=256) | 4 x { expand rows 128c.. (4 panels,
 *   K=256), squeeze columns 128c.. (8 panels, K=128) } | proj_q (8 panels, K=256; mode 1).
 * consts: 2568 floats: b_q | 1/softplus(scale) | b_lin | gamma1 | beta1 | b_expand (512) | b_squeeze | gamma2 | beta2 |
 *   {1/s_q, 1/s_lin, 1/s_exp, 1/(s_sq s_h), s_h, 0, 0, 0} with s_* the pack scales and s_h the scale of the FFN hidden row. */
int sam6d_pack_panels(const float* W, long ldw, int rows, int k0, int ksteps, float scale, void* dst, void* stream);
/* Front of RPEMultiHeadAttention.forward (PEM/model/transformer.py:395-405) in one launch: qkv (M,768) = x [Wq;Wk;Wv]^T + b, the
 * query folded through proj_p per head, qp (M,4,256) = Wp_h^T q_h (see sam6d_attention), and qd (M*4,32) = D_c^T qp_h (see
 * sam6d_rpe_scores).  wimage (sam6d_rpe_front_image_bytes() bytes) = sam6d_pack_panels of [Wq;Wk;Wv] (768 rows, ksteps 8), then per
 * head h of proj_p.weight^T (256 rows, k0 = 64 h, ksteps 2), then of D_c^T (32 rows, ksteps 8); inv_* = 1 / the three pack scales. */
long sam6d_rpe_front_image_bytes(void);
int sam6d_rpe_front(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc, float* qkv,
                    float* qp, float* qd, long M, void* stream);
/* The same launch with the values written TRANSPOSED per cloud of n tokens, vT (M / n, 256, ldp) -- the W operand of the P.v product
 * (sam6d_gemm_nt_b2), i.e. the `torch.matmul(attention_scores, v)` of PEM/model/transformer.py:416 -- instead of into the v third of
 * qkv (which is then left untouched): no separate sam6d_transpose pass. */
int sam6d_rpe_front_vt(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc, float* qkv,
                       float* qp, float* qd, long M, float* vT, int n, int ldp, void* stream);
long sam6d_token_block_image_bytes(int mode);
long sam6d_linattn_kv_image_bytes(void);
int sam6d_linattn_kv_pack(const float* kvT, int B, void* image, float* inv, void* stream);
/* The k / v side of LinearAttention (PEM/model/transformer.py:547-572) in one launch: kv (B clouds, J rows of [256 k | 256 v] channels,
 * row stride ld, cloud stride `stride`, UNfocused keys) -> phi(k), kv^T and the key sums of the 4 heads, and the packed image of kv^T:
 * the same bits as sam6d_linattn_focus_k + sam6d_linattn_kv + sam6d_linattn_kv_pack (kv itself is left untouched). */
int sam6d_linattn_kv_image(const float* kv, const float* scale, int B, int J, long ld, long stride, void* image, float* inv,
                           float* ksum, void* stream);
int sam6d_token_block(const float* hidden, const float* x, const void* wimage, const float* consts, float* out, long M, float eps,
                      void* stream);
int sam6d_linattn_layer(const float* D, const void* wimage, const float* consts, const void* kvimage, const float* kvinv,
                        const float* ksum, float* Dout, int B, int I, int row0, float eps, void* stream);

/* replaces GeometricStructureEmbedding.forward (PEM/model/transformer.py:343-363; indices :306-341; sinusoid :259-285).
 * points (B,n,3) (bg point already prepended) -> out (B,n,n,256).  Workspaces: knn_ws (B*n*3 + 1) i32 (the last int is
 * the "index beyond the fast-sincos range" flag), idx_ws (B*n*n*4) f32. */
int sam6d_geo_embedding(const float* points, int B, int n, const float* div_term, const float* Wd, const float* bd,
                        const float* Wa, const float* ba, float sigma_d, float factor_a, int angle_k, int hidden,
                        int* knn_ws, float* idx_ws, float* out, void* stream);
/* the two halves of sam6d_geo_embedding: get_embedding_indices (PEM/model/transformer.py:306-341) -> idx_ws
 * (B,n,n,4) = {d_idx, a_idx[0..2]}, and the sinusoid + proj_d/proj_a + max contraction (:343-363) over `pairs` rows. */
int sam6d_geo_indices(const float* points, int B, int n, float sigma_d, float factor_a, int angle_k, int* knn_ws,
                      float* idx_ws, void* stream);
int sam6d_geo_embed(const float* idx_ws, long pairs, const float* div_term, const float* Wd, const float* bd,
                    const float* Wa, const float* ba, int hidden, const int* flag, int only_if_large, float* out,
                    void* stream);

/* fp16 x3 split-precision form of sam6d_geo_embed (same call site, PEM/model/transformer.py:343-363): w_packed =
 * [16][2][256][40] halves = per 16-wide K chunk, per matrix (proj_d, proj_a), per output column: 16 hi | 16 lo | 8 pad halves of
 * weight*1024, produced with sam6d_split_f16 (x -> fp16(x*scale), fp16(x*scale - hi)) at weight-load time.
 * `flag` = the last int of knn_ws written by sam6d_geo_indices: when an embedding index exceeds the range of the
 * branch-free sincos (1e5; never for radius-normalised clouds) the h3 kernel returns at once and a following
 * sam6d_geo_embed(..., flag, only_if_large = 1, ...) launch produces the result with the library sincosf. */
int sam6d_split_f16(const float* x, long n, float scale, void* hi, void* lo, void* stream);
int sam6d_geo_embed_h3(const float* idx_ws, long pairs, const float* div_term, const void* w_packed, const float* bd,
                       const float* ba, int hidden, const int* flag, float* out, void* stream);
/* Same contract as sam6d_geo_embed_h3 (PEM/model/transformer.py:343-363) with the sinusoid contraction replaced by a
 * 32-term Chebyshev expansion of proj_d(sinusoid(x)) / proj_a(sinusoid(x)) on [0, xmax] (w_cheb: 2 x 256 rows of
 * 144 B = 32 hi | 32 lo | 8 pad fp16 halves of coefficient * 1024, built at weight-load time in float64).  Pairs with
 * an index outside [0, xmax] are marked in pos_ws (pairs ints) / collected in list_ws (1 + pairs ints) and computed by the sinusoid kernel
 * (div_term, w_packed as for sam6d_geo_embed_h3), so the result is independent of xmax up to the ~1e-7 split-precision
 * error.  No-op when *flag != 0 (indices beyond the fast-sincos range: follow with sam6d_geo_embed(only_if_large=1)). */
int sam6d_geo_embed_cheb(const float* idx_ws, long pairs, const void* w_cheb, float xmax, const float* div_term,
                         const void* w_packed, const float* bd, const float* ba, int hidden, const int* flag, int* pos_ws,
                         int* list_ws, float* out, void* stream);

/* Fused RPE self-attention (RPETransformerLayer, PEM/model/transformer.py:366-420, on top of
 * GeometricStructureEmbedding :343-363) WITHOUT a materialised embedding.  Three entry points:
 *
 * sam6d_geo_outliers: after sam6d_geo_indices.  pos_ws[pair] (pairs ints) = -1 when all four indices of the pair lie in
 *   [0, xmax], else its row in `rows` (pairs x 256 floats capacity; only the listed rows are written): the bias-free
 *   embedding proj_d(sin(d)) + max_k proj_a(sin(a_k)) of that pair from the sinusoid kernels (w_packed / Wd / Wa, flag as
 *   for sam6d_geo_embed_h3 / sam6d_geo_embed).  list_ws: 1 + pairs ints of scratch.
 * sam6d_rpe_scores: P[q][h][0..n) (row stride ldp) = softmax_m((qk[q][h][m] + qp[q][h][:] . E[q][m][:]) / 8) for the Q = B*n
 *   queries, E rebuilt per key tile from the 32-term Chebyshev basis (wa_cheb = the proj_a half of the sam6d_geo_embed_cheb
 *   image; qd[q][h][0..32) = D_c^T qp[q][h], D_c = Chebyshev coefficients of proj_d) or read from `rows` for listed pairs.
 *   qp (Q,4,256) = proj_p folded into the query (see sam6d_attention), qk (Q,4,ldp) = q_h . k_h[m]; qk is updated in place
 *   (the geometric term of the listed pairs is added to it first, one wave per pair of list_ws).
 * sam6d_transpose: dst[b][c][j] = src[b][j][c] (the values as the N x K operand of the P.V GEMM).
 * sam6d_geo_outliers2 / sam6d_rpe_scores2: the same two steps with a range of their own for the three angular indices
 *   ([0, xmax_a]; the angle of GeometricStructureEmbedding.get_embedding_indices, transformer.py:326-341, times 180 / (sigma_a pi) never
 *   exceeds 180 / sigma_a = 12) -- wa_cheb then holds proj_a's expansion on [0, xmax_a] -- and `products` = 3 (every split product over
 *   all 32 orders) or 2 (the cross terms of the orders 0..15 in one MFMA, those of the orders >= 16 dropped: the caller guarantees
 *   that 2^-10 sum_{p >= 16} |c[ch][p]| is negligible for every channel).  The entries without the suffix = xmax_a = xmax, products = 3. */
int sam6d_geo_outliers(const float* idx_ws, long pairs, float xmax, const float* div_term, const void* w_packed, const float* Wd,
                       const float* Wa, const int* flag, int* pos_ws, int* list_ws, float* rows, void* stream);
int sam6d_rpe_scores(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb, float xmax,
                     const float* qp, const float* qd, float* qk, float* P, long Q, int n, int ldp, void* stream);
int sam6d_geo_outliers2(const float* idx_ws, long pairs, float xmax, float xmax_a, const float* div_term, const void* w_packed,
                        const float* Wd, const float* Wa, const int* flag, int* pos_ws, int* list_ws, float* rows, void* stream);
int sam6d_rpe_scores2(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb, float xmax,
                      float xmax_a, int products, const float* qp, const float* qd, float* qk, float* P, long Q, int n, int ldp,
                      void* stream);
/* The same layer with the attention itself in one launch per (cloud, head) (RPEMultiHeadAttention.forward,
 * PEM/model/transformer.py:405-416): sam6d_rpe_geo_scores writes only the geometric score term G[q][h][0..n) = qp[q][h] . E[q][m]
 * (row stride ldp, not yet divided by 8; arguments as sam6d_rpe_scores2 without qk); sam6d_rpe_self_attention then computes, for
 * qkv (B n, 768) = q | k | v of sam6d_rpe_front, hidden[:, 64h .. 64h+64) = softmax((q_h k_h^T + G) / 8) v_h  (n <= 208, ldp a
 * multiple of 4).  Replaces the q.k^T GEMM, the softmax inside sam6d_rpe_scores2 and the P.v GEMM. */
int sam6d_rpe_geo_scores(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb, float xmax,
                         float xmax_a, int products, const float* qp, const float* qd, float* G, long Q, int n, int ldp, void* stream);
int sam6d_rpe_self_attention(const float* qkv, const float* G, float* hidden, int B, int n, int ldp, void* stream);
int sam6d_transpose(const float* src, long ld_src, long stride_src, int B, int n, int ncol, float* dst, long ld_dst,
                    long stride_dst, void* stream);

/* Cross attention of a TransformerLayer with the query projection inside (PEM/model/transformer.py:95-150: proj_q, per-head
 * softmax(q k^T / 8) v for 4 heads x 64) on the matrix cores, one workgroup per (cloud, head).  x (B,n,256) query-side tokens;
 * kv (B,m,512) = [proj_k | proj_v] of the memory tokens (one sam6d_gemm_nt); wq_image = sam6d_pack_panels(proj_q.weight, 256 rows,
 * k0 = 0, ksteps = 8, scale s) (256 KiB), inv_wq_scale = 1 / s; bq (256).  out (B,n,256) = the attention output before
 * AttentionLayer.linear.  n <= 256, m <= 208. */
int sam6d_cross_attention(const float* x, const float* kv, const void* wq_image, const float* bq, float inv_wq_scale, float* out,
                          int B, int n, int m, void* stream);
/* sam6d_cross_attention with the key-side projection inside as well (MultiHeadAttention.forward, PEM/model/transformer.py:127-129:
 * proj_k / proj_v of the memory tokens): mem (B,m,256) the memory tokens; wkv_image (sam6d_cross_attention_kv_image_bytes() bytes) =
 * per head h the sam6d_pack_panels images of proj_k.weight rows 64h..64h+64 then proj_v.weight rows 64h..64h+64 (K = 256, ksteps 8,
 * one common pack scale, inv_wkv_scale its inverse); bkv (512) = proj_k.bias | proj_v.bias.  k and v never reach HBM. */
long sam6d_cross_attention_kv_image_bytes(void);
int sam6d_cross_attention_kv(const float* x, const float* mem, const void* wq_image, const float* bq, float inv_wq_scale,
                             const void* wkv_image, const float* bkv, float inv_wkv_scale, float* out, int B, int n, int m,
                             void* stream);

/* replaces MultiHeadAttention.forward / RPEMultiHeadAttention.forward core (PEM/model/transformer.py:131-148,395-418):
 * 4 heads x 64, softmax((q.k [+ qp.E]) / 8) v.  q (B,n,256) ldq/sq; k,v (B,m,256); out (B,n,256).
 * RPE form: qp (B,n,4,256) = per-head query folded through proj_p, E (B,n,m,256); pass both NULL for the plain form. */
int sam6d_attention(const float* q, const float* k, const float* v, const float* qp, const float* E, float* out, int B,
                    int n, int m, long ldq, long ldk, long ldv, long ldo, long sq, long sk, long sv, long so,
                    void* stream);
/* Stand-alone pieces for callers of a single attention sub-module (the fused kernels never materialise the probabilities):
 * sam6d_scaled_softmax: out[r][0..n) = softmax_m((a[r][m] + b[r][m]) * scale), b may be NULL -- the softmax of MultiHeadAttention.forward
 *   (PEM/model/transformer.py:134-143) and, with b = the q . proj_p(E) term, of RPEMultiHeadAttention.forward (:404-412).
 * sam6d_sinusoid_embed: SinusoidalPositionalEmbedding.forward (PEM/model/transformer.py:269-285): x (n) -> out (n, d_model),
 *   out[i][2j] = sin(x_i div_term_j), out[i][2j+1] = cos(x_i div_term_j). */
int sam6d_scaled_softmax(const float* a, const float* b, float scale, long rows, int n, long lda, long ldb, float* out, long ldo,
                         void* stream);
int sam6d_sinusoid_embed(const float* x, long n, const float* div_term, int d_model, float* out, void* stream);

/* LinearAttention.forward pieces (PEM/model/transformer.py:546-578, kv path :569-572). */
int sam6d_linattn_focus_k(float* k, const float* scale, long rows, long ld, void* stream);
int sam6d_linattn_kv(const float* k, const float* v, int B, int J, long ldk, long ldv, long sk, long sv, float* kvT,
                     float* ksum, void* stream);
int sam6d_linattn_focus_q(float* q, const float* scale, const float* ksum, int B, long rows_per_b, long ld,
                          void* stream);

/* PositionalEncoding, one scale (PEM/model/fine_point_matching.py:126-139): QueryAndGroup
 * (PEM/model/pointnet2/pointnet2_utils.py:383-396) + SharedMLP 6->32->64->128 with eval BatchNorm folded to per-channel
 * scale/shift (PEM/model/pointnet2/pytorch_utils.py:25-50) + max over the ball, fused.  idx (B,N,S) from sam6d_ball_query
 * with new_xyz = pts + 1e-8; writes out[(b*N+j)*ldo + off + c], c < 128.  S must be a multiple of 32.
 * Range (split-precision mode): the six input features of a neighbour are scaled by a power of two chosen per neighbour, so layer 1
 * is accurate for coordinates of any magnitude; the hidden activations of layers 2 / 3 are split into fp16 halves unscaled and must
 * stay below 65504 -- coordinates below ~1e4 for weights of order one (PEM encodes radius-normalised clouds, |x| < 10).  Matmul mode 0
 * (exact fp32) has no such limit. */
int sam6d_pe_mlp_max(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1,
                     const float* sh1, const float* W2, const float* sc2, const float* sh2, const float* W3,
                     const float* sc3, const float* sh3, float* out, long ldo, int off, void* stream);
/* sam6d_pe_mlp_max (PositionalEncoding's SharedMLPs, PEM/model/fine_point_matching.py:113-144) with an upper bound on the persistent
 * workgroups of the split-precision kernel (0 = fill the chip, 3 per CU): a launch that runs beside another stream's kernels leaves
 * them LDS and registers with 256 or 512. */
int sam6d_pe_mlp_max_wg(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1, const float* sh1,
                        const float* W2, const float* sc2, const float* sh2, const float* W3, const float* sc3, const float* sh3,
                        float* out, long ldo, int off, int max_wg, void* stream);

/* y = (x - t) @ R per batch element (PEM/model/fine_point_matching.py:45). */
int sam6d_rigid_inverse(const float* x, const float* R, const float* t, int B, int N, float* y, void* stream);
/* strided row-block copy (the torch.cat call sites that add the bg token, PEM/model/coarse_point_matching.py:36-38). */
int sam6d_put_rows(const float* src, long s_src_b, long ld_src, float* dst, long s_dst_b, long ld_dst, int B, int rows,
                   int C, void* stream);
/* cat([bg_point(100,100,100), pts]) (PEM/model/pose_estimation_model.py:30-34). */
int sam6d_prepend_bg_point(const float* pts, int B, int n, float* out, void* stream);
/* y = x + s (new_xyz = pts + 1e-8, PEM/model/fine_point_matching.py:117) and a flat device copy of n floats
 * (batch stacking: the .repeat / torch.cat call sites PEM/run_inference_custom_pytorch.py:445-446). */
int sam6d_add_scalar(const float* x, float s, long n, float* y, void* stream);
/* flag[0] = 1 if x (B, n) holds B bitwise-identical rows, else 0: recognises the template tensors the reference's caller
 * `.repeat`s per instance (PEM/run_inference_custom_pytorch.py:445-446), whose pose-independent work then runs once. */
int sam6d_batch_rows_equal(const float* x, int B, long n, int* flag, void* stream);
int sam6d_copy_f32(const float* src, float* dst, long n, void* stream);
/* F.normalize(dim=-1) on 256-wide rows (PEM/utils/model_utils.py:141-142). */
int sam6d_l2norm256(const float* x, float* y, long rows, long ldx, long ldy, void* stream);

/* Soft assignment statistics and labels (PEM/utils/model_utils.py:229-235, 320-324): att (B,R,C);
 * rmax/rsum (B,R), cmax/csum (B,C), label1 (B,R-1) i32, label2 (B,C-1) i32; ws: scratch of >= 32*B*C floats. */
int sam6d_soft_assign(const float* att, int B, int R, int C, float* rmax, float* rsum, float* cmax, float* csum,
                      int* label1, int* label2, float* ws, long ws_floats, void* stream);
/* sam6d_soft_assign + sam6d_coarse_weights in one launch for matrices that fit the 160 KB of LDS of a CU (R * C + 3 R + 3 C floats:
 * the 197 x 197 coarse attention), one workgroup per proposal; every output has the bits of the two-call form (same per-element
 * arithmetic, same summation orders: PEM/utils/model_utils.py:229-240).  rc < 0 for larger matrices. */
int sam6d_coarse_soft_assign(const float* att, int B, int R, int C, float* rmax, float* rsum, float* cmax, float* csum, int* label1,
                             int* label2, float* weights, float* w1, void* stream);
/* replaces pairwise_distance(x, y) (PEM/utils/model_utils.py:101-128; normalized = False, channel-last): x (B,N,3), y (B,M,3) ->
 * out (B,N,M) = clamp((|x|^2 - 2 x.y) + |y|^2, min = 0) with the bits of torch's CPU evaluation (K = 3 matmul = an fma chain, plain
 * sums of squares: SURVEY 8c n1/n2).  The path's kernels inline the same device function; this entry is the direct check. */
int sam6d_pairwise_distance(const float* x, const float* y, int B, int N, int M, float* out, void* stream);
/* Sampling weights (S[1:,1:] * w1 * w2) ** 1.5 -> (B,(R-1)*(C-1)), w1 (B,R-1) (PEM/utils/model_utils.py:234-238). */
int sam6d_coarse_weights(const float* att, int B, int R, int C, const float* rmax, const float* rsum, const float* cmax,
                         const float* csum, const int* label1, const int* label2, float* weights, float* w1,
                         void* stream);
/* replaces cumsum + weighted_sampling_onnx_compatible (PEM/utils/model_utils.py:241-250, 277-305): `rand` (B,ns) are
 * the uniforms the reference draws with torch.rand (:292); idx (B,ns) i32; cum_ws (B,L) f32 scratch. */
int sam6d_weighted_sample(const float* weights, const float* rand, int B, int L, int ns, float* cum_ws, int* idx,
                          void* stream);
/* 3-point Procrustes per hypothesis + residual (PEM/utils/model_utils.py:244-257): idx (B,3*nh) -> Rs (B,nh,9),
 * ts (B,nh,3), dis (B,nh). */
int sam6d_coarse_hypotheses(const int* idx, const float* pts1, const float* pts2, int B, int N1, int N2, int nh,
                            float* Rs, float* ts, float* dis, void* stream);
/* torch.topk(k, largest=False) indices, ascending (PEM/utils/model_utils.py:258). */
int sam6d_select_smallest(const float* dis, int B, int n, int k, int* sel, void* stream);
/* Hypothesis scoring + argmax (PEM/utils/model_utils.py:261-270); model (B,P,3) raw, radius (B). */
int sam6d_score_select_hypotheses(const int* sel, const float* Rs, const float* ts, const float* pts1, const float* w1,
                                  const float* model, const float* radius, int B, int N1, int P, int nh, int k,
                                  float* scores, float* R, float* t, int* best, void* stream);
/* The same scoring + arg-max (PEM/utils/model_utils.py:258-275; the distance contraction is the reference's own torch.matmul,
 * :117) with the K = 3 products on the fp32 matrix cores -- identical distance bits -- and the weighted distances staged in `ws`
 * (sam6d_score_select_workspace_bytes(B, N1, k) bytes), summed per hypothesis in a fixed order.  P <= 4096. */
size_t sam6d_score_select_workspace_bytes(int B, int N1, int k);
int sam6d_score_select_hypotheses_ws(const int* sel, const float* Rs, const float* ts, const float* pts1, const float* w1,
                                     const float* model, const float* radius, int B, int N1, int P, int nh, int k, float* scores,
                                     float* R, float* t, int* best, float* ws, size_t ws_bytes, void* stream);
/* Fine-stage similarity + soft assignment as one pipeline (csrc/finematch.hip): compute_feature_similarity
 * (PEM/utils/model_utils.py:131-153: F.normalize both sides, f1 f2^T / temp) followed by the head of compute_fine_Rt (:308-331: the
 * two softmaxes, both arg-max label vectors, the masked assignment's row sums and weighted targets) without materialising the
 * attention matrix more than once.  f (2B, n, 256): out_proj outputs, scene clouds 0..B-1 then template clouds; n = 2049 (fine_npoint
 * 2048) or 4097 (BASELINE config 5: 4096-point fine matching; the label and assignment passes then run per 2048-column chunk and two
 * small kernels merge the chunks in ascending column order) -- rc < 0 otherwise: use sam6d_gemm_nt + sam6d_soft_assign +
 * sam6d_fine_assign.  pts2 (B, n-1, 3) template points.
 * -> label1, label2 (B, n-1) i32, pred (B, n-1, 3), weight (B, n-1).  ws: sam6d_fine_match_workspace_bytes_n(B, n) bytes
 * (sam6d_fine_match_workspace_bytes(B): the n = 2049 size). */
size_t sam6d_fine_match_workspace_bytes(int B);
size_t sam6d_fine_match_workspace_bytes_n(int B, int n);
int sam6d_fine_match(const float* f, int B, int n, float temp, const float* pts2, int* label1, int* label2, float* pred,
                     float* weight, void* ws, size_t ws_bytes, void* stream);
/* y = x W^T + b for M token rows of 256 channels on the panel kernel (the sparse-token projections: CoarsePointMatching's in_proj /
 * out_proj, PEM/model/coarse_point_matching.py:35-38, 61; [proj_k; proj_v] of a LinearAttention layer's memory tokens,
 * PEM/model/transformer.py:556-558).  wimage = sam6d_pack_panels(W, 32 npanels rows, k0 = 0, ksteps = 8, scale s), npanels 8 or 16
 * (256 or 512 outputs), inv_w_scale = 1 / s, bias may be null.  Row R = (cloud R / rows_per_cloud, token R % rows_per_cloud) is read
 * at row cloud * x_cloud_rows + x_row0 + token of x and written at row cloud * out_cloud_rows + out_row0 + token of out (32 npanels
 * floats per row): the bg slot of a token buffer is skipped in place on either side. */
int sam6d_rows_linear(const float* x, const void* wimage, int npanels, const float* bias, float inv_w_scale, float* out, long M,
                      int rows_per_cloud, long x_cloud_rows, long x_row0, long out_cloud_rows, long out_row0, void* stream);

/* sam6d_fine_match with the operands already prepared: sam6d_linear_norm_split computes y = x W^T + b for the M rows of x (M, 256), W as the
 * 8-panel image of sam6d_pack_panels (256 rows, k0 = 0, ksteps = 8, scale s; inv_w_scale = 1 / s) -- FinePointMatching's out_proj,
 * PEM/model/fine_point_matching.py:70-72 -- and writes fh | fl (M, 256) fp16 = hi / lo halves of (y / max(|y|, 1e-12)) * 2^10, the
 * F.normalize of compute_feature_similarity (PEM/utils/model_utils.py:141-142) and the operand split of the similarity product in one
 * pass; sam6d_fine_match_split takes those halves instead of f (same workspace size). */
int sam6d_linear_norm_split(const float* x, const void* wimage, const float* bias, float inv_w_scale, void* fh, void* fl, long M,
                            void* stream);
int sam6d_fine_match_split(const void* fh, const void* fl, int B, int n, float temp, const float* pts2, int* label1, int* label2,
                           float* pred, float* weight, void* ws, size_t ws_bytes, void* stream);
/* Fine soft-assignment reduction (PEM/utils/model_utils.py:325-331): pred (B,R-1,3), weight (B,R-1). */
int sam6d_fine_assign(const float* att, int B, int R, int C, const float* rmax, const float* rsum, const float* cmax,
                      const float* csum, const int* label1, const int* label2, const float* pts2, float* pred,
                      float* weight, void* stream);
/* replaces weighted_procrustes (PEM/utils/model_utils.py:343-436; CustomSVD/CustomDet :469-526); weights may be NULL. */
int sam6d_weighted_procrustes(const float* src, const float* ref, const float* weights, int B, int N,
                              float weight_thresh, float eps, float* R, float* t, void* stream);
/* Fine pose score and translation rescale (PEM/utils/model_utils.py:331-339; PEM/model/fine_point_matching.py:78);
 * t is scaled in place by (radius + 1e-6); cnt_ws (2*B) f32 scratch. */
int sam6d_fine_score(const float* pts1, const float* R, float* t, const float* model, const float* radius,
                     const int* label1, int B, int N, int P, float dis_thres, float* cnt_ws, float* score, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * B3: ISM template scoring (Instance_Segmentation_Model methods + model.loss classes).
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces PairwiseSimilarity.forward (ISM/model/loss.py:27-44): query (Nq,D), ref (No,Nt,D) -> scores (Nq,No,Nt) in [0,1]. */
int sam6d_ism_cosine(const float* query, const float* ref, int Nq, int No, int Nt, int D, float* scores, void* stream);
/* replaces the tail of compute_semantic_score + best_template_pose (ISM/model/detector.py:198-207, 265-296):
 * per query the aggregated score (mode 0 avg_5 / 1 mean / 2 max), arg-max object and its best template; `sel` = ascending
 * indices of the queries with score > thresh, *nsel their count (device int). */
int sam6d_ism_semantic(const float* scores, int Nq, int No, int Nt, int mode, float thresh, float* sem, int* obj, int* best,
                       int* sel, int* nsel, void* stream);
/* sam6d_ism_semantic with the survivors written out compacted as the tensors the reference's caller holds after its boolean-mask
 * indexing (ISM/model/detector.py:284-296, best_template_pose :198-207): sel / obj_sel / best_sel (Nq) i64, sem_sel (Nq) f32, the first
 * nsel[0] entries valid; sem_ws / obj_ws / best_ws (Nq) are per-query scratch. */
int sam6d_ism_semantic_compact(const float* scores, int Nq, int No, int Nt, int mode, float thresh, float* sem_ws, int* obj_ws,
                               int* best_ws, long long* sel, long long* obj_sel, float* sem_sel, long long* best_sel, int* nsel,
                               void* stream);
/* replaces compute_straight + compute_visible_ratio reductions (ISM/model/loss.py:52-76) over sim (Ns,P,P). */
int sam6d_ism_patch_scores(const float* sim, const float* q_appe, int Ns, int P, int D, float thred, float* appe, float* vis,
                           void* stream);
/* compute_appearance_score + compute_visible_ratio (ISM/model/detector.py:298-308, 310-322; ISM/model/loss.py:52-76) WITHOUT the
 * (Ns,P,P) similarity tensor and without gathering the chosen templates' descriptors: proposal p multiplies query patches
 * q[qsel ? qsel[p] : p] (P,D) with ref[obj[p]][best[p]] (P,D) read in place (ref: (No,Nt,P,D)); the tile epilogue keeps row / column
 * maxima only.  sam6d_ism_patch_fused fills the workspace (sam6d_ism_patch_fused_workspace_bytes(Ns, P) bytes), sam6d_ism_patch_fused_scores
 * turns it into appe (Ns) and / or vis (Ns) for a threshold (either output may be NULL).  Descriptors must be L2-normalised rows
 * (|x| <= 1, as ISM/model/dinov2.py:322-324 produces them; masked patches all-zero); P a multiple of 128, D of 32; indices int64. */
size_t sam6d_ism_patch_fused_workspace_bytes(int Ns, int P);
int sam6d_ism_patch_fused(const float* q, const long long* qsel, const float* ref, const long long* obj, const long long* best, int Ns,
                          int Nt, int P, int D, void* ws, size_t ws_bytes, void* stream);
int sam6d_ism_patch_fused_scores(const void* ws, int Ns, int P, float thred, float* appe, float* vis, void* stream);
/* replaces project_template_to_image + Calculate_the_query_translation (ISM/model/detector.py:209-246,
 * ISM/utils/trimesh_utils.py:77-105): masks (Ns,H,W) f32, depth (H,W) i32, K (3,3) f64, poses (Nt,4,4) f32,
 * pointcloud (No,Npc,3) -> image_vu (Ns,Npc,2) i32, xyxy (Ns,4) i32, translate (Ns,3); part_ws (Ns*64*4) f64 scratch. */
int sam6d_ism_project(const float* masks, const int* depth, const double* K, double depth_scale, const float* poses,
                      const float* pointcloud, const int* best, const int* obj, int Ns, int H, int W, int Npc,
                      double* part_ws, int* image_vu, int* xyxy, float* translate, void* stream);
/* sam6d_ism_project with the masks read IN PLACE as a 16-byte-per-lane stream (same call sites: ISM/model/detector.py:209-246,
 * ISM/utils/trimesh_utils.py:77-105): masks (Nq,H,W) with mask_bytes = 1 (uint8 / bool, non-zero = inside: SAM's binary proposals) or
 * 4 (float32, as Detections.masks holds them, ISM/model/utils.py:80-95); mask_index (Ns) i64 or NULL: proposal i uses mask
 * mask_index[i] -- the class-token selection of detector.py:289-296 applied without the gathered (Ns,H,W) copy Detections.filter makes.
 * The masked-depth sums are exact integer sums in float64 and the divisions by fx / fy happen once per proposal.  The fast path needs
 * W % 16 == 0, depth_scale > 0 and 16-byte aligned masks / depth; otherwise float32 masks without an index fall back to
 * sam6d_ism_project.  part_ws: sam6d_ism_project_workspace_doubles(Ns, H, W) doubles. */
size_t sam6d_ism_project_workspace_doubles(int Ns, int H, int W);
int sam6d_ism_project2(const void* masks, int mask_bytes, const long long* mask_index, const int* depth, const double* K,
                       double depth_scale, const float* poses, const float* pointcloud, const int* best, const int* obj, int Ns, int H,
                       int W, int Npc, double* part_ws, int* image_vu, int* xyxy, float* translate, void* stream);

/* replaces depth_image_to_pointcloud_translate_torch(depth, scale, K) itself (ISM/utils/trimesh_utils.py:77-105) for a direct caller:
 * masked_depth (N,H,W) f32 = N already-masked depth maps (mm), K (3,3) f64 row-major, -> translate (N,3) f32 = the mean back-projected
 * point of each map over its pixels with Z > 0 (count + 1e-8 in the denominator); float64 per-pixel terms and sums like the reference's
 * real caller.  part_ws: N * 64 * 4 doubles of scratch.  One launch pair for all maps. */
int sam6d_ism_translate_maps(const float* masked_depth, const double* K, double depth_scale, int N, int H, int W, double* part_ws,
                             float* translate, void* stream);
/* replaces compute_iou (ISM/utils/bbox_utils.py:197-222): boxes (Ns,4) int64; *all_positive = 0 when any pair has a
 * non-positive overlap (the reference then returns the scalar 0.0). */
int sam6d_ism_iou(const int* xyxy, const long long* boxes, int Ns, float* iou, int* all_positive, void* stream);
/* final score (ISM/model/detector.py:384): (sem[sel] + appe + geo*vis) / (2 + vis); geo NULL = the scalar-0.0 IoU case. */
int sam6d_ism_final_score(const float* sem, const float* appe, const float* geo, const float* vis, const int* sel, int Ns,
                          float* out, void* stream);
/* sam6d_ism_final_score (ISM/model/detector.py:384) with the IoU quirk of ISM/utils/bbox_utils.py:214-220 decided on the DEVICE: geo is
 * used only while all_positive[0] (the flag sam6d_ism_iou leaves) is non-zero, else the geometric term is 0 for every proposal -- no host
 * read-back between the IoU and the final score. */
int sam6d_ism_final_score_flag(const float* sem, const float* appe, const float* geo, const float* vis, const int* all_positive, int Ns,
                               float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Rows SURVEY 8f marks "next" (callers either side of the path).
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces the radius normalisation of ViTEncoder.forward (PEM/model/feature_extraction.py:133-137):
 * radius (B) = max_n |dense_po[b,n]|; po_out = dense_po / (radius + 1e-6); pm_out = pts / (radius + 1e-6). */
int sam6d_radius_normalize(const float* dense_po, const float* pts, int B, int Npo, int Npm, float* radius, float* po_out,
                           float* pm_out, void* stream);
/* replaces the masked patch-descriptor post-processing of CustomDINOv2 (ISM/model/dinov2.py:265-269, 322-324):
 * out = F.normalize(feats * [AvgPool2d(patch)(masks) > thresh], dim=-1); feats (N,P,D), masks (N,H,W) f32. */
int sam6d_masked_patch_normalize(const float* feats, const float* masks, int N, int P, int D, int H, int W, int patch,
                                 float thresh, float* out, void* stream);
/* replaces get_point_cloud_from_depth (PEM/utils/data_utils.py:92-110) on a resident depth map (H,W) f32, metres:
 * cloud (r1-r0, c1-c0, 3) for the crop bbox [rmin=r0, rmax=r1, cmin=c0, cmax=c1] (the whole image: 0,H,0,W). */
int sam6d_depth_to_cloud(const float* depth, int H, int W, int r0, int r1, int c0, int c1, float fx, float fy, float cx,
                         float cy, float* cloud, void* stream);
/* replaces the test of Detections.remove_very_small_detections (ISM/model/utils.py:96-102): keep[i] (u8) =
 * box_area(boxes[i]) / (H*W) > thr_box && masks[i].sum() / (H*W) > thr_mask; boxes (N,4) int64 xyxy, masks (N,H,W) f32.
 * thr_box = min_box_size**2, thr_mask = min_mask_size. */
int sam6d_detections_small_keep(const long long* boxes, const float* masks, int N, int H, int W, float thr_box,
                                float thr_mask, unsigned char* keep, void* stream);
/* boolean-mask indexing (ISM/model/utils.py:104-105): idx (N) i64 = positions of the non-zero keep bytes, count[0] = K. */
int sam6d_mask_to_indices(const unsigned char* keep, int N, long long* idx, int* count, void* stream);
/* replaces `getattr(self, key)[idxs]` of Detections.filter / apply_nms* (ISM/model/utils.py:105,119,126,190) for a
 * tensor of any dtype: dst[j,:] = src[idx[j],:], rows of row_bytes bytes, idx (M) i64 (negative = from the end;
 * out-of-range rows come back zero). */
int sam6d_take_rows(const void* src, const long long* idx, long n_src, int M, long row_bytes, void* dst, void* stream);
/* replaces mask_to_rle(force_binary_mask(mask)) of convert_npz_to_json, the `segmentation` field of detection_ism.json
 * (ISM/model/utils.py:25-43, 199-216; ISM/utils/bbox_utils.py:190-192): uncompressed COCO RLE of (mask > 0), runs counted in
 * column-major order starting with the zero run.  masks (N,H,W) f32.  Two calls: nruns[i] (N) i32 = number of counts of
 * mask i; then, with offsets (N+1) i64 = exclusive prefix sum of nruns, counts[offsets[i] .. offsets[i+1]) i32 = its runs
 * (a mask whose offsets do not match its run count is left unwritten). */
int sam6d_mask_rle_count(const float* masks, int N, int H, int W, int* nruns, void* stream);
int sam6d_mask_rle_encode(const float* masks, int N, int H, int W, const long long* offsets, int* counts, void* stream);
/* the inverse, as PEM's get_test_data reads the file back (cocomask.decode of an uncompressed RLE,
 * PEM/run_inference_custom_pytorch.py:308-317; same format as ISM/segment_anything/utils/amg.py:138-150):
 * masks (N,H,W) u8 from counts / offsets as above; ws_ends = i32 scratch as long as counts.  Positions beyond the
 * encoded runs come back 0. */
int sam6d_mask_rle_decode(const int* counts, const long long* offsets, int N, int H, int W, int* ws_ends,
                          unsigned char* masks, void* stream);
/* replaces torchvision.ops.nms as Detections.apply_nms (ISM/model/utils.py:121-124, group == NULL) and
 * apply_nms_per_object_id (:107-117, group = object_ids) call it: boxes (N,4) f32 xyxy, scores (N) f32.
 * keep_idx (N) i64 receives the surviving indices, ids ascending, score descending inside an id (stable); count[0] = K.
 * ws: sam6d_nms_workspace_bytes(N) bytes of device memory. */
size_t sam6d_nms_workspace_bytes(int N);
int sam6d_nms(const float* boxes, const float* scores, const long long* group, int N, float thresh, long long* keep_idx,
              int* count, void* ws, size_t ws_bytes, void* stream);
/* Proposal geometry of get_test_data (PEM/run_inference_custom_pytorch.py:316-355) on resident data; masks (N,H,W) u8, depth (H,W)
 * f32 metres.
 * sam6d_mask_bbox: bbox (N,4) i32 = get_bbox((mask > 0) & (depth > 0)) (PEM/utils/data_utils.py:125-160), count (N) valid pixels.
 * sam6d_crop_masked_points: the valid pixels of each crop in row-major order: choose (N,cap) i32 flat crop indices, cloud (N,cap,3)
 *   back-projected points (get_point_cloud_from_depth, data_utils.py:92-110), n_valid (N) (:326-330).
 * sam6d_radius_filter: in-place removal of the points farther than radius * 1.2 from their mean (:331-337); n_keep (N), center (N,3).
 * sam6d_choose_points: pts (N,ns,3) = cloud[sel], rgb_choose (N,ns) i64 = get_resize_rgb_choose(choose[sel], bbox, img_size)
 *   (data_utils.py:113-123); sel (N,ns) i32 is the caller's np.random.choice (:339-345). */
int sam6d_mask_bbox(const unsigned char* masks, const float* depth, int N, int H, int W, int* bbox, int* count, void* stream);
int sam6d_crop_masked_points(const unsigned char* masks, const float* depth, int N, int H, int W, const int* bbox, float fx, float fy,
                             float cx, float cy, int cap, int* choose, float* cloud, int* n_valid, void* stream);
int sam6d_radius_filter(int N, int cap, const int* n_valid, float radius, int* choose, float* cloud, int* n_keep, float* center,
                        void* stream);
int sam6d_choose_points(int N, int cap, const int* choose, const float* cloud, const int* bbox, const int* sel, int ns, int img_size,
                        float* pts, long long* rgb_choose, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM6D_HIP_H */
