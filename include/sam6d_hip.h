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

#ifdef __cplusplus
extern "C" {
#endif

const char* sam6d_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int sam6d_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * B1: the `pointnet2._ext` seam (EXT/src/bindings.cpp:11-24).  Semantics = the reference's CPU loops, bit-exact.
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces `at::Tensor furthest_point_sampling(at::Tensor points, const int nsamples)`
 * (EXT/src/sampling.cpp:184-212; loop :76-118).  xyz (B,N,3) f32 -> idx (B,m) i32.
 * temp: (B,N) f32 scratch, required only when N > 4096 (the reference allocates the same `tmp`, :192-194). */
int sam6d_furthest_point_sampling(const float* xyz, int B, int N, int m, float* temp, int* idx, void* stream);

/* replaces `at::Tensor gather_points(at::Tensor points, at::Tensor idx)` (EXT/src/sampling.cpp:120-150; loop :23-44).
 * points (B,C,N) f32, idx (B,M) i32 -> out (B,C,M) f32; out-of-range index -> 0. */
int sam6d_gather_points(const float* points, const int* idx, int B, int C, int N, int M, float* out, void* stream);

/* replaces `at::Tensor ball_query(at::Tensor new_xyz, at::Tensor xyz, const float radius, const int nsample)`
 * (EXT/src/ball_query.cpp:64-93; loop :16-62).  new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32,
 * every slot written (empty ball -> zeros, like the reference's zero-initialised output). */
int sam6d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample, int* idx,
                     void* stream);

/* replaces `at::Tensor group_points(at::Tensor points, at::Tensor idx)` (EXT/src/group_points.cpp:79-108; loop :20-45).
 * points (B,C,N) f32, idx (B,M,S) i32 -> out (B,C,M,S) f32. */
int sam6d_group_points(const float* points, const int* idx, int B, int C, int N, int M, int S, float* out,
                       void* stream);

/* Row-major companion of gather_points used inside the pipeline (the reference transposes to (B,C,N), gathers and
 * transposes back: PEM/utils/model_utils.py:76-80, PEM/model/transformer.py:667-705):
 * out[b,j,:] = feats[b, idx[b,j] + idx_off, :], rows of C floats (C % 4 == 0); batch strides in floats. */
int sam6d_gather_rows(const float* feats, const int* idx, int B, int N, int M, int C, long in_stride_b,
                      long out_stride_b, int idx_off, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM6D_HIP_H */
