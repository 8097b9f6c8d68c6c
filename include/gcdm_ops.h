/* gcdm_ops.h -- C ABI of libgcdm_ops.so: the module-level operators of the GCPNet block as stand-alone device functions, forward and
 * backward (gfx950 / MI355X).  Plain C99: device pointers, sizes, a hipStream_t passed as void*; every function returns 0 on success,
 * -1 for a bad argument, -2 if the launch failed.  All tensors fp32, row-major; index tensors int64 (torch's edge_index), CSR pointers
 * int32.  Work is enqueued on `stream`; nothing is allocated, nothing synchronises.
 *
 * What each entry replaces in the reference (BioinfoMachineLearning/bio-diffusion, src/models/components/):
 *   gcdm_op_gemm            every nn.Linear of GCP / GCP2 (gcpnet.py:85-118, 320-348) and its autograd: y = x W^T + b, dx = dy W, dW = dy^T x
 *   gcdm_op_colsum[_slices] the bias gradient of those Linears (with many rows: partial sums over row slices on the whole chip + gcdm_op_reduce_slices)
 *   gcdm_op_act[_bwd]       get_nonlinearity(...) (components/__init__.py: relu / leakyrelu / selu / silu) and torch.sigmoid (gcpnet.py:135,404)
 *   gcdm_op_norm3[_bwd]     safe_norm over the spatial axis (components/__init__.py:275-286; gcpnet.py:231,402,406,448)
 *   gcdm_op_scalarize[_bwd] scalarize (components/__init__.py:174-224): frames x vectors -> 3 x CH scalars; node mode = the same kernel on the
 *                           row-mean of the frames (the scatter-mean of a product that is linear in the frame)
 *   gcdm_op_vectorize[_bwd] vectorize (components/__init__.py:227-272), used by frame_gate (gcpnet.py:163-175, 393-403)
 *   gcdm_op_rowscale[_bwd]  `vector_rep * gate.unsqueeze(-1)` (gcpnet.py:136,404,175)
 *   gcdm_op_rowptr / gather / segment_sum / segment_bwd / scatter_add
 *                           ScalarVector.idx(row / col) (gcpnet.py:688-689), torch_scatter.scatter(sum | mean) (gcpnet.py:723; components/__init__.py:214,262)
 *   gcdm_op_localize        localize (components/__init__.py:123-171)
 *   gcdm_op_edge_features / gcdm_op_orientations
 *                           _edge_features / _orientations of the featuriser the dynamics call (gcpnet.py:1105-1109)
 *   gcdm_op_centralize      centralize (components/__init__.py:45-92) and the per-molecule mean removal of gcpnet.py:1222-1228
 *   gcdm_op_fc_edges        get_fully_connected_edge_index (gcpnet.py:1054-1066)
 *
 * Reference-side binding: none needed -- bio-diffusion_amd/ops.py wraps each pair as a torch.autograd.Function, and
 * bio-diffusion_amd/gcpnet.py composes them in modules with the reference's names and signatures (plug point 3). */
#ifndef GCDM_OPS_H
#define GCDM_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] = A[M,K] . B[K,N] (+ bias[N]); A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; C row-major.  slices > 1 splits K: slice z
 * writes C + z*M*N (bias in slice 0) and gcdm_op_reduce_slices adds them in slice order (deterministic; used for dW, where K = #entities). */
int gcdm_op_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, const float* bias, int64_t M, int32_t N,
                 int64_t K, int32_t slices, void* stream);
int gcdm_op_reduce_slices(const float* part, float* out, int64_t n, int32_t slices, void* stream);
int gcdm_op_colsum(const float* dy, float* db, int64_t M, int32_t N, void* stream);
/* partial column sums over `slices` contiguous row ranges, part [slices][N] (then gcdm_op_reduce_slices(part, db, N, slices)): the bias gradient when M = #edges */
int gcdm_op_colsum_slices(const float* dy, float* part, int64_t M, int32_t N, int32_t slices, void* stream);

/* kind: 0 identity, 1 silu, 2 relu, 3 sigmoid, 4 leakyrelu(0.01), 5 selu */
int gcdm_op_act(int32_t kind, const float* x, float* y, int64_t n, void* stream);
int gcdm_op_act_bwd(int32_t kind, const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* out[m][c] = sqrt(sum_xyz v^2 + 1e-8) + 1e-8; rep_layout 1: v is [M][C][3], 0: v is [M][3][C] */
int gcdm_op_norm3(const float* v, float* out, int64_t M, int32_t C, int32_t rep_layout, void* stream);
int gcdm_op_norm3_bwd(const float* v, const float* out, const float* dout, float* dv, int64_t M, int32_t C, int32_t rep_layout, void* stream);

/* u [M][3][CH] ("pre" layout), F [M][9] -> out [M][3*CH], out[m][3c + r] = F[m][r][:] . u[m][:][c] */
int gcdm_op_scalarize(const float* u, const float* F, float* out, int64_t M, int32_t CH, void* stream);
int gcdm_op_scalarize_bwd(const float* dout, const float* F, float* du, int64_t M, int32_t CH, void* stream);
/* gate [M][3*KC], F [M][9] -> out [M][KC][3] */
int gcdm_op_vectorize(const float* gate, const float* F, float* out, int64_t M, int32_t KC, void* stream);
int gcdm_op_vectorize_bwd(const float* dout, const float* F, float* dgate, int64_t M, int32_t KC, void* stream);

/* out[m][c][:] = v[m][c][:] * g[m][c] */
int gcdm_op_rowscale(const float* v, const float* g, float* out, int64_t M, int32_t C, void* stream);
int gcdm_op_rowscale_bwd(const float* v, const float* g, const float* dout, float* dv, float* dg, int64_t M, int32_t C, void* stream);

/* rowptr [N+1] of a row-sorted edge list; *flag |= 1 if it is not sorted (caller zeroes flag) */
int gcdm_op_rowptr(const int64_t* row, int64_t E, int64_t N, int32_t* rowptr, int32_t* flag, void* stream);
int gcdm_op_gather(const float* x, const int64_t* idx, float* out, int64_t E, int32_t C, void* stream);
int gcdm_op_segment_sum(const float* x, const int32_t* rowptr, float* out, int64_t N, int32_t C, int32_t mean, void* stream);
int gcdm_op_segment_bwd(const float* dout, const int64_t* row, const int32_t* rowptr, float* dx, int64_t E, int32_t C, int32_t mean, void* stream);
/* out (pre-zeroed by the caller) [N][C] += dy[e] at idx[e]: fp32 atomics */
int gcdm_op_scatter_add(const float* dy, const int64_t* idx, float* out, int64_t E, int32_t C, void* stream);

int gcdm_op_localize(const float* x, const int64_t* row, const int64_t* col, float* F, int64_t E, int32_t norm_x_diff, void* stream);
int gcdm_op_edge_features(const float* x, const int64_t* row, const int64_t* col, float* e_out, float* xi_out, int64_t E, void* stream);
int gcdm_op_orientations(const float* x, float* out, int64_t N, void* stream);
int gcdm_op_centralize(const float* x, const int64_t* batch_index, const uint8_t* mask, float* out, int64_t N, int32_t D, void* stream);
/* noff [B+1] node offsets (int32), eoff [B+1] edge offsets (int64; eoff[b+1] - eoff[b] = n_b^2), both on the device */
int gcdm_op_fc_edges(const int32_t* noff, const int64_t* eoff, int32_t B, int64_t* row, int64_t* col, int64_t E, void* stream);

#ifdef __cplusplus
}
#endif
#endif
