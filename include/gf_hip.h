/*
 * gf_hip.h -- C ABI of libgf_hip.so: MI355X (gfx950) kernels for GraphFlow's second-order CCN/SMP hot path.
 *
 * Boundary contract.  Plain pointers and sizes only; no C++/torch types.  Every entry point
 * returns a gf_status (never aborts, unlike the reference's assert()s) and records a message retrievable with
 * gf_last_error().  Each declaration cites the reference interface it replaces (paths relative to the
 * HyTruongSon/GraphFlow tree).
 *
 * Two calling modes share the same kernels:
 *   mode B "device"  *_f32   : device pointers, a batch of independent graphs, asynchronous on the context's
 *                              HIP stream.  This is the measured path.
 *   mode A "host"    *_host_*: host pointers in the reference's own container layout (N separate Tensor3D
 *                              buffers, double or float); the call stages H2D, runs the same kernels, stages D2H
 *                              and synchronises -- what an Entity-style op's forward()/backward() calls, exactly
 *                              as GraphFlow_gpu/RisiContraction_18_gpu.h:1509-1562 does around its CUDA kernel.
 *
 * Layouts (row-major, channel fastest):
 *   P   [batch][N][N][N][C]   P[g][a][b][c][f]  == tensors[a]->value[(b*N+c)*C+f]     (RisiContraction_18.h:48-53)
 *   A   [batch][N][N]         adj->value[d*N+e]                                        (Matrix.h:34-36)
 *   Out [batch][N][N][K][C]   value[(x*N+y)*K*C + k*C + f]                             (RisiContraction_18.h:29,103)
 *   G   same shape as Out (the op's own `gradient`), dP same shape as P (the inputs' `gradient`).
 */
#ifndef GF_HIP_H_INCLUDED
#define GF_HIP_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gf_ctx gf_ctx;

typedef enum {
    GF_OK = 0,
    GF_ERR_INVALID = 1,     /* bad argument (null pointer, non-positive size, unknown K) */
    GF_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed */
    GF_ERR_NOMEM = 3,       /* workspace allocation failed */
    GF_ERR_UNSUPPORTED = 4, /* shape outside what the kernels implement */
    GF_ERR_TIMEOUT = 5      /* a data-parallel rank waited longer than GF_DIST_TIMEOUT_S for its peers (gf_dist_*) */
} gf_status;

/* ---- context: device + stream + workspace ---------------------------------------------------------------------
 * Replaces the per-op-object cudaMalloc'd buffers and optional cudaStream_t of RisiContraction_18_gpu
 * (GraphFlow_gpu/RisiContraction_18_gpu.h:853-874, 947-955).  One context per host thread / per GPU; not thread-safe,
 * matching the reference's "one model clone per worker thread" rule (SMP_omega.h:115-129).
 * `stream` is a hipStream_t; NULL is the device's default (null) stream, which is where the reference's GPU ops run
 * until set_gpu_stream is called.  gf_ctx_use_private_stream gives the context its own non-blocking stream, the
 * analogue of the per-worker cudaStream_t of SMP_omega_gpu_multistreams.h:130-135.                                 */
gf_status gf_ctx_create(gf_ctx **out, int device, void *stream);
gf_status gf_ctx_use_private_stream(gf_ctx *ctx);
gf_status gf_ctx_destroy(gf_ctx *ctx);
gf_status gf_ctx_set_stream(gf_ctx *ctx, void *stream);     /* RisiContraction_18_gpu::set_gpu_stream (:947) */
void     *gf_ctx_get_stream(gf_ctx *ctx);
gf_status gf_ctx_synchronize(gf_ctx *ctx);                  /* cudaStreamSynchronize in SMP_omega_gpu_multistreams.h:765-771 */
gf_status gf_ctx_reserve(gf_ctx *ctx, size_t workspace_bytes); /* pre-size the scratch so timed regions never allocate */
/* Tracing: per-kernel HIP-event timing on the context's stream (the reference only has wall-clock gettimeofday in its
 * tests, tests/test_RisiContraction_18_gpu.cu:31-40).  Enabling resets the table; reading synchronises the stream. */
gf_status gf_ctx_set_timing(gf_ctx *ctx, int enable);
/* Restrict the timing to launches of ONE kernel name (NULL or "" = all).  The two events around a launch cost about as
 * much as a small kernel and keep neighbouring kernels from overlapping their ramp-up/tail: timing every launch slowed a
 * 72-launch SMP step by 5 %; timing only the kernel under study does not. */
gf_status gf_ctx_set_timing_filter(gf_ctx *ctx, const char *kernel_name);
int       gf_ctx_timing_count(gf_ctx *ctx);
gf_status gf_ctx_timing_get(gf_ctx *ctx, int index, const char **name, double *total_ms, long long *launches);
/* Measurement aid ("a measured device-to-device copy ceiling from the box"): `iters` copies of n floats src -> dst by a
 * hand-written kernel on the context's stream, bracketed by HIP events; *ms_per_copy = their average.  mode 0: float4 loads / stores,
 * grid-stride, one workgroup slot per CU x 8; mode 1: the same with non-temporal loads and stores.  n % 4 == 0.  No reference
 * counterpart (its tests time with gettimeofday, tests/test_RisiContraction_18_gpu.cu:31-40).                                       */
gf_status gf_hbm_copy_probe_f32(gf_ctx *ctx, float *dst, const float *src, size_t n, int mode, int iters, double *ms_per_copy);
const char *gf_last_error(gf_ctx *ctx);                     /* a per-thread copy: valid until this thread's next gf_last_error; ctx may be NULL for create errors */
const char *gf_version(void);
/* Per-context options.  GF_OPT_R18_GENERIC_KERNELS != 0 routes RisiContraction_18 through the layout-agnostic generic
 * kernels (any N, any C, one thread per table / output element) instead of the slab kernels: the independent second
 * implementation the parity tests hold the fast path against.  (The reference's GPU op has a comparable switch -- its CPU
 * fallback under a complexity threshold, RisiContraction_18_gpu.h:961-968 -- but both routes here run on the device.) */
typedef enum {
    GF_OPT_R18_GENERIC_KERNELS = 1,
    /* != 0: the block products of the fused SMP level at 64 channels run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) instead of
     * the f16 pipe with two-half fp32 operands (the default; component-wise fp32-grade inside a 2^17 window per 64-column block,
     * DESIGN.md section 5).  The environment variable GF_SMP_SPLIT=0 selects the same for every context of the process. */
    GF_OPT_SMP_FP32_PRODUCTS = 2
} gf_option;
gf_status gf_ctx_set_option(gf_ctx *ctx, int option, int value);

/* ---- data parallelism: one RCCL communicator per context ----------------------------------------------------------------
 * Replaces the master/worker exchange of SMP_omega::Threaded_BatchLearn (GraphFlow/SMP_omega.h:750-792): copy_value of the
 * master's parameters into every worker clone (:710-728, :771-773) = gf_dist_broadcast_f32; the serial add_gradient loop over
 * the clones (:730-740, :784-786) = gf_dist_allreduce_sum_f32.  A worker is one GPU: one context per GPU, in one process per
 * GPU (torchrun / mpirun style) or one host thread per GPU (the reference's std::thread style).
 * Rank 0 calls gf_dist_unique_id and hands the GF_DIST_ID_BYTES bytes to the other ranks by any host channel (a file, a
 * socket, MPI, torch.distributed, shared memory between threads); then EVERY rank calls gf_dist_init with the same id.
 * Collectives are in place, asynchronous and ordered on the context's stream.  RCCL is loaded on the first gf_dist_* call;
 * when it is missing the call fails with GF_ERR_UNSUPPORTED (nothing else in the library needs it).
 * Once a context has a communicator, gf_smp_backward on it returns the gradient SUMMED OVER ALL RANKS: each level's weight
 * gradients are all-reduced on the communicator's own stream as soon as the reverse sweep has produced them, while the sweep
 * continues below (gf_smp_set_grad_allreduce(smp, 0) turns that off for a handle).                                          */
#define GF_DIST_ID_BYTES 128
gf_status gf_dist_unique_id(gf_ctx *ctx, void *id_out /* GF_DIST_ID_BYTES */);
gf_status gf_dist_init(gf_ctx *ctx, const void *id /* GF_DIST_ID_BYTES */, int rank, int world);
gf_status gf_dist_finalize(gf_ctx *ctx);                   /* also done by gf_ctx_destroy */
int       gf_dist_rank(const gf_ctx *ctx);                 /* 0 when the context has no communicator */
int       gf_dist_world(const gf_ctx *ctx);                /* 1 when the context has no communicator */
gf_status gf_dist_allreduce_sum_f32(gf_ctx *ctx, float *buf, size_t n);
gf_status gf_dist_broadcast_f32(gf_ctx *ctx, float *buf, size_t n, int root);
/* Watchdog.  gf_dist_init, the join at the end of gf_smp_backward's overlapped all-reduces, and this call wait for the peers at most
 * GF_DIST_TIMEOUT_S seconds (environment, default 180, 0 = for ever) and then fail with GF_ERR_TIMEOUT; gf_last_error names the
 * rank, the world size, the device and the last collective handed to RCCL.  gf_dist_quiesce waits (by polling, under that limit)
 * until everything issued so far on the communicator's stream and on the context's stream has completed: call it before a
 * blocking hipStreamSynchronize / hipDeviceSynchronize that would otherwise hang on a peer that never joined.  No-op without a
 * communicator. */
gf_status gf_dist_quiesce(gf_ctx *ctx);

/* ---- tensor contractions, mode B (device pointers, batched) -----------------------------------------------------
 * K selects the family: 4, 10, 18 or 50.
 *   K=18: RisiContraction_18::forward/backward (GraphFlow/RisiContraction_18.h:73-331, 333-560), A gated by A>0 (:90,:345)
 *   K=10: RisiContraction_10 (RisiContraction_10.h:73-153, 155-225), no gate
 *   K=50: RisiContraction_50 (RisiContraction_50.h:73-441, 443-802), no gate
 *   K=4 : RisiContraction_4  (RisiContraction_4.h:68-125, 127-173), A ignored (may be NULL)
 * forward : Out is overwritten (the reference zeroes value first).
 * backward: accumulate != 0 -> dP += vjp (the reference's `+=` contract); accumulate == 0 -> dP = vjp (write-only
 *           fast path, legal when dP has a single consumer, as in the SMP DAG).  A receives no gradient.             */
gf_status gf_contract_forward_f32(gf_ctx *ctx, int K, const float *P, const float *A, float *Out,
                                  int N, int C, int batch);
gf_status gf_contract_backward_f32(gf_ctx *ctx, int K, const float *G, const float *A, float *dP,
                                   int N, int C, int batch, int accumulate);
/* Scratch bytes the two calls above need for (K,N,C,batch); gf_ctx_reserve(max over your shapes) up front. */
size_t gf_contract_workspace_bytes(int K, int N, int C, int batch);

/* ---- tensor contractions, mode A (host pointers, one graph, reference container layout) --------------------------
 * tensors[a] points at the a-th neighbour's Tensor3D::value ([N][N][C]); grads[a] at its ::gradient (always `+=`).
 * Blocking: returns after the result is in host memory.                                                              */
gf_status gf_contract_forward_host_f64(gf_ctx *ctx, int K, const double *const *tensors, const double *A,
                                       double *out_value, int N, int C);
gf_status gf_contract_backward_host_f64(gf_ctx *ctx, int K, const double *out_gradient, const double *A,
                                        double *const *grads, int N, int C);
gf_status gf_contract_forward_host_f32(gf_ctx *ctx, int K, const float *const *tensors, const float *A,
                                       float *out_value, int N, int C);
gf_status gf_contract_backward_host_f32(gf_ctx *ctx, int K, const float *out_gradient, const float *A,
                                        float *const *grads, int N, int C);

/* ---- RisiContraction_18_dropout (GraphFlow/RisiContraction_18_dropout.h:106-477 / :479-783) ----------------------------
 * keep_mask bit k = use[k]: a dropped slice is 0 in forward and ignored in backward.  Train mode: scale = 1 and the mask
 * the host drew (:113-125); test mode: keep_mask = all 18 bits, scale = nKept/18 (:465-471).  The host side
 * (graphflow_amd/host/RisiContraction_hip.h) draws the mask with rand() exactly as the reference does.                  */
gf_status gf_contract18_dropout_forward_f32(gf_ctx *ctx, unsigned keep_mask, float scale, const float *P, const float *A,
                                            float *Out, int N, int C, int batch);
gf_status gf_contract18_dropout_backward_f32(gf_ctx *ctx, unsigned keep_mask, const float *G, const float *A, float *dP,
                                             int N, int C, int batch, int accumulate);
gf_status gf_contract18_dropout_forward_host_f64(gf_ctx *ctx, unsigned keep_mask, double scale, const double *const *tensors,
                                                 const double *A, double *out_value, int N, int C);
gf_status gf_contract18_dropout_backward_host_f64(gf_ctx *ctx, unsigned keep_mask, const double *out_gradient, const double *A,
                                                  double *const *grads, int N, int C);
gf_status gf_contract18_dropout_forward_host_f32(gf_ctx *ctx, unsigned keep_mask, double scale, const float *const *tensors,
                                                 const float *A, float *out_value, int N, int C);
gf_status gf_contract18_dropout_backward_host_f32(gf_ctx *ctx, unsigned keep_mask, const float *out_gradient, const float *A,
                                                  float *const *grads, int N, int C);

/* ---- dense feature mixers, mode B (device pointers) ------------------------------------------------------------------
 * All row-major.  One strided-batched fp32 MFMA GEMM underneath (v_mfma_f32_32x32x2_f32, exact fp32).
 * backward: a NULL gradient pointer skips that operand; accumulate != 0 -> `+=` (the reference contract), else `=`.
 *   MatMul        C[M,N] = A[M,K] B[K,N]                       GraphFlow/MatMul.h:48-67 / :69-82
 *                 (GPU prior art: GraphFlow_gpu/MatMul_gpu.h:28-111)
 *   MatTensorMul  Out[R,J,D] = sum_k X[R,Kd] F[Kd,J,D]         GraphFlow/MatTensorMul.h:47-68 / :70-85
 *   TensorMatMul  Out[R,J,D] = sum_k F[R,Kd,D] Y[Kd,J]         GraphFlow/TensorMatMul.h:46-67 / :69-84                  */
gf_status gf_matmul_forward_f32(gf_ctx *ctx, const float *A, const float *B, float *C, int M, int K, int N);
gf_status gf_matmul_backward_f32(gf_ctx *ctx, const float *dC, const float *A, const float *B, float *dA, float *dB,
                                 int M, int K, int N, int accumulate);
gf_status gf_mattensormul_forward_f32(gf_ctx *ctx, const float *X, const float *F, float *Out, int R, int Kd, int J, int D);
gf_status gf_mattensormul_backward_f32(gf_ctx *ctx, const float *G, const float *X, const float *F, float *dX, float *dF,
                                       int R, int Kd, int J, int D, int accumulate);
gf_status gf_tensormatmul_forward_f32(gf_ctx *ctx, const float *F, const float *Y, float *Out, int R, int Kd, int J, int D);
gf_status gf_tensormatmul_backward_f32(gf_ctx *ctx, const float *G, const float *F, const float *Y, float *dF, float *dY,
                                       int R, int Kd, int J, int D, int accumulate);
/* CustomMatMulTensor (GraphFlow/CustomMatMulTensor.h:47-68 / :70-85), the channel mix of the SMP_2D_ver6-8 drivers:
 *   Out[rows,Kout] = T[rows,V] W^T,  W = [Kout,V] row-major, rows = nRows*nColumns positions of the Tensor3D.
 *   backward: dW (+)= G^T T,  dT (+)= G W.                                                                           */
gf_status gf_custommatmultensor_forward_f32(gf_ctx *ctx, const float *W, const float *T, float *Out, long long rows, int V,
                                            int Kout);
gf_status gf_custommatmultensor_backward_f32(gf_ctx *ctx, const float *G, const float *W, const float *T, float *dW,
                                             float *dT, long long rows, int V, int Kout, int accumulate);
/* StackTensor3D (GraphFlow/StackTensor3D.h:54-73 / :75-90): `tensors` / `grads` are DEVICE arrays of nRows device
 * pointers, each to per_tensor floats; forward copies them into one contiguous buffer, backward scatter-adds back. */
gf_status gf_stack_forward_f32(gf_ctx *ctx, const float *const *tensors, float *out, int nRows, size_t per_tensor);
gf_status gf_stack_backward_f32(gf_ctx *ctx, const float *G, float *const *grads, int nRows, size_t per_tensor);

/* ---- dense feature mixers, mode A (host pointers; gradients are always `+=`; NULL gradient pointers are skipped) --- */
gf_status gf_matmul_forward_host_f64(gf_ctx *ctx, const double *A, const double *B, double *C, int M, int K, int N);
gf_status gf_matmul_backward_host_f64(gf_ctx *ctx, const double *dC, const double *A, const double *B, double *dA,
                                      double *dB, int M, int K, int N);
gf_status gf_mattensormul_forward_host_f64(gf_ctx *ctx, const double *X, const double *F, double *Out, int R, int Kd, int J, int D);
gf_status gf_mattensormul_backward_host_f64(gf_ctx *ctx, const double *G, const double *X, const double *F, double *dX,
                                            double *dF, int R, int Kd, int J, int D);
gf_status gf_tensormatmul_forward_host_f64(gf_ctx *ctx, const double *F, const double *Y, double *Out, int R, int Kd, int J, int D);
gf_status gf_tensormatmul_backward_host_f64(gf_ctx *ctx, const double *G, const double *F, const double *Y, double *dF,
                                            double *dY, int R, int Kd, int J, int D);
gf_status gf_custommatmultensor_forward_host_f64(gf_ctx *ctx, const double *W, const double *T, double *Out, long long rows,
                                                 int V, int Kout);
gf_status gf_custommatmultensor_backward_host_f64(gf_ctx *ctx, const double *G, const double *W, const double *T, double *dW,
                                                  double *dT, long long rows, int V, int Kout);
gf_status gf_custommatmultensor_forward_host_f32(gf_ctx *ctx, const float *W, const float *T, float *Out, long long rows,
                                                 int V, int Kout);
gf_status gf_custommatmultensor_backward_host_f32(gf_ctx *ctx, const float *G, const float *W, const float *T, float *dW,
                                                  float *dT, long long rows, int V, int Kout);
gf_status gf_stack_forward_host_f64(gf_ctx *ctx, const double *const *tensors, double *out, int nRows, size_t per_tensor);
gf_status gf_stack_backward_host_f64(gf_ctx *ctx, const double *G, double *const *grads, int nRows, size_t per_tensor);
gf_status gf_stack_forward_host_f32(gf_ctx *ctx, const float *const *tensors, float *out, int nRows, size_t per_tensor);
gf_status gf_stack_backward_host_f32(gf_ctx *ctx, const float *G, float *const *grads, int nRows, size_t per_tensor);
gf_status gf_matmul_forward_host_f32(gf_ctx *ctx, const float *A, const float *B, float *C, int M, int K, int N);
gf_status gf_matmul_backward_host_f32(gf_ctx *ctx, const float *dC, const float *A, const float *B, float *dA, float *dB,
                                      int M, int K, int N);
gf_status gf_mattensormul_forward_host_f32(gf_ctx *ctx, const float *X, const float *F, float *Out, int R, int Kd, int J, int D);
gf_status gf_mattensormul_backward_host_f32(gf_ctx *ctx, const float *G, const float *X, const float *F, float *dX,
                                            float *dF, int R, int Kd, int J, int D);
gf_status gf_tensormatmul_forward_host_f32(gf_ctx *ctx, const float *F, const float *Y, float *Out, int R, int Kd, int J, int D);
gf_status gf_tensormatmul_backward_host_f32(gf_ctx *ctx, const float *G, const float *F, const float *Y, float *dF,
                                            float *dY, int R, int Kd, int J, int D);

/* ---- batched SMP_omega driver (mode B): the caller of the ops above ------------------------------------------------
 * Reproduces the op DAG of SMP_omega::complete_computation_graph (GraphFlow/SMP_omega.h:584-693) for a batch of
 * molecules: host graph preparation (Floyd-Warshall :358, WL features :382, ranking :406, receptive fields with the
 * omega cap :476-537, selection maps :461, reduced adjacency :556) + device levels (promotion as an index gather,
 * RisiContraction_18, K-projection, bias, LeakyReLU) + readout (:676-692) + the reverse sweep.
 * Parameters are ONE flat fp32 device buffer in the reference's registration / save_model order
 * (SMP_omega.h:289-295, 1033-1055): H[C][F(D+1)], then K_l[18C][C], b_l[C] for l = 1..L, then W[C];
 * gradients have the same layout and hold the SUM over the batch (what sum_gradients accumulates, :808-820), so a
 * data-parallel step is one all-reduce of that buffer.                                                                */
typedef struct gf_smp gf_smp;
typedef struct {
    int nLevels, nChanels, nFeatures, nDepth, max_receptive_field, has_WL_ordering;
    /* The SMP_2D_ver6 / ver7 / ver8 wirings of the same DAG (GraphFlow/SMP_2D_ver6.h:456-560): contraction family
     * nContractions = 10 / 50 / 18 (0 means 18) and, with custom_matmul = 1, the level weight K_l stored [C][nContractions C]
     * and applied by CustomMatMulTensor instead of Reshape2D + MatMul on [nContractions C][C].  Zero-initialised trailing
     * fields give SMP_omega.  Those models have no receptive-field cap: pass max_receptive_field = max_nVertices.
     * Since round 5 the `_10` and `_50` families (nChanels <= 32) are computed on the fused RisiContraction_18 level: for a symmetric
     * reduced adjacency with a unit diagonal their slices are slices of `_18` on the activations and on their per-node transposes (plus
     * three extra products for `_50`); the parameter / gradient layout at this interface stays the caller's.  A batch the embedding cannot
     * take -- an asymmetric or negative adjacency, a `_50` Coulomb batch, a `_10` Coulomb batch with an entry <= 0 -- is computed on the
     * op-by-op `_10` / `_50` levels instead: gf_smp_prepare picks the plan per batch (round 6; it used to refuse with GF_ERR_UNSUPPORTED),
     * nothing changes at this interface.  GF_SMP_VER6_FUSED=0 / GF_SMP_VER7_FUSED=0 at create time select the op-by-op levels for every batch. */
    int nContractions, custom_matmul;
    /* physics = 1: one TOWER of the `_physics` / `_pairgraphs` models (GraphFlow/SMP_omega_physics.h:29-170, :480-606;
     * SMP_omega_pairgraphs.h builds two of them): raw vertex features (nDepth must be 0; no WL histogram or ordering, the
     * receptive-field cap orders by hop distance only, :436-450), channels halve from level to level (C_l = max(1, C_{l-1} / 2),
     * K_l = [18 C_{l-1}][C_l], :141-156), and EVERY level is read out (:572-588).  Parameters: H[C][F], (K_l, b_l) l = 1..L -- no
     * readout vector.  gf_smp_forward then only fills graph_feature = [nMol][gf_smp_feature_width] (the ConcatVectors row, :590;
     * targets / predict / loss must be NULL) and the reverse sweep starts from gf_smp_backward_features.  The fully-connected
     * head on top is gf_head_*.  SMP_beta_physics / _pairgraphs = the same with max_receptive_field = max_nVertices. */
    int physics;
} gf_smp_config;
gf_status gf_smp_create(gf_ctx *ctx, const gf_smp_config *cfg, gf_smp **out);
gf_status gf_smp_destroy(gf_smp *smp);
size_t    gf_smp_param_count(const gf_smp *smp);
/* Host pointers: nVertices[nMol]; adj = the molecules' V x V int adjacency matrices back to back (DenseGraph::adj);
 * feature = their V x nFeatures matrices back to back (DenseGraph::feature).  Blocking (uploads index tables).
 * Thread safety: DIFFERENT handles of one context may be prepared at the same time from different host threads (each calling
 * thread has its own pool of preparation workers), also while another thread enqueues forward / backward on a third handle:
 * the data-loader pattern of bench.py's end_to_end loop.  One handle is used by one thread at a time. */
gf_status gf_smp_prepare(gf_smp *smp, int nMol, const int *nVertices, const int *adj, const double *feature);
/* The use_coulomb constructors of SMP_omega (SMP_omega.h:71-113): the reduced adjacency of every receptive field is taken
 * from the molecules' V x V Coulomb matrices (DenseGraph::coulomb, back to back; :568-579) instead of 1 / adj.  The
 * bonds in `adj` still define hops, features and fields.  coulomb == NULL is gf_smp_prepare. */
gf_status gf_smp_prepare_coulomb(gf_smp *smp, int nMol, const int *nVertices, const int *adj, const double *feature,
                                 const double *coulomb);
/* Device pointers.  targets may be NULL (Predict / Feature); predict, loss [nMol] and graph_feature [nMol][C] are
 * optional outputs (graph_feature is what SMP_omega::Feature returns, :984-996). */
gf_status gf_smp_forward(gf_smp *smp, const float *params, const float *targets, float *predict, float *loss,
                         float *graph_feature);
gf_status gf_smp_backward(gf_smp *smp, const float *params, float *grads, int accumulate);
/* Data-parallel runs (the context has a communicator, gf_dist_init): 1 (default) = gf_smp_backward leaves the gradient
 * summed over all ranks, the all-reduce of each level's [K_l | b_l] segment overlapped with the rest of the reverse sweep
 * (accumulate must be 0 then); 0 = local gradients only (the caller reduces them).  No effect without a communicator. */
gf_status gf_smp_set_grad_allreduce(gf_smp *smp, int on);
/* Physics towers: columns of a feature row (sum of the levels' channel counts), and the reverse sweep from the gradient of
 * the feature rows, d_feature [nMol][width] (what the head's backward hands down: ConcatVectors::backward). */
size_t    gf_smp_feature_width(const gf_smp *smp);
gf_status gf_smp_backward_features(gf_smp *smp, const float *params, float *grads, const float *d_feature, int accumulate);
/* The fully-connected head of the `_physics` / `_pairgraphs` models for a batch of feature rows x [n][width[0]]:
 *   h_i = LeakyReLU(W_i h_{i-1}) for i = 1..nLayers (W_i = [width[i]][width[i-1]], MatVecMul + LeakyReLU), y = <h_nLayers, w>,
 *   loss = (y - t)^2 / 2      (SMP_omega_physics.h:229-238, :592-606: one hidden layer of nTotal / 2;
 *                              SMP_omega_pairgraphs.h:330-345: two, max(nTotal / 2, 10) and max(that / 2, 10)).
 * params / dparams: W_1, ..., W_nLayers, w back to back (the models' registration order).  `work` holds the activations between
 * forward and backward: gf_head_work_floats floats.  backward: dx = gradient w.r.t. x, dparams += the weight gradients. */
size_t    gf_head_param_count(int nLayers, const int *width);
size_t    gf_head_work_floats(int nLayers, const int *width, int n);
gf_status gf_head_forward_f32(gf_ctx *ctx, int nLayers, const int *width, const float *x, int n, const float *params,
                              const float *targets, float *predict, float *loss, float *work);
gf_status gf_head_backward_f32(gf_ctx *ctx, int nLayers, const int *width, const float *x, int n, const float *params, float *work,
                               float *dx, float *dparams);
/* 1 (default): fused level kernels (no promoted stack, no 18-slice contraction output in HBM) where the shape allows;
 * 0: the op-by-op pipeline.  Same results within fp32 rounding; kept switchable for parity tests. */
gf_status gf_smp_set_fused(gf_smp *smp, int on);
/* The block products of one fused level at 64 channels as stand-alone operators on caller-supplied device matrices: the same
 * kernels gf_smp_forward / gf_smp_backward launch on a level's rows (the regrouped K-projection MatMul of GraphFlow/SMP_omega.h:654-657,
 * MatMul.h:48-82).  rows x 64-column blocks, row-major:  T = [S_ab|S_bc|T6|T10] (256 columns),  O / dO = [O_loc | U] (128),
 * rowscale [rows][2] = (tot, tr) of the row's node, trow [rows] = the row holding the transposed position (any permutation of the
 * rows), Wst [8][64][64] = stacked weight blocks W0..W7:
 *   forward  (backward == 0):  O_loc = tot (S_ab W0 + S_bc W1) + tr S_ab W2 + T6 W3 + T10 W4,   U = S_ab W5 + S_bc W6 + S_ab[trow] W7
 *   backward (backward != 0):  dT from dO, the transposed products (dS_ab = tot L W0^T + tr L W2^T + dU W5^T + dU[trow] W7^T, ...)
 *   wgrad:  dWst[p] = sum over rows of (T block of p)^T (dO block of p, with the factor of p)
 * On the f16 matrix pipe with two-half fp32 operands by default, on the fp32 pipe with GF_SMP_SPLIT=0 (read per call). */
gf_status gf_smp_level_products_f32(gf_ctx *ctx, int backward, int rows, const float *A, const float *rowscale, const float *Wst,
                                    const int *trow, float *Out);
gf_status gf_smp_level_wgrad_f32(gf_ctx *ctx, int rows, const float *T, const float *dO, const float *rowscale, const int *trow,
                                 float *dWst);
/* Device memory of the handle's buffer pool: bytes held by the current batch, and bytes the pool keeps in total (idle
 * blocks included).  Sizing aid for batch selection (GraphFlow has no counterpart: its tensors live in host `new[]`). */
gf_status gf_smp_device_bytes(const gf_smp *smp, size_t *in_use, size_t *reserved);
/* Host-pointer mode of the driver (what graphflow_amd/host/SMP_omega_hip.h uses): the handle owns the model -- a device
 * parameter buffer and its gradient.  Wherever the entry points of this section take `params` / `grads`, NULL selects the
 * handle-owned buffers.  gf_smp_forward_host runs gf_smp_forward on them and returns per-molecule results as doubles
 * (targets NULL: predict / Feature only; loss is 0.5 (y - t)^2 per molecule, SquaredLoss.h:45-53).  Blocking. */
gf_status gf_smp_parameters_upload(gf_smp *smp, const float *host_params);
gf_status gf_smp_parameters_download(gf_smp *smp, float *host_params, float *host_grads);
gf_status gf_smp_forward_host(gf_smp *smp, const double *targets, double *predict, double *loss, double *graph_feature);
/* Optimiser step of SMP_omega::BatchLearn (GraphFlow/SMP_omega.h:820-821 -> Adam::Learn(alpha, nBatch), Adam.h:106-133,
 * including its per-element advance of the bias-correction powers): params -= ..., from grads = the batch SUM written by
 * gf_smp_backward (all-reduced over ranks first in data-parallel runs; nBatch is then the GLOBAL batch).  The moment
 * buffers live in the handle and survive gf_smp_prepare.  Device pointers. */
gf_status gf_smp_adam_step(gf_smp *smp, float *params, const float *grads, double learning_rate, int nBatch);
gf_status gf_smp_adam_reset(gf_smp *smp);   /* zeroes the moment buffers (Adam and Momentum) and Adam's element counter */
/* Momentum::Learn(learning_rate, nBatch) (GraphFlow/Momentum.h:64-71), the optimiser of SMP_2D_ver6-8
 * (SMP_2D_ver6.h:204, :593): m = gamma m + learning_rate * grads / nBatch, params -= m. */
gf_status gf_smp_momentum_step(gf_smp *smp, float *params, const float *grads, double learning_rate, int nBatch, double gamma);
/* SMP_omega::weights_initialization (SMP_omega.h:334-338; GraphFlow.h:1297-1306) into a HOST buffer of
 * gf_smp_param_count floats, drawing from rand() in the reference's order: same srand() -> same initial weights. */
gf_status gf_smp_uniform_init_host(const gf_smp_config *cfg, float *params);
/* Text checkpoints interchangeable with SMP_omega::save_model / load_model (GraphFlow/SMP_omega.h:1033-1055):
 * whitespace-separated values in the flat parameter order above.  `params` is a device pointer.  Blocking. */
gf_status gf_smp_save_model(const gf_smp *smp, const float *params, const char *path);
gf_status gf_smp_load_model(gf_smp *smp, float *params, const char *path);
gf_status gf_smp_prepare_molecule_host(const gf_smp_config *cfg, int V, const int *adj, const double *feature,
                                       int *phi_out, double *wl_out);  /* host only; phi_out [L+1][V][cap+1], slot 0 = size */
int       gf_smp_receptive_field(const gf_smp *smp, int mol, int level, int v, int *out, int capacity);
/* Introspection for parity checks against the reference's per-level state: f_l[v] of molecule `mol` as [s][s][C] floats
 * (level[l]->f[v]->value after forward(), SMP_omega.h:667-669) and its reduced adjacency [s][s] (level[l]->adj[v], :556-581),
 * copied to HOST buffers.  Return the element count, or -1 (bad argument, capacity too small, nothing forwarded).  Blocking. */
long long gf_smp_read_activation(gf_smp *smp, int mol, int level, int v, float *host_out, size_t capacity);
long long gf_smp_read_reduced_adjacency(gf_smp *smp, int mol, int level, int v, float *host_out, size_t capacity);
gf_status gf_smp_level_sizes(const gf_smp *smp, int level, long long *nodes, long long *rows, long long *ppos);
/* sum of the receptive-field sizes s over the level's nodes: the (node, neighbour) pairs = promoted tensors of the level
 * (SMP_omega.h:641-647), and the rows of the per-(node, x) tables of the fused level; -1 on a bad argument */
long long gf_smp_level_pairs(const gf_smp *smp, int level);
/* Rows (a, b) of the level whose promoted slab row is not structurally zero -- vertex b lies inside the receptive field of a's
 * source (the selection matrices of SMP_omega.h:461-474 have a 1 in that column) -- out of gf_smp_level_sizes' `rows`.  The fused
 * level neither writes, reads nor back-propagates the other rows' S_ab / T6 table blocks; the bench prices its kernels with it. */
long long gf_smp_level_present_rows(const gf_smp *smp, int level);
/* ... and the rows (b, c) that SOME source covers (both vertices inside one source's field): the others' S_bc / T10 blocks are
 * structural zeros as well.  Equal to `rows` where the handle keeps no row flags (the flags exist for the channel counts the
 * row-panel kernels serve: models computed at 32 or 64 channels, i.e. every nChanels <= 64 through gf_smp_create's padding).
 * Both calls are introspection for tests and the bench: the first call after a prepare blocks on a device-to-host copy (the
 * result is then cached until the next prepare), and neither may run concurrently with gf_smp_prepare / gf_smp_destroy on the
 * SAME handle (they read the tables prepare replaces).  Other handles of the context are unaffected. */
long long gf_smp_level_covered_rows(const gf_smp *smp, int level);

/* RisiContraction_18_dropout inside a physics tower (SMP_sigma_pairgraphs.h:248-265, :632-651): masks[(l-1) * nVertices + gv] =
 * kept-slice bits of the contraction of global vertex gv (molecules back to back) at level l, drawn by the caller in the
 * reference's order (gf_smp_model_forward does); scale 1 in train mode, nKept / 18 with every bit set in test mode.  NULL: off. */
gf_status gf_smp_dropout_masks(gf_smp *smp, const unsigned *masks, float scale);
/* Adam::Learn(learning_rate, nBatch) (GraphFlow/Adam.h:106-133) on any flat device buffer with caller-owned moment buffers;
 * element i uses the bias-correction powers beta^(elements_before + i + 1), as the reference's per-element advance gives. */
gf_status gf_adam_step_f32(gf_ctx *ctx, float *params, const float *grads, float *m, float *v, size_t n, double learning_rate,
                           int nBatch, unsigned long long elements_before);

/* ---- the `_physics` / `_pairgraphs` models as one handle ------------------------------------------------------
 * nTowers 1: SMP_omega_physics (GraphFlow/SMP_omega_physics.h:29-606) -- SMP_beta_physics with max_receptive_field =
 *            max_nVertices; head = one hidden layer of nTotal / 2 units (:229-238).
 * nTowers 2: SMP_omega_pairgraphs / SMP_beta_pairgraphs (GraphFlow/SMP_omega_pairgraphs.h:81-730): a graph and its partner (e.g.
 *            its line graph) through two towers with their own weights, level features concatenated level by level (:699-704),
 *            head = two hidden layers (:330-345).  nKept > 0: SMP_sigma_pairgraphs (RisiContraction_18_dropout keeping nKept
 *            slices; masks drawn with rand() in the reference's order in train mode, all slices x nKept / 18 in test mode).
 * params / grads: flat device buffers in the class's registration order -- physics H, (K_l, b_l)..., W1, W2 (:254-262);
 * pairgraphs H_1, H_2, (K1_l, b1_l, K2_l, b2_l)..., W1, W2, W3 (SMP_omega_pairgraphs.h:361-375).  grads = the batch SUM.
 * prepare: host pointers as gf_smp_prepare, one set per tower (the second set NULL for nTowers 1).                             */
typedef struct gf_smp_model gf_smp_model;
typedef struct {
    int nTowers, nLevels, nChanels, max_receptive_field;
    int nFeatures[2];
    int nKept;   /* 0: RisiContraction_18; 1..18: RisiContraction_18_dropout */
} gf_smp_model_config;
gf_status gf_smp_model_create(gf_ctx *ctx, const gf_smp_model_config *cfg, gf_smp_model **out);
gf_status gf_smp_model_destroy(gf_smp_model *model);
size_t    gf_smp_model_param_count(const gf_smp_model *model);
gf_status gf_smp_model_set_mode(gf_smp_model *model, int train);   /* SMP_sigma_pairgraphs::setMode (:139-149); default train */
gf_status gf_smp_model_prepare(gf_smp_model *model, int nMol, const int *nVertices1, const int *adj1, const double *feature1,
                               const int *nVertices2, const int *adj2, const double *feature2);
gf_status gf_smp_model_forward(gf_smp_model *model, const float *params, const float *targets, float *predict, float *loss);
gf_status gf_smp_model_backward(gf_smp_model *model, const float *params, float *grads, int accumulate);
gf_status gf_smp_model_uniform_init_host(const gf_smp_model *model, float *host_params);
/* host-pointer mode (graphflow_amd/host/SMP_physics_hip.h): the handle owns parameters, gradients and Adam moments; NULL
 * params / grads in the two calls above select them */
gf_status gf_smp_model_parameters_upload(gf_smp_model *model, const float *host_params);
gf_status gf_smp_model_parameters_download(gf_smp_model *model, float *host_params, float *host_grads);
gf_status gf_smp_model_forward_host(gf_smp_model *model, const double *targets, double *predict, double *loss);
gf_status gf_smp_model_adam_step(gf_smp_model *model, double learning_rate, int nBatch);

#ifdef __cplusplus
}
#endif
#endif /* GF_HIP_H_INCLUDED */
