/*
 * pvlm.h — C ABI of the MI355X-native association + residual/Jacobian engine (libpvlm.so).
 *
 * This is the drop-in boundary for ONE hot path of 3dv-casia/PanoVLM: the ICP-style inner loop
 * (LiDAR<->LiDAR / camera<->LiDAR correspondence search + residual/Jacobian evaluation).  Every
 * entry point names the reference interface it replaces (paths relative to the PanoVLM tree).
 * Plain pointers and sizes only; no exceptions cross the boundary; every function returns a
 * pvlm_status (0 = OK, negative = error, message via pvlm_last_error).
 *
 * Threading: one pvlm_ctx per GPU; a ctx is NOT thread-safe, different ctxs are independent.  A call may use a few short-lived host
 * threads of its own for host-side passes over its inputs (pvlm_scan_upload_batch: bounding boxes and the staging copy of a large batch);
 * they are joined before the call returns.
 * Memory: buffers passed in are caller-owned and only read during the call unless stated.
 * Host pointers unless a parameter is prefixed d_ (device pointer, same GPU as the ctx).
 */
#ifndef PVLM_H_
#define PVLM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvlm_ctx pvlm_ctx;
typedef struct pvlm_resset pvlm_resset;   /* a device-resident set of residual blocks of one functor */
typedef struct pvlm_neq pvlm_neq;         /* block-sparse normal equations (per-pose 6x6 blocks)     */
typedef struct pvlm_scan pvlm_scan;       /* device-resident feature clouds of one LiDAR scan        */

typedef enum {
  PVLM_OK = 0,
  PVLM_ERR_ARG = -1,      /* bad argument (NULL, negative size, unknown enum)             */
  PVLM_ERR_HIP = -2,      /* HIP runtime error (no device, launch failure, ...)           */
  PVLM_ERR_NOMEM = -3,    /* host or device allocation failed                             */
  PVLM_ERR_STATE = -4,    /* call order violated (e.g. evaluate before pvlm_set_poses)    */
  PVLM_ERR_CAPACITY = -5, /* caller-provided output buffer too small                      */
  PVLM_ERR_REFUSED = -6   /* the input is valid for the reference but outside what this entry point handles (pvlm_ring_extract_batch: a non-finite
                             coordinate, more undecided segmentation edges than its tables hold): the caller takes its own path for it; nothing else failed */
} pvlm_status;

/* Residual functors — base/CostFunction.h.  All are 1 residual x 4 parameter blocks of 3
 * (angleAxis_rw, t_rw, angleAxis_nw, t_nw), i.e. AutoDiffCostFunction<F,1,3,3,3,3>. */
typedef enum {
  PVLM_POINT2PLANE_METER = 0,   /* base/CostFunction.h:567-619 ; row = [P_n(3) plane(4)]            stride 7  */
  PVLM_POINT2PLANE_ANGLE = 1,   /* base/CostFunction.h:630-729 ; row = [P_n(3) plane(4)]            stride 7  */
  PVLM_POINT2LINE_METER = 2,    /* base/CostFunction.h:769-829 ; row = [P_n(3) A(3) B(3)]           stride 9  */
  PVLM_POINT2LINE_ANGLE = 3,    /* base/CostFunction.h:836-934 ; row = [P_n(3) A(3) B(3)]           stride 9  */
  PVLM_PLANE2PLANE_GLOBAL = 4,  /* base/CostFunction.h:350-425 ; row = [plane_ref(3) a(3) b(3) w]   stride 10 */
  PVLM_PLANE_IOU = 5            /* base/CostFunction.h:433-507 ; row = [plane(4) mid_n(3) mid_r(3) angle w] stride 12 */
} pvlm_functor;

#define PVLM_FLAG_NORMALIZE_DISTANCE 1u /* `normalize_distance` of the *_Angle functors (Config.h:117) */

typedef enum {
  PVLM_LOSS_NONE = 0,  /* loss_function == nullptr (util/Optimization.cpp:304,417)         */
  PVLM_LOSS_HUBER = 1  /* ceres::HuberLoss(a)       (util/Optimization.cpp:513-517)        */
} pvlm_loss;

/* ---- context ------------------------------------------------------------------------------ */
pvlm_status pvlm_create(int device, pvlm_ctx** ctx);
pvlm_status pvlm_destroy(pvlm_ctx* ctx);
const char* pvlm_last_error(const pvlm_ctx* ctx);   /* valid until the next call on ctx */
const char* pvlm_version(void);
/* Launch on an externally owned hipStream_t (e.g. the caller's framework stream).  The handle is
 * used as given: NULL is HIP's default (null) stream, which is what a framework hands out for its
 * default stream.  pvlm_use_own_stream returns to the context's private non-blocking stream (the
 * state after pvlm_create).  Kernels of one ctx are always issued on exactly one stream. */
pvlm_status pvlm_set_stream(pvlm_ctx* ctx, void* hip_stream);
pvlm_status pvlm_use_own_stream(pvlm_ctx* ctx);
pvlm_status pvlm_synchronize(pvlm_ctx* ctx);
/* HIP-event timing on the ctx stream (bench.py measures kernel time with these). */
pvlm_status pvlm_timer_start(pvlm_ctx* ctx);
pvlm_status pvlm_timer_stop(pvlm_ctx* ctx, float* elapsed_ms);   /* records, synchronises, returns ms */
pvlm_status pvlm_device_info(pvlm_ctx* ctx, int* cu_count, int64_t* hbm_bytes, char* name, int name_cap);
/* Device memory.  Every device buffer of a context comes from a caching sub-allocator owned by the context (hipMalloc
 * maps pages at 40-70 ms per GB on MI355X; the reference's own loop re-associates every outer iteration,
 * lidar_mapping/LidarOdometry.cpp:38-110, so scratch and outputs of similar size are needed again and again).
 * pvlm_reserve makes sure ONE free range of `bytes` exists (a production host reserves its working set once, next to
 * uploading the scans); pvlm_trim synchronises and returns every unused slab to the driver; pvlm_mem_info reports bytes
 * held from the driver, bytes handed out, the high-water mark and the number of hipMalloc calls so far (a steady state
 * adds none).  Any pointer may be NULL.  PVLM_NO_POOL=1 in the environment bypasses the pool (debugging). */
pvlm_status pvlm_reserve(pvlm_ctx* ctx, int64_t bytes);
/* The pinned staging window of pvlm_scan_upload[_batch] (64 MB by default, PVLM_UPLOAD_STAGE_MB) is allocated by the first upload that needs it:
 * hipHostMalloc + first touch of 64 MB cost 25 ms.  A host that knows it will upload reserves the window at start-up.  The same call allocates the
 * 32 MB pinned arena every other small host <-> device copy of the context goes through (8 ms inside whichever call needs it first otherwise).      */
pvlm_status pvlm_reserve_staging(pvlm_ctx* ctx, int64_t bytes);
/* Loads the code objects of the library's kernels now (one empty launch per translation unit + a synchronisation).  HIP loads a code object at the first launch of
 * one of its kernels — 5-20 ms each, ~60 ms for all of them — which otherwise lands inside the first call that needs it (the first EstimatePose of a process).  Like
 * pvlm_reserve / pvlm_reserve_staging: once per context, after pvlm_create; never needed for correctness. */
pvlm_status pvlm_preload(pvlm_ctx* ctx);
pvlm_status pvlm_trim(pvlm_ctx* ctx);
pvlm_status pvlm_mem_info(const pvlm_ctx* ctx, int64_t* reserved, int64_t* in_use, int64_t* peak, int64_t* device_allocs);
/* HIP graph of a step: the calls issued between _begin and _end on this context (pvlm_set_poses_dev,
 * pvlm_neq_accumulate_dev / pvlm_eval_pair_blocks_dev / pvlm_eval_dev, pvlm_allreduce_sum_f64 — the `_dev` forms, which
 * neither allocate nor synchronise once they have run once with the same objects) are captured instead of executed;
 * pvlm_graph_launch replays them as one submission on the context's stream with the CURRENT contents of the device
 * buffers they were given (pose arrays, packed buffer).  One LM step is five small kernels around the fused one: replayed
 * as a graph they cost one launch.  A call that would have to allocate or synchronise inside a capture fails with
 * PVLM_ERR_STATE (run the step once eagerly first).  Needs a real stream (the context's own, or one given to
 * pvlm_set_stream), not the legacy NULL stream. */
typedef struct pvlm_graph pvlm_graph;
pvlm_status pvlm_graph_begin(pvlm_ctx* ctx);
pvlm_status pvlm_graph_end(pvlm_ctx* ctx, pvlm_graph** out);
pvlm_status pvlm_graph_launch(pvlm_ctx* ctx, pvlm_graph* graph);
pvlm_status pvlm_graph_destroy(pvlm_ctx* ctx, pvlm_graph* graph);
/* Per-kernel timing of the dominant kernels: when enabled, every launch of the fused
 * residual/Jacobian kernel (which=0), the materialise kernel (which=1) and the k-NN + plane-fit
 * association kernel (which=2) and of the batched camera-LiDAR vote kernel (which=3) is bracketed by HIP events on the ctx stream.  pvlm_profile_read
 * synchronises and returns the accumulated milliseconds and launch count since the last
 * pvlm_profile_enable(ctx, 1). */
pvlm_status pvlm_profile_enable(pvlm_ctx* ctx, int on);
pvlm_status pvlm_profile_read(pvlm_ctx* ctx, int which, double* total_ms, int64_t* launches);

/* ---- parameter blocks ---------------------------------------------------------------------- *
 * The pose table the residual blocks index: angleAxis_lw_list / t_lw_list of
 * lidar_mapping/LidarOdometry.cpp:23-33 (and angleAxis_cw_list / t_cw_list of
 * joint_optimization/CameraLidarOptimizer.cpp:387-548 appended behind them).  n x 3 each. */
pvlm_status pvlm_set_poses(pvlm_ctx* ctx, int n, const double* angle_axis, const double* translation);
pvlm_status pvlm_set_poses_dev(pvlm_ctx* ctx, int n, const double* d_angle_axis, const double* d_translation);

/* ---- residual sets ---------------------------------------------------------------------------- *
 * Replaces the per-correspondence `X::Create(...)` + `problem.AddResidualBlock(cost, loss, aa_r,
 * t_r, aa_n, t_n)` loops of util/Optimization.cpp:538-557 (point-to-plane), :410-434
 * (line-to-line), :583-602 (camera-LiDAR).  Residual blocks are grouped in `n_pairs` segments;
 * segment p = rows [pair_offsets[p], pair_offsets[p+1]) all share the parameter blocks
 * (pair_ref[p], pair_nei[p]) — exactly the (i, n_idx) loop nest of the reference.  `rows` is
 * n x stride doubles with the per-functor row layout given at pvlm_functor.  `weight` is the
 * scalar weight of kinds 0..3 (ignored by the *_Angle functors, as in the reference). */
pvlm_status pvlm_resset_upload(pvlm_ctx* ctx, pvlm_functor kind, unsigned flags, double weight, int64_t n,
                               int n_pairs, const int64_t* pair_offsets, const int* pair_ref, const int* pair_nei,
                               const double* rows, int stride, pvlm_resset** out);
pvlm_status pvlm_resset_destroy(pvlm_ctx* ctx, pvlm_resset* rs);
pvlm_status pvlm_resset_info(const pvlm_resset* rs, int64_t* n, int* n_pairs, int* kind, unsigned* flags);
/* Copies the segment table / rows back (rows in the upload layout); any pointer may be NULL. */
pvlm_status pvlm_resset_download(pvlm_ctx* ctx, const pvlm_resset* rs, int64_t* pair_offsets, int* pair_ref,
                                 int* pair_nei, double* rows);
/* Replaces the pose ids of the set's segments (pair_ref / pair_nei, n_pairs each).  A set produced by the association carries the ids
 * of the scans it was built from (lidars[i].id, util/Optimization.cpp:527-528); a problem that mixes such sets with uploaded ones
 * (camera poses + LiDAR poses, joint_optimization/CameraLidarOptimizer.cpp:394-417) renumbers them once into ONE pose table, so that a
 * single pvlm_set_poses and a single packed buffer serve every set.  Queued on the context stream; no synchronisation. */
pvlm_status pvlm_resset_set_pose_ids(pvlm_ctx* ctx, pvlm_resset* rs, const int* pair_ref, const int* pair_nei);

/* ceres::CostFunction::Evaluate for every block of the set at the poses of pvlm_set_poses:
 * residuals[n] and jacobians[n x 12] = [d/daa_r | d/dt_r | d/daa_n | d/dt_n] (row-major per block,
 * the layout AutoDiffCostFunction<F,1,3,3,3,3> hands to Ceres).  jacobians may be NULL (cost-only
 * evaluation).  Values are the RAW residuals (no loss applied), as Evaluate returns them. */
pvlm_status pvlm_eval(pvlm_ctx* ctx, const pvlm_resset* rs, double* residuals, double* jacobians);
/* Same, outputs left in device memory (async on the ctx stream; no host copies). */
pvlm_status pvlm_eval_dev(pvlm_ctx* ctx, const pvlm_resset* rs, double* d_residuals, double* d_jacobians);

/* ---- host-visible evaluation: the Ceres-feeding boundary ------------------------------------------------------- *
 * ceres::CostFunction::Evaluate reads host memory, so the last hop of this mode is a PCIe link (30 GB/s measured on the
 * MI355X box, against 5 TB/s for the kernel): the API is built around that hop.
 *   pvlm_host_alloc / _free     page-locked host memory (the copies run at link rate and asynchronously only into it;
 *                               pageable destinations work but serialise).
 *   pvlm_eval_host_async        residuals[n] and jacobians[n x 12] (NULL = cost only) as pvlm_eval, delivered
 *                               asynchronously on the context's stream: 104 B per block.
 *   pvlm_eval_wrench_host_async the same information in 56 B per block: wrench_rows[n x 7] = [r | c(3) | g(3)] plus the
 *                               per-pair tables pair_tables[n_pairs x PVLM_PAIR_TABLE] = [R_rn(9) | t_rn(3) | t_rw(3) |
 *                               J_l(aa_r)(9) | M_n(9)], from which the host forms the row on demand (36 multiply-adds):
 *                                 d r/d aa_r = c^T J_l      d r/d t_r = g^T      d r/d aa_n = c^T M_n      d r/d t_n = -g^T R_rn
 *                               (row-major 3x3 blocks; integration/pvlm_ceres.hpp does it inside Evaluate).
 *   pvlm_eval_force_host_async  kinds 0..3 (the point functors), 32 B per block: force_rows[n x 4] = [r | g(3)].  The moment is
 *                               c = (R_rn P_n + t_rn - t_rw) x g with P_n = the first three entries of the block's row (the host holds them:
 *                               pvlm_resset_download) and R_rn, t_rn, t_rw from the pair table — 15 more multiply-adds in Evaluate for
 *                               43 % fewer bytes over the link, which is the roof of this mode.
 * All three go through a bounded device staging buffer (PVLM_STAGE_ROWS rows, 32 M by default) slice by slice; the results
 * are complete after pvlm_synchronize.  Residuals are RAW (no loss), in the compact order of pvlm_resset_download. */
#define PVLM_PAIR_TABLE 33
pvlm_status pvlm_host_alloc(pvlm_ctx* ctx, int64_t bytes, void** out);
pvlm_status pvlm_host_free(pvlm_ctx* ctx, void* p);
pvlm_status pvlm_eval_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* residuals, double* jacobians_or_null);
pvlm_status pvlm_eval_wrench_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* wrench_rows, double* pair_tables);
pvlm_status pvlm_eval_force_host_async(pvlm_ctx* ctx, const pvlm_resset* rs, double* force_rows, double* pair_tables);

/* Fused evaluation: residual + Jacobian in registers, loss-corrected (Ceres corrector for
 * rho'' <= 0: scale r and J by sqrt(rho')) and contracted into per-pair normal-equation blocks
 *   out[p] = [ H_rr(36) | H_rn(36) | H_nn(36) | g_r(6) | g_n(6) | cost(1) ]   (121 doubles, row-major)
 * with H = J^T J, g = J^T r, cost = sum 1/2 rho(r^2) over the segment.  Deterministic. */
#define PVLM_PAIR_BLOCK 121
pvlm_status pvlm_eval_pair_blocks(pvlm_ctx* ctx, const pvlm_resset* rs, pvlm_loss loss, double loss_a, double* out);
pvlm_status pvlm_eval_pair_blocks_dev(pvlm_ctx* ctx, const pvlm_resset* rs, pvlm_loss loss, double loss_a, double* d_out);

/* ---- block-sparse normal equations ------------------------------------------------------------ *
 * The Gauss-Newton system the outer trust-region solver needs (what Ceres assembles internally
 * from the blocks of ceres::Problem; LidarOdometry.cpp:36-80).  Packed layout (doubles):
 *   [ Hdiag n_poses x 36 | Hoff n_upairs x 36 | g n_poses x 6 | cost 1 ]
 * Hoff[u] is the 6x6 block d2/dx_i dx_j for the unordered pose pair (upair_i[u] < upair_j[u]).
 * One RCCL all-reduce(sum) of this buffer per LM iteration is the only multi-GPU exchange. */
pvlm_status pvlm_neq_create(pvlm_ctx* ctx, int n_poses, int n_upairs, const int* upair_i, const int* upair_j,
                            pvlm_neq** out);
pvlm_status pvlm_neq_destroy(pvlm_ctx* ctx, pvlm_neq* neq);
int64_t pvlm_neq_size(const pvlm_neq* neq); /* number of doubles in the packed buffer */
/* packed (+)= contribution of one residual set.  d_packed is a device buffer of pvlm_neq_size()
 * doubles (e.g. the tensor handed to the all-reduce); zero_first!=0 clears it before adding. */
pvlm_status pvlm_neq_accumulate_dev(pvlm_ctx* ctx, pvlm_neq* neq, const pvlm_resset* rs, pvlm_loss loss,
                                    double loss_a, int zero_first, double* d_packed);
pvlm_status pvlm_neq_accumulate(pvlm_ctx* ctx, pvlm_neq* neq, const pvlm_resset* rs, pvlm_loss loss, double loss_a,
                                int zero_first, double* packed_host_inout);
/* The same contribution, queued: the set is linearised into the structure's OWN device buffer (kept with the handle) and its copy
 * into `packed` (host, pvlm_neq_size() doubles) is queued behind it — nothing is allocated after the first call and nothing
 * waits.  An LM driver calls it once per residual set of its problem (pvlm_set_poses in that set's pose numbering first) and then
 * pvlm_synchronize ONCE: every `packed` is complete when that returns.  One submission and one synchronisation per
 * linearisation, whatever the number of residual sets (lidar_mapping/LidarOdometry.cpp:36-80 evaluates point-to-plane and
 * line-to-line blocks in the same ceres::Solve). */
pvlm_status pvlm_neq_accumulate_async(pvlm_ctx* ctx, pvlm_neq* neq, const pvlm_resset* rs, pvlm_loss loss, double loss_a, double* packed);
/* A whole problem in ONE packed buffer: the n residual sets are linearised one after the other and SUMMED on the device into the
 * buffer of neq[0] (all n structures must have been created with the same n_poses and pair list, and the sets must share one pose
 * numbering — pvlm_resset_set_pose_ids below); one copy into `packed` (host, pvlm_neq_size(neq[0]) doubles) is queued behind them and
 * is complete after the next pvlm_synchronize.  Nothing is allocated after the first call, nothing waits. */
pvlm_status pvlm_neq_accumulate_sets(pvlm_ctx* ctx, int n, pvlm_neq* const* neq, const pvlm_resset* const* rs, const pvlm_loss* loss,
                                     const double* loss_a, double* packed);

/* ---- panoramic reprojection blocks with point elimination ---------------------------------------- *
 * PanoramaReprojResidual_1Angle (base/CostFunction.h:218-247): r = w * angle(R(aa_cw) X + t_cw, bearing),
 * three parameter blocks (aa_cw, t_cw, point_3d), added per track observation by AddCameraResidual
 * (util/Optimization.cpp:172-222) inside CameraLidarOptimizer::Optimize (CameraLidarOptimizer.cpp:431-432),
 * solved upstream by Ceres *_SCHUR (util/Optimization.cpp:608-634).  Here the 3-D points live on the GPU
 * and are eliminated there: the host only sees the reduced 6x6 camera blocks.
 *   observations are grouped by point: point p owns [point_offsets[p], point_offsets[p+1]);
 *   cam_ids index the pose table of pvlm_set_poses (angleAxis_cw, t_cw);
 *   bearings: n_obs x 3, any norm (normalised like the functor's constructor); points: n_points x 3.
 * Packed output of pvlm_ba_reduce (pvlm_ba_packed_size doubles), F = n_cams, U = n_upairs (co-visible pairs,
 * ui < uj, from pvlm_ba_structure):
 *   [ S_diag F x 36 | S_off U x 36 (rows ui, cols uj) | g F x 6 | cost | U_diag F x 6 | g_cam F x 6 | gmax_points ]
 * S, g = Schur complement of the point blocks damped as the LM driver damps every column:
 *   V* = V + diag(clamp(V_kk s_k^2, min_diag, max_diag) / (radius s_k^2)),  s_k = 1/(1+sqrt(V_kk)) at the
 *   first call (init_scale = 1; Ceres' Jacobi scaling), cameras are left undamped/unscaled for the caller;
 * U_diag / g_cam = diagonal and gradient of the camera blocks BEFORE elimination (for the caller's scaling,
 * damping and gradient test);
 * gmax_points = max |gradient| over the point blocks; cost = sum rho(r^2)/2.
 * pvlm_ba_step back-substitutes the points for the camera steps dcam (F x 6, zeros for constant blocks)
 * into the candidate points and returns out3 = [model cost decrease of these blocks, |dX|^2, |X|^2];
 * pvlm_ba_cost evaluates the cost at the current (candidate = 0) or candidate points with the poses of the
 * last pvlm_set_poses; pvlm_ba_accept makes the candidate current.  pvlm_ba_eval materialises r and the
 * 1x9 Jacobian rows [d/daa_cw | d/dt_cw | d/dX] (Ceres-feeding / parity mode).  pvlm_ba_set_constant marks
 * points (mask[p] != 0) as constant parameter blocks (Problem::SetParameterBlockConstant, the
 * refine_structure = false case of CameraLidarOptimizer.cpp:462-466): they are neither eliminated nor moved;
 * mask = NULL frees all. */
typedef struct pvlm_baset pvlm_baset;
pvlm_status pvlm_ba_create(pvlm_ctx* ctx, int n_points, int64_t n_obs, const int64_t* point_offsets, const int* cam_ids,
                           const double* bearings, const double* points, double weight, pvlm_baset** out);
pvlm_status pvlm_ba_destroy(pvlm_ctx* ctx, pvlm_baset* set);
pvlm_status pvlm_ba_structure(const pvlm_baset* set, int* n_points, int64_t* n_obs, int* n_cams, int* n_upairs, int* ui, int* uj);
int64_t pvlm_ba_packed_size(const pvlm_baset* set);
pvlm_status pvlm_ba_get_points(pvlm_ctx* ctx, const pvlm_baset* set, int candidate, double* points);
pvlm_status pvlm_ba_set_points(pvlm_ctx* ctx, pvlm_baset* set, const double* points);
pvlm_status pvlm_ba_set_constant(pvlm_ctx* ctx, pvlm_baset* set, const unsigned char* mask_or_null);
pvlm_status pvlm_ba_eval(pvlm_ctx* ctx, const pvlm_baset* set, double* r, double* J_or_null);
pvlm_status pvlm_ba_reduce(pvlm_ctx* ctx, pvlm_baset* set, pvlm_loss loss, double loss_a, int init_scale, double radius,
                           double min_diag, double max_diag, double* packed);
pvlm_status pvlm_ba_step(pvlm_ctx* ctx, pvlm_baset* set, pvlm_loss loss, double loss_a, const double* dcam, double* out3);
pvlm_status pvlm_ba_cost(pvlm_ctx* ctx, const pvlm_baset* set, pvlm_loss loss, double loss_a, int candidate, double* cost);
pvlm_status pvlm_ba_accept(pvlm_ctx* ctx, pvlm_baset* set);

/* ---- dense SPD solve for an LM driver (not hot path) ------------------------------------------------------------ *
 * Blocked fp64 Cholesky + triangular solves on the GPU (hand-written: 32-wide block columns, 64 x 64 register-tiled
 * trailing updates).  Solves A X = B for a symmetric positive definite n x n matrix (dense, host, full symmetric storage)
 * and B = n x nrhs (column-major, host, overwritten by X).  *info = 0 on success, k > 0 when the leading minor of order k
 * is not positive definite (X is then undefined).  Upstream this step is inside ceres::Solve (SPARSE_SCHUR,
 * util/Optimization.cpp:608-666); the host mirror's stand-in LM driver uses it once the reduced pose system is too large
 * for a host factorisation (a Room-sized joint problem has 5442 unknowns). */
pvlm_status pvlm_spd_solve(pvlm_ctx* ctx, int n, int nrhs, const double* A, double* B, int* info);
/* Block-sparse input: M = D (sum of the 6x6 blocks) D + diag(diag_add), D = diag(scale), assembled on the device, then
 * M x = rhs solved in place (rhs host, n doubles).  blocks: n_blocks x 36 row-major; row_idx / col_idx: n_blocks x 6
 * scalar indices of the block's rows / columns (-1 = constant parameter, dropped).  mirror[b] != 0 (block between two
 * different poses) also adds the transposed block to the other triangle; a block of a pose with itself is given in full
 * with mirror = 0.  Repeated blocks add up.  A block between two DIFFERENT poses must be given with mirror != 0 (or in both
 * triangles): from 1500 unknowns on the unknowns are reordered before the factorisation, which reads one triangle of the
 * REORDERED matrix — a block given in one triangle only may land in the other one.  *info = 0 on success; k > 0 = the leading
 * minor of order k of the matrix AS FACTORISED (reordered when pvlm_spd_last_plan reports a tile-sparse plan) is not positive
 * definite: take it as "not positive definite", not as the index of an unknown. */
pvlm_status pvlm_spd_solve_blocks(pvlm_ctx* ctx, int n, int n_blocks, const int* row_idx, const int* col_idx, const int* mirror, const double* blocks,
                                  const double* scale, const double* diag_add, double* rhs, int* info);
/* Makes the plan of a structure AHEAD of the solve that needs it: the host half (ordering, symbolic factorisation, schedule lists: ~10 ms for a Floor-sized
 * pose graph) starts on a thread of its own and the call returns at once; the lists are copied.  The next pvlm_spd_solve_blocks whose n / row_idx / col_idx /
 * mirror are EXACTLY these takes the plan (waiting for the thread if it is still at work); with any other structure the prefetch is dropped and the plan is made
 * inside the solve as without this call — a hint, never a change of result.  The host mirror's LM driver calls it on entry to a Solve, so that the plan is made
 * beside the upload of the residual sets and the first linearisation (upstream: inside ceres::Solve's preprocessing, util/Optimization.cpp:638-666).
 * pvlm_spd_plan_prefetch_hits: how many solves of this context took a prefetched plan. */
pvlm_status pvlm_spd_plan_prefetch(pvlm_ctx* ctx, int n, int n_blocks, const int* row_idx, const int* col_idx, const int* mirror);
pvlm_status pvlm_spd_plan_prefetch_hits(const pvlm_ctx* ctx, long long* hits);
/* How the last pvlm_spd_solve_blocks structure of this context is factorised: *tile_sparse = 1 when the dense kernels skip the
 * structurally zero 64-row tiles (ordering of the pose graph by minimum degree + elimination-tree postorder; the reference selects
 * SPARSE_SCHUR for 50 < lidars <= 2000, util/Optimization.cpp:641-658), *update_fraction = its tile updates / those of the dense
 * factorisation.  Systems under 1500 unknowns, or whose fraction exceeds 0.6, keep the natural order and the dense kernels. */
pvlm_status pvlm_spd_plan_info(const pvlm_ctx* ctx, int* tile_sparse, double* update_fraction);
/* The schedule of that factorisation.  From 1500 unknowns on the pose graph is ordered by NESTED DISSECTION (recursive bisection by breadth-first level sets; the
 * halves first, the separator last; every group aligned to a 64-row tile by identity rows) and the block columns are factorised LEVEL by level: the columns of a level
 * do not depend on each other — one launch factorises them all, one launch applies their trailing updates (a workgroup per target tile, sources added in list order:
 * bit-reproducible), the triangular solves run level by level too.  *levels = dependent steps (0: the column-by-column factorisation is in use — small systems,
 * PVLM_SPD_LEVELS=0, or a graph whose schedule is not shorter than two thirds of its block columns), *block_columns = block columns of the (padded) system,
 * *padded_rows = its rows.  The reference leaves this to Ceres' SPARSE_SCHUR (util/Optimization.cpp:638-666).  Any pointer may be NULL. */
pvlm_status pvlm_spd_plan_schedule(const pvlm_ctx* ctx, int* levels, int* block_columns, int* padded_rows);
/* The dense TAIL of that schedule.  The last group of the dissection (the top separator) is dense once everything below it is eliminated and every one of its block
 * columns is a level of its own; from eight such columns on they are factorised by ONE launch instead (a tile Cholesky in 64 x 64 tiles whose workgroups hand the
 * finished tiles to each other inside the launch; forward substitution inside it, backward substitution in a second launch).  *tail_block_columns = the block columns
 * done that way (0: none — PVLM_SPD_TAIL=0, no level schedule, or a short separator), *launched_levels = the levels that still run launch by launch.  With the
 * one-launch form of the whole factorisation (pvlm_spd_one_launch, the default) EVERY block column is done inside one launch: *tail_block_columns = all of them,
 * *launched_levels = 0.  Bit-reproducible like the levels; last-bit differences against PVLM_SPD_TAIL=0 (another order of the same sums).  Any pointer may be NULL. */
pvlm_status pvlm_spd_plan_tail(const pvlm_ctx* ctx, int* tail_block_columns, int* launched_levels);
/* ONE LAUNCH for the whole tile-sparse factorisation (the default with a level schedule): every 64 x 64 tile of the factor is a task (its sources = the earlier tile
 * columns that hold both tiles; the sources of a tile that exist two levels ahead of it are split off into chunk tasks that subtract them from the tile in memory as
 * soon as they exist), tasks are ordered by dependency depth, workgroups take them by a ticket and hand finished tiles to each other inside the launch (payload stored
 * write-through, one flag per task); the forward substitution rides along and the backward substitution is a second launch over the tile columns.  Floor system: 278
 * dependent launches -> 2, 6.9 -> 3.2 ms per solve.  pvlm_spd_plan_tail then reports every block column as done inside one launch (launched_levels 0).
 * The workgroups of such a launch WAIT for each other, which presumes that the launch gets the GPU's workgroup slots: processes that share one GPU should switch it off
 * (enable = 0: level launches + the dense tail; enable < 0: no change, query only).  A solve whose launch does not get through within 2 s is redone with the level
 * launches by itself, the context keeps them from then on, and *fallbacks (may be NULL) counts such solves.  Both forms are bit-reproducible; they differ from each
 * other in the last bits (another order of the same sums) — ranks that must agree bit for bit use the same form.
 * Test hook: enable = 2 + k makes task k of the one launch withhold its tile (what a launch that does not get through looks like): the next solve runs into the limit
 * and exercises the recovery; tests/test_linalg_gpu.py. */
pvlm_status pvlm_spd_one_launch(pvlm_ctx* ctx, int enable, long long* fallbacks);

/* ---- multi-GPU exchange (RCCL over xGMI) -------------------------------------------------------- *
 * The reference is single-process (OpenMP only); sharding scan pairs across GPUs introduces exactly one
 * exchange: all-reduce(sum) of the packed normal-equation buffer per LM iteration.
 * One process per GPU.  Rank 0 obtains a 128-byte id (pvlm_comm_unique_id), the host's launcher carries
 * it to the other ranks, every rank calls pvlm_comm_create (collective).  The all-reduce is enqueued on
 * the ctx stream, in place, on a device buffer of `count` doubles (e.g. the d_packed of
 * pvlm_neq_accumulate_dev).  RCCL is loaded lazily on the first pvlm_comm_* call. */
typedef struct pvlm_comm pvlm_comm;
pvlm_status pvlm_comm_unique_id(pvlm_ctx* ctx, unsigned char id_out[128]);
pvlm_status pvlm_comm_create(pvlm_ctx* ctx, int world_size, int rank, const unsigned char id[128], pvlm_comm** out);
pvlm_status pvlm_comm_destroy(pvlm_ctx* ctx, pvlm_comm* comm);
pvlm_status pvlm_allreduce_sum_f64(pvlm_ctx* ctx, pvlm_comm* comm, double* d_buf, int64_t count);
/* Same on a HOST buffer (staged through the context's device buffer, synchronous): for hosts whose LM driver keeps the
 * summed system in host memory, like the mirrored one (panovlm_amd/host: Exchange / MakeRcclExchange). */
pvlm_status pvlm_allreduce_sum_f64_host(pvlm_ctx* ctx, pvlm_comm* comm, double* buf, int64_t count);

/* ---- scans and LiDAR<->LiDAR association -------------------------------------------------------- *
 * Input contract = the public members of sensors/Velodyne.h:80-91 during association: feature
 * clouds in the WORLD frame as float32 (the float buffers pcl::transformPointCloud left behind,
 * sensors/Velodyne.cpp:1773-1808), class tag = PointXYZI::intensity, pose (R_wl, t_wl) double. */
typedef struct {
  int id;                     /* Velodyne::id — index into the pose table                         */
  const double* R_wl;         /* 9, row-major                                                     */
  const double* t_wl;         /* 3                                                                */
  int n_surf_flat;            const float* surf_flat_xyz;      const float* surf_flat_tag;       /* queries  (xyz interleaved) */
  int n_surf_less_flat;       const float* surf_less_flat_xyz; const float* surf_less_flat_tag;  /* targets                     */
  int n_corner;               const float* corner_xyz;         /* cornerLessSharp                 */
  const int* p2s_offsets;     const int* p2s_ids;              /* point_to_segment as CSR (n_corner+1) */
  int n_segments;             const int* segment_size;         /* edge_segmented[s].size()        */
  const double* segment_coeffs; /* 6 per segment, LiDAR-local (point, unit direction)             */
  const double* end_points;     /* 2x3 per segment, LiDAR-local                                   */
  const float* seg_points_xyz;  /* optional: the points of edge_segmented[0], [1], ... one after the other (sum of
                                   segment_size x 3 floats, WORLD frame like the clouds) — needed by pvlm_line2line_residuals */
  int point_stride_floats;      /* distance in floats between consecutive points of surf_flat_xyz / surf_less_flat_xyz / corner_xyz AND between
                                   consecutive entries of the two tag arrays.  0 (or 3 for the xyz arrays): packed, as documented above.  4: the arrays point
                                   INTO pcl::PointXYZI-style records {x, y, z, intensity} (xyz = &cloud[0].x, tag = &cloud[0].intensity): the upload gathers them
                                   itself and the caller flattens nothing.  seg_points_xyz is always packed.                                                  */
} pvlm_scan_desc;

pvlm_status pvlm_scan_upload(pvlm_ctx* ctx, const pvlm_scan_desc* desc, pvlm_scan** out);
/* The same for `n_scans` scans at once (descs[k] -> out[k]; all or nothing): one staging copy, one device allocation
 * shared by the batch (released when the last of its scans is destroyed) and one set of grid-build launches — what the
 * loop over lidars of AddLidarPointToPlaneResidual / AddLidarLineToLineResidual2 (util/Optimization.cpp:521, :345) needs
 * at every outer iteration of LidarOdometry::EstimatePose, when all 454 re-posed Room scans change at once
 * (lidar_mapping/LidarOdometry.cpp:150-170).  pvlm_scan_upload is the n_scans = 1 case. */
pvlm_status pvlm_scan_upload_batch(pvlm_ctx* ctx, int n_scans, const pvlm_scan_desc* descs, pvlm_scan** out);
pvlm_status pvlm_scan_destroy(pvlm_ctx* ctx, pvlm_scan* scan);

/* Velodyne::Transform2LidarWorld / Velodyne::Transform2Local (sensors/Velodyne.cpp:1773-1808, :1810-1848) on the RESIDENT clouds of n_scans
 * scans, as LidarOdometry::RefinePose calls them around every solve (lidar_mapping/LidarOdometry.cpp:17-21, :100-110): scan k's float clouds
 * (surfFlat, surfLessFlat, cornerLessSharp and the segments' points) are replaced IN PLACE by T_k applied the way
 * pcl::transformPointCloud(cloud, cloud, Eigen::Matrix4d) applies it — per coordinate float(((m0 x + m1 y) + m2 z) + m3), the float point promoted
 * to double — so the device's clouds equal the host's after any number of World <-> Local round trips, bit for bit, and nothing is uploaded again.
 * T_rowmajor12: n_scans x 12 doubles, the upper 3 x 4 of T_k row-major ([R | t]; T_wl = [R_wl | t_wl] for Transform2LidarWorld,
 * [R_wl^T | -R_wl^T t_wl] for Transform2Local).  rebuild_grids != 0: the voxel grids of the transformed clouds are rebuilt — bounding boxes
 * reduced on the device, the same plan and tables an upload of the same floats gets; rebuild_grids == 0 leaves them stale (a scan on its way
 * to the local frame is not searched): pvlm_knn / pvlm_assoc_point2plane on such a scan return PVLM_ERR_STATE until a call with
 * rebuild_grids != 0.  The pose of a scan (R_wl, t_wl of its descriptor) is NOT touched: pvlm_scan_set_pose.  A scan may be listed once.
 * On an error after the launch (a coordinate left the float range: PVLM_ERR_ARG) the clouds stay transformed and the grids stale: destroy
 * the scans. */
pvlm_status pvlm_scan_transform_batch(pvlm_ctx* ctx, int n_scans, pvlm_scan* const* scans, const double* T_rowmajor12, int rebuild_grids);
/* Velodyne::SetRotation / SetTranslation (sensors/Velodyne.h:213-233) for a resident scan: replaces the pose the association and the line
 * kernels read (R_wl 9 row-major, t_wl 3).  Host-side state only; the clouds are not moved. */
pvlm_status pvlm_scan_set_pose(pvlm_ctx* ctx, pvlm_scan* scan, const double* R_wl, const double* t_wl);
/* Tests / visualisation: the resident arrays of one cloud of a scan.  which: 0 surfFlat, 1 surfLessFlat, 2 cornerLessSharp, 3 the segments'
 * points.  pvlm_scan_cloud_info: the cloud's size and (clouds 1, 2) the parameters of its voxel grid; pvlm_scan_cloud_fetch: xyz (n x 3), and of
 * the grid cell_count / cell_start (table_size ints each), keys (table_size 64-bit words; hashed tables only) and the cell-sorted points
 * (n x 4 floats: x, y, z, original index as int bits; the order INSIDE a cell is not deterministic).  NULL = skip. */
typedef struct pvlm_grid_info {
  int n;                       /* points                                                                   */
  int has_grid, stale;         /* clouds 1, 2 with n > 0; stale: transformed without rebuild              */
  int dense, nx, ny, nz, xf;   /* dense table: (iz ny + iy) nx + ix, cells xf times finer along x          */
  int table_size;              /* dense: cells + 1; hashed: slots (power of two)                           */
  float cell, origin[3];
} pvlm_grid_info;
pvlm_status pvlm_scan_cloud_info(const pvlm_scan* scan, int which, pvlm_grid_info* info);
pvlm_status pvlm_scan_cloud_fetch(pvlm_ctx* ctx, const pvlm_scan* scan, int which, float* xyz, int* cell_count, int* cell_start,
                                  unsigned long long* keys, float* sorted_xyzi);

/* Exact k-nearest-neighbour search of `queries` (nq x 3 float, world frame) in the scan's
 * surfLessFlat cloud (which=0) or cornerLessSharp cloud (which=1): what
 * pcl::KdTreeFLANN<PointXYZI>::nearestKSearch returns at LidarFeatureAssociate.cpp:575 / :496 —
 * float32 squared distances accumulated as ((dx*dx)+dy*dy)+dz*dz, ascending, ties by index.
 * Neighbours farther than `max_dist` are not searched: rows with fewer than k neighbours within
 * max_dist get idx = -1 / sqd = +inf in the missing slots.  k <= 16. */
pvlm_status pvlm_knn(pvlm_ctx* ctx, const pvlm_scan* scan, int which, const float* queries, int nq, int k,
                     float max_dist, int32_t* idx, float* sqd);

/* The searches of FindNeighbors (lidar_mapping/LidarFeatureAssociate.cpp:19-111) for ALL scans at once.  Upstream builds a pcl::KdTreeFLANN over the scan centres
 * (float32 PointXYZI) and asks, per scan, for nearestKSearch(neighbor_size) and radiusSearch(20 m) — both return ascending L2_Simple distances, ties in index order.
 * Here: row i of the result = the positions j of all centres ordered by the 64-bit key (bits of the float32 squared distance ((dx*dx)+dy*dy)+dz*dz from centre i)
 * << 32 | j, ascending — the first k entries of a row are the k nearest (the scan itself first), the entries whose distance lies below a squared radius are the radius
 * search (the caller recomputes the distance of an entry it looks at: three multiply-adds).  One workgroup per row (distances + a bitonic sort in LDS).
 * xyz: n x 3 floats (host); order: n x n positions (uint16, host, the caller's).  n <= 4096 (PVLM_ERR_CAPACITY beyond: the caller keeps its own search).  The set
 * logic of FindNeighbors (loop closures, temporal neighbours) stays with the caller: it walks sorted rows. */
pvlm_status pvlm_centre_orders(pvlm_ctx* ctx, int n, const float* xyz, uint16_t* order);

/* AssociatePoint2Plane(ref, nei, plane_tolerance, dist_threshold) of
 * lidar_mapping/LidarFeatureAssociate.cpp:550-630 for a BATCH of ordered scan pairs
 * (ref[p], nei[p]); the result is the residual set util/Optimization.cpp:506-562
 * (AddLidarPointToPlaneResidual) would have added for those pairs, in the same order:
 * segment p holds the accepted correspondences of pair p in query order.
 * kind = PVLM_POINT2PLANE_ANGLE or _METER (config.angle_residual), flags/weight as for upload. */
pvlm_status pvlm_assoc_point2plane(pvlm_ctx* ctx, int n_pairs, pvlm_scan* const* ref, pvlm_scan* const* nei,
                                   double plane_tolerance, float dist_threshold, pvlm_functor kind, unsigned flags,
                                   double weight, pvlm_resset** out);
/* The plane of a correspondence (lidar_mapping/LidarFeatureAssociate.cpp:592-602: FormPlane, base/Geometry.hpp:345-373 — Eigen's column-pivoted
 * Householder QR of the 10 x 3 system A n = -1) is computed in one of two ways:
 *   default                      normal equations + an a-posteriori bound on the distance to the QR's solution (csrc/pvlm_assoc_core.h:
 *                                form_plane_fast).  The reference's accept / reject decision is taken from it only when it holds for every solution
 *                                inside the bound; otherwise — the largest point distance within ~1e-6 of plane_tolerance, an ill-conditioned
 *                                neighbourhood — the QR runs for that query.  Same correspondences, same order; the stored plane is within 5e-7
 *                                relative of the QR's by construction (1e-11 typically; the bar of the path is 1e-6).
 *   PVLM_FLAG_ASSOC_EXACT_FIT    the QR for every query, operation by operation, unfused: records bit-identical to an x86-64 build of the
 *                                reference (what the parity tests pin).  About 1.6 x the time of the plane-fit kernel.
 * pvlm_assoc_point2plane_stats: how many queries of the set took the QR because the fast fit refused (0 in exact mode). */
#define PVLM_FLAG_ASSOC_EXACT_FIT 0x200u
pvlm_status pvlm_assoc_point2plane_stats(const pvlm_resset* rs, int64_t* exact_fits);
/* The same + how the call was run: its batches (column blocks) and how many of them ran the exact plane-fit kernel.  In the default mode the first batch is a
 * probe of about a million queries (PVLM_ASSOC_PROBE_ROWS): when the fast fit refuses more than 2 % of its queries — neighbourhoods the normal equations cannot
 * handle, e.g. raw scans as targets — the other batches run the QR kernel directly, which is then the faster one.  Decided from the data alone. */
pvlm_status pvlm_assoc_point2plane_stats2(const pvlm_resset* rs, int64_t* exact_fits, int* batches, int* exact_kernel_batches);
/* Optional debug readback of the last pvlm_assoc_point2plane call: query index and the 10
 * neighbour indices of every accepted correspondence (n x 1, n x 10). */
pvlm_status pvlm_assoc_point2plane_debug(pvlm_ctx* ctx, const pvlm_resset* rs, int32_t* query_idx, int32_t* nn_idx);

/* Vote matrix of AssociateLine2Line (lidar_mapping/LidarFeatureAssociate.cpp:457-473):
 * votes[nei_seg][ref_seg] = number of nei cornerLessSharp points of that nei segment whose
 * PointToLineDistance3D to the ref segment's world line is <= dist_threshold.
 * votes is n_nei_segments x n_ref_segments, row-major, int32. */
pvlm_status pvlm_line2line_votes(pvlm_ctx* ctx, const pvlm_scan* ref, const pvlm_scan* nei, float dist_threshold,
                                 int32_t* votes);

/* Batched form — every (ref, nei) pair of an outer iteration in ONE launch and one copy back (the reference
 * calls AssociateLine2Line twice per pair per outer iteration: LidarLineMatch.cpp:68, util/Optimization.cpp:379).
 * The vote block of pair p starts at votes + vote_offsets[p] (n_nei_segments(p) x n_ref_segments(p), row-major);
 * vote_offsets (n_pairs + 1) is an output.  votes == NULL only fills vote_offsets (sizing call);
 * PVLM_ERR_CAPACITY if capacity (int32 elements) < vote_offsets[n_pairs]. */
pvlm_status pvlm_line2line_votes_batch(pvlm_ctx* ctx, int n_pairs, pvlm_scan* const* ref, pvlm_scan* const* nei,
                                       float dist_threshold, int64_t* vote_offsets, int32_t* votes, int64_t capacity);

/* The batched votes reduced on the device to what FindAssociations (LidarFeatureAssociate.cpp:126-133) reads first: per neighbour segment (row of the
 * pair's n_nei_segments x n_ref_segments block) the reference segment with the most votes — the first of equals — and that count.  Rows of pair p:
 * [row_offsets[p], row_offsets[p + 1]) (a pair whose reference scan has no segment has none); n_pairs + 1 offsets.  best_col / best_count null =
 * sizing call; PVLM_ERR_CAPACITY if capacity (entries) < row_offsets[n_pairs]. */
pvlm_status pvlm_line2line_best_batch(pvlm_ctx* ctx, int n_pairs, pvlm_scan* const* ref, pvlm_scan* const* nei, float dist_threshold,
                                      int64_t* row_offsets, int32_t* best_col, int32_t* best_count, int64_t capacity);

/* Residual blocks of the line-to-line term, built on the device (the body of the innermost loops of
 * AddLidarLineToLineResidual2, util/Optimization.cpp:404-434): match m says that segment match_nei_seg[m] of scan
 * nei[match_pair[m]] was associated with segment match_ref_seg[m] of scan ref[match_pair[m]] (AssociateLine2Line +
 * the line-track filter, decided by the caller).  Every point of the neighbour segment becomes one Point2Line block
 *   [ World2Local_nei(point) | ref_local_point + 0.1 dir | ref_local_point - 0.1 dir ]      (Optimization.cpp:410-434)
 * on the parameter blocks (ref, nei), in match order and, inside a match, in the order of edge_segmented[seg] — the
 * order the reference's AddResidualBlock calls have.  Matches must be sorted by pair; consecutive matches of one pair form
 * one segment of the result.  kind = PVLM_POINT2LINE_ANGLE / _METER.  Both scans must have been uploaded with
 * seg_points_xyz.  Upstream this loop costs one heap-allocated AutoDiffCostFunction per POINT (600 k per outer iteration
 * at Room scale); here the rows never exist on the host. */
pvlm_status pvlm_line2line_residuals(pvlm_ctx* ctx, int n_pairs, pvlm_scan* const* ref, pvlm_scan* const* nei, int n_matches,
                                     const int* match_pair, const int* match_nei_seg, const int* match_ref_seg, pvlm_functor kind,
                                     unsigned flags, double weight, pvlm_resset** out);

/* ---- equirectangular camera model + camera<->LiDAR voting --------------------------------------- */
/* Equirectangular::CamToImage<T> (sensors/Equirectangular.h:173-182, FastAtan2 variant) for n
 * points; cam n x 3, pixels n x 2.  _f32 mirrors the cv::Point3f overloads, _f64 the Eigen ones. */
pvlm_status pvlm_cam_to_image_f32(pvlm_ctx* ctx, int rows, int cols, int64_t n, const float* cam, float* pixels);
pvlm_status pvlm_cam_to_image_f64(pvlm_ctx* ctx, int rows, int cols, int64_t n, const double* cam, double* pixels);
/* Equirectangular::ImageToCam<T> (sensors/Equirectangular.h:149-170). */
pvlm_status pvlm_image_to_cam_f32(pvlm_ctx* ctx, int rows, int cols, int64_t n, const float* pixels, float r, float* cam);
pvlm_status pvlm_image_to_cam_f64(pvlm_ctx* ctx, int rows, int cols, int64_t n, const double* pixels, double r, double* cam);

/* Same maps on device-resident buffers (async on the ctx stream, no copies) — whole-panorama use. */
pvlm_status pvlm_cam_to_image_f32_dev(pvlm_ctx* ctx, int rows, int cols, int64_t n, const float* d_cam, float* d_pixels);
pvlm_status pvlm_image_to_cam_f32_dev(pvlm_ctx* ctx, int rows, int cols, int64_t n, const float* d_pixels, float r, float* d_cam);

/* ProjectLidar2PanoramaDepth (util/Visualization.h:407-441; the LiDAR-seeded depth prior of mvs/MVS.cpp:510-514):
 * transforms the cloud (n x 3 float, LiDAR frame) by T_cl like pcl::transformPointCloud, projects with
 * Equirectangular::CamToImage<float> and paints depth * 256 (uint16) over the window
 * [floor(px) - size/2, ceil(px) + size/2]^2; points whose window leaves the image are skipped; where windows
 * overlap the LAST point of the cloud wins, as in the reference's sequential loop.  depth: rows x cols uint16. */
pvlm_status pvlm_project_lidar_depth(pvlm_ctx* ctx, int rows, int cols, int64_t n, const float* xyz, const double* T_cl_rowmajor16,
                                     unsigned size, uint16_t* depth);

/* MVS::InitDepthNormal (mvs/MVS.cpp:496-584; config 5 "LiDAR-seeded depth priors"): the depth image of
 * pvlm_project_lidar_depth (uint16, depth * 256; size 2 upstream, :512) seeds the depth map, every pixel without a LiDAR depth
 * gets a uniform random depth in [min_depth, max_depth], keep_lidar_constant != 0 marks the seeded pixels in depth_constant
 * (config.keep_lidar_constant, :561-565), `mask` (float, 1 = keep) zeroes excluded pixels (:571), and every kept pixel gets a
 * random unit normal facing the camera (GenerateRandomNormal :1404-1431).  lidar_depth == NULL is the use_lidar = false branch
 * (:566-569), mask == NULL keeps everything.  Random draws: upstream uses one cv::RNG seeded with time(NULL); here draw k of
 * pixel e is a hash of (seed, e, k), as in pvlm_mvs_propagate — the same arguments give the same maps.
 * depth: rows x cols, normal: rows x cols x 3 (outputs); depth_constant: rows x cols uint8, written only with a LiDAR image. */
pvlm_status pvlm_mvs_init_depth_normal(pvlm_ctx* ctx, int rows, int cols, const uint16_t* lidar_depth_or_null, const float* mask_or_null, float min_depth,
                                       float max_depth, int keep_lidar_constant, unsigned long long seed, float* depth, float* normal,
                                       unsigned char* depth_constant_or_null);
/* MVS::RemoveSmallSegments (mvs/MVS.cpp:1504-1577, called at the end of a view's estimation, :102 / :138): 4-connected regions of
 * similar depth (relative difference < depth_diff_threshold, config 0.01) grown from seeds in column-major order; regions with fewer
 * than min_segment pixels (config/Room.txt:92: 100) lose depth (0), normal (0) and confidence (-1) — pixels without depth are
 * regions of size one and are reset as well.  In place on HOST arrays, executed on the host: the growth rule divides by the depth
 * of the pixel a neighbour is reached from, so the result depends on the sequential seed order (see csrc/pvlm_mvs.hip). */
pvlm_status pvlm_mvs_remove_small_segments(pvlm_ctx* ctx, int rows, int cols, float depth_diff_threshold, int min_segment, float* depth, float* normal,
                                           float* conf, int64_t* removed_or_null);

/* Photometric scoring pass of the panoramic PatchMatch MVS: MVS::InitPatchMap + MVS::InitConfMap(use_geometry = false)
 * (mvs/MVS.cpp:586-680) with the photometric term of ScorePixel (:774-923): for every pixel with depth > 0 the
 * bilaterally weighted NCC of its (2 half_window + 1)^2 / step^2 window against each neighbour panorama through the
 * plane-induced homography R_nr + t_nr n^T / d, averaged over the two best neighbours; conf = -1 (and depth / normal
 * zeroed) where the reference patch is invalid, the plane faces away (d > 0) or no neighbour sees the window.
 * ref_gray / nei_gray[b]: rows x cols uint8; R_nr: n x 9 row-major, t_nr: n x 3 (reference -> neighbour camera);
 * depth (rows x cols), normal (rows x cols x 3, camera frame), conf (rows x cols): float32, in-out. n_neighbors <= 16.
 * nei_depth != NULL = InitConfMap(use_geometry = true): nei_depth[b] is neighbour b's photometric depth map
 * (Frame::depth_filter) and every neighbour score gets the geometric-consistency adjustment of ScorePixel :857-893
 * (forward / backward projection through that depth map, 0.2 x min(angle in degrees, 2) penalty). */
pvlm_status pvlm_mvs_init_conf_map(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                   const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal,
                                   float* conf, const float* const* nei_depth_or_null);

/* PatchMatch sweep: MVS::EstimateDepthMapSingle(ref, Propagate::CHECKER_BOARD, max_iter, conf_threshold, use_geometry)
 * (mvs/MVS.cpp:682-720) = max_iter x PropagateCheckerBoard (:1098-1129; per pixel ProcessPixel :721-772: the four direct
 * neighbours' hypotheses interpolated to the pixel and re-scored with the smoothness term, then PerturbDepthNormal3
 * :1254-1320: up to 6 random + 6 refining perturbations), then hypotheses with conf < conf_threshold are dropped
 * (depth 0, conf -1, normal 0; depth_constant pixels are kept).  depth / normal / conf are IN-OUT and must hold an
 * initialised state (InitDepthNormal + pvlm_mvs_init_conf_map).  nei_depth != NULL = use_geometry; depth_constant may
 * be NULL; min_depth / max_depth = config.min_depth / max_depth.
 * RANDOM DRAWS: upstream all OpenMP threads pull from one cv::RNG seeded with time(NULL) (mvs/MVS.cpp:30), a data race
 * that makes every run different; here draw k of pixel e in colour pass p is a hash of (seed, p, e, k) — the same
 * arguments give the same maps. */
pvlm_status pvlm_mvs_propagate(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                               const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal, float* conf,
                               const float* const* nei_depth_or_null, const unsigned char* depth_constant_or_null, float min_depth, float max_depth,
                               unsigned long long seed, int max_iter, float conf_threshold);
/* The same with Propagate::SEQUENTIAL — the strategy config/Room.txt:90 and config/Floor.txt:88 select (propagate_strategy = 2):
 * MVS::PropagateSequential (mvs/MVS.cpp:1057-1097) walks the image in raster order on even iterations (a pixel takes the
 * hypotheses of its LEFT and UPPER neighbour, which the walk has just updated) and backwards on odd ones (right, lower); pixels
 * with patch.sq0 <= 0 are skipped.  Here every anti-diagonal col + row = d is one launch, ascending (descending) d: a pixel
 * reads only its four direct neighbours, so the pixels of a diagonal are independent and each sees exactly what the raster walk
 * shows it — same maps as the sequential loop, bit for bit.  Draw k of pixel e in iteration i = hash(seed, i, e, k). */
pvlm_status pvlm_mvs_propagate_sequential(pvlm_ctx* ctx, int rows, int cols, int half_window, int step, const unsigned char* ref_gray, int n_neighbors,
                                          const unsigned char* const* nei_gray, const float* R_nr, const float* t_nr, float* depth, float* normal,
                                          float* conf, const float* const* nei_depth_or_null, const unsigned char* depth_constant_or_null,
                                          float min_depth, float max_depth, unsigned long long seed, int max_iter, float conf_threshold);

/* Depth-map fusion filter: MVS::FilterDepthImage (mvs/MVS.cpp:1735-1790) with ProjectDepthConfToRef (:2011-2070, depth):
 * every neighbour depth map is forward-projected into the reference view (4-pixel splat, nearest range wins); a reference
 * depth survives when >= 2 neighbours agree within 0.8 x threshold at the pixel and >= 5 (neighbour, 4-neighbourhood)
 * samples agree within 1.2 x threshold (or depth_constant marks it).  depth_filter / conf_filter (rows x cols float)
 * are outputs, 0 where rejected; conf / conf_filter and depth_constant may be NULL. */
pvlm_status pvlm_mvs_filter_depth(pvlm_ctx* ctx, int rows, int cols, int n_neighbors, const float* const* nei_depth, const float* R_nr, const float* t_nr,
                                  const float* depth, const float* conf_or_null, const unsigned char* depth_constant_or_null,
                                  float depth_diff_threshold, float* depth_filter, float* conf_filter_or_null);

/* The fusion filter the reference's MVS pipeline actually runs (mvs/MVS.cpp:194): MVS::FilterDepthImageRefine
 * (mvs/MVS.cpp:1794-1890) with ProjectDepthConfToRef projecting depth and confidence (:2011-2070).  A reference depth is
 * replaced by the confidence-weighted average of itself and the neighbour ranges that agree within 1.2 x threshold when
 * >= 2 neighbours agree, the agreeing confidence exceeds the disagreeing one (occlusions / free-space violations) and
 * the average lies in [min_depth, max_depth] (config.min_depth / max_depth, base/Config.h:65-66); conf_filter = positive -
 * negative confidence.  depth_constant pixels that fail keep their depth with confidence 1.  nei_conf[b] = neighbour
 * b's conf_map after ConvertNCC2Conf (:2343); conf (the reference frame's conf_map) is IN-OUT: zeroed where depth <= 0,
 * as upstream.  depth_filter / conf_filter: rows x cols float outputs, 0 where rejected. */
pvlm_status pvlm_mvs_filter_depth_refine(pvlm_ctx* ctx, int rows, int cols, int n_neighbors, const float* const* nei_depth,
                                         const float* const* nei_conf, const float* R_nr, const float* t_nr, const float* depth, float* conf,
                                         const unsigned char* depth_constant_or_null, float depth_diff_threshold, float min_depth, float max_depth,
                                         float* depth_filter, float* conf_filter);

/* Depth map -> coloured world points: MVS::DepthImageToCloud (mvs/MVS.cpp:2073-2107; filter_sky = 1, normal_out = NULL) and
 * MVS::DepthNormalToCloud (:2109-2142; filter_sky = 0, normal_out != NULL), the per-frame bodies of MVS::MergeDepthImages
 * (:2144-2166; what MVS::FuseDepthMaps :224-227 saves as MVS-merge.pcd) and of the per-frame .pcd export.  A pixel with
 * 0 < depth < 0.8 max_depth becomes TranslatePoint<float, double>(ImageToCam(col, row) * depth, T_wc) (base/Geometry.hpp:545-551)
 * with its colour — unless filter_sky and BGR2HSV (util/Visualization.cpp:57-77) puts the colour in the reference's "sky blue"
 * box (H 100..124, S 43..200, V 150..255) — and, with normal_out, the normal R_wc n.  Points come out in raster order, as
 * the reference's loops push them.  bgr: rows x cols x 3 (cv::Vec3b order); T_wc: the frame pose, 3 x 4 or 4 x 4 row-major
 * (12 doubles read); xyz / normal_out: capacity rows x cols x 3 float, rgb: rows x cols x 3 bytes (r, g, b); *n_points = points written.
 * pvlm_mvs_views_depth_to_cloud (declared below) reads depth (or depth_filter) and normal of a resident view instead. */
pvlm_status pvlm_mvs_depth_to_cloud(pvlm_ctx* ctx, int rows, int cols, const float* depth, const unsigned char* bgr, const float* normal_or_null,
                                    const double* T_wc, float max_depth, int filter_sky, float* xyz, unsigned char* rgb, float* normal_out_or_null,
                                    long long* n_points);

/* Resident view set: the maps of n_views equally sized views (grey image, depth, normal, conf, depth_filter, conf_filter —
 * the cv::Mat members of sensors/Frame.h the MVS touches) stay in HBM across the calls, so that a view's scoring pass,
 * sweeps and fusion filter — and its use as somebody's neighbour — cost no PCIe traffic.  Same kernels and results as
 * the per-call entry points above.  `nei` holds view indices (each != ref); R_nr / t_nr as above.
 *   pvlm_mvs_views_estimate      max_iter < 0: MVS::InitConfMap(ref, ., use_geometry); max_iter >= 0: MVS::EstimateDepthMapSingle
 *                                (checkerboard; pvlm_mvs_views_estimate_sequential: the sequential sweep).  use_geometry reads the neighbours' depth_filter (the photometric depth the reference
 *                                keeps there, mvs/MVS.cpp:470-473): pvlm_mvs_views_snapshot_depth copies depth -> depth_filter of a view.
 *   pvlm_mvs_views_filter_refine MVS::FilterDepthImageRefine(ref): writes depth_filter / conf_filter of ref, zeroes its conf where depth <= 0.
 * estimate / filter_refine / snapshot_depth are asynchronous on the context's stream; upload / download synchronise.
 * NULL pointers in upload / download skip that map. */
typedef struct pvlm_mvs_views pvlm_mvs_views;
pvlm_status pvlm_mvs_views_create(pvlm_ctx* ctx, int rows, int cols, int n_views, pvlm_mvs_views** out);
pvlm_status pvlm_mvs_views_destroy(pvlm_ctx* ctx, pvlm_mvs_views* views);
pvlm_status pvlm_mvs_views_depth_to_cloud(pvlm_ctx* ctx, pvlm_mvs_views* views, int view, int use_filtered_depth, const unsigned char* bgr, const double* T_wc,
                                          float max_depth, int filter_sky, float* xyz, unsigned char* rgb, float* normal_out_or_null, long long* n_points);
pvlm_status pvlm_mvs_views_upload(pvlm_ctx* ctx, pvlm_mvs_views* views, int view, const unsigned char* gray_or_null, const float* depth_or_null,
                                  const float* normal_or_null, const float* conf_or_null);
pvlm_status pvlm_mvs_views_download(pvlm_ctx* ctx, pvlm_mvs_views* views, int view, float* depth_or_null, float* normal_or_null, float* conf_or_null,
                                    float* depth_filter_or_null, float* conf_filter_or_null);
pvlm_status pvlm_mvs_views_snapshot_depth(pvlm_ctx* ctx, pvlm_mvs_views* views, int view);
pvlm_status pvlm_mvs_views_estimate(pvlm_ctx* ctx, pvlm_mvs_views* views, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                    int half_window, int step, int use_geometry, const unsigned char* depth_constant_or_null, float min_depth,
                                    float max_depth, unsigned long long seed, int max_iter, float conf_threshold);
/* EstimateDepthMapSingle with Propagate::SEQUENTIAL on a resident view (see pvlm_mvs_propagate_sequential); max_iter >= 0 */
pvlm_status pvlm_mvs_views_estimate_sequential(pvlm_ctx* ctx, pvlm_mvs_views* views, int ref, int n_neighbors, const int* nei, const float* R_nr,
                                               const float* t_nr, int half_window, int step, int use_geometry, const unsigned char* depth_constant_or_null,
                                               float min_depth, float max_depth, unsigned long long seed, int max_iter, float conf_threshold);
/* The same for n_jobs DISTINCT reference views in one go (one launch per anti-diagonal covers all of them): a single view's
 * diagonal is too short to fill the GPU, and upstream runs this strategy with one image per thread for the same reason
 * (mvs/MVS.cpp:87-93).  Job j: reference refs[j], its nei_counts[j] neighbours follow those of job j - 1 in nei / R_nr / t_nr,
 * random stream seeds[j]; depth_constant: NULL or n_jobs pointers (NULL entries allowed).  Results = n_jobs calls of
 * pvlm_mvs_views_estimate_sequential, bit for bit.  Synchronises before returning. */
pvlm_status pvlm_mvs_views_estimate_sequential_batch(pvlm_ctx* ctx, pvlm_mvs_views* views, int n_jobs, const int* refs, const int* nei_counts, const int* nei,
                                                     const float* R_nr, const float* t_nr, int half_window, int step, int use_geometry,
                                                     const unsigned char* const* depth_constant_or_null, float min_depth, float max_depth,
                                                     const unsigned long long* seeds, int max_iter, float conf_threshold);
pvlm_status pvlm_mvs_views_filter_refine(pvlm_ctx* ctx, pvlm_mvs_views* views, int ref, int n_neighbors, const int* nei, const float* R_nr, const float* t_nr,
                                         const unsigned char* depth_constant_or_null, float depth_diff_threshold, float min_depth, float max_depth);

/* Hot loop #3 of CameraLidarLineAssociate::AssociateByAngle
 * (joint_optimization/CameraLidarLineAssociate.cpp:394-426): for every image line (x1,y1,x2,y2
 * pixels, n_lines x 4 float) and every LiDAR corner point (LiDAR-local float xyz, transformed by
 * T_cl as pcl::transformPointCloud does) apply the 15 m range test and the two 3-degree angle tests
 * and count votes per LiDAR segment.  votes is n_lines x n_segments int32. */
pvlm_status pvlm_cam_lidar_votes(pvlm_ctx* ctx, int rows, int cols, const float* lines, int n_lines,
                                 const pvlm_scan* lidar_local, const double* T_cl_rowmajor16, int32_t* votes);

/* Batched form — every (frame, LiDAR) pair of AssociateLineMulti (CameraLidarOptimizer.cpp:345-377, an omp loop
 * over frames upstream) in one launch: pair p uses the image lines [line_offsets[p], line_offsets[p+1]) of
 * `lines`, the scan lidar_local[p] and T_cl + 16 p; its votes (n_lines(p) x n_segments(p)) start at
 * votes + vote_offsets[p].  Sizing / capacity rules as for pvlm_line2line_votes_batch. */
pvlm_status pvlm_cam_lidar_votes_batch(pvlm_ctx* ctx, int n_pairs, int rows, int cols, const int64_t* line_offsets,
                                       const float* lines, pvlm_scan* const* lidar_local, const double* T_cl_rowmajor16,
                                       int64_t* vote_offsets, int32_t* votes, int64_t capacity);

/* The same launch with the votes returned SPARSE: a few per cent of the counters are non-zero (43 MB of dense blocks for the 1 362 pairs of a Room
 * sequence).  vote_offsets as above (the dense layout the indices refer to); nz_index[k] = dense position of the k-th non-zero counter (ascending:
 * pair by pair, line by line, segment by segment), nz_count[k] = its value; *n_nz = how many there are.  capacity = room in nz_index / nz_count:
 * PVLM_ERR_CAPACITY when it is too small (*n_nz is set: call again). */
pvlm_status pvlm_cam_lidar_votes_batch_sparse(pvlm_ctx* ctx, int n_pairs, int rows, int cols, const int64_t* line_offsets, const float* lines,
                                              pvlm_scan* const* lidar_local, const double* T_cl_rowmajor16, int64_t* vote_offsets, int64_t* nz_index,
                                              int32_t* nz_count, int64_t capacity, int64_t* n_nz);

/* ---- LiDAR feature extraction, range-image stages ------------------------------------------------------- *
 * The per-point / per-ring stages in front of the association, for a BATCH of raw scans in one go (the reference runs them scan by
 * scan under `omp parallel for`, lidar_mapping/LidarOdometry.cpp:131-147):
 *   Velodyne::ReOrderVLP      sensors/Velodyne.cpp:371-526    firing order -> ring order; range image; (ring, column) of every point
 *   Velodyne::Segmentation    sensors/Velodyne.cpp:1438-1586  range-image labelling; points of small components dropped (segment != 0)
 *   adaptive-window curvature sensors/Velodyne.cpp:623-657    the first loop of Velodyne::ExtractFeatures (method ADAPTIVE)
 * What they produce — cloud_scan, point_idx_to_image, image_to_point_idx, range_image, scanStartInd / scanEndInd, cloudCurvature,
 * left_neighbor / right_neighbor, cloudDistance (sensors/Velodyne.h:97-120 and the locals of ExtractFeatures) — is what the
 * sort-dependent picks (ExtractEdgeFeatures2 :883-1000, ExtractPlaneFeatures2 :1098-1189) read.
 * xyzi: the scan as LoadLidar left it (sensors/Velodyne.cpp:92-145; camera-style axes, float x y z intensity), point i at
 * xyzi + i * stride_floats (pcl::PointXYZI: stride 8; packed: 4).  Non-finite coordinates are an error (LoadLidar removed them).
 * Decisions that go through the float libm of the reference's host (atan / atan2 with `using namespace std`) are certified on
 * the device in fp64 interval form; the few points / edges that cannot be certified are decided by the libm of THIS host
 * inside the call (resolved_points / resolved_edges count them; replayed = 1 when a scan's +z crossing needed every point). */
typedef struct pvlm_ring_batch pvlm_ring_batch;
typedef struct pvlm_raw_scan {
  const float* xyzi;
  int n;
  int stride_floats;
} pvlm_raw_scan;
typedef struct pvlm_ring_result {
  int n_raw;                        /* points handed in                                                                          */
  int n_reordered;                  /* cloud_scan.size() after ReOrderVLP                                                        */
  int n_kept;                       /* cloud_scan.size() after Segmentation (== n_reordered when segment == 0)                   */
  int resolved_points, resolved_edges, replayed;
  const int* ring_count_reordered;  /* 64 entries: points per ring after ReOrderVLP                                              */
  const int* ring_count;            /* 64 entries: points per ring of the kept cloud: scanStartInd[r] = sum_{q<r} + 5,
                                       scanEndInd[r] = sum_{q<=r} - 6 (:520-522, :1575-1580)                                      */
  /* the kept cloud, n_kept entries each, valid until pvlm_ring_batch_destroy (pinned host memory owned by the batch):           */
  const int* source;                /* index of the point in the raw scan: cloud_scan[i] = (raw xyz, intensity = ring)           */
  const int* ring_col;              /* point_idx_to_image[i] as (ring << 16) | column                                            */
  const float* curvature;           /* cloudCurvature[i]; -1 where upstream leaves it unset                                      */
  const int* half_window;           /* left_neighbor[i] = i - half_window[i], right_neighbor[i] = i + half_window[i]; -1 = unset  */
  const float* range;               /* cloudDistance[i] = range_image(point_idx_to_image[i])  (:566-569)                         */
  const int* sorted;                /* cloudSortInd: the six sectors of every ring (:707-723) ordered by curvature as ExtractEdgeFeatures2
                                       :896 / ExtractPlaneFeatures2 :1110 sort them; index order outside the sectors                */
  const unsigned char* sector_host; /* n_rings x 6: 1 = the sector holds equal curvatures (or a NaN, or > 2048 points): std::sort's
                                       order of equal keys is the library's, `sorted` is the index order there and the caller sorts */
  /* pvlm_ring_extract_batch_picks only (picks == 1): the picks of ExtractEdgeFeatures2 (:883-1000) / ExtractPlaneFeatures2 (:1098-1189) and the
   * voxel grid of the less-flat points, ring by ring.  A scan with any ring_host[r] != 0 (incidence angle within libm distance of the threshold,
   * a sector left to the host, a ring beyond the kernel's bounds) is to be picked by the caller from the arrays above.                          */
  int picks;
  float max_curvature, intersect_angle_threshold;   /* what the picks were made with                                                */
  const unsigned char* state;       /* PointClassification bits of every kept point after both pick passes                         */
  const int* corner;                /* n_rings x 181: count, then the edge picks in pick order: point | 0x80000000 when sharp       */
  const int* flat;                  /* n_rings x 25: count, then the plane picks (surfFlat) in pick order                           */
  const int* voxel_span;            /* n_rings x 2: first, count of the ring's surfLessFlat centroids in `voxels`                  */
  const float* voxels;              /* x, y, z, class (1 = POINT_NORMAL) per centroid, ascending voxel order inside a ring          */
  const unsigned char* ring_host;   /* n_rings flags                                                                               */
} pvlm_ring_result;
pvlm_status pvlm_ring_extract_batch(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* scans, int n_rings, int horizon_scans, int segment,
                                    pvlm_ring_batch** out);
/* The same, plus K24: the feature picks and the voxel grid (max_curvature, intersect_angle_threshold: the arguments of Velodyne::ExtractFeatures).  With the picks made
 * on the device the five per-point arrays curvature / half_window / range / sorted / sector_host (16 of the 27 B per point of the download) are delivered only for the
 * scans that have a ring left to the host (ring_host != 0) — they are NULL for the others — unless bit 1 of `segment` is set (segment = 2 | run_segmentation: always). */
pvlm_status pvlm_ring_extract_batch_picks(pvlm_ctx* ctx, int n_scans, const pvlm_raw_scan* scans, int n_rings, int horizon_scans, int segment,
                                          float max_curvature, float intersect_angle_threshold, pvlm_ring_batch** out);
pvlm_status pvlm_ring_batch_scan(const pvlm_ring_batch* batch, int scan, pvlm_ring_result* result);
/* The device-resident arrays of one scan (tests, visualisation).  state 0: after ReOrderVLP, 1: after Segmentation.  cloud_xyzi: n x 4
 * (intensity = ring), ring_col_pairs: n x 2, range_image / image_to_point: n_rings x horizon_scans (0 / -1 = empty).  NULL = skip. */
pvlm_status pvlm_ring_batch_fetch(pvlm_ctx* ctx, const pvlm_ring_batch* batch, int scan, int state, float* cloud_xyzi, int* ring_col_pairs,
                                  float* range_image, int* image_to_point);
/* HIP-event milliseconds of the last run: [0] upload  [1] K16 ring/azimuth + host libm  [2] K17 columns (+ replays)  [3] K18 scatter
 * [4] K19 edges + host libm  [5] K20 components  [6] K21 compaction + K22 curvature  [7] download. */
pvlm_status pvlm_ring_batch_timing(const pvlm_ring_batch* batch, double* ms8);
pvlm_status pvlm_ring_batch_destroy(pvlm_ctx* ctx, pvlm_ring_batch* batch);
/* Tests: the device's restatement of std::sort (csrc/pvlm_stdsort.h, sort_wave) on caller-given keys, n <= 4096: order[k] = index of the k-th element
 * as std::sort of the indices 0 .. n-1 by `keys[a] < keys[b]` leaves them — equal keys in libstdc++'s order.                                      */
pvlm_status pvlm_ring_debug_sort(pvlm_ctx* ctx, const unsigned* keys, int n, int* order);

/* ---- growth of the line segments of a batch of scans (N3, K27) -----------------------------------------------------------------------------
 * The growth phase of ExtractLineFeatures / ExpandLine (sensors/LidarLineExtraction.cpp:296-389, :10-70; called by Velodyne::EdgeToLine,
 * sensors/Velodyne.cpp:1269-1324) for every scan of a batch.  The segment grown from edge point i with its neighbours (a, b), 1 <= a < b <= 4 among its four nearest
 * edge points, never looks at what other segments took — only upstream's walk over the start points does (a point an earlier segment took is skipped) — so the
 * device grows the tasks of the next start points side by side and replays the walk over them (csrc/pvlm_linegrow.hip).  clouds[s]: the edge points of scan s
 * (cornerLessSharp before the filter), n <= 65535, stride_floats >= 3 floats between points (pcl::PointXYZI: 4).
 * pvlm_line_grow_scan: the segments upstream's walk keeps for one scan, in its order (start point ascending, then the combinations (1,2) (1,3) (1,4) (2,3) (2,4)
 * (3,4)): seg_task[q] = 6 i + combination, members[seg_offset[q] .. seg_offset[q + 1]) = ascending point ids of segment q (>= 5), coeffs[6 q ..] =
 * FormLine(members, 1.0) (zeros when that fit refuses: kept, as upstream keeps it).  status != 0: a segment of this scan outgrew the kernel's lists (2: more than 64
 * members, or more than 8192 edge points) or met a turn angle the host's acos has to decide (3) — the scan is to be grown by the caller.  PVLM_ERR_REFUSED: the
 * batch's segment pool was exhausted (nothing is returned; grow on the host).  The result lives in host memory until pvlm_line_grow_destroy. */
typedef struct pvlm_edge_cloud { const float* xyz; int n; int stride_floats; } pvlm_edge_cloud;
typedef struct pvlm_line_grow pvlm_line_grow;
typedef struct pvlm_line_grow_result {
  int status, n_points, n_segments;
  const int* seg_task;
  const int* seg_offset;            /* n_segments + 1 entries, offsets into `members` */
  const int* members;
  const double* coeffs;
  double kernel_ms;                 /* HIP-event time of the batch's two kernels */
  long long tasks_run;              /* (start point, neighbour pair) tasks the batch grew, the speculative ones included */
} pvlm_line_grow_result;
pvlm_status pvlm_line_grow_batch(pvlm_ctx* ctx, int n_scans, const pvlm_edge_cloud* clouds, pvlm_line_grow** out);
/* The same in two halves: _begin copies the clouds, queues the batch on a stream of its own (ordered behind what the context's stream holds at that moment) and returns
 * without waiting — the caller's next calls on this context (the range-image stages of the next device batch, say) run beside it; _finish waits and brings the
 * segments down.  One batch in flight per context (PVLM_ERR_STATE otherwise); pvlm_line_grow_scan serves finished batches only. */
pvlm_status pvlm_line_grow_begin(pvlm_ctx* ctx, int n_scans, const pvlm_edge_cloud* clouds, pvlm_line_grow** out);
pvlm_status pvlm_line_grow_finish(pvlm_ctx* ctx, pvlm_line_grow* grow);
pvlm_status pvlm_line_grow_scan(const pvlm_line_grow* grow, int scan, pvlm_line_grow_result* result);
pvlm_status pvlm_line_grow_destroy(pvlm_ctx* ctx, pvlm_line_grow* grow);

/* ---- motion compensation of the sweeps (N5) ---------------------------------------------------------------------------------------------
 * Velodyne::UndistortCloud(R_we, t_we) (sensors/Velodyne.cpp:1642-1674) for every scan of LidarOdometry::UndistortLidars
 * (lidar_mapping/LidarOdometry.cpp:189-263) in one call: point i of n moves by the share i / n of the motion from the sweep's start pose (R_wl, t_wl)
 * to its end pose (R_we, t_we) — rotation by Identity.slerp(i / n, q_se), translation (i / n) t_se.  xyzi: n points of stride_floats >= 4 floats
 * (x, y, z first; pcl::PointXYZI: 8), updated in place; rotations row-major 3x3, world <- sensor.  The caller picks the end pose (next scan's pose
 * through SlerpPose, ...) and clears the feature clouds as upstream does.                                                                    */
typedef struct pvlm_undistort_scan {
  float* xyzi;
  int n;
  int stride_floats;
  const double* R_wl; const double* t_wl;
  const double* R_we; const double* t_we;
} pvlm_undistort_scan;
pvlm_status pvlm_undistort_batch(pvlm_ctx* ctx, int n_scans, const pvlm_undistort_scan* scans);

#ifdef __cplusplus
}
#endif
#endif /* PVLM_H_ */
