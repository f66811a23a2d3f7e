/*
 * voxgraph_amd.h -- C ABI of libvoxgraph_amd.so: voxgraph's two data-parallel
 * inner loops as hand-written HIP kernels for MI355X (gfx950).
 *
 *   REG  : voxgraph::RegistrationCostFunction::Evaluate
 *          (voxgraph/src/backend/constraint/cost_functions/registration_cost_function.cpp:58-298)
 *   TSDF : voxblox::FastTsdfIntegrator::integratePointCloud, called at
 *          voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:83
 *
 * Plain C: opaque handles, pointers and sizes, int status codes.  No
 * exceptions or aborts cross this boundary.  All host pointers are ordinary
 * (pageable) memory unless a parameter is documented as a DEVICE pointer.
 * There is no CPU fallback: without a gfx950 device vgx_ctx_create fails with
 * VGX_ERR_NO_DEVICE and nothing else can be called.
 *
 * Reference citations are relative to /root/reference/voxgraph/.
 * INTEGRATION.md shows the reference-side C++ that binds these entry points.
 */
#ifndef VOXGRAPH_AMD_H_
#define VOXGRAPH_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGX_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------ */
#define VGX_OK 0
/* Evaluate() would `return false`: summed reference weight == 0
 * (registration_cost_function.cpp:273).  Outputs are unspecified. */
#define VGX_EVALUATE_FALSE 1
#define VGX_ERR_INVALID (-1)     /* bad argument / handle / state            */
#define VGX_ERR_HIP (-2)         /* a HIP runtime call failed                */
#define VGX_ERR_NOMEM (-3)       /* host or device allocation failed         */
#define VGX_ERR_UNSUPPORTED (-4) /* e.g. voxels_per_side not in {8,16}       */
#define VGX_ERR_NO_DEVICE (-5)   /* no HIP device / not gfx950               */

typedef struct vgx_ctx_s* vgx_ctx;
typedef struct vgx_submap_s* vgx_submap;
typedef struct vgx_reg_s* vgx_reg;
typedef struct vgx_reg_batch_s* vgx_reg_batch;
typedef struct vgx_reg_multi_s* vgx_reg_multi;
typedef struct vgx_reg_assembler_s* vgx_reg_assembler;

/* ---- context ----------------------------------------------------------- */
/* One context per process per GPU (one process per GPU is the multi-GPU
 * model; constraint shards meet in an RCCL all-reduce issued by the host on
 * the buffer vgx_reg_batch_evaluate_normal fills). */
VGX_API int vgx_ctx_create(int device, vgx_ctx* out);
VGX_API int vgx_ctx_destroy(vgx_ctx ctx);
/* Human-readable description of the last error on this context (or of the
 * last failed vgx_ctx_create when ctx == NULL). Never NULL. */
VGX_API const char* vgx_last_error(vgx_ctx ctx);
/* THREADING AND STREAMS.  A context has two sides, each with its own HIP stream and its own lock, as the reference has
 * two threads (voxgraph_mapper.cpp:218-238: the ROS thread integrates scans while optimizePoseGraph runs on a
 * std::async thread):
 *   registration side   everything on FINISHED submaps -- vgx_submap_*, vgx_reg_*, vgx_find_overlapping_pairs,
 *                       vgx_map_file_load_submap ...; stream: vgx_ctx_set_stream / _get_stream.
 *   TSDF side           the ACTIVE submap -- vgx_tsdf_layer_*, vgx_tsdf_integrator_*, vgx_tsdf_integrate*;
 *                       stream: vgx_ctx_set_tsdf_stream / _get_tsdf_stream.
 * A scan therefore neither queues behind a solver evaluation nor waits for its lock: one thread may integrate while
 * another evaluates, on the same context (tests/test_concurrency_gpu.py).  What makes that safe is the reference's own
 * invariant (voxgraph_mapper.cpp:464-471): a finished submap is immutable, and only the active layer is written.  The
 * two sides meet in vgx_submap_from_tsdf_layer (finishSubmap(), voxgraph_submap.cpp:84-107), which orders the
 * registration stream behind the TSDF stream with an event; the layer must not be integrated into while that call
 * runs (the reference finishes a submap on the thread that integrates).  Any entry point may be called from any
 * thread; calls on one side are serialised among themselves (except the drop-in vgx_reg_evaluate, which overlaps on up
 * to eight evaluation streams).  DEVICE pointers handed to vgx_tsdf_integrate*_device must be ready with respect to the
 * TSDF stream (complete, produced on / ordered before vgx_ctx_get_tsdf_stream(), or ordered behind their producer with
 * vgx_ctx_tsdf_wait_for_stream).  NOTE for callers that set the context's stream to their own (vgx_ctx_set_stream) and
 * produce scan points on it: since round 5 scans do NOT run on that stream -- order them with
 * vgx_ctx_tsdf_wait_for_stream(ctx, NULL) before the vgx_tsdf_integrate*_device call, or hand the same stream to
 * vgx_ctx_set_tsdf_stream as well.
 *
 * PRIORITY.  The TSDF side goes first: the context's own TSDF stream is created with the device's highest stream
 * priority and its own registration stream with the lowest, so that a scan (one short kernel the sensor's cadence waits
 * for) is dispatched as workgroups of a running solver evaluation (thousands, which nobody waits for one by one) retire
 * instead of behind all of them -- voxgraph optimises in the background of its mapping thread
 * (voxgraph_mapper.cpp:218-238).  Priorities alone did not do it on gfx950 (measured); what does: while the context has a
 * TSDF integrator, the fused pass's tile kernel is launched FIVE workgroups deep per CU instead of six, which leaves the
 * registers of one scan workgroup free on every CU -- a scan under a running solve then takes 0.14-0.32 ms (median)
 * instead of 0.7 ms, at + 3 % per solver evaluation; a context without an integrator runs the solver at full depth.
 * Measured per-scan latency under a running solve: bench.py `tsdf.*.latency_under_solve_us`,
 * profiles/r06_scan_latency.txt.  Streams handed in by the caller keep the priority
 * the caller gave them.  vgx_ctx_stream_priorities: 1 when the own streams were created that way (0: the device offers
 * one level only, or VGX_STREAM_PRIORITY=0 in the environment -- A/B aid).
 *
 * vgx_ctx_set_stream: launch the registration side on an existing hipStream_t (e.g. the caller's PyTorch stream);
 * vgx_ctx_set_tsdf_stream: the same for the TSDF side (waits for what that side has queued so far).  NULL restores
 * the context's own stream.  vgx_ctx_synchronize waits for both, vgx_ctx_synchronize_tsdf for the TSDF side alone (the
 * mapping thread's "is my scan in?", whatever the solver has queued).  vgx_ctx_tsdf_wait_for_stream: the TSDF stream
 * waits ON THE DEVICE for what producer_stream (NULL: the context's registration stream) holds at the time of the call
 * (HIP's legacy default stream, whose handle IS NULL, cannot be named here: produce on a created stream). */
VGX_API int vgx_ctx_set_stream(vgx_ctx ctx, void* hip_stream);
VGX_API void* vgx_ctx_get_stream(vgx_ctx ctx);
VGX_API int vgx_ctx_set_tsdf_stream(vgx_ctx ctx, void* hip_stream);
VGX_API void* vgx_ctx_get_tsdf_stream(vgx_ctx ctx);
VGX_API int vgx_ctx_synchronize_tsdf(vgx_ctx ctx);
VGX_API int vgx_ctx_tsdf_wait_for_stream(vgx_ctx ctx, void* producer_stream);
VGX_API int vgx_ctx_stream_priorities(vgx_ctx ctx);
/* How the sampling grids of the submaps created on this context FROM NOW ON are laid out in HBM (set it
 * once, before the first submap; a batch refuses to mix layouts).  Results never depend on it.
 *   VGX_BRICKS_APRON (default)  17^3 floats per block; fewest bytes: fastest where every registration
 *                               point is evaluated (sampling_ratio -1).
 *   VGX_BRICKS_QUAD             a 2x2x2 neighbourhood is 32 contiguous bytes (4.25 x the memory):
 *                               fastest where evaluations are scattered -- the reference's shipped
 *                               sampling_ratio 0.05 (voxgraph_mapper.yaml:34): -14 % per solver evaluation,
 *                               at +18-25 % on the all-points passes. */
#define VGX_BRICKS_APRON 0
#define VGX_BRICKS_QUAD 1
VGX_API int vgx_ctx_set_brick_layout(vgx_ctx ctx, int32_t layout);
/* Sampling sessions (round 4).  A vgx_reg_batch whose constraints ALL sample (sampling_ratio != -1: the
 * reference's shipped 0.05, voxgraph_mapper.yaml:34) evaluates scattered points, where every 128-byte line a
 * neighbourhood touches is an HBM fetch of its own; on a context whose submaps hold apron bricks such a batch
 * therefore reads QUAD bricks, made on demand from the apron bricks of the submaps it reads (a device-side
 * copy at batch creation, kept with the submap: + 4.25 x the grid memory of those submaps only) -- the
 * all-points passes of the same context keep their apron bricks.  Results never depend on it.
 *   VGX_SAMPLING_BRICKS_QUAD (default)   as described
 *   VGX_SAMPLING_BRICKS_SAME             sampling batches read the bricks everything else reads */
#define VGX_SAMPLING_BRICKS_SAME 0
#define VGX_SAMPLING_BRICKS_QUAD 1
VGX_API int vgx_ctx_set_sampling_bricks(vgx_ctx ctx, int32_t mode);
/* which bricks a batch reads (VGX_BRICKS_APRON / VGX_BRICKS_QUAD); -1 for a NULL handle */
VGX_API int32_t vgx_reg_batch_brick_layout(vgx_reg_batch batch);
VGX_API int vgx_ctx_synchronize(vgx_ctx ctx);
/* hipEvent-based timer on the context's stream (used by bench.py so that
 * the kernel time is measured on the stream the kernels run on). */
VGX_API int vgx_ctx_timer_start(vgx_ctx ctx);
VGX_API int vgx_ctx_timer_stop(vgx_ctx ctx, float* elapsed_ms);

/* ---- submaps ----------------------------------------------------------- */
/* Stands in for a *finished* voxgraph::VoxgraphSubmap
 * (include/voxgraph/frontend/submap_collection/voxgraph_submap.h:16): the
 * TSDF and ESDF voxblox layers (same voxel size and voxels_per_side,
 * voxgraph_submap.cpp:26-29) plus its cached registration points.  A submap
 * is immutable once uploaded -- the invariant the reference relies on to
 * optimise while integrating (voxgraph_mapper.cpp:464-471).
 *
 * Layer layout = voxblox's: block b covers block_index[b] * (vps*voxel_size);
 * per-block arrays hold vps^3 voxels, linear index x + vps*(y + vps*z).
 * Any of the four voxel arrays may be NULL when that layer is not needed
 * (REG with use_esdf_distance reads only esdf_*; point extraction reads
 * tsdf_* and, with use_esdf_distance, esdf_distance).
 * Distances of valid voxels must be finite. */
VGX_API int vgx_submap_create(vgx_ctx ctx, int32_t submap_id, float voxel_size,
                              int32_t voxels_per_side, int32_t n_blocks,
                              const int32_t* block_index /* [n_blocks][3] */,
                              const float* tsdf_distance /* [n_blocks][vps^3] */,
                              const float* tsdf_weight,
                              const float* esdf_distance,
                              const uint8_t* esdf_observed,
                              vgx_submap* out);
/* Lifetimes follow the reference's ownership: a RegistrationCostFunction holds VoxgraphSubmap::ConstPtr to both submaps
 * (registration_cost_function.h), a ceres::Problem owns its cost functions.  So destroying a submap that cost functions
 * were built on is DEFERRED to the destruction of the last of them (the call returns VGX_OK at once and the handle must
 * not be used again by the caller); likewise vgx_reg_destroy on a cost function that batches still list is carried out
 * by the last vgx_reg_batch_destroy.  Contexts are not counted: destroy a context last. */
VGX_API int vgx_submap_destroy(vgx_submap submap);
VGX_API int32_t vgx_submap_id(vgx_submap submap);
VGX_API int32_t vgx_submap_num_blocks(vgx_submap submap);

/* VoxgraphSubmap::RegistrationPointType (voxgraph_submap.h:65) */
#define VGX_POINTS_ISOSURFACE 0
#define VGX_POINTS_VOXELS 1

/* vgx_submap_set_points flags */
#define VGX_POINTS_KEEP_ORDER 0u
/* Re-order the points along a Morton curve of their voxel coordinates so a
 * wavefront's 64 points gather from neighbouring cache lines.  Residual i
 * then refers to uploaded point order[i] (vgx_submap_point_order); Ceres is
 * indifferent to residual order, and every residual row still pairs with its
 * own Jacobian row. */
#define VGX_POINTS_SORT_MORTON 1u

/* Upload cached registration points (RegistrationPoint,
 * registration_point.h:6-12): position in the submap frame, distance, weight. */
VGX_API int vgx_submap_set_points(vgx_submap submap, int32_t point_type,
                                  int64_t n, const float* xyz /* [n][3] */,
                                  const float* distance, const float* weight,
                                  uint32_t flags);
/* Device-side VoxgraphSubmap::findRelevantVoxelIndices
 * (voxgraph_submap.cpp:144-201): stream-compacts every TSDF voxel with
 * weight > min_voxel_weight && |distance| < max_voxel_distance into the
 * VGX_POINTS_VOXELS set, in block order then (by default) linear-index order.
 * Needs tsdf_* (and esdf_distance if use_esdf_distance). */
VGX_API int vgx_submap_extract_voxel_points(vgx_submap submap,
                                            double min_voxel_weight,
                                            double max_voxel_distance,
                                            int32_t use_esdf_distance,
                                            int64_t* n_points_out);
/* Device-side VoxgraphSubmap::findIsosurfaceVertices (voxgraph_submap.cpp:203-243), the
 * VGX_POINTS_ISOSURFACE set the shipped "explicit_to_implicit" method registers with:
 * zero crossings of the TSDF along the edges of every fully observed dual cell (all 8
 * corner voxels with weight > min_weight: voxblox MeshIntegrator), merged per
 * 0.5-voxel cell (MeshLayer::getConnectedMesh(mesh, 0.5 * voxel_size)), each carrying the
 * trilinearly interpolated TSDF distance and weight (Interpolator::getVoxel).  Which of
 * several vertices in one cell survives is implementation-defined in the reference
 * (unordered_map order); here it is the first in (block, linear index, axis) order.
 * Needs the raw TSDF layer.  Also records the blocks that contain vertices
 * (isosurface_blocks_, voxgraph_submap.cpp:237-240). */
VGX_API int vgx_submap_extract_isosurface_points(vgx_submap submap, double min_voxel_weight,
                                                 int64_t* n_points_out);
VGX_API int64_t vgx_submap_num_points(vgx_submap submap, int32_t point_type);
/* order[i] = index (in upload / extraction order) of the point residual i uses */
VGX_API int vgx_submap_point_order(vgx_submap submap, int32_t point_type,
                                   int64_t* order /* [n] */);
/* Copy the (re-ordered) device point set back: xyz[n][3], distance, weight. */
VGX_API int vgx_submap_download_points(vgx_submap submap, int32_t point_type,
                                       float* xyz, float* distance,
                                       float* weight);
/* Copy the raw voxel layers back ([n_blocks][vps^3] each, any pointer may be
 * NULL) and the block index list ([n_blocks][3]). */
VGX_API int vgx_submap_download_layers(vgx_submap submap, float* tsdf_distance,
                                       float* tsdf_weight, float* esdf_distance,
                                       uint8_t* esdf_observed);
VGX_API int vgx_submap_block_index(vgx_submap submap, int32_t* block_index);
/* Device-side cblox::TsdfEsdfSubmap::generateEsdf() (voxgraph_submap.cpp:86), i.e.
 * voxblox::EsdfIntegrator::updateFromTsdfLayerBatch [recalled]: TSDF voxels with
 * weight >= min_weight become observed; |tsdf| < min_distance_m is copied and fixed;
 * the rest starts at sign * default_distance_m and is lowered by the quasi-Euclidean
 * 26-neighbour wavefront (steps 1, sqrt2, sqrt3 voxels) from voxels closer than
 * max_distance_m, never across a sign change.  voxblox runs a bucketed label-correcting
 * queue that ignores improvements below min_diff_m (1 mm); the GPU relaxes to the exact
 * fixed point of the same recurrence, so results agree to within a few min_diff_m
 * (min_diff_m and num_buckets are accepted for layout compatibility and ignored): worst 2.54 mm over 174.7 M fuzzed voxels.  One
 * discontinuity, outside voxblox's defaults only: when default_distance_m > max_distance_m, a voxel of the shell beyond the limit
 * whose neighbour sits within that slack of it can take another neighbour's longer path or keep the default.
 * Fills the ESDF raw layer from the resident TSDF layer and rebuilds the ESDF sampling
 * grid.  cfg == NULL uses voxblox's defaults.  sweeps (nullable) = global passes used. */
typedef struct vgx_esdf_config {
  float max_distance_m;     /* 2.0   */
  float min_distance_m;     /* 0.2   */
  float default_distance_m; /* 2.0   */
  float min_diff_m;         /* 0.001 (ignored) */
  float min_weight;         /* 1e-6  */
  int32_t num_buckets;      /* 20    (ignored) */
} vgx_esdf_config;
VGX_API void vgx_esdf_config_default(vgx_esdf_config* cfg);
VGX_API int vgx_submap_generate_esdf(vgx_submap submap, const vgx_esdf_config* cfg,
                                     int32_t* sweeps);
/* Drop the raw voxel layers after extraction (keeps the sampling grids). */
VGX_API int vgx_submap_release_raw_layers(vgx_submap submap);

/* ---- REG: one registration constraint ---------------------------------- */
/* RegistrationCostFunction::Config (registration_cost_function.h:17-41).
 * jacobian_evaluation_method is always analytic; visualize_* are ignored. */
typedef struct vgx_reg_config {
  int32_t registration_point_type; /* VGX_POINTS_*, default ISOSURFACE (h:20) */
  float sampling_ratio;            /* -1 disables sampling (h:28)             */
  double no_correspondence_cost;   /* default 0 (h:32)                        */
  int32_t use_esdf_distance;       /* default 1 (h:35)                        */
  uint32_t sampler_seed;           /* 0 (default): draw from the reference submap's own
                                    * sampler stream -- one default-seeded (5489)
                                    * std::mt19937 per point set, advanced by every cost
                                    * function sampling that set, as WeightedSampler does
                                    * (weighted_sampler.h:36-39).  != 0: a private engine
                                    * seeded with this value                              */
} vgx_reg_config;
VGX_API void vgx_reg_config_default(vgx_reg_config* cfg);

/* new RegistrationCostFunction(reference_submap, reading_submap, config)
 * (registration_cost_function.cpp:12-56; construction sites
 * registration_constraint.cpp:33-35, submap_registration_helper.cpp:44-46,
 * map_evaluation.cpp:143-144).  Cheap: no device allocation proportional to
 * the point count happens until the first evaluate. */
VGX_API int vgx_reg_create(vgx_ctx ctx, vgx_submap reference_submap,
                           vgx_submap reading_submap, const vgx_reg_config* cfg,
                           vgx_reg* out);
VGX_API int vgx_reg_destroy(vgx_reg reg);
/* num_residuals() (registration_cost_function.cpp:45-55) */
VGX_API int64_t vgx_reg_num_residuals(vgx_reg reg);

/* Drop-in for ceres::CostFunction::Evaluate
 * (registration_cost_function.h:47-48, .cpp:58-298):
 *   parameters[0] = ref_pose  {x,y,z,yaw} of the reference (first) submap,
 *   parameters[1] = read_pose {x,y,z,yaw} of the reading (second) submap,
 *   residuals[N]; jac_ref / jac_read are jacobians[0] / jacobians[1], each
 *   [N][4] row-major f64, either or both may be NULL.
 * All outputs are already scaled by N / sum(w) (.cpp:274-291).
 * Returns VGX_OK (true), VGX_EVALUATE_FALSE (false) or an error. */
VGX_API int vgx_reg_evaluate(vgx_reg reg, const double ref_pose[4],
                             const double read_pose[4], double* residuals,
                             double* jac_ref, double* jac_read);

/* Same evaluation, results left on the device as f32 (the 88 B/evaluation
 * form): DEVICE pointers residuals[N], jac_ref[N][4], jac_read[N][4]
 * (16-byte aligned), either Jacobian may be NULL. Asynchronous on the
 * context's stream. */
VGX_API int vgx_reg_evaluate_device_f32(vgx_reg reg, const double ref_pose[4],
                                        const double read_pose[4],
                                        void* d_residuals, void* d_jac_ref,
                                        void* d_jac_read);

/* ---- REG: all constraints of a pose graph in one launch ----------------- */
/* Mirrors one pass of the Ceres evaluator over every registration residual
 * block (pose_graph.cpp:101): constraint c links node_pair[c][0] (reference,
 * first submap) to node_pair[c][1] (reading, second submap).
 * global_index (nullable) gives each constraint's index in the whole graph
 * when the constraint list is sharded across processes; n_global is the
 * unsharded constraint count (values below n are read as n when global_index == NULL; a
 * shard that owns no constraint at all passes n = 0, global_index = NULL and the real n_global,
 * so that its assembled buffer has -- and zeroes -- the full size). */
VGX_API int vgx_reg_batch_create(vgx_ctx ctx, int32_t n, const vgx_reg* regs,
                                 const int32_t* node_pair /* [n][2] */,
                                 const int32_t* global_index /* [n] or NULL */,
                                 int32_t n_global, vgx_reg_batch* out);
VGX_API int vgx_reg_batch_destroy(vgx_reg_batch batch);
VGX_API int64_t vgx_reg_batch_num_residuals(vgx_reg_batch batch);
/* row_offset[c] = first row of constraint c in the stacked outputs; [n+1] */
VGX_API int vgx_reg_batch_row_offsets(vgx_reg_batch batch, int64_t* row_offset);

/* Materialising pass: residual + both Jacobians of every constraint as f32
 * into DEVICE arrays stacked by row_offset: residuals[R], jac_ref[R][4],
 * jac_read[R][4].  poses: host [n_nodes][4] f64.  status[c] (host, nullable)
 * receives VGX_OK / VGX_EVALUATE_FALSE per constraint.  Asynchronous.
 * Alignment: 16 bytes is required (float4 stores); more buys nothing (measured
 * with the placement held fixed: profiles/r05_points_placement.txt).  WHERE the
 * arrays lie physically does matter: vgx_reg_batch_choose_outputs below. */
VGX_API int vgx_reg_batch_evaluate_points(vgx_reg_batch batch,
                                          const double* poses, int32_t n_nodes,
                                          void* d_residuals, void* d_jac_ref,
                                          void* d_jac_read, int32_t* status);

/* The same pass in Ceres' own types: residuals[R] and jac_*[R][4] as F64, every value the f64 the reference's Evaluate writes
 * (registration_cost_function.cpp:163-166, 254-267, scaled as :274-291) -- what vgx_reg_evaluate returns for one constraint,
 * for the whole list in one launch and left on the device (72 B per row written instead of 36: SURVEY.md 8d's "124 B"
 * variant).  Same arguments and status as vgx_reg_batch_evaluate_points; jac_* 32-byte aligned (one row). */
VGX_API int vgx_reg_batch_evaluate_points_f64(vgx_reg_batch batch, const double* poses /* [n_nodes][4] */, int32_t n_nodes,
                                              void* d_residuals, void* d_jac_ref, void* d_jac_read, int32_t* status);

/* The same rows KEPT BY THE BATCH, and one constraint's slice of them fetched to the host -- SURVEY.md 8b's "vgx_reg_fetch(h,
 * residuals, jac_ref, jac_read) for the cached per-constraint slice": ONE launch per solver evaluation, and every residual block
 * still the reference's own N-residual block (f64, Ceres layout, every value the reference's: a ceres::LossFunction or a
 * covariance estimate sees exactly what RegistrationCostFunction::Evaluate would have given it), where the drop-in
 * vgx_reg_evaluate is one launch per block.  evaluate_rows_f64: want_jac_* = 0 leaves that block's Jacobians out (Ceres passes
 * jacobians == NULL, or the block is constant); the arrays are the batch's own (72 B per row, allocated at first use, with a
 * pinned host mirror filled by one copy per evaluation while they are below 2 GiB).  fetch_rows_f64(c, ...): any output may
 * be NULL; waits for the evaluation; VGX_ERR_INVALID without one, or for a Jacobian block it was not asked for.
 * voxgraph_amd/cpp/gpu_registration_rows.h is the ceres::EvaluationCallback built on the pair. */
VGX_API int vgx_reg_batch_evaluate_rows_f64(vgx_reg_batch batch, const double* poses /* [n_nodes][4] */, int32_t n_nodes,
                                            int32_t want_jac_ref, int32_t want_jac_read, int32_t* status);
VGX_API int vgx_reg_batch_fetch_rows_f64(vgx_reg_batch batch, int32_t constraint, double* residuals, double* jac_ref,
                                         double* jac_read);


/* The same pass into ONE output stream: an array of tile blocks, one block per rows_per_block (1024) consecutive residuals
 * of a constraint -- block = [residual f32 x 1024][jac_ref f32x4 x 1024][jac_read f32x4 x 1024], 36 KiB, every constraint
 * padded to whole blocks (its last block's unused rows are not written).  Residual k of constraint c is row k % 1024 of
 * block first_block[c] + k / 1024.  For consumers that live on the device and do not need three Ceres-shaped arrays: one
 * write front instead of three (profiles/r05_points_placement.txt says what three cost on an unlucky placement).
 * vgx_reg_batch_blocked_layout: bytes = size of the array; first_block (nullable) = [n + 1]. */
VGX_API int vgx_reg_batch_blocked_layout(vgx_reg_batch batch, int64_t* bytes, int32_t* rows_per_block,
                                         int64_t* first_block);
VGX_API int vgx_reg_batch_evaluate_points_blocked(vgx_reg_batch batch, const double* poses, int32_t n_nodes,
                                                  void* d_blocks, int32_t* status);

/* Placement by measurement.  WHERE the output arrays of the materialising pass lie in physical memory decides which of
 * two speeds the kernel runs at -- 200 x 256^3 submaps, 1176 constraints: 4.4-4.7 ms or 5.3-5.7 ms per launch; about half
 * of the sets of three an allocator hands out are slow ones, a matter of how the arrays lie relative to each other: every
 * array's own fill and read rate is the same (measured: profiles/r05_points_placement.txt).  The physical address is not the caller's to choose, but which of several
 * allocations to keep is: given n_candidates device pointers for each array (each large enough for the batch; d_jac_ref /
 * d_jac_read may be NULL as for the pass itself), this call times the batch's own launch -- one warm-up and `launches`
 * timed launches per trial -- first on whole sets (the k-th candidate of each array), then array by array against the
 * best combination so far, and returns in chosen[] the index to keep for residuals, jac_ref and jac_read.  ms_chosen
 * (nullable): ms per launch of that combination.  ms_trials (nullable, [n_candidates * 4]): every trial in order -- the n
 * sets, then jac_read's, jac_ref's and the residuals' candidates (-1 where a candidate needed no new trial).  The arrays
 * are overwritten.  Synchronous.  4 candidates and 3 launches cost 13 trials of 4 launches.
 * REFUSED (VGX_ERR_INVALID) for a batch with sampling constraints: every trial is launches + 1 evaluations of the batch,
 * and an evaluation of a sampling batch DRAWS -- it would leave every reference point set's std::mt19937 dozens of
 * evaluations further on and break "one evaluation of the batch = one Evaluate of every constraint in list order".  Where
 * the arrays lie does not depend on which points are drawn: choose with an all-points batch of the same sizes. */
VGX_API int vgx_reg_batch_choose_outputs(vgx_reg_batch batch, const double* poses, int32_t n_nodes,
                                         int32_t n_candidates, void* const* d_residuals, void* const* d_jac_ref,
                                         void* const* d_jac_read, int32_t launches, int32_t chosen[3],
                                         float* ms_chosen, float* ms_trials);

/* The same, with the arrays the library's own: allocates n_candidates (1..16) sets of the row arrays the batch needs (f32:
 * residuals [R], jac_* [R][4]; want_jac_* = 0 leaves that array out), chooses among them as vgx_reg_batch_choose_outputs does
 * (3 launches per trial), frees the unchosen ones and returns the three pointers to keep -- one call instead of "allocate a
 * few, choose, free the rest"; transient memory n_candidates x 36 B x R (fewer sets are tried when the device runs out).  A
 * sampling batch gets the first set untimed (its trial evaluations would advance the engines).  ms_chosen (nullable): ms per
 * launch on the arrays returned (0 when nothing was timed).  Release them with vgx_reg_batch_free_outputs while the batch
 * lives (it waits for the batch's stream first). */
VGX_API int vgx_reg_batch_alloc_outputs(vgx_reg_batch batch, const double* poses, int32_t n_nodes, int32_t n_candidates,
                                        int32_t want_jac_ref, int32_t want_jac_read, void** d_residuals, void** d_jac_ref,
                                        void** d_jac_read, float* ms_chosen);
VGX_API int vgx_reg_batch_free_outputs(vgx_reg_batch batch, void* d_residuals, void* d_jac_ref, void* d_jac_read);

/* Fused pass: no per-point outputs.  Per constraint c, 45 f64:
 *   [0]      sum r^2
 *   [1..8]   J^T r      over the stacked parameters [ref(4), read(4)]
 *   [9..44]  upper triangle of J^T J (row-major, 8x8)
 * d_normal: DEVICE pointer [n][45] (nullable); normal_host: host [n][45]
 * (nullable; implies a stream synchronisation).
 * Deterministic: fixed reduction tree, no atomics. */
VGX_API int vgx_reg_batch_evaluate_normal(vgx_reg_batch batch,
                                          const double* poses, int32_t n_nodes,
                                          void* d_normal, double* normal_host,
                                          int32_t* status);

/* Fused pass, COST ONLY: per constraint c one f64, sum r^2 -- element [0] of vgx_reg_batch_evaluate_normal's block at the
 * same poses, BIT FOR BIT (the same f32 operations in the same order, the same reduction tree; all-points batches -- a
 * sampling batch draws anew, see below).  The reference does no Jacobian work when Ceres passes `jacobians == nullptr`
 * (registration_cost_function.cpp:179), and Ceres' Levenberg-Marquardt evaluates every TRIAL step that way
 * (pose_graph.cpp:90-101): this is that evaluation for the whole constraint list -- no gradient, no pose-Jacobian
 * products, one running sum per lane instead of 21, nothing to compress afterwards.
 * d_cost: DEVICE pointer [n] f64 (nullable); cost_host: host [n] (nullable; implies a stream synchronisation).
 * With d_cost == NULL the batch's internal block array serves as scratch: the blocks a previous
 * vgx_reg_batch_evaluate_normal(d_normal == NULL) left there (what vgx_reg_batch_assemble / _scatter_normal read when
 * THEY are given NULL) are gone -- evaluate the blocks again before assembling.  Deterministic.
 * SAMPLING constraints (sampling_ratio != -1): EVERY evaluation of a batch -- points, normal or cost -- is one Evaluate of
 * every constraint in list order and DRAWS its points from the reference point sets' engines, as every call of the
 * reference's Evaluate does (registration_cost_function.cpp:113-122): a cost-only evaluation followed by a full one at the
 * same poses sees two different draws, exactly like Ceres on the reference.  A caller that serves a second request at
 * an unchanged point from what it cached (the C++ adapters, when Ceres says new_evaluation_point == false and what it
 * asks for is cached) does NOT redraw where the reference would: fewer draws, each still a legal one. */
VGX_API int vgx_reg_batch_evaluate_cost(vgx_reg_batch batch, const double* poses, int32_t n_nodes, void* d_cost,
                                        double* cost_host, int32_t* status);

/* Measurement aid for the fused pass: the number of residuals whose registration points the
 * fused kernel actually reads at these poses, i.e. the points of every 512-point chunk whose
 * bounding sphere can touch the reading submap's block box (chunks that cannot are skipped
 * without loading their points when no_correspondence_cost == 0); and (nullable) the number of
 * DISTINCT points behind them -- constraints that share a reference submap read the same points,
 * which the fused pass's launch order lets them share through one XCD's L2.  Synchronous. */
VGX_API int vgx_reg_batch_count_live(vgx_reg_batch batch, const double* poses, int32_t n_nodes,
                                     int64_t* live_residuals, int64_t* unique_points);

/* The same count per constraint (live_each[n], batch order): what a constraint costs at these poses is
 * roughly 36 B x its residuals + 45 B x its live residuals, which is a better weight for
 * vgx_lpt_shards than the residual count alone when much of the constraint list is culled. */
VGX_API int vgx_reg_batch_count_live_each(vgx_reg_batch batch, const double* poses, int32_t n_nodes,
                                          int64_t* live_each);

/* Measurement aid: which launch order the batch's tiles took at their first evaluation.  pass 0 =
 * fused (evaluate_normal), 1 = materialising (evaluate_points).  *grouped = 1: constraints that share
 * a reference submap run side by side on one XCD and read its points through that XCD's L2 once;
 * 0: plain constraint-major order (nothing worth sharing, or too much culled); -1: that pass has not
 * run yet.  Results never depend on the order. */
VGX_API int vgx_reg_batch_launch_order(vgx_reg_batch batch, int32_t pass, int32_t* grouped);

/* Scatter-adds this process's [n][45] blocks into the fused buffer every
 * process all-reduces once per solver evaluation, DEVICE f64:
 *   [0]                               sum of costs
 *   [1 .. 4*n_nodes]                  J^T r per node
 *   [.. + 16*n_nodes]                 diagonal 4x4 blocks of J^T J per node
 *   [.. + 16*n_global]                off-diagonal 4x4 block (ref rows, read
 *                                     cols) per constraint, written by exactly
 *                                     one process
 * The buffer is zeroed first when `zero_first` != 0. */
VGX_API int vgx_reg_batch_assemble(vgx_reg_batch batch, const void* d_normal,
                                   int32_t n_nodes, void* d_fused,
                                   int32_t zero_first);
VGX_API int64_t vgx_reg_fused_size(int32_t n_nodes, int32_t n_global);

/* Sharding-independent assembly (round 4).  vgx_reg_batch_assemble sums a node's entries over the shard's
 * own constraints, so a sum of per-shard fused buffers depends, in its last bits, on how the list was
 * sharded and on the order a collective adds in.  Exchanging the per-constraint BLOCKS instead makes the
 * result the single-GPU one bit for bit, for any number of shards:
 *   1. every shard:  vgx_reg_batch_evaluate_normal, then vgx_reg_batch_scatter_normal writes its [n][45]
 *      blocks (d_normal NULL: the batch's own) into rows global_index[c] of a DEVICE [n_global][45] f64
 *      array, zeroed first when zero_first != 0;
 *   2. ONE all-reduce(sum) of that array (n_global x 360 B: 423 KB for 1176 constraints), its words taken as
 *      64-bit INTEGERS: every row is written by exactly one shard and is all-zero-bits everywhere else, so
 *      the integer sum is that shard's bit pattern in ANY order (an f64 sum is exact too, but turns a -0.0
 *      into +0.0);
 *   3. vgx_reg_assembler_assemble builds the fused buffer (layout above, vgx_reg_fused_size(n_nodes, n))
 *      from the complete array in list order -- what a single vgx_reg_batch over the whole list computes.
 * The assembler holds the list's node structure (node_pair[n][2], the caller's constraint order) on `ctx`. */
VGX_API int vgx_reg_batch_scatter_normal(vgx_reg_batch batch, const void* d_normal, void* d_normal_all,
                                         int32_t zero_first);
VGX_API int vgx_reg_assembler_create(vgx_ctx ctx, int32_t n, const int32_t* node_pair /* [n][2] */,
                                     vgx_reg_assembler* out);
VGX_API int vgx_reg_assembler_assemble(vgx_reg_assembler assembler, const void* d_normal_all, int32_t n_nodes,
                                       void* d_fused);
VGX_API int vgx_reg_assembler_destroy(vgx_reg_assembler assembler);

/* Host-side helper for solvers that want residual blocks (Ceres), not normal equations:
 * turns one constraint's 45-number block N = [J r]^T [J r] into a 9-residual block with
 * the same normal equations, r_c[9] and J_c[9][8] row-major (J_c^T J_c = J^T J,
 * J_c^T r_c = J^T r, r_c^T r_c = r^T r), via a symmetric eigen-decomposition
 * N = V L V^T, [J_c | r_c] = sqrt(max(L, 0)) V^T.  Exact for least squares without a robust
 * loss, which is how the reference adds registration constraints
 * (registration_constraint.cpp:10, constraint.h:34).  Pure host arithmetic. */
VGX_API int vgx_reg_compress_normal(const double normal[45], double residuals9[9],
                                    double jacobian9x8[72]);

/* ---- REG on several GPUs of one process --------------------------------- */
/* voxgraph is one process (voxgraph_mapping_node.cpp:6-26); given the poses its registration
 * constraints are independent, so the list is pair-sharded over N contexts -- one per
 * GPU, every finished submap uploaded to each -- and each solver evaluation ends in ONE reduction.
 *
 * vgx_lpt_shards: greedy longest-processing-time partition; weight[c] = the constraint's residual
 * count (keep both directions of a mirrored pair together by giving them one entry).  Create each
 * constraint's cost function on the context of its shard, then hand the whole list over:
 * vgx_reg_multi_create sorts the constraints by owning context, builds one vgx_reg_batch per
 * context and starts one host thread per context.  Contexts may share a device (testing). */
VGX_API int vgx_lpt_shards(int32_t n, const int64_t* weight, int32_t n_shards, int32_t* shard_of /* [n] */);
/* The locality-aware alternative: the list cut into n_shards CONSECUTIVE runs of (nearly) equal weight
 * (constraint c goes to the shard its weight midpoint falls in).  voxgraph creates constraints in submap
 * order, i.e. along the trajectory, so a shard then touches the submaps of one stretch of the map and only
 * those need to be resident on its GPU -- a third of the map instead of three quarters at N = 8 on the bench
 * graphs, for a balance a few per cent behind LPT's (profiles/r04_shard_balance.json).
 * Results never depend on the placement (see "Sharding-independent assembly"). */
VGX_API int vgx_contiguous_shards(int32_t n, const int64_t* weight, int32_t n_shards, int32_t* shard_of /* [n] */);
VGX_API int vgx_reg_multi_create(int32_t n_ctx, const vgx_ctx* ctxs, int32_t n, const vgx_reg* regs,
                                 const int32_t* node_pair /* [n][2] */, vgx_reg_multi* out);
VGX_API int vgx_reg_multi_destroy(vgx_reg_multi multi);
VGX_API int32_t vgx_reg_multi_num_shards(vgx_reg_multi multi);
/* How the contexts' results meet on context 0.  Default VGX_REDUCE_PEER_SUM (the name is round 2's: since
 * round 4 nothing is summed): context 0 GATHERS every constraint's [45] block from the context that computed
 * it, through xGMI peer mappings, and assembles the fused buffer once, in list order.
 * VGX_REDUCE_RCCL: ONE ncclAllReduce(sum) per solver evaluation over xGMI (BASELINE north_star) of the
 * [n][45] array of blocks (words summed as int64), every context contributing its own rows and zero bits
 * elsewhere -- the contributing context's bit pattern in any order
 * -- then the same assembly (librccl.so is opened at run time, one communicator per context from
 * ncclCommInitAll, so every context needs its own device; VGX_ERR_UNSUPPORTED if RCCL cannot be opened or two
 * contexts share a device).  Either way the buffer is the one a single vgx_reg_batch over the whole list
 * assembles, BIT FOR BIT, whatever the number of contexts and the placement. */
#define VGX_REDUCE_PEER_SUM 0
#define VGX_REDUCE_RCCL 1
VGX_API int vgx_reg_multi_set_reduction(vgx_reg_multi multi, int32_t reduction);
VGX_API int vgx_reg_multi_shard_of(vgx_reg_multi multi, int32_t* shard_of /* [n] */);
/* One solver evaluation: every context runs vgx_reg_batch_evaluate_normal on its share concurrently (own
 * thread, own stream); the per-constraint blocks meet on context 0 (see vgx_reg_multi_set_reduction), which
 * assembles the fused buffer of vgx_reg_batch_assemble's layout (vgx_reg_fused_size(n_nodes, n) doubles) in
 * list order and returns it to the host.  Bitwise reproducible and independent of the sharding.
 * status: [n], nullable.  A context whose evaluation fails fails the call with that context's message. */
VGX_API int vgx_reg_multi_evaluate_fused(vgx_reg_multi multi, const double* poses, int32_t n_nodes,
                                         double* fused_host, int32_t* status);
/* The same pass for solvers that want residual blocks (Ceres through vgx_reg_compress_normal): the
 * [n][45] normal blocks in the caller's constraint order; needs no reduction at all. */
VGX_API int vgx_reg_multi_evaluate_normal(vgx_reg_multi multi, const double* poses, int32_t n_nodes,
                                          double* normal_host /* [n][45] */, int32_t* status);
/* ... and its cost-only form (vgx_reg_batch_evaluate_cost on every context's share): cost_host[c] = element [0] of the
 * block above, bit for bit, in the caller's constraint order. */
VGX_API int vgx_reg_multi_evaluate_cost(vgx_reg_multi multi, const double* poses, int32_t n_nodes,
                                        double* cost_host /* [n] */, int32_t* status);

/* ---- overlap detection (callers' side of REG) -------------------------- */
/* VoxgraphSubmap::getSubmapFrameSurfaceObb (voxgraph_submap.cpp:280-321): box around the
 * kVoxels registration voxels (centres -+ half a voxel), submap frame.  Needs the
 * VGX_POINTS_VOXELS set.  Returns VGX_ERR_INVALID when the set is empty. */
VGX_API int vgx_submap_surface_obb(vgx_submap submap, float min_xyz[3], float max_xyz[3]);
/* VoxgraphSubmap::getMissionFrameSurfaceAabb = BoundingBox::getAabbFromObbAndPose
 * (bounding_box.cpp:28-42) for a 4-DoF pose {x,y,z,yaw}. */
VGX_API int vgx_submap_mission_surface_aabb(vgx_submap submap, const double pose[4],
                                            float min_xyz[3], float max_xyz[3]);
/* PoseGraphInterface::updateOverlappingSubmapList (pose_graph_interface.cpp:109-147) over
 * VoxgraphSubmap::overlapsWith (voxgraph_submap.cpp:245-278): all pairs i < j whose
 * mission-frame surface AABBs intersect and for which at least one isosurface block
 * centre of submap i, carried into submap j's frame, falls in an allocated block of j.
 * The AABB stage runs on the host, the block stage is one launch over the surviving pairs.
 * Needs both point sets on every submap.  pairs: [max_pairs][2] indices into `submaps`,
 * in the reference's loop order. */
VGX_API int vgx_find_overlapping_pairs(vgx_ctx ctx, int32_t n, const vgx_submap* submaps,
                                       const double* poses /* [n][4] */, int32_t* pairs,
                                       int32_t max_pairs, int32_t* n_pairs);

/* ---- TSDF: voxblox::FastTsdfIntegrator ---------------------------------- */
/* Replaces the integrator voxgraph constructs and drives at
 * voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83
 * (`new voxblox::FastTsdfIntegrator(config, layer)`, `setLayer`,
 * `integratePointCloud(T_submap_sensor, pointcloud, colors)`).  The arithmetic is
 * voxblox's (not vendored in the reference; restated in oracle/tsdf_oracle.c). */
typedef struct vgx_tsdf_layer_s* vgx_tsdf_layer;           /* voxblox::Layer<TsdfVoxel> */
typedef struct vgx_tsdf_integrator_s* vgx_tsdf_integrator; /* voxblox::FastTsdfIntegrator */

/* voxblox::TsdfIntegratorBase::Config (voxblox defaults; voxgraph_mapper.yaml:21-28
 * overrides truncation 0.60, max ray 16 m, const weight, drop-off, sparsity
 * compensation 20).  integrator_threads / max_integration_time_s have no meaning on the GPU: every ray is
 * its own thread; integration_order_mode (voxgraph_mapper.yaml:29) is `integration_order` below. */
#define VGX_TSDF_ORDER_MIXED 0  /* integration_order_mode "mixed" (voxblox's and voxgraph's default) */
#define VGX_TSDF_ORDER_SORTED 1 /* "sorted": points visited by ascending squared norm of point_C    */
/* ABI note: the struct grew at its END in rounds 3 (deterministic) and 4 (integration_order), and every field is
 * validated (an integration_order that is neither value is refused with VGX_ERR_INVALID).  Fill it with
 * vgx_tsdf_config_default() -- or zero it -- before setting fields; a caller compiled against an older header must be
 * rebuilt (the library reads sizeof(vgx_tsdf_config) bytes). */
typedef struct vgx_tsdf_config {
  float default_truncation_distance;    /* 0.1   */
  float max_weight;                     /* 10000 */
  int32_t voxel_carving_enabled;        /* 1     */
  float min_ray_length_m;               /* 0.1   */
  float max_ray_length_m;               /* 5.0   */
  int32_t use_const_weight;             /* 0     */
  int32_t allow_clear;                  /* 1     */
  int32_t use_weight_dropoff;           /* 1     */
  int32_t use_sparsity_compensation_factor; /* 0 */
  float sparsity_compensation_factor;   /* 1.0   */
  float start_voxel_subsampling_factor; /* 2.0   */
  int32_t max_consecutive_ray_collisions; /* 2   */
  int32_t clear_checks_every_n_frames;  /* 1     */
  int32_t enable_anti_grazing;          /* 0     (merged integrator only) */
  /* 0: every ray is its own thread and rays race on the approximate sets and the voxels exactly as
   * voxblox's worker threads do (a legal order, different from run to run on dense scans).
   * 1: REPRODUCIBLE mode -- the scan is integrated as voxblox does with integrator_threads = 1 and
   * integration_order_mode "mixed": the same rays are cast, stop at the same voxel and update
   * every voxel in the same order, so the same scans give the same layer bit for bit, run after
   * run (and the layer oracle/tsdf_oracle.c computes).  Slower (several sorts and a fixed-point
   * iteration per scan instead of one kernel); meant for regression tests and reproducible maps. */
  int32_t deterministic;                /* 0     */
  /* voxblox's integration_order_mode: VGX_TSDF_ORDER_MIXED (1024-point groups visited round-robin) or
   * VGX_TSDF_ORDER_SORTED (ascending f32 squaredNorm() of the sensor-frame point; voxblox sorts with the
   * unstable std::sort, so the order of points at EQUAL range is unspecified there: here, and in the
   * oracle, equal ranges are visited by ascending point index).  The visiting order is what the
   * reproducible mode reproduces and what the merged integrator merges a group's points in; the racing
   * fast integrator (deterministic = 0) has no visiting order and ignores it. */
  int32_t integration_order;            /* 0     */
} vgx_tsdf_config;
VGX_API void vgx_tsdf_config_default(vgx_tsdf_config* cfg);

/* An active (unfinished) submap's TSDF layer, resident on the GPU between scans
 * (SURVEY.md 3.1).  Unbounded, like voxblox::Layer<TsdfVoxel>: blocks (12 B per voxel) are allocated
 * on demand wherever rays go.  lut_min / lut_dim (block coordinates; both may be NULL) and max_blocks
 * (<= 0: a default) are only an initial reservation: before each scan the integrator enlarges the
 * block table and the block pool, on the stream, to hold everything that scan can reach
 * (origin +- max_ray_length + truncation), so no update is ever dropped for lack of room.  If the
 * GPU itself runs out of memory the integrate call fails with VGX_ERR_NOMEM. */
VGX_API int vgx_tsdf_layer_create(vgx_ctx ctx, float voxel_size, int32_t voxels_per_side,
                                  const int32_t lut_min[3], const int32_t lut_dim[3],
                                  int32_t max_blocks, vgx_tsdf_layer* out);
VGX_API int vgx_tsdf_layer_destroy(vgx_tsdf_layer layer);
/* allocated blocks; *dropped_updates = voxel updates that found no block (always 0 unless an
 * allocation failed, in which case the next integrate call reports VGX_ERR_NOMEM).  Waits for the
 * scans in flight. */
VGX_API int vgx_tsdf_layer_stats(vgx_tsdf_layer layer, int32_t* n_blocks,
                                 int64_t* dropped_updates);
/* Optional: make room NOW for scans taken from around `origin` (layer frame) that reach up to
 * reach_m metres (max_ray_length + truncation), so that the first scans there do not pay for the
 * enlargement.  Scans reserve for themselves anyway. */
VGX_API int vgx_tsdf_layer_reserve(vgx_tsdf_layer layer, const float origin[3], float reach_m);
/* Acknowledges dropped updates (the GPU ran out of memory during an earlier scan: integrate calls keep
 * failing with VGX_ERR_NOMEM until then) and resets the counter, after the caller has made room or
 * decided to live with the hole.  Waits for the scans in flight. */
VGX_API int vgx_tsdf_layer_clear_dropped(vgx_tsdf_layer layer);
/* how often the layer has enlarged its block table or pool so far (diagnostics) */
VGX_API int64_t vgx_tsdf_layer_growths(vgx_tsdf_layer layer);
/* block_index[n][3], distance / weight [n][vps^3], rgba [n][vps^3][4]; any may be NULL */
VGX_API int vgx_tsdf_layer_download(vgx_tsdf_layer layer, int32_t* block_index,
                                    float* distance, float* weight, uint8_t* rgba);
/* Replaces the layer's contents with host blocks in the same layout (rgba may be NULL): hands a
 * voxblox::Layer<TsdfVoxel> that already holds data over to the GPU integrator. */
VGX_API int vgx_tsdf_layer_upload(vgx_tsdf_layer layer, int32_t n_blocks, const int32_t* block_index,
                                  const float* distance, const float* weight, const uint8_t* rgba);

VGX_API int vgx_tsdf_integrator_create(vgx_ctx ctx, const vgx_tsdf_config* cfg,
                                       vgx_tsdf_layer layer, vgx_tsdf_integrator* out);
VGX_API int vgx_tsdf_integrator_destroy(vgx_tsdf_integrator integrator);
/* FastTsdfIntegrator::setLayer (pointcloud_integrator.cpp:77) */
VGX_API int vgx_tsdf_integrator_set_layer(vgx_tsdf_integrator integrator, vgx_tsdf_layer layer);
/* Optional hint: the scans to come are ORGANISED clouds of `width` points per row (sensor_msgs/PointCloud2.width, which
 * voxgraph's callback receives -- pointcloud_integrator.cpp:23 -- and voxblox's flat Pointcloud drops; 0 = unorganised, the
 * default).  The racing integrator then gives a workgroup a 16 x 16 tile of beams instead of 256 consecutive ones: the
 * beams that end in one voxel are neighbours in both directions, so most of a voxel's updates meet inside one workgroup.
 * Only the assignment of points to workgroups depends on it -- which rays are cast and what they write is one of voxblox's
 * legal orders either way; a scan whose length is not a multiple of `width` is treated as unorganised. */
VGX_API int vgx_tsdf_integrator_set_cloud_width(vgx_tsdf_integrator integrator, int32_t width);
/* integratePointCloud(T_G_C, points_C, colors, freespace_points)
 * (pointcloud_integrator.cpp:83).  T_G_C = {qw,qx,qy,qz, tx,ty,tz} f32
 * (voxblox::Transformation); points_C [n][3] sensor frame; rgba [n][4] or NULL.
 * Host pointers (pageable is fine: the arrays are read before the call returns, through pinned staging buffers of the
 * integrator's own, and are the caller's again afterwards).  n_updates == NULL -- what voxblox's void call corresponds to:
 * the call returns with the scan QUEUED on the context's TSDF stream; the layer is the device's, and every reader of it
 * (vgx_tsdf_layer_download / _stats, vgx_submap_from_tsdf_layer, the next scan) is ordered behind the scan on that stream;
 * vgx_ctx_synchronize_tsdf waits for it explicitly.  n_updates != NULL: a COUNTED scan -- the call waits for it and
 * returns the number of voxel updates performed (a slower kernel instantiation: diagnostics, tests). */
VGX_API int vgx_tsdf_integrate(vgx_tsdf_integrator integrator, const float T_G_C[7],
                               const float* points_C, const uint8_t* rgba, int64_t n,
                               int32_t freespace_points, int64_t* n_updates);
/* Same with DEVICE pointers (scan already resident in HBM); asynchronous unless
 * n_updates != NULL. */
VGX_API int vgx_tsdf_integrate_device(vgx_tsdf_integrator integrator, const float T_G_C[7],
                                      const void* d_points_C, const void* d_rgba, int64_t n,
                                      int32_t freespace_points, int64_t* n_updates);

/* voxblox::MergedTsdfIntegrator::integratePointCloud (north_star names it; voxgraph itself
 * constructs the Fast integrator, pointcloud_integrator.h:30) on the same integrator object -- config
 * and layer; the approximate sets of the fast integrator are not involved: the valid points are
 * grouped by the voxel their end point falls in (clearing rays separately), every group is merged
 * into one weighted-mean point in the reference's visiting order and ONE ray is cast for it through
 * all its voxels with the summed weight; with enable_anti_grazing a ray skips voxels that are the end
 * voxel of another group.  Same argument conventions as vgx_tsdf_integrate[_device].
 * Inputs the reference leaves undefined (all integrators): a coordinate that is NaN indexes voxel 0 on its axis,
 * one beyond +-2^31 voxels saturates (getGridIndexFromPoint's cast, defined the same way in oracle/tsdf_oracle.c);
 * such points pass isPointValid as in the reference and, where they are clearing rays (allow_clear / freespace
 * scans), carve along their direction up to max_ray_length_m like any other return beyond the range.
 * VGX_ERR_UNSUPPORTED: a ray of more than 2^24 voxel steps; a voxel beyond +-2^20 voxels of the layer origin on a
 * ray's walk (reproducible mode and merged integrator). */
VGX_API int vgx_tsdf_integrate_merged(vgx_tsdf_integrator integrator, const float T_G_C[7],
                                      const float* points_C, const uint8_t* rgba, int64_t n,
                                      int32_t freespace_points, int64_t* n_updates);
VGX_API int vgx_tsdf_integrate_merged_device(vgx_tsdf_integrator integrator, const float T_G_C[7],
                                             const void* d_points_C, const void* d_rgba, int64_t n,
                                             int32_t freespace_points, int64_t* n_updates);

/* finishSubmap() hand-off without a host round trip: turns the active layer's blocks
 * into a (not yet finished) submap holding the raw TSDF layer and its TSDF sampling
 * grid; follow with vgx_submap_generate_esdf and vgx_submap_extract_voxel_points.  The
 * layer itself is left untouched (the mapper moves on to a new active submap). */
VGX_API int vgx_submap_from_tsdf_layer(vgx_ctx ctx, vgx_tsdf_layer layer, int32_t submap_id,
                                       vgx_submap* out);

/* ---------------------------------------------------------------------------
 * Saved maps: cblox submap-collection files and voxblox layer files.
 *
 * voxgraph saves its map with SubmapCollection::saveToFile (voxgraph_mapper.cpp:412-417)
 * and reads collections back with cblox::io::LoadSubmapCollection<VoxgraphSubmap>
 * (registration_test_bench.cpp:173-175 -> VoxgraphSubmap::LoadFromStream,
 * voxgraph_submap.cpp:398-415).  The container is a sequence of length-prefixed
 * protobuf messages (varint32 size, then the message):
 *   collection : SubmapCollectionProto, then per submap a SubmapProto header followed by
 *                its TSDF BlockProtos and (TsdfEsdfSubmap) its ESDF BlockProtos
 *   layer file : varint32 message count, LayerProto, BlockProtos   (voxblox::io::SaveLayer)
 * This is a hand-written wire-format reader/writer (protobuf is not a dependency).
 * [recalled] The message schemas live in un-vendored voxblox / cblox and could not be
 * checked against a real file in this environment; they are isolated in one table
 * (voxgraph_amd/csrc/vgx_mapfile_schema.h).  Unknown fields are skipped, packed and
 * unpacked repeated encodings are both accepted.
 * Host-only (no device needed) except vgx_map_file_load_submap.
 * ------------------------------------------------------------------------- */
typedef struct vgx_map_file_s* vgx_map_file;
#define VGX_FILE_CBLOX_COLLECTION 0
#define VGX_FILE_VOXBLOX_LAYER 1
typedef struct vgx_map_file_submap_info {
  int64_t id;              /* SubmapProto.id (0 for a layer file)                       */
  double T_M_S[7];         /* submap pose {qw,qx,qy,qz, tx,ty,tz} (identity for a layer) */
  double voxel_size;
  int32_t voxels_per_side;
  int32_t n_tsdf_blocks;
  int32_t n_esdf_blocks;   /* 0 when the file holds no ESDF for this submap              */
  int32_t layer_is_esdf;   /* layer files only: LayerProto.type == "esdf"                */
} vgx_map_file_submap_info;
/* Indexes the file (headers and message offsets; voxel payloads are decoded on read). */
VGX_API int vgx_map_file_open(const char* path, int32_t format, vgx_map_file* out);
VGX_API int vgx_map_file_close(vgx_map_file file);
/* Last error of this file handle (or of the last failed open when file == NULL). */
VGX_API const char* vgx_map_file_last_error(vgx_map_file file);
VGX_API int32_t vgx_map_file_num_submaps(vgx_map_file file);
VGX_API int vgx_map_file_get_submap_info(vgx_map_file file, int32_t index, vgx_map_file_submap_info* info);
/* Decodes submap `index` into caller arrays (any may be NULL): TSDF blocks in file order
 * (block_index [n_tsdf][3], distance / weight / rgba [n_tsdf][vps^3]), and the ESDF values
 * of the SAME blocks (esdf_distance, esdf_observed [n_tsdf][vps^3]; blocks without an ESDF
 * counterpart read distance 0 / observed 0) -- the layout vgx_submap_create takes. */
VGX_API int vgx_map_file_read_submap(vgx_map_file file, int32_t index, int32_t* block_index,
                                     float* tsdf_distance, float* tsdf_weight, uint8_t* tsdf_rgba,
                                     float* esdf_distance, uint8_t* esdf_observed);
/* read + vgx_submap_create: the device-side counterpart of VoxgraphSubmap::LoadFromStream.
 * The submap is NOT finished: follow with vgx_submap_generate_esdf (if the file has no ESDF)
 * and the extract calls, as finishSubmap() does after loading. */
VGX_API int vgx_map_file_load_submap(vgx_ctx ctx, vgx_map_file file, int32_t index, vgx_submap* out);
/* Writer (round trips, and saving maps built on the device).  One entry per submap;
 * esdf_* may be NULL (TSDF-only submap). */
typedef struct vgx_map_file_submap_data {
  int64_t id;
  double T_M_S[7];
  int32_t n_blocks;
  const int32_t* block_index;     /* [n][3] */
  const float* tsdf_distance;     /* [n][vps^3] */
  const float* tsdf_weight;
  const uint8_t* tsdf_rgba;       /* [n][vps^3][4] or NULL */
  const float* esdf_distance;     /* or NULL */
  const uint8_t* esdf_observed;
} vgx_map_file_submap_data;
VGX_API int vgx_map_file_write(const char* path, int32_t format, double voxel_size,
                               int32_t voxels_per_side, int32_t n_submaps,
                               const vgx_map_file_submap_data* submaps);

#ifdef __cplusplus
}
#endif
#endif /* VOXGRAPH_AMD_H_ */
