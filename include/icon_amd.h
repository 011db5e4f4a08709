/*
 * icon_amd.h - C ABI of the MI355X-native implicit-surface query engine for ICON.
 *
 * The reference (YuliangXiu/ICON) has no FFI / plugin registry on this path: the seam is two
 * Python call signatures plus module state.  This header is the boundary a
 * maintainer binds with ctypes; every entry point cites the reference
 * code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all `d_*` pointers are DEVICE pointers (HIP), all `h_*` pointers are HOST pointers;
 *   - float data is float32, indices int64 (as in the reference) unless stated;
 *   - every call that launches work enqueues it on `stream` (a hipStream_t passed as void*) and
 *     returns without synchronising, except where the comment says "synchronises";
 *   - return value: 0 on success, non-zero error code otherwise; icon_last_error() returns a
 *     thread-local human-readable message.  No CPU fallback exists: without a GPU every compute
 *     entry point fails with ICON_ERR_HIP.
 */
#ifndef ICON_AMD_H
#define ICON_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICON_AMD_VERSION 100          /* 0.1.0 */

enum {
    ICON_OK = 0,
    ICON_ERR_ARG = 1,                 /* bad argument (null, shape, range)         */
    ICON_ERR_HIP = 2,                 /* HIP runtime error (message has details)   */
    ICON_ERR_UNSUPPORTED = 3,         /* shape outside what the kernels are built for */
    ICON_ERR_STATE = 4                /* call order violated (e.g. finish before features) */
};

/* prior_type of lib/net/HGPIFuNet.py:63 */
enum { ICON_PRIOR_ICON = 0, ICON_PRIOR_PAMIR = 1, ICON_PRIOR_PIFU = 2 };

/* How clipped ("outlier") points get their cmap channels, lib/net/HGPIFuNet.py:298-305.
 * REFERENCE reproduces the reference bit for bit: smpl_sdf[outlier].repeat(1,1,3) tiles the
 * list of outlier signs of the whole call, so the j-th outlier's channel k receives the sign of
 * outlier (3j+k) mod K.  LOCAL is the per-point rule (cmap := own sign). */
enum { ICON_CMAP_REFERENCE = 0, ICON_CMAP_LOCAL = 1 };

/* MLP arithmetic.  F32: v_mfma_f32_32x32x2_f32, bit-for-bit an f32 fma chain.  F16X3 (the DEFAULT
 * of the host layer): every product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16
 * with f32 accumulation (22-bit operands; ~1e-6 of the f32 result, 5x the rate) - f32-class.
 * Its range is f16's: an input or hidden activation beyond 65504 (a thousand times anything a body mesh produces; a mesh
 * squashed flat reaches it) would make the result NaN - the kernels flag such points and redo them in plain f32
 * (k_rescue_*, mlp_plain_device.h), so F16X3 returns a number wherever F32 does.
 * (Rounds 1-3 carried a third, explicitly NOT f32-equivalent mode - f16 main term + block-scaled fp6 cross terms, value 2 -
 *  that only ever ran the materialising pipeline; removed in round 4: it could never be the headline.) */
enum { ICON_PRECISION_F32 = 0, ICON_PRECISION_F16X3 = 1 };

/* nearest-triangle search strategy (both give identical results; BRUTE is the validation path) */
enum { ICON_SEARCH_BVH = 0, ICON_SEARCH_BRUTE = 1 };

/* Bad data in device buffers never faults: query points whose (projected) coordinates are NaN / Inf / beyond +-64 are
 * evaluated at +-64 - far outside the cube, occupancy 0 (the reference returns 0 * NaN there), an ordinary "outside" entry
 * of the call's outlier list; faces / tetrahedra that name a vertex that does not exist are skipped (icon_visibility,
 * icon_semantic_voxelize) or reported (icon_mesh_components, icon_mesh_create, which also refuses non-finite vertices). */

typedef struct icon_mesh icon_mesh_t;   /* per-image SMPL body: triangles, normals, BVH, ray bins   */
typedef struct icon_feat icon_feat_t;   /* per-image feature planes (and PaMIR volume), repacked    */
typedef struct icon_mlp  icon_mlp_t;    /* if_regressor weights, BatchNorm folded, MFMA operand order */
typedef struct icon_work icon_work_t;   /* reusable device workspace                                  */

const char *icon_last_error(void);
int icon_version(void);
/* number of HIP devices visible; 0 (not an error) when there is none */
int icon_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Mesh: the per-image half of cal_sdf_batch (lib/dataset/mesh_util.py:357-372):
 *   Meshes(verts, faces).verts_normals_padded()            :367  (pytorch3d leaf)
 *   face_vertices(verts|normals|cmaps|vis, faces)          :369-372, lib/common/render_utils.py:149-163
 * plus the acceleration structures our kernels need (BVH over triangles, (y,z) ray bins).
 * The reference redoes this work on every query() call; here it is once per image.
 * d_verts [V,3] f32, d_faces [F,3] i64, d_cmap [V,3] f32, d_vis [V] f32 (the [1,V,1] tensor).
 * Built ON THE DEVICE (round 4), as the reference's own prologue runs on the device (mesh_util.py:367-372): kernels on
 * `stream`, no copy of the mesh to the host, no synchronisation - the handle is usable by later calls on the same
 * stream as soon as the function returns.  icon_mesh_create allocates the mesh's device memory itself (one hipMalloc;
 * icon_mesh_destroy frees it, which waits for the device); icon_mesh_create_arena builds into memory the CALLER owns
 * (>= icon_mesh_arena_bytes(V, F) bytes, 256-byte aligned, alive and untouched until the last call that uses the
 * handle has run) - with a caching allocator behind it nothing is allocated per image and nothing ever waits.
 * What the build finds wrong with its input (a face naming a missing vertex, a non-finite coordinate) cannot be
 * returned by a call that does not wait: such input is made harmless (vertex 0 / coordinate 0 - no fault) and
 * reported by icon_mesh_status.
 * ------------------------------------------------------------------------------------------- */
int icon_mesh_create(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F,
                     const float *d_cmap, const float *d_vis, void *stream, icon_mesh_t **out);
int icon_mesh_arena_bytes(int64_t V, int64_t F, int64_t *bytes);
int icon_mesh_create_arena(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F,
                           const float *d_cmap, const float *d_vis, void *d_arena, int64_t arena_bytes,
                           void *stream, icon_mesh_t **out);
int icon_mesh_destroy(icon_mesh_t *mesh);
/* Input check of the build.  wait != 0: blocks until the build has run; wait == 0: answers from a pinned host word
 * the build's last copy writes - *bits = -1 while it has not run yet (returns ICON_OK), otherwise a bit set:
 * 1 bad face index, 2 bad vertex coordinate (both: ICON_ERR_ARG, the checks the reference's tensors would fail in
 * kaolin), 4 the ray-bin lists did not fit (not an error: inside tests count crossings over all triangles), 8 an
 * internal check of the build failed (ICON_ERR_STATE). */
int icon_mesh_status(const icon_mesh_t *mesh, int wait, int *bits);
/* copy the area-weighted unit vertex normals [V,3] to a device buffer (for tests) */
int icon_mesh_vertex_normals(const icon_mesh_t *mesh, float *d_out, void *stream);
/* BVH statistics (waits for the build): out[0]=nodes, out[1]=max depth, out[2]=ray-bin entries, out[3]=max bin length,
 * out[4]=leaves, out[5]=triangle slots (= F: a slot is a position in the BVH order) */
int icon_mesh_stats(const icon_mesh_t *mesh, int64_t out[6]);

/* ---------------------------------------------------------------------------------------------
 * cal_sdf_batch, per-point half (lib/dataset/mesh_util.py:374-396):
 *   point_to_mesh_distance (kaolin leaf) :374, gathers :375-382,
 *   barycentric_coordinates_of_projection :319-354,383, weighted sums :385-391,
 *   check_sign (kaolin leaf) :393.
 * d_points [N,3]; outputs d_sdf [N], d_norm [N,3], d_cmap [N,3], d_vis [N] (0/1 as float),
 * optional d_face [N] int64 (nearest face index) and d_inside [N] uint8.  Unclipped, as
 * cal_sdf_batch returns them.  The points may come in any order: batches below 98,304 points are
 * searched one wavefront per point, larger ones as 64-point packets over a Morton ordering built on
 * the device (that path synchronises the stream once to free its scratch).
 * ------------------------------------------------------------------------------------------- */
int icon_sdf_query(const icon_mesh_t *mesh, const float *d_points, int64_t N,
                   float *d_sdf, float *d_norm, float *d_cmap, float *d_vis,
                   int64_t *d_face, uint8_t *d_inside, int search, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Feature planes: what index() samples (lib/net/geometry.py:21-43):
 *   d_planes [C,H,W] f32 - features[-1][0] from HGPIFuNet.filter (lib/net/HGPIFuNet.py:204-266);
 *   d_vol [Cv,D,H,W] f32 or NULL - VolumeEncoder output (PaMIR, lib/net/HGPIFuNet.py:321-325).
 * n_select = 2 for the icon prior (front/back halves chosen by feat_select,
 * lib/dataset/mesh_util.py:266-277), 1 otherwise - including the icon prior with smpl_feats that lack 'vis'
 * (configs/train/icon-mvp.yaml:40): every channel is then an MLP input (lib/net/HGPIFuNet.py:345-346).  Repacks to
 * channel-last so one bilinear tap is one to four 16-byte loads.
 * ------------------------------------------------------------------------------------------- */
int icon_feat_create(const float *d_planes, int C, int H, int W, int n_select,
                     const float *d_vol, int Cv, int Dv, int Hv, int Wv,
                     void *stream, icon_feat_t **out);
int icon_feat_destroy(icon_feat_t *feat);
/* icon prior: cfg.net.smpl_feats (lib/net/HGPIFuNet.py:301-309).  The MLP input is [img | sdf | cmap if has_cmap | norm if
 * has_norm]; 'sdf' is always present; 'vis' (it selects the feature half, :334-336) is n_select = 2 above, its absence
 * n_select = 1.  At most 15 input channels in all, or the query entry points return ICON_ERR_UNSUPPORTED.
 * Default: both (every configs/ *.yaml). */
int icon_feat_set_smpl_feats(icon_feat_t *feat, int has_cmap, int has_norm);

/* ---------------------------------------------------------------------------------------------
 * MLP (lib/net/MLP.py:8-72 as built by lib/net/HGPIFuNet.py:128-133): Conv1d(k=1) stack with
 * BatchNorm1d (eval) + LeakyReLU(0.01) after all but the last layer, raw input re-concatenated
 * (after the activations, MLP.py:62) before the layers flagged in is_res, no last_op (test mode).
 * HOST pointers, reference state_dict layout:
 *   h_W[l] [cout[l], cin[l]], h_b[l] [cout[l]],
 *   h_bn_gamma/beta/mean/var[l] [cout[l]] for l < n_layers-1 (NULL arrays => no norm).
 * Supported: n_layers == 4, cin[0] <= 15, cout = {512,256,128,1}, is_res = {0,0,1,1}
 * (every config in configs/ *.yaml).  BatchNorm is folded in float64 on the host.
 * ------------------------------------------------------------------------------------------- */
int icon_mlp_create(int n_layers, const int *cin, const int *cout, const int *is_res,
                    const float *const *h_W, const float *const *h_b,
                    const float *const *h_bn_gamma, const float *const *h_bn_beta,
                    const float *const *h_bn_mean, const float *const *h_bn_var,
                    float bn_eps, void *stream, icon_mlp_t **out);
int icon_mlp_destroy(icon_mlp_t *mlp);
/* MLP.last_op (lib/net/MLP.py:68-70): HGPIFuNet builds the regressor with last_op = nn.Sigmoid() unless cfg.test_mode
 * (lib/net/HGPIFuNet.py:133; every configs/ *.yaml sets test_mode: True, training / validation runs do not).  Applied to the
 * network output before the in_cube mask, in every precision. */
enum { ICON_LASTOP_NONE = 0, ICON_LASTOP_SIGMOID = 1 };
int icon_mlp_set_last_op(icon_mlp_t *mlp, int last_op);
/* MLP.forward on point-major input rows: d_x [N,16] f32 (channels cin[0]..15 ignored),
 * d_out [N].  precision: ICON_PRECISION_*. */
int icon_mlp_forward(const icon_mlp_t *mlp, const float *d_x, int64_t N, float *d_out,
                     int precision, void *stream);

/* workspace: grows on demand, reusable across calls on one stream */
int icon_work_create(icon_work_t **out);
int icon_work_destroy(icon_work_t *work);
/* Stage timing with HIP events recorded on the caller's stream (bench.py's live roofline leg).
 * After enabling, every icon_query_points / icon_grid_* call brackets its stages with events;
 * icon_work_stage_ms synchronises on the last one and returns milliseconds of the most recent
 * call: out_ms[0] = geometry pre-pass (nearest search, sign / outlier codes, count/scan/compact),
 *       out_ms[1] = row materialisation + outlier cmap patch (0 on the fused path),
 *       out_ms[2] = the MLP kernel alone (fused path: k_fused_f16x3, which also assembles the rows). */
int icon_work_profile(icon_work_t *work, int enable);
/* The fused MLP kernel is a persistent grid of one workgroup per CU that takes a CU whole (132 KiB of LDS, every register):
 * the kernels of a collective enqueued on another stream (the all_gather of the first half of a Z-slab, recon.py) cannot run
 * beside it.  n > 0 makes its grid n CUs smaller - the collective gets them; the MLP pays n / CUs.  Default 0. */
int icon_work_set_reserve_cus(icon_work_t *work, int n);
int icon_work_stage_ms(icon_work_t *work, float out_ms[3]);
/* More of the same profiled call: out[0] = the nearest-triangle search kernel ALONE (ms, HIP events around its launch; 0 if
 * the call had none), out[1] = shader cycles and out[2] = wall milliseconds of the fused MLP kernel's workgroup 0 (it brackets
 * its run with s_memtime and the constant-rate s_memrealtime), out[3] = out[1] / out[2] in MHz: the EFFECTIVE clock the
 * kernel ran at under its own load.  cycles = the code's invariant, MHz = the box's: bench.py prints both so that a slower
 * line can be told from a slower build without PMC files.  Synchronises like icon_work_stage_ms. */
int icon_work_profile_detail(icon_work_t *work, double out[4]);
/* ... and what EVERY workgroup of that launch of the fused MLP kernel did (it is a persistent grid of one workgroup per CU;
 * the launch ends with the slowest one).  rec[5 b + 0] = the XCD (HW_REG_XCC_ID) workgroup b ran on, [1] = its start in ms
 * after the earliest start, [2] = its span in ms (constant-rate counter), [3] = its shader cycles, [4] = the tiles of 256
 * lattice points it evaluated (static run + drawn from the pool).  *n = workgroups of the launch; at most cap records are
 * written.  bench.py derives roofline.wg_span_ms / per_xcd_clock_mhz / tail_ms from it. */
int icon_work_profile_workgroups(icon_work_t *work, double *rec, int cap, int *n);
/* The fused MLP kernel's tile partition: the tiles are cut into one contiguous span per workgroup; a workgroup evaluates the first
 * (1000 - permille) / 1000 of its span itself, the rest of every span is drawn in contiguous groups of `group` tiles (1 .. 127) by
 * the workgroups that finish first - from the spans of their own XCD first (one L2 per XCD), then from the others'.
 * permille = 0: all static.  Default 150, 2.  The result does not depend on the setting (bit for bit). */
int icon_work_set_steal(icon_work_t *work, int permille, int group);

/* ---------------------------------------------------------------------------------------------
 * HGPIFuNet.query (lib/net/HGPIFuNet.py:268-367) for explicit points:
 *   xyz = orthogonal(points, calib)                lib/net/geometry.py:46-61  (h_calib: 12 floats,
 *         row-major [3,4] = calibs[0,:3,:4]; NULL = identity as query_func passes,
 *         lib/common/train_util.py:340)
 *   icon : cal_sdf_batch + clipping (:285-305), index + feat_select (:335-336), cat (:343,359)
 *   pamir: index(im_feat, xy) ++ index(vol_feat, xyz)   (:348-354)
 *   pifu : index(im_feat, xy) ++ z                       (:356-357)
 *   preds = in_cube * regressor(point_feat)              (:274-275,361-363)
 * d_points [N,3] (the [1,3,N] tensor transposed), d_occ [N].
 * mesh may be NULL for the pamir / pifu priors.
 * ------------------------------------------------------------------------------------------- */
int icon_query_points(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                      int prior_type, float sdf_clip, int cmap_mode, const float *h_calib,
                      const float *d_points, int64_t N, float *d_occ,
                      int search, int precision, icon_work_t *work, void *stream);

/* The same with the calibration left on the device: d_calib = 12 floats, calibs[0,:3,:4] row-major
 * (HGPIFuNet.query receives `calibs` as a device tensor, lib/common/train_util.py:340-343); the
 * kernels read it themselves, so query() never synchronises the stream to copy it to the host. */
int icon_query_points_dcalib(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                             int prior_type, float sdf_clip, int cmap_mode, const float *d_calib,
                             const float *d_points, int64_t N, float *d_occ,
                             int search, int precision, icon_work_t *work, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense lattice evaluation: one rank's Z-slab of reconEngine (lib/common/seg3d_lossless.py).
 * Lattice of `res`^3 points (res odd, :84-86), world mapping of batch_eval (:125-137) with
 * align_corners=True, b_min=[-1,1,-1], b_max=[1,-1,1] (apps/ICON.py:78-90), identity calibration
 * (lib/common/train_util.py:340); output layout [z, y, x] as `occupancys.view(..., D, H, W)`
 * (seg3d_lossless.py:171).  Evaluates planes z0 <= z < z1 into d_occ [(z1-z0), res, res].
 *
 * The result equals ONE query() call over the whole lattice in z,y,x order, i.e. what
 * Seg3dLossless computes with resolutions=[res].
 *
 * Single call (single GPU, or ICON_CMAP_LOCAL on any number of GPUs):
 * ------------------------------------------------------------------------------------------- */
int icon_grid_eval_slab(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                        int prior_type, float sdf_clip, int cmap_mode,
                        int res, int z0, int z1, float *d_occ,
                        int search, int precision, icon_work_t *work, void *stream);

/* Split form for Z-slab sharding with ICON_CMAP_REFERENCE, where the outlier sign list of the
 * WHOLE lattice is needed before any slab can finish:
 *   1. icon_grid_slab_features: SDF + gather for the slab; writes the slab's outlier signs
 *      (int8, +1/-1/0, lattice order) to d_signs_local (capacity (z1-z0)*res*res) and the
 *      count to *d_count_local (device int64).
 *   2. caller exchanges counts and sign lists between ranks (RCCL all_gather) and builds the
 *      concatenated list d_signs_global [K_total] in rank order;
 *   3. icon_grid_slab_finish: patches the slab's cmap channels from the global list
 *      (rank_offset = number of outliers in lower slabs) and runs the MLP. */
int icon_grid_slab_features(const icon_mesh_t *mesh, const icon_feat_t *feat,
                            int prior_type, float sdf_clip, int cmap_mode,
                            int res, int z0, int z1, int8_t *d_signs_local, int64_t *d_count_local,
                            int search, icon_work_t *work, void *stream);
int icon_grid_slab_finish(const icon_mlp_t *mlp, int res, int z0, int z1,
                          const int8_t *d_signs_global, int64_t k_total, int64_t rank_offset,
                          float *d_occ, int precision, icon_work_t *work, void *stream);
/* The single-collective form of the same protocol (no host read between the exchange and the MLP):
 *   1. icon_grid_slab_features_msg: phase 1, and the slab's message written to d_msg =
 *        [int64 K][K outlier signs in lattice order, 2 bits each (sign + 1), four to a byte, low bits first]
 *      (msg_bytes >= 8 + ceil(points of the slab / 4); every rank uses the same msg_bytes = stride);
 *   2. ONE all_gather of the fixed-size messages: rank r's message at d_gathered + r * stride;
 *   3. icon_grid_slab_finish_gathered: K, this rank's offset and the segment of every index are derived on the
 *      device from the headers.  Evaluates the planes [za, zb) of the slab (z0 <= za < zb <= z1) into the SLAB's
 *      buffer d_occ [(z1-z0), res, res]; it may be called several times for one prepared slab - the multi-GPU driver
 *      gathers the first half of a slab while the second half is computed.  d_gathered == NULL: no exchange
 *      (ICON_CMAP_LOCAL, or a single rank) - the slab's own sign list is the whole list.
 * Shell skip: in_cube is strict (lib/net/HGPIFuNet.py:274-275,363), so every lattice point with a coordinate of
 * exactly +-1 is multiplied by 0.  When the body's bounding box is farther from the cube's boundary than the clip
 * band is wide - then every shell point is an "outside" outlier without looking at the mesh - the search and the MLP
 * cover the interior only and the shell is written as 0 (results identical, 2.3 % less work at 257^3). */
int icon_grid_slab_features_msg(const icon_mesh_t *mesh, const icon_feat_t *feat,
                                int prior_type, float sdf_clip, int cmap_mode,
                                int res, int z0, int z1, void *d_msg, int64_t msg_bytes,
                                int search, icon_work_t *work, void *stream);
int icon_grid_slab_finish_gathered(const icon_mlp_t *mlp, int res, int z0, int z1, int za, int zb,
                                   const void *d_gathered, int64_t stride, int world, int rank,
                                   float *d_occ, int precision, icon_work_t *work, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The reference's own schedule: Seg3dLossless._forward_faster (lib/common/seg3d_lossless.py:152-265, the mode
 * apps/ICON.py:89 selects) - the coarsest lattice dense, every finer level only the voxels near the 0.5 boundary of the
 * trilinearly upsampled field (dilated by a 9^3 / 7^3 / 3^3 box, minus what was evaluated before), the last level
 * interpolated - entirely as kernels on `stream`: boundary test, dilation, compaction in the reference's point order,
 * one query() call per level (own outlier sign list), scatter, upsample; nothing is read back between the levels.
 * resolutions[n_levels]: ascending, odd, each 2 r - 1 of the one before (apps/ICON.py:62-66: 33, 65, 129, 257);
 * standard box / align_corners=True / identity calibration as icon_grid_eval_slab.  d_out [res_last^3] f32, [z][y][x].
 * h_counts [n_levels + 1] (host) or NULL: the points queried per level and, last, 1 when some voxel of the coarsest level
 * exceeds 0.5 (otherwise the reference returns None, :173-177) - filling it synchronises the stream once, at the end;
 * with NULL the call is asynchronous and icon_adaptive_counts reads them later.  ICON_PRECISION_F16X3 + ICON_SEARCH_BVH
 * only (ICON_ERR_UNSUPPORTED otherwise: the host layer drives the schedule itself then).
 * ------------------------------------------------------------------------------------------- */
int icon_adaptive_eval(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp, int prior_type, float sdf_clip,
                       int cmap_mode, const int *resolutions, int n_levels, float balance, float *d_out, int64_t *h_counts,
                       int search, int precision, icon_work_t *work, void *stream);
int icon_adaptive_counts(icon_work_t *work, int n_levels, int64_t *h_counts, void *stream);
/* With h_counts the call synchronises at its end; the range safety net of the split-precision MLP (operands beyond the f16
 * range are redone in f32, per launch) is then checked ONCE for the whole schedule and a schedule that met such operands is
 * run a second time the per-launch way - same results, three launches fewer in every other case.  *n = how often that
 * happened on this workspace (diagnostics; no shipped checkpoint produces such operands). */
int icon_adaptive_reruns(icon_work_t *work, int *n);

/* ---------------------------------------------------------------------------------------------
 * The MLP input rows of a call, materialised: what HGPIFuNet.query concatenates into point_feat before the regressor
 * (lib/net/HGPIFuNet.py:329-359), point-major d_rows [N,16] f32: slots [0,c0) in the reference's channel order, zeros,
 * slot 15 = integer bits, value 8 set = in_cube (:274-275).  For regressors whose normalisation runs over the points of the call
 * (norm_mlp 'group' / 'instance', lib/net/MLP.py:35-41: the host derives the call's statistics, folds them like
 * BatchNorm and runs icon_mlp_forward on these rows) and for diagnostics.  Exactly one of h_calib / d_calib (or neither:
 * identity); icon_grid_rows: the lattice planes [z0,z1) as one call, shell included.
 * ------------------------------------------------------------------------------------------- */
int icon_query_rows(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior_type, float sdf_clip, int cmap_mode,
                    const float *h_calib, const float *d_calib, const float *d_points, int64_t N, float *d_rows,
                    int search, icon_work_t *work, void *stream);
int icon_grid_rows(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior_type, float sdf_clip, int cmap_mode,
                   int res, int z0, int z1, float *d_rows, int search, icon_work_t *work, void *stream);

/* Diagnostics: with precision F16X3 and the BVH search the query runs FUSED - the MLP input rows are
 * assembled in LDS by the MLP kernel itself and never reach HBM (icon_amd/csrc/fused_f16x3.hip).  on != 0
 * forces the materialising path (rows written by a feature kernel, patched, read back by the MLP kernel) that
 * the other precisions use; both give bit-identical results (tests/test_gpu_parity.py).  Process-wide. */
int icon_debug_set_unfused(int on);
/* Diagnostics: host != 0 makes icon_mesh_create* build on the HOST (copy to the host, sequential builder, copy back,
 * synchronises) - the checker of the device build: both emit the same arrays bit for bit.  Process-wide; default device
 * (or ICON_AMD_MESH_BUILD=host in the environment).  icon_debug_mesh_layout: byte offsets of the arena sections
 * [dyn, vnormals, nodes, leaves, tris, attr, slot2face, face2slot, bin_start, bin_slots, end of bin_slots, total]. */
int icon_debug_set_mesh_build(int host);
int icon_debug_mesh_layout(int64_t V, int64_t F, int64_t out[12]);
/* the host builder on HOST buffers, no device involved: fills h_arena (zeroed by the caller) in the arena layout */
int icon_debug_host_mesh_build(const float *h_verts, int64_t V, const int64_t *h_faces, int64_t F,
                               const float *h_cmap, const float *h_vis, void *h_arena, int64_t arena_bytes);
/* Diagnostics: on == 0 evaluates and masks the shell of the lattice like every other point (the reference's own
 * order of operations) instead of skipping it; both give identical results.  Process-wide; default on
 * (or ICON_AMD_SHELL_SKIP=0 in the environment). */
int icon_debug_set_shell_skip(int on);
/* Test / A-B switches by name, process-wide; production leaves them alone.  "lattice_fast" (default 1): 0 makes every packet
 * of the 257^3-class search derive its own tile set-up instead of reading the per-call record.  "share_waves" (-1 = by launch
 * size): wavefronts that share one packet's walk, 1 / 8 / 16 (4: adaptive schedule only).  "share_ring" (0 = 64): forced
 * ring size of the shared walks' hand-over queue; "share_lose_push" (0 = none): ticket of the push that is announced but never
 * stored; "share_spin_log2" (0 = 18): the wait bound, 2^n polls - the torture / fault-injection tests of the error path below. */
int icon_debug_set_option(const char *key, int value);
/* The searches of coarse lattices and of the adaptive schedule share one packet's BVH walk between the wavefronts of a
 * workgroup through an LDS queue (csrc/geom_device.h nearest_shared).  Every wait in that hand-over is bounded; a wave that
 * gives a wait up records what it waited for in the workspace's host-mapped error record, the walk terminates, and the host
 * reports it: ICON_ERR_STATE (message: code, workgroup, wave, ticket, queue counters) from icon_work_status - or from the
 * next icon_query_points / icon_grid_* / icon_adaptive_* call on the workspace, and from icon_adaptive_eval (with h_counts) /
 * icon_adaptive_counts for the schedule they have just synchronised on.  The record is cleared by the report.  Never
 * synchronises: synchronise the stream first for the verdict on a particular launch. */
int icon_work_status(icon_work_t *work);

/* ---- tie sensitivity of the nearest-triangle choice ---------------------------------------------------
 * lib/dataset/mesh_util.py:374-390: the winner of kaolin's point_to_mesh_distance decides the triangle whose
 * UNCLAMPED barycentric extrapolation gives norm / cmap / vis; around a vertex or an edge several triangles are
 * mathematically equidistant and the last bits of d^2 decide (PARITY UNPINNED: kaolin is not in the tree).
 * icon_sdf_query_ties reports, per point: d_face = the winner (S3: smallest d^2, lowest index on exact ties),
 * d_face2 = the runner-up (next-smallest (d^2, index) key; -1 if none) and d_ulps = how many float32 ulps the
 * runner-up's d^2 lies above the winner's, clipped to 255.  Synchronises the stream.
 * icon_work_set_tie_rule(work, 1, ulps) makes every BVH search issued through `work` pick, among the faces within
 * `ulps` ulps of the minimum d^2, the one with the HIGHEST index (rule 0 = the definition): running the pipeline
 * both ways measures how much of the output depends on the unpinned tie behaviour. */
int icon_sdf_query_ties(const icon_mesh_t *mesh, const float *d_points, int64_t N,
                        int32_t *d_face, int32_t *d_face2, uint8_t *d_ulps, void *stream);
int icon_work_set_tie_rule(icon_work_t *work, int rule, int ulps);

/* Diagnostics (synchronises): BVH work of the lattice traversal over planes [z0,z1):
 * out[0] = wavefronts (point blocks: 4^3 on fine lattices, 2^3 on coarse ones), out[1] = BVH nodes visited, out[2] = triangles
 * tested, both summed over wavefronts (every visit serves all lanes of the wavefront), out[3] = nodes + triangles of the
 * LONGEST walk (what a latency-bound launch - fewer packets than wave slots - waits for). */
int icon_debug_traversal_stats(const icon_mesh_t *mesh, int res, int z0, int z1, uint64_t out[4]);

/* Seg3dLossless._forward_faster's None rule (lib/common/seg3d_lossless.py:173-177: the call returns None when nothing exceeds
 * 0.5 on the COARSEST lattice) for a dense device volume d_occ [res,res,res]: the coarsest lattice is the sub-lattice of strides
 * (sx, sy, sz).  *h_any = 1 when one of its points exceeds `level`, else 0.  One kernel writing a host-mapped word, then the call
 * waits for the stream - the same synchronisation point the reference's `(occupancys > 0.5).sum() == 0` is. */
int icon_volume_any_above(const float *d_occ, int res, int sx, int sy, int sz, float level, icon_work_t *work, void *stream, int *h_any);

/* ---------------------------------------------------------------------------------------------
 * Seg3dLossless.export_mesh (lib/common/seg3d_lossless.py:583-604): marching cubes at 0.5 on
 * occ[1:,1:,1:]; vertices returned as (x,y,z) in voxel units of the cropped grid
 * (verts[:, [2,1,0]]), faces with flipped winding (faces[:, [0,2,1]]).
 * h_occ: HOST [res,res,res] f32.  Two-call protocol: call with h_verts == NULL to get the counts,
 * then with buffers of that size.
 * ------------------------------------------------------------------------------------------- */
int icon_export_mesh(const float *h_occ, int res, float level,
                     float *h_verts, int64_t *n_verts, int64_t *h_faces, int64_t *n_faces);

/* The same on the device (the reference uses a CUDA marching cubes for grids <= 256^3,
 * lib/common/seg3d_lossless.py:597-602): d_occ is the DEVICE volume [res,res,res].
 * icon_mc_count classifies and scans, synchronises and returns the sizes; the caller allocates
 * d_verts [n_verts,3] f32 and d_faces [n_faces,3] i64 and calls icon_mc_emit on the same workspace.
 * Same vertices and triangles as icon_export_mesh (as sets; the order differs). */
int icon_mc_count(const float *d_occ, int res, float level, icon_work_t *work, void *stream,
                  int64_t *n_verts, int64_t *n_faces);
int icon_mc_emit(float *d_verts, int64_t *d_faces, icon_work_t *work, void *stream);
/* One Z-slab's share of the same triangulation (the sharded driver's gather="mesh": meshes instead of the volume cross xGMI).
 * Cell layers [zc0, zc1) only (a layer z reads the planes z + 1 and z + 2 of the volume, export_mesh's occ[1:,1:,1:] view);
 * d_occ is the address plane 0 of the full [res,res,res] volume WOULD have - only the planes the layers read (and, with halo,
 * plane zc1 + 1, the neighbour's first) are dereferenced, so a rank passes slab_ptr - first_plane * res^2.  halo = 1: the x / y
 * edge crossings of plane zc1 + 1 are emitted too (the layer below refers to them; the neighbour emits them as well).
 * icon_mc_emit_keyed also writes d_keys [n_verts] i64 = 3 * cell + edge direction: the vertex's place in the vertex order of the
 * whole-volume call - concatenating the ranks' outputs in rank order and merging equal keys (sorted unique) reproduces
 * icon_mc_count / icon_mc_emit on the whole volume, vertex for vertex and face for face. */
int icon_mc_count_range(const float *d_occ, int res, float level, int zc0, int zc1, int halo, icon_work_t *work, void *stream,
                        int64_t *n_verts, int64_t *n_faces);
int icon_mc_emit_keyed(float *d_verts, int64_t *d_faces, int64_t *d_keys, icon_work_t *work, void *stream);

/* ---- PaMIR semantic voxelisation ---------------------------------------------------------------------
 * replaces voxelize_cuda.forward_semantic_voxelization (external CUDA wheel, requirements.txt:34; call site
 * lib/net/voxelize.py:57-59, arguments of Voxelization.forward :119-137; volume_res 128, sigma 0.05 at
 * lib/net/HGPIFuNet.py:109-118).  PARITY UNPINNED (source and test vectors absent): semantics as defined in
 * oracle/icon_accel.c - voxel centre p = ((x,y,z)+0.5)/res - 0.5, inside = p in some tetrahedron,
 * out = inside * sum_v w_v code_v / (1e-3 + sum_v w_v), w_v = exp(-|p-v|^2 / (2 sigma^2)) over the surface vertices.
 * d_verts [V,3] f32 (the first V_surf are the SMPL surface vertices, the rest the added interior ones,
 * lib/dataset/TestDataset.py:160-163), d_code [V_surf,3] f32, d_tets [T,4] int64 vertex indices,
 * d_out [res,res,res,3] f32 in (z,y,x,c) order - the layout lib/net/voxelize.py:22 documents.  Synchronises. */
int icon_semantic_voxelize(const float *d_verts, int64_t V, int64_t V_surf, const float *d_code,
                           const int64_t *d_tets, int64_t T, int res, float sigma, float *d_out, void *stream);

/* ---- connected components of a triangle mesh --------------------------------------------------------
 * the engine of clean_mesh (lib/dataset/mesh_util.py:778-791: trimesh split, keep the component with the
 * most vertices; called on the marching-cubes output at apps/ICON.py:755-756).
 * d_faces [F,3] int64, d_labels [V] int32 out: the smallest vertex index of the vertex's component.
 * Synchronises the stream (reports out-of-range face indices). */
int icon_mesh_components(const int64_t *d_faces, int64_t F, int64_t V, int32_t *d_labels, void *stream);

/* clean_mesh (lib/dataset/mesh_util.py:778-791, called at apps/ICON.py:755-756 on the marching-cubes output) as ONE call:
 * trimesh's split(only_watertight=False) - components over FACE adjacency (two faces are adjacent when they share an edge that
 * exactly two faces use; faces meeting in a vertex only are not) - and the component with the most vertices kept (a pinch
 * vertex counts for every component it touches; ties: the component holding the lowest-index face); vertices and faces keep
 * their relative order, vertex indices are renumbered.  d_verts [V,3] f32, d_faces [F,3] i64 -> d_out_verts [V,3] f32,
 * d_out_faces [F,3] i32 (caller-allocated at the INPUT sizes; the first h_counts[0] vertices / h_counts[1] faces are
 * valid).  All device pointers; synchronises the stream once, for the two counts.  ICON_ERR_ARG for a face naming a vertex
 * that does not exist.  Scratch lives in `work`. */
int icon_clean_mesh(const float *d_verts, int64_t V, const int64_t *d_faces, int64_t F, float *d_out_verts, int32_t *d_out_faces,
                    int64_t *h_counts, icon_work_t *work, void *stream);

/* ---- SMPL vertex visibility ----------------------------------------------------------------------
 * replaces get_visibility (lib/dataset/mesh_util.py:280-316: pytorch3d rasterisation at 2^12 squared,
 * cull_backfaces, 1 face per pixel; vis[faces[unique(pix_to_face)]] = 1, faces[-1] included).
 * d_xy [V,2], d_z [V] exactly as the reference passes them (TestDataset.py:136-137 passes -z),
 * d_faces [F,3] int64, d_vis [V] float32 out in {0,1}.  Synchronises the stream (frees its z-buffer). */
int icon_visibility(const float *d_xy, const float *d_z, int64_t V, const int64_t *d_faces, int64_t F,
                    int image_size, float *d_vis, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ICON_AMD_H */
