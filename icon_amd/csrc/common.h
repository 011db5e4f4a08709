// common.h - shared declarations of the icon_amd native library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/icon_amd.h"

namespace icon {
constexpr int kWgRec = 6;                // words per workgroup record behind icon_work::d_clock[4]: 4 stamps, (XCC_ID << 32 | HW_ID), tiles done
constexpr int kMaxProfGrid = 1024;       // workgroups with a record (the fused kernel's grid is one per CU)

// ---- error handling -------------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define ICON_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::icon::fail(ICON_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define ICON_ARG(cond, msg)                                                  \
    do {                                                                     \
        if (!(cond)) return ::icon::fail(ICON_ERR_ARG, std::string(msg));    \
    } while (0)

// ---- device data layouts (HBM) --------------------------------------------------------------
// BVH2 node, 64 B (one cache line): both children's boxes live in the parent so one load
// decides both.  child >= 0: internal node index; child < 0: leaf, ~child = (leaf id << 2) |
// (triangle count - 1).
// IDs ARE POSITIONS (round 4: the tree is built on the device, where nothing can be counted first):
// the triangles are permuted into `order` (slot2face), a node owns a contiguous range [begin, end) of
// it and splits it at `mid`: node id = mid - 1 (every position is the split point of at most one
// node), leaf id = begin, triangle SLOT = position in `order` - the leaf's triangles are slots
// begin .. begin + count - 1.  All arrays have F entries; the root reference lives in MeshDyn.
struct alignas(64) BvhNode {
    float lo[3][2], hi[3][2];   // [axis][child]: the two children's bounds interleaved (packed-f32 operands)
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// Triangle record in BVH leaf ("slot") order, 48 B = three 16-byte loads:
// positions of the three corners followed by their vertex ids (needed by the canonical
// edge rule of the ray-parity test).
struct alignas(16) TriRec {
    float a[3], b[3], c[3];
    int32_t ia, ib, ic;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// Per-triangle attributes in slot order, 96 B = six 16-byte loads: what face_vertices()
// gathers for normals / cmaps / vis (lib/dataset/mesh_util.py:369-372).
struct alignas(16) TriAttr {
    float n[3][3];     // vertex normals of the three corners
    float cm[3][3];    // cmap
    float vis[3];      // visibility
    int32_t face;      // original face index
    int32_t pad[2];
};
static_assert(sizeof(TriAttr) == 96, "TriAttr must be 96 bytes");

// Per-triangle constants of the distance test, 96 B, evaluated once on the host:
// corners a, b; edges ab, ac, bc; reciprocal squared edge lengths; Gram matrix of (ab, ac) and
// the reciprocal of its determinant (reciprocals are 0 for zero-length / zero-area cases).
struct alignas(32) TriPre {
    float a[3], b[3], ab[3], ac[3], bc[3];
    float i00, i11, ibc, a00, a01, a11, inn;
    int32_t face;       // original face index
    int32_t pad;
};
static_assert(sizeof(TriPre) == 96, "TriPre must be 96 bytes");

// BVH leaf (indexed by leaf id = first slot) as the packet traversal reads it: two PAIRS of triangles, each pair stored
// field-interleaved ([field 0..23][triangle 0..1]) so that one 8-byte scalar load yields the same
// constant of both triangles - the operand shape of the packed-f32 VALU (v_pk_fma_f32 & co), which
// evaluates the distance test for two triangles per instruction.  Short leaves repeat their last
// triangle.  Field order = TriPre's 24 dwords.
struct alignas(32) LeafRec {
    float pair[2][24][2];
};
static_assert(sizeof(LeafRec) == 384, "LeafRec must be 384 bytes");

constexpr int kLeafMax = 4;        // triangle slots per BVH leaf (short leaves are padded)
// wavefronts sharing ONE packet's search (geom_device.h: nearest_shared) by the packets of the launch - a launch with fewer
// packets than the GPU has wave slots lasts as long as its longest walk.  Measured on whole-slab calls, MI355X (search + MLP,
// 4^3 packets; one wave per packet / 8 / 16 waves): 33^3 (972 packets) 0.58 / 0.20 / 0.18 ms, 65^3 (5,780) 0.66 / 0.48 /
// 0.65 ms, 129^3 (39,204) 2.31 / 2.97 / 4.49 ms.
constexpr int kShareWavesMany = 8, kShareWavesFew = 16;
inline int share_waves(int64_t packets) { return packets <= 1500 ? kShareWavesFew : (packets <= 8000 ? kShareWavesMany : 1); }
constexpr int kStackDepth = 48;    // per-wave traversal stack entries (LDS)
constexpr int kXRow = 16;          // floats per point row of the MLP input buffer
constexpr int kCodeSlot = 15;      // row slot holding the per-point code word

// code word bits (row slot 15)
constexpr uint32_t kCodeOutlier = 1u;          // |sdf| >= sdf_clip
constexpr uint32_t kCodeSignShift = 1;         // 2 bits: sign(sdf) + 1  (0,1,2)
constexpr uint32_t kCodeInCube = 8u;           // all(-1 < xyz < 1)
constexpr uint32_t kCodeInside = 16u;          // check_sign: the point is inside the body (k_sign only)

constexpr int kScanBlock = 256;    // points per block of the outlier count / scan / compact passes == one tile of the fused kernel
constexpr int kMaxWorld = 64;      // ranks whose sign messages one gathered buffer may hold

// What the BUILD decides lives in device memory (the host never reads it back on the hot path: icon_mesh_create
// enqueues kernels and returns); the query kernels read it through wave-uniform scalar loads.
struct MeshDyn {
    int32_t root;               // reference of the BVH root: node id >= 0, or a leaf code (meshes of <= kLeafMax triangles)
    int32_t gy, gz;             // ray bins over (y, z); gy == 0: the bin lists overflowed their buffer -> brute-force parity
    float bin_y0, bin_z0, bin_y1, bin_z1, bin_inv_y, bin_inv_z;
    float box_lo[3], box_hi[3]; // bounding box of the vertices
    // statistics / status (icon_mesh_stats, icon_mesh_status)
    int32_t n_nodes, n_leaves, depth, bin_entries, max_bin;
    int32_t status;             // kMeshBad* bits
    int32_t pad[11];
};
static_assert(sizeof(MeshDyn) == 128, "MeshDyn layout");
constexpr int kMeshBadFace = 1;       // a face names a vertex that does not exist (treated as vertex 0)
constexpr int kMeshBadVertex = 2;     // a non-finite / absurdly large coordinate (treated as 0)
constexpr int kMeshBinOverflow = 4;   // ray-bin lists did not fit (inside tests fall back to brute force): not an error
constexpr int kMeshInternal = 8;      // an invariant of the device build failed (a bug: partition count mismatch, queue overflow)
constexpr int kMeshBuilt = 0x100;     // the last kernel of the build has run (pinned host mirror only)

struct MeshDev {
    const BvhNode *nodes;       // [F] indexed by node id = mid - 1
    const TriRec *tris;         // [F] by slot
    const TriAttr *attr;        // [F] by slot
    const int32_t *slot2face;   // [F] the permutation `order`
    const int32_t *face2slot;   // [F] its inverse: the packet traversal tracks (d^2, face) only
    const LeafRec *leaves;      // [F] indexed by leaf id = first slot of the leaf
    const MeshDyn *dyn;         // device
    int32_t n_tris;             // triangle slots = F
    // ray bins over (y, z)
    const int32_t *bin_start;   // [gy*gz + 1]
    const int32_t *bin_slots;   // triangle slots
};

struct FeatDev {
    const float *planes;   // [n_select][H][W][cpad]
    const float *vol;      // [D][H][W][vpad] or null
    int32_t C, H, W, n_select, csel, cpad;
    int32_t Cv, Dv, Hv, Wv, vpad;
    // icon prior: which SMPL features follow the sdf in the MLP input (cfg.net.smpl_feats, lib/net/HGPIFuNet.py:301-309):
    // bit 0 = cmap (3 channels), bit 1 = norm (3 channels); 'sdf' is always there, 'vis' only selects the feature half
    int32_t smpl_mask;
};
constexpr int kSmplCmap = 1, kSmplNorm = 2;

// affine calibration (rot | trans), row-major [3][4]; when `d` is set the 12 floats are read from
// device memory by the kernel itself (wave-uniform scalar loads), so a caller holding the calibration
// on the device never has to copy it to the host (no stream synchronisation in query())
struct Calib { float m[12]; const float *d; };

// lattice descriptor: point i -> (x, y, z) index; world = idx/(R-1) * (bmax-bmin) + bmin
struct Lattice {
    int32_t res, z0, z1;
};

// Lattice tiling of the geometry kernels (see geom_device.h: lattice_point)
struct LatticeMap {
    int res, z0, nz;           // the slab: planes [z0, z0+nz); a point's linear index is relative to plane z0
    int tx, ty, tz;            // tile counts of the nearest-triangle search
    int remap;                 // 0: single tiles alternate over the XCDs, x rotated per row (default), 3: the same without the rotation, 2: x-rows of tiles, 1: contiguous run of tiles per XCD
    // Shell skip (lib/net/HGPIFuNet.py:274-275,363: in_cube is strict, so every lattice point with a coordinate of
    // exactly +-1 - the outermost shell, 2.3 % of a 257^3 lattice - is multiplied by 0).  off = 1: the MLP tiles cover
    // the interior [off, res-off)^3 only and the shell is written as 0; off = 0: everything is evaluated and masked, as
    // the reference does.  A shell point still needs its code byte (it is an entry of the call's outlier sign list), but
    // not its nearest triangle when it is farther from the body's bounding box than the clip band is wide: the faces of
    // the cube whose distance from the box says so for ALL their points are left out of the SEARCH region as well.
    int off;
    int zs, nzi;               // planes of the MLP tiles: global [zs, zs + nzi) = slab planes minus the z shell
    int sx0, sx1, sy0, sy1;    // search region of k_nearest: lattice indices [s?0, s?1) in x and y ...
    int sz0, sz1;              // ... and planes [sz0, sz1) RELATIVE to the slab
    // trim = 1: the region above is the WHOLE slab and the launch grid covers it; the kernel itself leaves out the far
    // faces (lattice_trim in geom_device.h) from the body's bounding box in MeshDyn - the host does not know the box
    // (device-built mesh, no read-back).  Workgroups beyond the trimmed tiling exit at once.
    int trim;
    float trim_need;           // distance from the cube's boundary beyond which a face is "far" (from sdf_clip)
    int pk;                    // points per wavefront of the search = pk^3 (4, 2 or 1; 0 = 4): see lattice_point
};

// inf or NaN, as a test on the bits of an OPAQUE copy: in a translation unit compiled with -fno-honor-nans a NaN result is
// poison to the optimiser, which may fold any test on it away - the empty asm hides where the value came from
__device__ __forceinline__ bool not_finite(float y)
{
    uint32_t u = __float_as_uint(y);
    asm volatile("" : "+v"(u));              // an opaque INTEGER: otherwise the mask-and-compare is matched back into |y| == inf
    return (u & 0x7f800000u) == 0x7f800000u;
}

// MLP.last_op (lib/net/MLP.py:68-70; nn.Sigmoid() unless cfg.test_mode, lib/net/HGPIFuNet.py:133)
__device__ __forceinline__ float apply_last_op(float y, int last_op)
{
    return last_op == ICON_LASTOP_SIGMOID ? 1.0f / (1.0f + expf(-y)) : y;
}

// Hand-over of the nearest-triangle search to k_sign / the feature phase, structure of arrays: 16 bits per point - the low 15
// bits of the triangle slot and, in bit 15, "outside the clip band" - plus, only for meshes with more than 32,768 slots (SMPL:
// 17 k, SMPL-X: 26 k), a byte with the higher slot bits, plus d^2 for the points inside the band (geom_device.h: store_near).
// `map` (the adaptive schedule, adaptive.hip): the search ran on a lattice, the call's points are a SUBSET of it - entry of
// point i = map[i] (its [z][y][x] linear index); null: entry of point i = i.
struct NearRef { uint16_t *lo; uint8_t *hi; float *d2; const int32_t *map; };

// where the fused kernel / the patch kernels find the outlier signs of the whole call (HGPIFuNet.py:303-305)
enum { kSignNone = 0, kSignSelf = 1, kSignGlobal = 2, kSignSeg = 3 };
struct FusedSigns {
    int mode;
    const int8_t *list;           // SELF / GLOBAL: signs in point order
    const int64_t *k_dev;         // SELF: device scalar K
    int64_t k_host, rank_offset;  // GLOBAL
    const int8_t *gathered;       // SEG: all_gather output, message r = [int64 count][2-bit packed signs]
    int64_t stride;
    int world, rank;
};

}  // namespace icon

// ---- opaque handle bodies -------------------------------------------------------------------
struct icon_mesh {
    int64_t V = 0, F = 0;
    char *arena = nullptr;            // ONE device allocation holding every array below (mesh_layout in mesh_device.hip)
    size_t arena_bytes = 0;
    bool owns_arena = false;          // icon_mesh_create: hipMalloc'ed here; icon_mesh_create_arena: the caller's memory
    float *d_vnormals = nullptr;
    icon::MeshDyn *d_dyn = nullptr;
    icon::MeshDyn *h_dyn = nullptr;   // pinned host mirror, written by the last copy of the build (never waited for on the hot path)
    hipEvent_t built = nullptr;       // recorded after that copy: icon_mesh_stats / icon_mesh_status wait on it
    int depth_bound = 0;              // the builder never exceeds it (depth_bound(F)): sizes the traversal stacks without a read-back
    icon::MeshDev dev{};
};

struct icon_feat {
    float *d_planes = nullptr;
    float *d_vol = nullptr;
    icon::FeatDev dev{};
};

struct icon_adaptive;
struct icon_mlp {
    int c0 = 0;               // input channels (<= 15)
    int last_op = 0;          // ICON_LASTOP_*: applied to the network output before the in_cube mask (MLP.py:68-70)
    float *d_blob = nullptr;  // all packed operands, one allocation
    size_t blob_bytes = 0;
    // offsets (in floats) into the blob
    size_t off_w0 = 0, off_b0 = 0, off_w1 = 0, off_b1 = 0, off_w2 = 0, off_w2x = 0, off_b2 = 0, off_w3 = 0;
    size_t off_plain = 0;     // [Cout][Cin] f32 copies of the folded layers (mlp_plain_device.h) - the range safety net
    size_t off_flag = 0;      // one int: "a split-precision kernel produced a non-finite in-cube result" (k_rescue_* recompute it)
    float b3 = 0.f;
    // 3xf16 split-precision path (mlp_f16x3.hip): chunked hi/lo operand image + f32 side arrays
    char *d_f16 = nullptr;
    float f16_inv[3] = {1.f, 1.f, 1.f};
};

namespace icon {
// host helper: fn(i) for i in [0, n) on up to 16 threads (operand packing, BVH subtrees)
void parallel_for(int n, const std::function<void(int)> &fn);
// ICON_AMD_DEBUG_SYNC=1: wait for the stream after the named launch and say so on stderr (localises a hung / faulting kernel)
void debug_sync(const char *what, hipStream_t st);
// mesh_device.hip: the per-image build on the device; pooled pinned mirrors of MeshDyn + events
struct MeshLayout;
int mesh_host_state_get(MeshDyn **h, hipEvent_t *ev);
void mesh_host_state_put(MeshDyn *h, hipEvent_t ev);
void mesh_bind_arena(icon_mesh *m, const MeshLayout &L);
int mesh_build_device(icon_mesh *m, const float *d_verts, const int64_t *d_faces, const float *d_cmap, const float *d_vis, hipStream_t st);
// adaptive.hip
void adaptive_destroy(icon_adaptive *a);
// points per wavefront (pk^3) of the lattice search: 4^3 blocks; ICON_AMD_PACKET=2 (diagnostics) makes them 2^3.
// (Round-4 history: 2^3 packets were the choice for the coarse lattices of the reference's schedule while a packet was ONE
//  wavefront's walk - fewer points, shorter union walk, but 8 of 64 lanes at work; with the walk shared by the workgroup
//  (share_waves) the 4^3 packet wins at every resolution: 33^3 0.35 -> 0.18 ms, 129^3 4.3 -> 2.3 ms per slab call.)
inline int lattice_packet()
{
    static const int pk_env = getenv("ICON_AMD_PACKET") ? atoi(getenv("ICON_AMD_PACKET")) : 0;
    return pk_env == 2 ? 2 : 4;
}
// query_kernels.hip: the outlier sign list of a point-mode call whose size is known on the device only (*n_dev <= n_max)
int outlier_list_dev(icon_work *w, const int *n_dev, int64_t n_max, hipStream_t st);
int ensure_work_points(icon_work *w, int64_t n_points);
// mc_device.hip
struct McDevState;
void mc_destroy(McDevState *s);
// clean_mesh.hip
struct CleanState;
void clean_destroy(CleanState *s);
// sort_points.hip: perm[k] = index of the k-th point in Morton order of the projected positions (device array owned by w)
int morton_order(icon_work *w, const float *d_points, const float *calib12, const float *d_calib12, int64_t N, hipStream_t st, const int32_t **perm);
// mlp_kernels.hip
int mlp_launch(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, int precision, hipStream_t st);
int mlp_launch_ex(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, bool mask, hipStream_t st);
// mlp_kernels.hip: after an f16x3 launch over materialised rows - recompute the points the flag was raised for in plain f32
int mlp_rescue_rows(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, hipStream_t st);
int mlp_flag_reset(const icon_mlp *mlp, hipStream_t st);
int rescue_always();
// mlp_f16x3.hip
int mlp_pack_f16x3(icon_mlp *m, const std::vector<std::vector<float>> &W, const std::vector<std::vector<float>> &B, hipStream_t st);
int mlp_launch_f16x3(const icon_mlp *mlp, const float *d_x, int64_t N, float *d_out, bool mask, hipStream_t st);
uint16_t f32_to_f16_rtn(float f);
float f16_to_f32(uint16_t h);
float pick_scale(const std::vector<float> &W);     // power of two that brings max|W| to ~8192
// fused_f16x3.hip
int launch_sign(const icon_mesh *mesh, const Calib &cal, int res, int z0, const float *d_points, int64_t N, float sdf_clip,
                const icon_work *work, bool lattice, hipStream_t st);
// lattice: evaluates the planes [za, zb) of the slab L (global plane numbers; the whole slab = [L.z0, L.z0 + L.nz)),
// d_occ is the SLAB's output buffer; points: the N points of the call
int launch_fused_f16x3(const icon_mesh *mesh, const icon_feat *feat, const icon_mlp *mlp, int prior, const Calib &cal,
                       const LatticeMap &L, int za, int zb, const float *d_points, int64_t N, float sdf_clip, int cmap_local,
                       const icon_work *work, const FusedSigns &fs, float *d_occ, bool lattice, hipStream_t st);
// per-device launch facts (CU count; one-off kernel attributes): a process may drive several devices
int device_cu_count(int *n_cu);
int once_per_device(int kernel_id, const std::function<hipError_t()> &set);   // runs `set` once per (kernel_id, current device), under a mutex
struct LatticeFast;                   // geom_device.h
// shared walks (nearest_shared, geom_device.h): where a wave reports a hand-over it gave up on + the test-only switches
struct ShareDbg {
    int *err;                                 // [8] host-mapped error record (zero = no error), or null
    int ring, lose, spin_log2;
};
// fills `out` for a launch on this workspace: the workspace's host-mapped error record (allocated on first use) and the process-
// wide test switches of icon_debug_set_option
int work_share_dbg(icon_work *w, ShareDbg *out);
// ICON_ERR_STATE (and the record cleared) if a shared walk of an earlier launch on this workspace reported; no synchronisation
int work_check_err(icon_work *w);
int share_waves_override();           // ICON_AMD_SHARE / icon_debug_set_option("share_waves"): -1 = by launch size
}  // namespace icon

struct icon_work {
    float *d_x = nullptr;                 // [cap_x][16] MLP input rows (materialising paths only: f32 / brute force)
    int64_t cap_x = 0;
    uint16_t *d_near16 = nullptr;         // nearest-triangle result (NearRef): low slot bits + far flag [cap_points]
    uint8_t *d_near_hi = nullptr;         //   higher slot bits [cap_points_hi], allocated for meshes with > 32,768 slots only
    float *d_near_d2 = nullptr;           //   d^2 [cap_points], written inside the clip band only
    int64_t cap_points_hi = 0;
    uint8_t *d_code8 = nullptr;           // [cap_points] byte copy of each row's code word
    int64_t cap_points = 0;
    uint64_t *d_grp_mask = nullptr;       // [4 * cap_blocks] outlier ballot of every 64-point group (k_sign): rank of a point = block offset + popcounts
    int32_t *d_block_counts = nullptr;    // outliers per 256-point scan block
    int64_t *d_block_offsets = nullptr;   // exclusive prefix of the above
    int32_t *d_scan_local = nullptr;      // scan scratch: prefix inside each 1024-entry chunk
    int64_t *d_scan_part = nullptr;       // scan scratch: chunk totals
    int64_t cap_blocks = 0;
    int8_t *d_signs = nullptr;            // compacted outlier signs of the call
    int64_t cap_signs = 0;
    int64_t *d_total = nullptr;           // device scalar: number of outliers
    int *d_flag = nullptr;                // "a split-precision launch of THIS workspace produced a non-finite in-cube result" (k_rescue_fused
                                          //   redoes those points in f32): per workspace, so that launches sharing one MLP handle on different
                                          //   streams / threads cannot clear each other's flag
    mutable bool flag_clean = false;      // k_sign of the call in progress has cleared d_flag (no memset launch before the fused kernel)
    int64_t *d_seg = nullptr;             // [kMaxWorld + 1] prefix of the per-rank counts of a gathered exchange
    int32_t *d_row_count = nullptr;       // lattice mode: per (y,z) row, triangles covering the row
    int32_t *d_row_slots = nullptr;       // [rows][kRowCap]
    int64_t cap_rows = 0;
    int *h_err = nullptr;                 // [8] host-mapped error record of the shared walks (zero = none): work_share_dbg / work_check_err
    icon::LatticeFast *d_lfast = nullptr; // lattice mode: the search's per-packet set-up, written once per call (geom_device.h)
    // point mode: Morton order of the query points (sort_points.hip), so that a wavefront's 64 points are neighbours
    uint32_t *d_sort_keys = nullptr;      // [2][cap_sort]
    int32_t *d_sort_idx = nullptr;        // [2][cap_sort]
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int64_t cap_sort = 0;
    // state of the split slab protocol (icon_grid_slab_features -> icon_grid_slab_finish)
    int slab_res = 0, slab_z0 = 0, slab_z1 = 0, slab_c0 = 0, slab_cmap_slot = 0;
    bool slab_ready = false, slab_needs_patch = false, slab_patched = false;
    // diagnostics: tie rule of the nearest-triangle choice (icon_work_set_tie_rule); 0 = the definition (S3)
    int tie_rule = 0, tie_ulps = 0;
    // what phase 1 of a query recorded for phase 2 (the handles must outlive the pair of calls)
    const icon_mesh *q_mesh = nullptr;
    const icon_feat *q_feat = nullptr;
    int q_prior = 0, q_cmap_mode = 0, q_search = 0;
    float q_sdf_clip = 0.f;
    icon::Calib q_cal{};
    icon::LatticeMap q_L{};
    const float *q_points = nullptr;
    int64_t q_N = 0;
    bool q_lattice = false, q_rows_ready = false;
    // lattice-subset calls of the adaptive schedule (adaptive.hip): the number of points lives on the device (no read-back
    // between the levels), the search results are indexed through q_map, the occupancies go to d_occ[q_map[i]]
    const int32_t *q_map = nullptr;
    const int *q_n_dev = nullptr;
    int reserve_cus = 0;                  // the persistent MLP kernel leaves this many CUs free (icon_work_set_reserve_cus)
    // icon_adaptive_eval with host counts (it synchronises at its end anyway): the fused kernels of the schedule raise THIS sticky
    // word instead of d_flag and no k_rescue_fused is launched behind them (three launches at the launch floor per schedule);
    // a raised word makes the call run the schedule again with the per-launch rescue.  Null everywhere else.
    int *h_any = nullptr;                 // host-mapped word of icon_volume_any_above
    int *defer_range_flag = nullptr;
    int range_reruns = 0;                 // schedules that had to be run again for that reason (icon_adaptive_reruns)
    struct icon_adaptive *ad = nullptr;   // level buffers of icon_adaptive_eval
    icon::McDevState *mc = nullptr;       // device marching-cubes scratch (icon_mc_count / icon_mc_emit)
    icon::CleanState *clean = nullptr;    // icon_clean_mesh scratch
    // optional stage timing: ev[0] start, ev[1] features done, ev[2] patch done, ev[3] MLP done
    bool prof = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // ev[4], ev[5]: around the nearest-triangle search kernel alone
    bool ev_search = false;
    unsigned long long *d_clock = nullptr;   // [4] cycle / wall counters of the fused kernel's workgroup 0 + [kWgRec] per workgroup (FusedGeom::clock)
    mutable int clock_grid = 0;                      // grid of the most recent profiled launch of the fused kernel (records behind d_clock[4])
    double clock_kernel_ms = 0.0;
    // the fused kernel's tile partition (icon_work_set_steal): the last steal_permille / 1000 of the tiles are not assigned
    // to a workgroup up front but drawn in contiguous groups of steal_grp by whoever finishes its static run first
    unsigned int *d_steal = nullptr;         // [9] one ticket counter per XCD's list, finished-workgroup counter (self-cleaning, fused_f16x3.hip)
    int steal_permille = 150, steal_grp = 2;
    bool ev_valid = false;
};

namespace icon {
constexpr int kNearLoSlots = 32768;      // slots addressable by the 15 low bits
inline NearRef work_near(const icon_work *w, const icon_mesh *mesh)
{
    NearRef r;
    r.lo = w->d_near16; r.d2 = w->d_near_d2;
    r.hi = (mesh && mesh->F > kNearLoSlots) ? w->d_near_hi : nullptr;
    r.map = w->q_map;
    return r;
}
}  // namespace icon
