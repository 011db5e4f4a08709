// fused_f16x3.hip - HGPIFuNet.query with NOTHING but the occupancy written to HBM per point:
// the 13-channel MLP input of a point is assembled in LDS by the same persistent workgroup that then
// carries it through the 3x-f16 MFMA chain (mlp_f16x3_device.h).  Default path (precision f16x3).
//
//   before (round 1)                                   HBM per point
//     k_nearest            -> (slot, d^2)                  8 B  w
//     k_features           -> X row [16] f32 + code      8 r + 65 w
//     count / scan / compact over the codes                2 r + <=1 w
//     k_outlier_patch_self -> X rows rewritten         1 + 64 r + 64 w     (reference cmap mode)
//     k_mlp_f16x3          <- X rows                      64 r + 4 w      => ~280 B/pt, 4.8 GB per 257^3
//   now
//     k_nearest            -> slot 4 B + d^2 4 B (structure of arrays)        8 B  w
//     k_sign               -> 1-byte code (outlier, sign, inside, in_cube)   4 r + 1 w
//     count / scan / compact over the codes                2 r + <=1 w
//     k_fused_f16x3        <- slot, code, d^2 only inside the clip band (~6 %), sign list; -> occupancy
//                                                                         4 + 1 + ~3 r + 4 w  => ~30 B/pt
// (the algorithmic traffic is the 4-byte occupancy; what is left on top is the
// nearest-triangle result handed from the VALU-bound traversal kernel to the MFMA-bound one.)
//
// Reference being replaced: lib/net/HGPIFuNet.py:268-367 (query), lib/dataset/mesh_util.py:357-396,
// lib/net/geometry.py:21-61, lib/net/MLP.py:49-72.  The geometry arithmetic is the shared, bit-exact code
// of geom_device.h (this file is compiled with -ffp-contract=off like query_kernels.hip); the MLP
// arithmetic is the shared code of mlp_f16x3_device.h, so the results are IDENTICAL to the unfused
// f16x3 path (tests/test_gpu_parity.py::test_fused_equals_unfused).
//
// Workgroup = 512 threads = 8 waves, ONE per CU (LDS: 80 KiB weight double buffer + 32 KiB resident W0 +
// side arrays + 16 KiB point tile), persistent: it walks tiles of 256 WORK ITEMS.  Point mode: work item q is point
// q of the call.  Lattice mode: the work items are the lattice points that can be non-zero - with the shell skip
// (LatticeMap::off = 1) the interior [1, R-2]^3 of the evaluated planes in z,y,x order, 2.3 % fewer tiles at 257^3;
// the shell itself is written by k_shell_zero (in_cube is strict: lib/net/HGPIFuNet.py:274-275,363).
// The outlier rank of a point (its position in the call's sign list) = the exclusive scan of the outlier counts
// per 256 points of the call's LINEAR order + popcounts of the 64-point outlier ballots k_sign left behind, so
// the tiles need not be aligned with that order, and the reference's tiled cmap rule
// cmap[j][k] = s[(3j+k) mod K] (HGPIFuNet.py:303-305) is three byte loads.
// W0 / biases are staged once per workgroup; chunk 0 of the next tile is DMA'd during the last layer-2 chunk.
#pragma clang fp contract(off)

#include "geom_device.h"
#include "mlp_f16x3_device.h"
#include "mlp_plain_device.h"

#include <algorithm>
#include <mutex>

namespace icon {

constexpr int kTilePts = kF16Pts;                     // 256 points per tile
// LDS map: [2 x 40 KiB weight buffers][W0 32 KiB][side arrays][point tile 16 KiB][wave sums]
constexpr int kXsOff = kSideOff + 4352;               // point tile [256][16] f32 behind the side arrays
constexpr int kFusedLds = kXsOff + kTilePts * kXRow * 4;

struct SignSrc {
    int mode;
    const int8_t *list;          // SELF / GLOBAL: the outlier signs of the call in point order
    const int64_t *k_dev;        // SELF: device scalar K
    int64_t k_host, rank_offset; // GLOBAL: K and the number of outliers in lower slabs
    const int8_t *gathered;      // SEG: all_gather output, message r = [int64 count][signs, 2 bits each: sign + 1]
    int64_t stride;
    int world, rank;
    const int64_t *seg;          // SEG: [world + 1] exclusive prefix of the counts (k_seg_offsets)
};

struct FusedGeom {
    MeshDev m;
    FeatDev f;
    Calib cal;
    int res, z0;                 // lattice mode: resolution, first plane of the slab (linear index i is relative to it)
    int off, nx, zs;             //   work item q -> (off + q % nx, off + (q / nx) % nx, zs + q / nx^2), nx = res - 2 off
    const float *pts;            // point mode
    int64_t N;                   // work items of this launch
    const int *n_dev;            // point mode, adaptive levels: the number of work items lives on the device (N is its upper bound)
    const int32_t *out_map;      //   ... and point i's occupancy goes to out[out_map[i]]
    float sdf_clip;
    int cmap_local;
    NearRef near;                // icon prior: slot of the nearest triangle (k_nearest / k_nearest_coop), far flag, d^2 inside the clip band
    const uint8_t *code8;        // icon prior: outlier / sign / inside / in_cube (k_nearest<.., SIGN> or k_sign)
    const int64_t *block_offsets;        // exclusive scan of the outlier counts per 256 points of the linear order
    const unsigned long long *grp_mask;  // outlier ballot of every 64-point group of the linear order (4 per block)
    SignSrc sg;
    // profiling (icon_work_profile): workgroup 0 brackets its run with the shader-cycle counter and the constant-rate wall
    // counter - [0] s_memtime, [1] s_memrealtime at the start, [2], [3] at the end: cycles / wall time = the EFFECTIVE clock
    // the matrix pipe ran at under this launch's load (bench.py roofline.effective_clock_mhz: separates a slow box from a slow build).
    // Behind them EVERY workgroup leaves a record of kWgRec words at [4 + kWgRec * blockIdx.x]: the same four stamps, the
    // hardware ids (XCC_ID, HW_ID) and the tiles it evaluated (static run + stolen) - icon_work_profile_workgroups: the span of
    // every workgroup, the clock of every XCD, the tail of the launch (kernel time - median span)
    unsigned long long *clock;
    // static runs + XCD-local pools: the tiles are cut into one contiguous SPAN per workgroup (as the static partition of rounds
    // 1-5 cut them); a workgroup runs the first steal_static tiles of its span itself, the rest of every span is cut into
    // steal_ngrp groups of steal_grp tiles that whoever finishes first draws - from the spans of ITS OWN XCD first (block b runs on
    // XCD b mod 8: list x = the groups of the workgroups x, x + 8, ... in order; steal[x] is its ticket counter), from the next
    // XCD's list once its own is exhausted.  A die that holds a lower clock under the power limit leaves groups to the others;
    // tiles change XCD - and with them the triangles and texels in that XCD's L2 - only for the few percent that really move.
    // steal[8] counts the workgroups that have finished; the last one zeroes all nine words (clean for the next launch, no
    // memset).  steal == nullptr: the whole launch is static (SMALL variants, device-side N).
    unsigned int *steal;
    int steal_static, steal_grp, steal_ngrp;
};
// the workgroup's draw state and the partition's parameters: nine LDS words in the slack behind the side arrays (as scalars they
// pushed the MFMA body's spills past the 64 a vector register holds: 20 bytes of scratch per lane)
constexpr int kPoolOff = kSideOff + kSideFloats * 4;
enum { kPoolNext = 0, kPoolMask, kPoolDone, kPoolPer, kPoolRem, kPoolStatic, kPoolGroup, kPoolNGrp, kPoolGrid, kPoolWords };
static_assert(kPoolOff + 4 * kPoolWords <= kSideOff + 4352, "the draw state lives in the slack behind the side arrays");
constexpr int kPoolLenShift = 24;                        // kPoolNext = first tile | tiles << 24 (tiles < 2^23: N < 2^31), -1: none
constexpr int kMaxStealGroup = 127;

// One draw (one lane): the next group of this workgroup - own XCD's list first.  Writes kPoolNext (and the exhausted-lists mask).
__device__ __forceinline__ void pool_draw(unsigned int *steal, volatile int *pool)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int per = pool[kPoolPer], rem = pool[kPoolRem], stat = pool[kPoolStatic], grp = pool[kPoolGroup], ngrp = pool[kPoolNGrp],
              grid = pool[kPoolGrid];
    int mask = pool[kPoolMask], word = -1;
    for (int s = 0; s < 8 && word < 0; ++s) {
        const int x = (int)((xcc + (unsigned)s) & 7u);
        if ((mask >> x) & 1) continue;
        const int nwg = (grid - x + 7) >> 3;                 // workgroups x, x + 8, ... below the grid size (<= 0: none)
        for (;;) {
            const int t = (int)atomicAdd(steal + x, 1u);
            if (t >= nwg * ngrp) { mask |= 1 << x; break; }  // this list is exhausted: never asked again by this workgroup
            const int k = t / ngrp, g = t - k * ngrp, b = x + 8 * k;
            const int sb = b * per + min(b, rem), eb = sb + per + (b < rem ? 1 : 0);
            const int a = sb + stat + g * grp;
            if (a < eb) { word = a | (min(grp, eb - a) << kPoolLenShift); break; }
            // (the last group of a span one tile shorter than the longest is empty: draw again)
        }
    }
    pool[kPoolMask] = mask;
    pool[kPoolNext] = word;
}

__device__ __forceinline__ void wg_stamp(unsigned long long *rec)
{
    rec[0] = __builtin_readcyclecounter();                   // s_memtime: shader cycles
    rec[1] = __builtin_amdgcn_s_memrealtime();               // constant rate (hipDeviceAttributeWallClockRate)
}

// ---------------------------------------------------------------------------------------------
// k_sign: outlier flag, sign, inside flag and in_cube flag of every point (icon prior), 1 byte.  Reads only
// the squared distances (4 B/pt of the structure-of-arrays result of k_nearest).  Folding this epilogue into
// k_nearest itself was measured: its registers (75 VGPRs / 106 SGPRs instead of 50 / 70) cost the traversal
// two of its eight waves per SIMD, 4.48 vs 4.31 ms for the pre-pass - so it stays a separate 0.07 ms launch.
// ---------------------------------------------------------------------------------------------
static_assert(kScanBlock == 256, "k_sign's 256-point blocks are the blocks of the outlier scan and the tiles of the fused kernel");
template <bool LATTICE>
__global__ __launch_bounds__(256) void k_sign(MeshDev m, Calib cal, int res, int z0, const float *__restrict__ pts, int64_t N,
                                              float sdf_clip, const int32_t *__restrict__ row_count, const int32_t *__restrict__ row_slots,
                                              NearRef near, uint8_t *__restrict__ code8,
                                              int32_t *__restrict__ block_counts, unsigned long long *__restrict__ grp_mask, float far_box2,
                                              const int *__restrict__ n_dev, int *__restrict__ range_flag)
{
    __shared__ int wsum[4];
    if (range_flag && blockIdx.x == 0 && threadIdx.x == 0) *range_flag = 0;      // the fused kernel's range flag (it runs after this one): no memset launch
    if (n_dev) { N = *n_dev; if ((int64_t)blockIdx.x * 256 >= N) return; }      // the call's size is on the device (adaptive levels)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < N;
    uint32_t code = 0;
    if (live) {
    f3 p; bool ins;
    if (LATTICE) {
        const int64_t row = i / res;
        const int ix = (int)(i - row * res), iy = (int)(row % res), iz = (int)(row / res);
        p = lattice_world(res, ix, iy, iz + z0);
        ins = inside_row(m, p, row_count, row_slots, row);
    } else {
        p = project(resolve_calib(cal), mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        ins = inside_bins(m, p);
    }
    // outside the clip band (~94 % of a lattice): the code needs the inside test only.  Nine points in ten are
    // farther from the body's bounding box than the band is wide - they are known to be outside it without
    // reading anything; for the rest the search left a flag in the slot word
    const MeshDyn &d = *m.dyn;
    const bool far = box_dist2(d.box_lo[0], d.box_lo[1], d.box_lo[2], d.box_hi[0], d.box_hi[1], d.box_hi[2], p) > far_box2 || near_is_far(near, i);
    code = far ? sign_code_far(p, ins) : sign_code(p, near_d2(near, i), ins, sdf_clip);
    code8[i] = (uint8_t)code;
    }
    // outliers of this 256-point block == one tile of the fused kernel (kScanBlock): the count pass for free
    const unsigned long long b = __ballot(live && (code & kCodeOutlier));
    if ((threadIdx.x & 63) == 0) {
        wsum[threadIdx.x >> 6] = __popcll(b);
        grp_mask[(int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = b;    // the fused kernel derives a point's outlier rank from these
    }
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Point mode, calls of few points (the levels of the reference's schedule): FOUR lanes per point walk the point's ray-bin list
// (every 4th entry each; the parities add up) - one thread per point was a ~30 us launch of pure latency (a chain of ~100
// dependent load pairs).  A workgroup is still one 256-point block of the outlier scan / one tile of the fused kernel.
__global__ __launch_bounds__(1024) void k_sign_wide(MeshDev m, Calib cal, const float *__restrict__ pts, int64_t N, float sdf_clip, NearRef near,
                                                    uint8_t *__restrict__ code8, int32_t *__restrict__ block_counts,
                                                    unsigned long long *__restrict__ grp_mask, float far_box2, const int *__restrict__ n_dev,
                                                    int *__restrict__ range_flag)
{
    __shared__ unsigned long long gm[4];
    if (range_flag && blockIdx.x == 0 && threadIdx.x == 0) *range_flag = 0;
    if (n_dev) { N = *n_dev; if ((int64_t)blockIdx.x * 256 >= N) return; }
    if (threadIdx.x < 4) gm[threadIdx.x] = 0ull;
    __syncthreads();
    const int s = threadIdx.x & 3, pt = threadIdx.x >> 2;        // point of the block, lane of the point
    const int64_t i = (int64_t)blockIdx.x * 256 + pt;
    const bool live = i < N;
    uint32_t code = 0;
    const MeshDyn &d = *m.dyn;
    f3 p = mk3(0.f, 0.f, 0.f);
    int beg = 0, end = 0;
    bool brute = false;
    if (live) {
        p = project(resolve_calib(cal), mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        if (d.gy == 0) brute = true;
        else if (p.y >= d.bin_y0 && p.y <= d.bin_y1 && p.z >= d.bin_z0 && p.z <= d.bin_z1) {
            const int cy = bin_cell(p.y, d.bin_y0, d.bin_inv_y, d.gy);
            const int cz = bin_cell(p.z, d.bin_z0, d.bin_inv_z, d.gz);
            const int cell = cz * d.gy + cy;
            beg = m.bin_start[cell]; end = m.bin_start[cell + 1];
        }
    }
    int cnt = 0;
    for (int k = beg + s; k < end; k += 4) {
        f3 a, b, c; int ia, ib, ic;
        load_tri_full(m.tris + m.bin_slots[k], a, b, c, ia, ib, ic);
        cnt += ray_hit(p, a, b, c, ia, ib, ic);
    }
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    if (live) {
        const bool ins = brute ? inside_brute(m, p) : ((cnt & 1) != 0);
        const bool far = box_dist2(d.box_lo[0], d.box_lo[1], d.box_lo[2], d.box_hi[0], d.box_hi[1], d.box_hi[2], p) > far_box2 || near_is_far(near, i);
        code = far ? sign_code_far(p, ins) : sign_code(p, near_d2(near, i), ins, sdf_clip);
        if (s == 0) code8[i] = (uint8_t)code;
    }
    // the outliers of the block's four 64-point groups as masks (the fused kernel derives a point's outlier rank from them)
    unsigned long long x = __ballot(live && s == 0 && (code & kCodeOutlier));      // bit 4 j = point j of this wave's 16
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000f000f000f000full;
    x = (x | (x >> 12)) & 0x000000ff000000ffull;
    x = (x | (x >> 24)) & 0xffffull;
    const int wave = threadIdx.x >> 6;                           // points 16 wave .. 16 wave + 15 of the block
    if ((threadIdx.x & 63) == 0 && x) atomicOr(&gm[wave >> 2], x << (16 * (wave & 3)));
    __syncthreads();
    if (threadIdx.x < 4) grp_mask[(int64_t)blockIdx.x * 4 + threadIdx.x] = gm[threadIdx.x];
    if (threadIdx.x == 0) block_counts[blockIdx.x] = __popcll(gm[0]) + __popcll(gm[1]) + __popcll(gm[2]) + __popcll(gm[3]);
}

// the shell of the planes [za, zb) of a slab whose MLP tiles skipped it: exact zeros (in_cube * pred, HGPIFuNet.py:363).
// The first `row_wgs` workgroups take one ROW per thread and zero its two end points; the workgroups behind them take one FACE
// row per wavefront (rows y = 0 / res - 1 of every plane, every row of the planes z = 0 / res - 1) and zero it whole with
// coalesced stores.  (Until round 6: one 64-thread workgroup per row - 66,049 workgroups at 257^3, 16 us of launch rate.)
__global__ __launch_bounds__(256) void k_shell_zero(float *__restrict__ out, int res, int z0, int za, int zb, int row_wgs)
{
    const int planes = zb - za;
    if ((int)blockIdx.x < row_wgs) {
        const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (row >= (int64_t)planes * res) return;
        const int iy = (int)(row % res), iz = za + (int)(row / res);
        float *o = out + ((int64_t)(iz - z0) * res + iy) * res;
        o[0] = 0.0f; o[res - 1] = 0.0f;
        return;
    }
    // face rows: first the two y-rows of every plane, then the whole planes z = 0 and z = res - 1 where the piece holds them
    const int f = ((int)blockIdx.x - row_wgs) * 4 + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int iy, iz;
    if (f < 2 * planes) { iz = za + (f >> 1); iy = (f & 1) ? res - 1 : 0; }
    else {
        const int g = f - 2 * planes;                              // row g of the face planes present: z = 0 first
        const bool has0 = za == 0, has1 = zb == res;
        const int pl = g / res;
        if (pl >= (has0 ? 1 : 0) + (has1 ? 1 : 0)) return;
        iz = (pl == 0 && has0) ? 0 : res - 1;
        iy = g % res;
    }
    float *o = out + ((int64_t)(iz - z0) * res + iy) * res;
    for (int x = lane; x < res; x += 64) o[x] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// the fused kernel
// ---------------------------------------------------------------------------------------------
// SEG mode: `off` = exclusive prefix of the per-rank counts (world + 1 entries, global memory written by
// k_seg_offsets, read here through wave-uniform scalar loads): segment of mm = number of boundaries <= mm
__device__ __forceinline__ float sign_at(const SignSrc &sg, int64_t mm)
{
    if (sg.mode == kSignSeg) {
        int r = 0;
        int64_t base = 0;
        for (int q = 1; q < sg.world; ++q) {
            const int64_t o = sg.seg[q];
            const bool ge = mm >= o;
            r += ge ? 1 : 0;
            base = ge ? o : base;
        }
        const int64_t e = mm - base;                             // 2 bits per sign (sign + 1), four to a byte
        const uint32_t byte = (uint8_t)sg.gathered[(int64_t)r * sg.stride + 8 + (e >> 2)];
        return (float)((int)((byte >> (2 * (int)(e & 3))) & 3u) - 1);
    }
    return (float)sg.list[mm];
}

// lattice mode: work item q -> lattice indices and the point's index in the linear order of the slab
__device__ __forceinline__ int64_t lattice_item(const FusedGeom &G, int64_t q, int &ix, int &iy, int &iz)
{
    const uint32_t nx = (uint32_t)G.nx, qq = (uint32_t)q;          // N < 2^31 (checked by the launcher)
    const uint32_t r = qq / nx, lx = qq - r * nx, lz = r / nx, ly = r - lz * nx;
    ix = G.off + (int)lx; iy = G.off + (int)ly; iz = G.zs + (int)lz;
    return ((int64_t)(iz - G.z0) * G.res + iy) * G.res + ix;
}

__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The MLP input row of one point of the icon prior, written slot by slot into the LDS tile (no 16-wide register tuple next
// to the MFMA accumulators); slots >= c0 are never read as data (masked by index in the MLP part), slot 15 = in_cube flag.
// (Requesting the texels of BOTH feature halves up front - their addresses depend on the position only - instead of after
//  the dependent chain slot -> triangle attributes -> visibility flag was measured: 14.69 vs 14.36 ms; the phase is bound by
//  the number of loads, not by the length of the chain.)
__device__ __forceinline__ void icon_row(const FusedGeom &G, f3 p, int64_t i, float *xrow, int64_t K, int64_t rank0)
{
    const uint32_t code = G.code8[i];
    Nearest nr;
    nr.slot = near_slot_of(G.near, i); nr.face = 0;
    nr.d2 = (code & kCodeOutlier) ? 0.0f : near_d2(G.near, i);    // an outlier's sdf is its sign
    const SdfOut o = sdf_attrs(G.m, p, nr, (code & kCodeInside) != 0);
    float s = o.sdf;
    f3 cmv = o.cm;
    if (code & kCodeOutlier) {                      // HGPIFuNet.py:298-305
        s = (float)((int)((code >> kCodeSignShift) & 3u) - 1);
        if (G.cmap_local) cmv = mk3(s, s, s);
        else if (K > 0 && (G.f.smpl_mask & kSmplCmap)) {
            // rank among the call's outliers: scan over 256-point blocks + ballots of the 64-point groups
            const int64_t blk = i >> 8;
            const int g = (int)(i >> 6) & 3;
            const unsigned long long *mk = G.grp_mask + blk * 4;
            int before = __popcll(mk[g] & ((1ull << (i & 63)) - 1ull));
            if (g > 0) before += __popcll(mk[0]);
            if (g > 1) before += __popcll(mk[1]);
            if (g > 2) before += __popcll(mk[2]);
            const int64_t jr = rank0 + G.block_offsets[blk] + before;
            float c3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int64_t mm = 3 * jr + k;            // jr < K  =>  mm < 3K
                if (mm >= K) mm -= K;
                if (mm >= K) mm -= K;
                c3[k] = sign_at(G.sg, mm);
            }
            cmv = mk3(c3[0], c3[1], c3[2]);
        }
    }
    gather_planes_dyn(G.f, (G.f.n_select == 2 && o.vis == 0.0f) ? 1 : 0, p.x, p.y, xrow);   // feat_select: vis==1 -> front half; no 'vis' in smpl_feats: all channels
    int hh = G.f.csel;                                  // [img | sdf | cmap (if) | norm (if)], HGPIFuNet.py:301-311
    xrow[hh++] = s;
    if (G.f.smpl_mask & kSmplCmap) { xrow[hh] = cmv.x; xrow[hh + 1] = cmv.y; xrow[hh + 2] = cmv.z; hh += 3; }
    if (G.f.smpl_mask & kSmplNorm) { xrow[hh] = o.nrm.x; xrow[hh + 1] = o.nrm.y; xrow[hh + 2] = o.nrm.z; }
    xrow[kCodeSlot] = __int_as_float((int)(code & kCodeInCube));
}

// The MLP input row of work item q of the launch (16 floats: reference channel order, zeros, slot kCodeSlot = the in_cube
// bit): what the feature phase of k_fused_f16x3 writes into LDS - and what k_rescue_fused rebuilds for a point to redo.
template <int PRIOR, bool LATTICE>
__device__ __forceinline__ void build_row(const FusedGeom &G, int64_t q, float *xrow, int64_t K, int64_t rank0)
{
    // work item -> point: its world position p and its index i in the linear order of the call
    f3 p;
    int64_t i;
    if (LATTICE) {
        int ix, iy, iz;
        i = lattice_item(G, q, ix, iy, iz);
        p = lattice_world(G.res, ix, iy, iz);
    } else {
        i = q;
        p = project(resolve_calib(G.cal), mk3(G.pts[3 * i], G.pts[3 * i + 1], G.pts[3 * i + 2]));
    }
    if (PRIOR == ICON_PRIOR_ICON) {
        icon_row(G, p, i, xrow, K, rank0);
    } else {
        gather_planes_dyn(G.f, 0, p.x, p.y, xrow);
        const int hh = G.f.csel;
        if (PRIOR == ICON_PRIOR_PAMIR) {
            float v[8];
            if (G.f.vpad == 8) gather_volume<2>(G.f, p.x, p.y, p.z, v); else gather_volume<1>(G.f, p.x, p.y, p.z, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < G.f.Cv) xrow[hh + k] = v[k];
        } else {
            xrow[hh] = p.z;
        }
        xrow[kCodeSlot] = __int_as_float((int)in_cube_bit(p));
    }
}

// the sign-list geometry of the launch (reference cmap mode): K outliers in the call, rank0 = rank of this slab's first
__device__ __forceinline__ void sign_list_extent(const FusedGeom &G, int64_t &K, int64_t &rank0)
{
    K = 0; rank0 = 0;
    if (G.cmap_local) return;
    if (G.sg.mode == kSignSelf) K = *G.sg.k_dev;
    else if (G.sg.mode == kSignGlobal) { K = G.sg.k_host; rank0 = G.sg.rank_offset; }
    else if (G.sg.mode == kSignSeg) { K = G.sg.seg[G.sg.world]; rank0 = G.sg.seg[G.sg.rank]; }
}

// SMALL (calls of at most 262,144 points: the levels of the reference's schedule, ordinary query() batches): a 256-point tile is
// ~50 us of ONE CU's matrix cores whatever the launch size, and a call of 20,000 points occupies 79 of the 256 CUs.  When the
// call fits the grid as 128-point tiles, the tiles are 128 points: waves 0-3 (one per SIMD) carry 32 points each through the
// MLP, waves 4-7 only issue their share of the weight DMA and keep the barriers - twice the CUs at work, half the chain per
// tile.  Decided in the kernel (the point count of a schedule level lives on the device).  SMALL = false compiles the
// 257^3 kernel as it was.
template <int PRIOR, bool LATTICE, bool SMALL = false>
__global__ __launch_bounds__(kF16Block, 2) void k_fused_f16x3(FusedGeom G, float *__restrict__ out, MlpF16Dev w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane0 = threadIdx.x & 63, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *Xs = reinterpret_cast<float *>(smem + kXsOff);
    if (G.clock && threadIdx.x == 0) {
        if (blockIdx.x == 0) wg_stamp(G.clock);
        if (blockIdx.x < kMaxProfGrid) wg_stamp(G.clock + 4 + kWgRec * blockIdx.x);
    }

    // ---- once per workgroup: resident layer-0 operands, side arrays, sign-list geometry ------------------
    issue_units(w.image, smem + kW0Off, kW0Bytes / 1024, wave0, lane0);
    float *side = reinterpret_cast<float *>(smem + kSideOff);
    for (int i = threadIdx.x; i < kSideFloats; i += kF16Block) side[i] = w.side[i];
    const float *sb0 = side, *sb1 = side + 512, *sb2 = side + 768, *sw3 = side + 896;
    int64_t K = 0, rank0 = 0;
    if (PRIOR == ICON_PRIOR_ICON) sign_list_extent(G, K, rank0);
    // wave-uniform 64-bit values that live across the whole MLP body: keep them in SGPRs, not in the
    // 250-register vector budget of the MFMA chain
    K = uniform64(K); rank0 = uniform64(rank0);

    if (G.n_dev) { G.N = *G.n_dev; if (G.N <= 0) return; }
    // every workgroup walks a CONTIGUOUS run of tiles: consecutive tiles share the cache lines at their common boundary
    // (a tile of interior rows does not end on a line) and the triangles / feature texels of neighbouring points, and a
    // workgroup stays on one XCD - interleaved over the grid, those lines were fetched into two L2s (105 vs 91 MB of HBM
    // reads per 257^3 launch when the tiles stopped being 1 KiB-aligned runs of the linear order)
    const int tp = (SMALL && G.N <= (int64_t)(kTilePts / 2) * gridDim.x) ? kTilePts / 2 : kTilePts;       // points per tile (wave-uniform)
    const int ntiles = (int)((G.N + tp - 1) / tp);                  // N < 2^31 (checked by the launcher)
    // the workgroup's span of the tiles, and of it the static run it evaluates itself (all of it when the launch is static)
    const bool stealing = !SMALL && G.steal != nullptr;
    const int per = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
    int tile = (int)blockIdx.x * per + min((int)blockIdx.x, rem);
    int tile_end = tile + (stealing ? min(G.steal_static, per) : per + ((int)blockIdx.x < rem ? 1 : 0));
    tile = __builtin_amdgcn_readfirstlane(tile);
    tile_end = __builtin_amdgcn_readfirstlane(tile_end);
    volatile int *pool = reinterpret_cast<volatile int *>(smem + kPoolOff);
    if (!SMALL) {
        if (threadIdx.x == 0) {
            pool[kPoolPer] = per; pool[kPoolRem] = rem; pool[kPoolStatic] = G.steal_static; pool[kPoolGroup] = G.steal_grp;
            pool[kPoolNGrp] = G.steal_ngrp; pool[kPoolGrid] = (int)gridDim.x; pool[kPoolDone] = 0; pool[kPoolMask] = 0; pool[kPoolNext] = -1;
            if (stealing && tile >= tile_end) pool_draw(G.steal, pool);      // no static run at all: draw now
        }
        __syncthreads();
        if (stealing && tile >= tile_end) {
            const int word = __builtin_amdgcn_readfirstlane(pool[kPoolNext]);
            __syncthreads();                                 // (every wave has read the word before a draw of the loop rewrites it)
            if (word >= 0) { tile = word & ((1 << kPoolLenShift) - 1); tile_end = tile + (word >> kPoolLenShift); }
        }
    }
    if (tile < tile_end) issue_chunk(w.image, smem, 0, wave0, lane0);

    // `nb`: the group drawn while the LAST tile of the current run is in flight (first tile | tiles << 24; -1: none, or no draw) -
    // a few returned atomics of wave 4, which has no work item in the feature phase; the barrier that publishes the tile publishes it
    for (int nb = -1; tile < tile_end;) {
        // Everything derived from the thread index is re-derived per tile from an opaque copy: hoisted out of
        // the loop, those ~20 lane-dependent addresses would have to stay live across the 250-register MFMA
        // body, i.e. be spilled to scratch (measured: 1.5 GB of scratch writes per 257^3 launch).
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int j = lane & 31, h = lane >> 5;
        // ---- feature phase: waves 0-3, one work item per thread -> Xs[t][16] -----------------------------
        const int t = tid;
        const bool draw = !SMALL && stealing && tile + 1 == tile_end;       // wave-uniform: this is the run's last tile
        if (!SMALL && draw && t == 256) pool_draw(G.steal, pool);           // (-1: every list is exhausted)
        const bool worker = t < tp;                             // wave-uniform
        int64_t q = (int64_t)tile * tp + (worker ? t : 0);
        if (q >= G.N) q = G.N - 1;                               // padding lanes of the last tile recompute its last item
        if (worker) build_row<PRIOR, LATTICE>(G, q, Xs + t * kXRow, K, rank0);
        __syncthreads();          // tile visible; chunk 0 (and, the first time, W0 + side arrays) landed

        nb = (!SMALL && draw) ? __builtin_amdgcn_readfirstlane(pool[kPoolNext]) : -1;
        if (!SMALL && tid == 0) pool[kPoolDone] = pool[kPoolDone] + 1;
        const bool more = tile + 1 < tile_end || nb >= 0;       // the first chunks of the next tile ride on the last ones
        if (SMALL && wave >= (tp >> 5)) {
            // a wave without points (128-point tiles): its share of the weight stream, the same barriers as the body below
            for (int c = 0; c < 16; ++c) {
                issue_chunk(w.image, smem + ((c + 1) & 1) * kBufBytes, c + 1, wave, lane);
                ICON_CHUNK_BARRIER();
            }
            issue_chunk(w.image, smem + kBufBytes, 17, wave, lane);
            ICON_CHUNK_BARRIER();
            issue_chunk(w.image, smem, 18, wave, lane);
            ICON_CHUNK_BARRIER();
            issue_chunk(w.image, smem + kBufBytes, 19, wave, lane);
            ICON_CHUNK_BARRIER();
            if (more) issue_chunk(w.image, smem, 0, wave, lane);
            ++tile;
            continue;
        }
        // ---- MLP: one wave = 32 points, lane (j,h) holds input slots 8h..8h+7 of point j ------------------
        const int pt = wave * 32 + j;
        float xr[8];
        {
            const float4 *q = reinterpret_cast<const float4 *>(Xs + pt * kXRow + 8 * h);
            const float4 a = q[0], b4 = q[1];
            xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b4.x; xr[5] = b4.y; xr[6] = b4.z; xr[7] = b4.w;
        }
        const float maskf = (__float_as_int(Xs[pt * kXRow + kCodeSlot]) & (int)kCodeInCube) ? 1.0f : 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = (s + 8 * h < w.c0) ? xr[s] : 0.0f;
        half8 xhi, xlo;
        split8(xr, xhi, xlo);

        f32x16 acc1[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc1[m] = ld16(sb1 + (m * 2 + h) * 16);
        half8 bh[2], bl[2];
        activate_split(l0_tile(smem + kW0Off, sb0, 0, xhi, xlo, h, lane), LeakyK{w.inv0, w.p0, w.q0}, bh, bl);
        f32x16 acc2[4];
        for (int c = 0; c < 16; ++c) {
            l01_chunk(smem + (c & 1) * kBufBytes, smem + ((c + 1) & 1) * kBufBytes, smem + kW0Off, sb0, w.image, c, acc1, xhi, xlo,
                      LeakyK{w.inv0, w.p0, w.q0}, h, lane, wave, bh, bl);
            ICON_CHUNK_BARRIER();
        }
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) acc2[m2] = ld16(sb2 + (m2 * 2 + h) * 16);
        activate_split(acc1[0], LeakyK{w.inv1, w.p1, w.q1}, bh, bl);
        l2_chunk<0>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
        ICON_CHUNK_BARRIER();
        l2_chunk<1>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
        ICON_CHUNK_BARRIER();
        l2_chunk<2>(smem, smem + kBufBytes, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl);
        ICON_CHUNK_BARRIER();
        l2_chunk<3>(smem + kBufBytes, smem, w.image, acc1, acc2, xhi, xlo, LeakyK{w.inv1, w.p1, w.q1}, lane, wave, bh, bl, more ? 0 : -1);

        // ---- layer 3 on the VALU (f32) ---------------------------------------------------------------------
        const float *w3 = sw3 + h * 72;
        float part = 0.0f;
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) {
            const f32x16 wv = ld16(w3 + m2 * 16);
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                part = fmaf(wv[tt], leaky_scaled(acc2[m2][tt], LeakyK{w.inv2, w.p2, w.q2}), part);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) part = fmaf(w3[64 + s], xr[s], part);
        const float other = __shfl_xor(part, 32);
        const float y = apply_last_op((part + other) + w.b3, w.last_op);
        // where this point's occupancy goes is re-derived from the work item (a handful of integer instructions):
        // nothing lane-dependent lives across the MFMA body
        const int64_t oq = (int64_t)tile * tp + pt;
        if (h == 0 && oq < G.N) {
            int ix, iy, iz;
            out[LATTICE ? lattice_item(G, oq, ix, iy, iz) : (G.out_map ? (int64_t)G.out_map[oq] : oq)] = masked_result(y, maskf != 0.0f, w.flag);
        }
        // the next tile: the one behind this, or the first of the group drawn during this one
        if (!SMALL && nb >= 0) {
            tile = nb & ((1 << kPoolLenShift) - 1);
            tile_end = tile + (nb >> kPoolLenShift);
        } else ++tile;
    }
    if (!SMALL && stealing && threadIdx.x == 0) {
        // the last workgroup to finish leaves the pair clean for the next launch (every draw of this launch precedes its
        // workgroup's arrival here in program order; agent-scope atomics on both words)
        __threadfence();
        if (atomicAdd(G.steal + 8, 1u) == gridDim.x - 1)
            for (int k = 0; k < 9; ++k) atomicExch(G.steal + k, 0u);
    }
    if (G.clock && threadIdx.x == 0 && blockIdx.x < kMaxProfGrid) {
        if (blockIdx.x == 0) wg_stamp(G.clock + 2);
        unsigned long long *rec = G.clock + 4 + kWgRec * blockIdx.x;
        wg_stamp(rec + 2);
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        rec[4] = ((unsigned long long)xcc << 32) | hw;
        rec[5] = SMALL ? 0ull : (unsigned long long)pool[kPoolDone];
    }
}

// The range safety net (mlp_plain_device.h): when the fused kernel raised the flag, find the work items whose result is not
// finite, rebuild their input rows exactly as the feature phase did and redo them in plain f32.  One wave per 64 work items.
template <int PRIOR, bool LATTICE>
__global__ __launch_bounds__(64) void k_rescue_fused(FusedGeom G, float *__restrict__ out, MlpPlain P, const int *flag, int always)
{
    __shared__ float s[kPlainLds];
    if (!always && *flag == 0) return;                        // the usual case: one word read per workgroup of a small grid
    if (G.n_dev) G.N = *G.n_dev;
    const int lane = threadIdx.x;
    int64_t K = 0, rank0 = 0;
    if (PRIOR == ICON_PRIOR_ICON) sign_list_extent(G, K, rank0);
    for (int64_t base = (int64_t)blockIdx.x * 64; base < G.N; base += (int64_t)gridDim.x * 64) {
        const int64_t q = base + lane;
        int ix, iy, iz;
        const int64_t o = q < G.N ? (LATTICE ? lattice_item(G, q, ix, iy, iz) : (G.out_map ? (int64_t)G.out_map[q] : q)) : 0;
        const float v = q < G.N ? out[o] : 0.0f;
        unsigned long long todo = __ballot(not_finite(v));
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if (lane == b) {
                float row[kXRow];
#pragma unroll
                for (int k = 0; k < kXRow; ++k) row[k] = 0.0f;
                build_row<PRIOR, LATTICE>(G, q, row, K, rank0);
#pragma unroll
                for (int k = 0; k < kXRow; ++k) s[k] = k < P.c0 ? row[k] : 0.0f;
            }
            const float y = mlp_plain_wave(P, s, lane);
            if (lane == b) out[o] = y;       // only in-cube results are ever non-finite (masked_result): no mask to apply
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// exclusive prefix of the per-rank outlier counts found in the headers of the gathered messages
__global__ void k_seg_offsets(const int8_t *__restrict__ gathered, int64_t stride, int world, int64_t *__restrict__ seg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t a = 0;
    for (int r = 0; r < world; ++r) { seg[r] = a; a += *reinterpret_cast<const int64_t *>(gathered + (int64_t)r * stride); }
    seg[world] = a;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int launch_sign(const icon_mesh *mesh, const Calib &cal, int res, int z0, const float *d_points, int64_t N, float sdf_clip,
                const icon_work *work, bool lattice, hipStream_t st)
{
    const int64_t nb = (N + 255) / 256;
    ICON_ARG(nb > 0 && nb < (1ll << 31), "too many workgroups for one launch");
    const float far_box2 = far_box_dist2(sdf_clip);
    if (lattice) hipLaunchKernelGGL(k_sign<true>, dim3((unsigned)nb), dim3(256), 0, st, mesh->dev, cal, res, z0, d_points, N, sdf_clip,
                                    work->d_row_count, work->d_row_slots, work_near(work, mesh), work->d_code8, work->d_block_counts,
                                    (unsigned long long *)work->d_grp_mask, far_box2, (const int *)nullptr, work->d_flag);
    else if (work->q_n_dev || N <= 262144)        // few points (a level of the schedule: the size on the device is a few percent of N)
        hipLaunchKernelGGL(k_sign_wide, dim3((unsigned)nb), dim3(1024), 0, st, mesh->dev, cal, d_points, N, sdf_clip, work_near(work, mesh), work->d_code8,
                           work->d_block_counts, (unsigned long long *)work->d_grp_mask, far_box2, work->q_n_dev, work->d_flag);
    else hipLaunchKernelGGL(k_sign<false>, dim3((unsigned)nb), dim3(256), 0, st, mesh->dev, cal, res, z0, d_points, N, sdf_clip,
                            (const int32_t *)nullptr, (const int32_t *)nullptr, work_near(work, mesh), work->d_code8, work->d_block_counts,
                            (unsigned long long *)work->d_grp_mask, far_box2, work->q_n_dev, work->d_flag);
    ICON_HIP(hipGetLastError());
    work->flag_clean = work->d_flag != nullptr;
    return ICON_OK;
}

// ---- per-device launch facts --------------------------------------------------------------------------
// A process may drive several devices (one image per GPU from one process, threads): the CU count and the
// "dynamic LDS attribute has been set" flags are per device, guarded by one mutex (touched once per launch).
namespace {
constexpr int kMaxDevices = 64, kMaxKernelIds = 16;
std::mutex g_dev_mutex;
int g_n_cu[kMaxDevices];
bool g_attr_set[kMaxKernelIds][kMaxDevices];
}  // namespace

int device_cu_count(int *n_cu)
{
    int dev = 0;
    ICON_HIP(hipGetDevice(&dev));
    ICON_ARG(dev >= 0 && dev < kMaxDevices, "device index out of range");
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    if (!g_n_cu[dev]) {
        int n = 0;
        ICON_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        g_n_cu[dev] = n > 0 ? n : 256;
    }
    *n_cu = g_n_cu[dev];
    return ICON_OK;
}

// Runs `set` (a hipFuncSetAttribute) exactly once per (kernel_id, current device), UNDER the mutex and before the pair
// is marked: a second host thread cannot launch the kernel with its dynamic LDS size before the attribute exists.
int once_per_device(int kernel_id, const std::function<hipError_t()> &set)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || kernel_id < 0 || kernel_id >= kMaxKernelIds) {
        const hipError_t e = set();
        return e == hipSuccess ? ICON_OK : fail(ICON_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    if (g_attr_set[kernel_id][dev]) return ICON_OK;
    const hipError_t e = set();
    if (e != hipSuccess) return fail(ICON_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    g_attr_set[kernel_id][dev] = true;
    return ICON_OK;
}

int launch_fused_f16x3(const icon_mesh *mesh, const icon_feat *feat, const icon_mlp *mlp, int prior, const Calib &cal,
                       const LatticeMap &L, int za, int zb, const float *d_points, int64_t N, float sdf_clip, int cmap_local,
                       const icon_work *work, const FusedSigns &fs, float *d_occ, bool lattice, hipStream_t st)
{
    FusedGeom G{};
    if (mesh) G.m = mesh->dev;
    G.f = feat->dev; G.cal = cal; G.pts = d_points; G.sdf_clip = sdf_clip; G.cmap_local = cmap_local;
    if (lattice) {
        // planes [za, zb) of the slab, minus the z shell when it is skipped; x / y interior likewise
        const int zs = std::max(za, L.off), ze = std::min(zb, L.res - L.off);
        if (L.off) {
            const int64_t rows = (int64_t)(zb - za) * L.res;
            const int row_wgs = (int)((rows + 255) / 256);
            const int64_t face_rows = 2 * (int64_t)(zb - za) + (int64_t)((za == 0 ? 1 : 0) + (zb == L.res ? 1 : 0)) * L.res;
            if (rows > 0) hipLaunchKernelGGL(k_shell_zero, dim3((unsigned)(row_wgs + (face_rows + 3) / 4)), dim3(256), 0, st, d_occ, L.res, L.z0, za, zb, row_wgs);
        }
        G.res = L.res; G.z0 = L.z0; G.off = L.off; G.nx = L.res - 2 * L.off; G.zs = zs;
        N = (ze > zs) ? (int64_t)(ze - zs) * G.nx * G.nx : 0;
    }
    if (N <= 0) { ICON_HIP(hipGetLastError()); return ICON_OK; }
    ICON_ARG(N < (1ll << 31), "fused: more than 2^31 points in one call");
    G.N = N;
    G.near = work_near(work, mesh); G.code8 = work->d_code8;
    if (!lattice) { G.n_dev = work->q_n_dev; G.out_map = work->q_map; }
    G.clock = work->prof ? work->d_clock : nullptr;
    G.steal = nullptr; G.steal_static = 0; G.steal_grp = 1; G.steal_ngrp = 0;
    G.block_offsets = work->d_block_offsets; G.grp_mask = (const unsigned long long *)work->d_grp_mask;
    G.sg.mode = fs.mode; G.sg.list = fs.list; G.sg.k_dev = fs.k_dev; G.sg.k_host = fs.k_host; G.sg.rank_offset = fs.rank_offset;
    G.sg.gathered = fs.gathered; G.sg.stride = fs.stride; G.sg.world = fs.world; G.sg.rank = fs.rank; G.sg.seg = nullptr;
    if (fs.mode == kSignSeg) {
        if (!work->d_seg) return fail(ICON_ERR_STATE, "fused: segment offsets buffer missing");
        hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(64), 0, st, fs.gathered, fs.stride, fs.world, work->d_seg);
        G.sg.seg = work->d_seg;
    }
    MlpF16Dev w;
    w.image = mlp->d_f16;
    w.side = reinterpret_cast<const float *>(mlp->d_f16 + kImageBytes);
    w.b3 = mlp->b3; w.inv0 = mlp->f16_inv[0]; w.inv1 = mlp->f16_inv[1]; w.inv2 = mlp->f16_inv[2]; w.c0 = mlp->c0; w.last_op = mlp->last_op;
    w.p0 = 0.505f * w.inv0; w.q0 = 0.495f * w.inv0; w.p1 = 0.505f * w.inv1; w.q1 = 0.495f * w.inv1; w.p2 = 0.505f * w.inv2; w.q2 = 0.495f * w.inv2;
    // the range flag of THIS workspace (the handle's own word serves icon_mlp_forward, which has no workspace)
    w.flag = work->d_flag ? work->d_flag : reinterpret_cast<int *>(mlp->d_blob + mlp->off_flag);
    const bool defer = work->defer_range_flag != nullptr;        // a schedule that checks ONE sticky word at its end (adaptive.hip)
    if (defer) w.flag = work->defer_range_flag;

    int n_cu = 0;
    int rc = device_cu_count(&n_cu);
    if (rc) return rc;
    if (defer) {}                                                // (the schedule zeroed its word once)
    else if (work->d_flag) { if (!work->flag_clean) ICON_HIP(hipMemsetAsync(work->d_flag, 0, sizeof(int), st)); }    // (k_sign of this call cleared it)
    else if ((rc = mlp_flag_reset(mlp, st))) return rc;
    work->flag_clean = false;
    const MlpPlain plain = mlp_plain_of(mlp);
    const int64_t n_resc = std::min<int64_t>((N + 63) / 64, 2048);
    // calls of few points: the kernel variant that may cut its tiles in half (k_fused_f16x3<..., SMALL>; icon prior only); its
    // grid is sized for 128-point tiles.  A schedule level's N is n_max here - its real size lives on the device
    const bool small = prior == ICON_PRIOR_ICON && (work->q_n_dev != nullptr || N <= 262144);
    const int64_t ntiles = small ? (N + kTilePts / 2 - 1) / (kTilePts / 2) : (N + kTilePts - 1) / kTilePts;
    // one persistent workgroup per CU (LDS-bound: 132 KiB, all registers).  Multi-GPU: a collective's kernels cannot co-reside
    // with it on a CU - icon_work_set_reserve_cus leaves some CUs to them
    const unsigned grid = (unsigned)std::min<int64_t>(ntiles, std::max(n_cu - work->reserve_cus, 1));
    {
        // static runs + XCD-local pools: a workgroup's time depends on its XCD's clock under the power limit (and a little on its
        // planes: the clip band costs more per tile than the far field); the launch ends with the slowest one.  The last
        // steal_permille / 1000 of every workgroup's span is handed out in groups to whoever finishes first.
        const int per = (int)(ntiles / grid);
        const int pool_len = (int)((int64_t)per * work->steal_permille / 1000);
        if (!small && work->d_steal && pool_len > 0 && ntiles > (int64_t)grid) {
            G.steal = work->d_steal; G.steal_grp = std::min(work->steal_grp, kMaxStealGroup);
            G.steal_static = per - pool_len;
            G.steal_ngrp = (pool_len + 1 + G.steal_grp - 1) / G.steal_grp;     // covers the spans that are one tile longer
        }
    }
    if (G.clock) work->clock_grid = (int)std::min<unsigned>(grid, (unsigned)kMaxProfGrid);
#define ICON_FUSED(P, L_, ID, ...)                                                                                         \
    do {                                                                                                                   \
        if ((rc = once_per_device(ID, [] { return hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_f16x3<P, L_, ##__VA_ARGS__>),   \
                                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds); }))) return rc; \
        hipLaunchKernelGGL((k_fused_f16x3<P, L_, ##__VA_ARGS__>), dim3(grid), dim3(kF16Block), kFusedLds, st, G, d_occ, w); \
        debug_sync("k_fused_f16x3", st);                                                                                   \
        if (!defer) {                                                                                                      \
            hipLaunchKernelGGL((k_rescue_fused<P, L_>), dim3((unsigned)n_resc), dim3(64), 0, st, G, d_occ, plain, w.flag, rescue_always()); \
            debug_sync("k_rescue_fused", st);                                                                              \
        }                                                                                                                  \
    } while (0)
    if (small) { if (lattice) ICON_FUSED(ICON_PRIOR_ICON, true, 9, true); else ICON_FUSED(ICON_PRIOR_ICON, false, 10, true); }
    else if (prior == ICON_PRIOR_ICON) { if (lattice) ICON_FUSED(ICON_PRIOR_ICON, true, 0); else ICON_FUSED(ICON_PRIOR_ICON, false, 1); }
    else if (prior == ICON_PRIOR_PAMIR) { if (lattice) ICON_FUSED(ICON_PRIOR_PAMIR, true, 2); else ICON_FUSED(ICON_PRIOR_PAMIR, false, 3); }
    else { if (lattice) ICON_FUSED(ICON_PRIOR_PIFU, true, 4); else ICON_FUSED(ICON_PRIOR_PIFU, false, 5); }
#undef ICON_FUSED
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

}  // namespace icon
