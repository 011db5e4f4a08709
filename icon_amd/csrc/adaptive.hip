// adaptive.hip - the reference's coarse-to-fine schedule (Seg3dLossless._forward_faster, lib/common/seg3d_lossless.py:152-265,
// the mode apps/ICON.py:89 selects) as HIP kernels on one stream, no host read between the levels.
//
//   level 0        : the coarsest lattice, dense (the lattice kernels: one query() call over the whole 33^3 lattice)
//   level l >= 1   : occ_l   = trilinear upsample of occ_{l-1} (F.interpolate, align_corners=True: :201, :213-216)
//                    valid_l = upsample of (occ_{l-1} > balance); a voxel is a BOUNDARY voxel when 0 < valid < 1 (:217-218) -
//                              with res_l = 2 res_{l-1} - 1 every weight is 0, 1/2 or 1, so that is exactly "the (up to 8)
//                              parent corners around the voxel are not all on one side of the level" (integer logic)
//                    dilate the boundary by a 9^3 / 7^3 / 3^3 box (SmoothConv3D(k) > 0, lib/common/seg3d_utils.py:169-181:
//                    a box is separable - three 1-D passes), drop the voxels evaluated at an earlier level (coords_accum)
//                    - boundary, dilation and removal in ONE kernel on 64-bit rows in LDS (k_ad_mask) -,
//                    compact in the reference's order (is_boundary.permute(2,1,0).nonzero(): x slowest, z fastest, :236-240),
//                    query those points as ONE call (own outlier sign list, same order), scatter (:256-262)
//   last level     : upsample only ("last step no examine", :186-203)
//
// What makes the sparse levels fast: their points are lattice points in a band around the surface, so the exact
// nearest-triangle search runs as 4x4x4 PACKETS over the blocks of the level lattice that hold at least one candidate
// (k_ad_nearest, the traversal of k_nearest<lattice>) instead of one wavefront per scattered point; the per-point phases
// (inside test, sign list, feature rows + MLP) are the point-mode kernels of the ordinary query with the point count left on
// the device (icon_work::q_n_dev) and the search results found through the lattice index (NearRef::map).
// Masks live in [x][y][z] order (z fastest): the compaction's linear order IS the reference's point order.
#pragma clang fp contract(off)

#include "geom_device.h"

#include <algorithm>
#include <vector>

constexpr int kAdMaxLevels = 8;

struct icon_adaptive {
    int n_levels = 0;
    int res[kAdMaxLevels] = {0};
    float *occ[kAdMaxLevels] = {nullptr};      // level volumes [z][y][x] (the last one is the caller's buffer, not owned)
    uint8_t *P = nullptr;                      // (occ_{l-1} > balance) as bytes, [x][y][z]
    uint8_t *M1 = nullptr;                     // the level's candidate mask, [x][y][z]
    uint8_t *D[kAdMaxLevels] = {nullptr};      // voxels evaluated up to and including level l, [x][y][z] (level 0: all - not stored)
    int32_t *map = nullptr;                    // compacted candidates: [z][y][x] linear index, in the reference's order
    float *pts = nullptr;                      // their world positions [n][3]
    int32_t *blk_count = nullptr;              // compaction scratch: candidates per 256 voxels, behind them per 64 such blocks (k_ad_mask)
    int32_t *blk_list = nullptr;               // the 4x4x4 blocks holding a candidate (any order; count in counters[8])
    int *h_counters = nullptr;                 // host-mapped mirror of `counters`: the last upsample launch of a schedule writes it (no copy launch)
    int *h_counters_dev = nullptr;             //   ... its device address
    int *counters = nullptr;                   // device: [0..levels) points queried per level, [8] n_blocks, [9] any-positive flag of level 0, [10] range flag
    int *counters_base = nullptr;              // TWO sets of 16: a call uses the set the call before did not, and its last upsample launch zeroes
    int cur_set = 0;                           //   the other one for the call after it (no memset launch ahead of the first kernel)
    bool other_clean = true;                   // the set the next call will use is zero (false after a call that ended early)
    int64_t cap = 0;                           // voxels of the largest queried level
};

namespace icon {

void adaptive_destroy(icon_adaptive *a)
{
    if (!a) return;
    for (int l = 0; l + 1 < a->n_levels; ++l) (void)hipFree(a->occ[l]);
    for (int l = 1; l < kAdMaxLevels; ++l) (void)hipFree(a->D[l]);
    (void)hipFree(a->P); (void)hipFree(a->M1); (void)hipFree(a->map); (void)hipFree(a->pts);
    (void)hipFree(a->blk_count); (void)hipFree(a->blk_list); (void)hipHostFree(a->h_counters); (void)hipFree(a->counters_base);
    delete a;
}

namespace {

// ---- F.interpolate(mode='trilinear', align_corners=True): ATen's upsample_trilinear3d expression, term for term --------------
// One WAVEFRONT = one (z, y) row of the output: the row's weights and source rows are wave-uniform, the lanes take consecutive
// x (coalesced 256-byte stores; the 257^3 level is 68 MB of output).  History: a thread per voxel spent its time on index
// arithmetic (75 us for the 257^3 level); four consecutive x per thread shared the row's setup but stored 4 B at a 16 B stride.
// Since round 6 the launch also carries the per-voxel bookkeeping of the SOURCE level that used to be two more launches at the
// launch floor (k_ad_pbits after level 0, k_ad_next after every queried level): its first ceil(rp^3 / 256) workgroups write
// P = (src > balance) in [x][y][z] order for the mask kernel that follows, D = the voxels evaluated up to and including the
// source level, and level 0's any-positive flag; and it zeroes the two-level counters the mask kernel adds its candidates to.
struct AdBook {
    uint8_t *P;                    // [x][y][z] bits of the source level (null: not needed - the level that starts here is the last)
    uint8_t *D;                    // evaluated-so-far of the source level (null: the source is level 0 - every voxel - or no queried level follows)
    const uint8_t *C, *Dprev;      // the source level's candidate mask and the level before's D (null: level 0's: every voxel)
    int rpp;                       // resolution of the level before the source
    float balance;
    int *any_pos;                  // level 0 only: some voxel exceeds 0.5 (the reference returns None otherwise, :173-177)
    int32_t *zero;                 // blk_count ++ sup_count of the level that starts here
    int n_zero;
    // the LAST upsample launch of a schedule of >= 3 levels (every counter is final by then): mirror the 16 counters into the
    // host-mapped record the call reads after its synchronisation, and zero the set the NEXT call will count in
    const int *mirror_src;
    int *mirror_dst, *zero_set;
};
__device__ __forceinline__ int64_t xyz(int r, int x, int y, int z);
__global__ __launch_bounds__(256) void k_ad_up(const float *__restrict__ src, int rp, float *__restrict__ dst, int r, int *__restrict__ n_blocks, AdBook bk)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_blocks) *n_blocks = 0;     // the block list of the level that starts here (k_ad_mask fills it)
    if (blockIdx.x == 0 && threadIdx.x < 16) {
        if (bk.mirror_dst) bk.mirror_dst[threadIdx.x] = bk.mirror_src[threadIdx.x];
        if (bk.zero_set) bk.zero_set[threadIdx.x] = 0;
    }
    {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i < bk.n_zero) bk.zero[i] = 0;
        if (bk.P || bk.any_pos) {
            const bool live = i < (int64_t)rp * rp * rp;
            bool pos = false;
            if (live) {
                const int z = (int)(i % rp), y = (int)((i / rp) % rp), x = (int)(i / ((int64_t)rp * rp));
                const float v = src[((int64_t)z * rp + y) * rp + x];
                if (bk.P) bk.P[i] = v > bk.balance ? 1 : 0;
                pos = v > 0.5f;
                if (bk.D) {
                    bool d = bk.C[i] != 0;
                    if (!d && !((x | y | z) & 1)) d = bk.Dprev ? bk.Dprev[xyz(bk.rpp, x >> 1, y >> 1, z >> 1)] != 0 : true;
                    bk.D[i] = d ? 1 : 0;
                }
            }
            if (bk.any_pos && __any(pos) && (threadIdx.x & 63) == 0) atomicOr(bk.any_pos, 1);
        }
    }
    const int row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (row >= r * r) return;
    const int lane = threadIdx.x & 63;
    const int h2 = row % r, t2 = row / r;
    const float scale = (r > 1) ? (float)(rp - 1) / (float)(r - 1) : 0.0f;      // area_pixel_compute_scale, align_corners
    const float t1r = scale * t2, h1r = scale * h2;
    const int t1 = (int)t1r, h1 = (int)h1r;
    const int t1p = (t1 < rp - 1) ? 1 : 0, h1p = (h1 < rp - 1) ? 1 : 0;
    const float t1l = t1r - t1, t0l = 1.0f - t1l, h1l = h1r - h1, h0l = 1.0f - h1l;
    const float *r00 = src + ((int64_t)t1 * rp + h1) * rp, *r01 = src + ((int64_t)t1 * rp + h1 + h1p) * rp;
    const float *r10 = src + ((int64_t)(t1 + t1p) * rp + h1) * rp, *r11 = src + ((int64_t)(t1 + t1p) * rp + h1 + h1p) * rp;
    float *out = dst + (int64_t)row * r;
    for (int w2 = lane; w2 < r; w2 += 64) {
        const float w1r = scale * w2;
        const int w1 = (int)w1r;
        const int w1p = (w1 < rp - 1) ? 1 : 0;
        const float w1l = w1r - w1, w0l = 1.0f - w1l;
        out[w2] = t0l * (h0l * (w0l * r00[w1] + w1l * r00[w1 + w1p]) + h1l * (w0l * r01[w1] + w1l * r01[w1 + w1p])) +
                  t1l * (h0l * (w0l * r10[w1] + w1l * r10[w1 + w1p]) + h1l * (w0l * r11[w1] + w1l * r11[w1 + w1p]));
    }
}

// [x][y][z] index of voxel (x, y, z)
__device__ __forceinline__ int64_t xyz(int r, int x, int y, int z) { return ((int64_t)x * r + y) * r + z; }

// P[x][y][z] = occ[z][y][x] > balance; any_pos: the reference returns None when nothing exceeds 0.5 on the coarsest lattice (:173-177)
__global__ __launch_bounds__(256) void k_ad_pbits(const float *__restrict__ occ, int r, float balance, uint8_t *__restrict__ P, int *any_pos)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < (int64_t)r * r * r;
    bool pos = false;
    if (live) {
        const int z = (int)(i % r), y = (int)((i / r) % r), x = (int)(i / ((int64_t)r * r));
        const float v = occ[((int64_t)z * r + y) * r + x];
        P[i] = v > balance ? 1 : 0;
        pos = v > 0.5f;
    }
    if (any_pos && __any(pos) && (threadIdx.x & 63) == 0) atomicOr(any_pos, 1);
}

// The candidate mask of level l in ONE pass over 16 x 16 x 32 tiles (x, y, z) with a halo of 4: rows along z are 64-bit masks in
// LDS, so the boundary test, the three 1-D passes of the box dilation and the removal of the voxels evaluated before are a few
// word operations per row (four separate kernels over bytes before: 37 us per level, most of it launch floor and strided bytes).
//   boundary(v)  = the 8 parent corners of v (P at [c >> 1] and [(c >> 1) + (c & 1)] per axis) are not all equal
//   dilation     = box of radius `rad` per axis, zeros outside the lattice (SmoothConv3D's padding, seg3d_lossless.py:105-112)
//   done(v)      = all coordinates even and D_prev[v / 2] (level 1: every such voxel) - coords_accum * 2, seg3d_lossless.py:230-234
constexpr int kMaskTX = 16, kMaskTY = 16, kMaskTZ = 32, kMaskH = 4;
constexpr int kMaskRX = kMaskTX + 2 * kMaskH, kMaskRY = kMaskTY + 2 * kMaskH, kMaskRZ = kMaskTZ + 2 * kMaskH;      // 24 x 24 x 40
constexpr int kMaskPX = kMaskRX / 2 + 2, kMaskPY = kMaskRY / 2 + 2, kMaskPZ = kMaskRZ / 2 + 2;                   // parents: 14 x 14 x 22
__device__ __forceinline__ unsigned long long dup_bits(unsigned long long x)      // bit q -> bits 2q and 2q + 1 (q < 32)
{
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}
// Also lists the P x P x P blocks of the search that hold a candidate (shift = log2 P; tiles are whole blocks): one atomic per
// tile reserves the tile's run of the list (any order: the blocks are independent).
// ... and counts the candidates per 256 voxels of the LINEAR [x][y][z] order (the compaction's blocks) and per 64 such blocks:
// a core row of the tile is 32 consecutive voxels of that order - at most two blocks - so a thread adds its row's popcounts to one
// or two counters of each level (zeroed by k_ad_up); k_ad_scatter sums the counters before its block itself.  (Until round 5:
// k_ad_count + k_ad_scan, two more launches at the launch floor per level.)
constexpr int kAdSupShift = 7;                                  // 128 blocks = 32,768 voxels per second-level counter: the 16 rows a tile
                                                                // holds for one x span 16 r + 32 voxels of the order - at most TWO of them
                                                                // for every r the entry admits, so a tile pre-adds them in LDS (cs) and issues at
                                                                // most 32 global atomics on the few, hot second-level counters
__global__ __launch_bounds__(256) void k_ad_mask(const uint8_t *__restrict__ P, int rp, const uint8_t *__restrict__ Dprev, uint8_t *__restrict__ C, int r, int rad,
                                                int shift, int nbk, int32_t *__restrict__ blk_list, int *__restrict__ n_blocks,
                                                int32_t *__restrict__ blk_count, int32_t *__restrict__ sup_count)
{
    __shared__ int wcnt[4], lbase, cs[2 * kMaskTX];
    __shared__ unsigned pm[kMaskPX * kMaskPY];                  // parent rows: bit c = P[x][y][pz0 + c]
    __shared__ unsigned long long ma[kMaskRX * kMaskRY], mb[kMaskRX * kMaskRY];
    const int ntx = (r + kMaskTX - 1) / kMaskTX, nty = (r + kMaskTY - 1) / kMaskTY;
    const int tx = blockIdx.x % ntx, ty = (blockIdx.x / ntx) % nty, tz = blockIdx.x / (ntx * nty);
    const int ox = tx * kMaskTX - kMaskH, oy = ty * kMaskTY - kMaskH, oz = tz * kMaskTZ - kMaskH;      // region origin (even)
    const int px0 = ox >> 1, py0 = oy >> 1, pz0 = oz >> 1;       // (arithmetic shifts: -4 -> -2)
    const int t = threadIdx.x;
    if (t < 2 * kMaskTX) cs[t] = 0;
    if (t < kMaskPX * kMaskPY) {
        const int a = t / kMaskPY, b = t % kMaskPY;
        const int x = px0 + a, y = py0 + b;
        unsigned m = 0;
        if (x >= 0 && x < rp && y >= 0 && y < rp) {
            const uint8_t *row = P + ((int64_t)x * rp + y) * rp;
            for (int c = 0; c < kMaskPZ; ++c) { const int z = pz0 + c; if (z >= 0 && z < rp && row[z]) m |= 1u << c; }
        }
        pm[t] = m;
    }
    __syncthreads();
    // boundary rows; bit k of a region row = voxel z = oz + k
    const int zlo = max(0, -oz), zhi = min(kMaskRZ, r - oz);     // region z inside the lattice: [zlo, zhi)
    const unsigned long long zmask = zhi > zlo ? ((zhi >= 64 ? ~0ull : ((1ull << zhi) - 1ull)) & ~((1ull << zlo) - 1ull)) : 0ull;
    for (int q = t; q < kMaskRX * kMaskRY; q += 256) {
        const int i = q / kMaskRY, j = q % kMaskRY;
        const int x = ox + i, y = oy + j;
        unsigned long long bm = 0;
        if (x >= 0 && x < r && y >= 0 && y < r) {
            const int a0 = (x >> 1) - px0, a1 = a0 + (x & 1), b0 = (y >> 1) - py0, b1 = b0 + (y & 1);
            const unsigned m00 = pm[a0 * kMaskPY + b0], m01 = pm[a0 * kMaskPY + b1], m10 = pm[a1 * kMaskPY + b0], m11 = pm[a1 * kMaskPY + b1];
            const unsigned long long o = dup_bits(m00 | m01 | m10 | m11), n = dup_bits(m00 & m01 & m10 & m11);
            // voxel k: parents k >> 1 and (k + 1) >> 1 = bits k and k + 1 of the duplicated masks
            bm = ((o | (o >> 1)) & ~(n & (n >> 1))) & zmask;
        }
        unsigned long long mz = bm;                              // z pass
        for (int d = 1; d <= rad; ++d) mz |= (bm << d) | (bm >> d);
        ma[q] = mz & zmask;
    }
    __syncthreads();
    for (int q = t; q < kMaskRX * kMaskRY; q += 256) {           // y pass
        const int i = q / kMaskRY, j = q % kMaskRY;
        unsigned long long m = 0;
        for (int d = -rad; d <= rad; ++d) if (j + d >= 0 && j + d < kMaskRY) m |= ma[i * kMaskRY + j + d];
        mb[q] = m;
    }
    __syncthreads();
    {                                                            // x pass: one core row per thread
        const int i = kMaskH + t / kMaskTY, j = kMaskH + t % kMaskTY;
        unsigned long long m = 0;
        for (int d = -rad; d <= rad; ++d) m |= mb[(i + d) * kMaskRY + j];
        ma[i * kMaskRY + j] = m;                                 // (core rows of `ma` are read by the y pass only, which is over)
    }
    __syncthreads();
    {                                                            // drop what was evaluated before (this thread's own row)
        const int i = kMaskH + t / kMaskTY, j = kMaskH + t % kMaskTY;
        const int x = ox + i, y = oy + j;
        unsigned long long m = (x < r && y < r) ? ma[i * kMaskRY + j] : 0ull;
        if (m && !((x | y) & 1)) {
            unsigned long long dm = 0;
            if (!Dprev) dm = 0x5555555555555555ull;              // region bit k = voxel z = oz + k, oz even
            else {
                const uint8_t *drow = Dprev + ((int64_t)(x >> 1) * rp + (y >> 1)) * rp;
                for (int k = kMaskH; k < kMaskH + kMaskTZ; k += 2) { const int z = oz + k; if (z < r && drow[z >> 1]) dm |= 1ull << k; }
            }
            m &= ~dm;
        }
        ma[i * kMaskRY + j] = m;
        // the row's candidates, counted into the compaction's blocks
        const int z0 = oz + kMaskH;
        unsigned core = (unsigned)(m >> kMaskH);
        if (z0 + 32 > r) core &= (z0 < r) ? ((1u << (r - z0)) - 1u) : 0u;
        if (core) {
            const int64_t l0 = xyz(r, x, y, z0);
            const int first = 256 - (int)(l0 & 255);             // voxels of the row that fall into the first block
            const int c1 = first >= 32 ? __popc(core) : __popc(core & ((1u << first) - 1u));
            const int c2 = __popc(core) - c1;
            const int b0 = (int)(l0 >> 8);
            const int s_first = (int)(xyz(r, x, oy + kMaskH, z0) >> (8 + kAdSupShift));      // of this x's first core row
            int *sx = cs + 2 * (i - kMaskH);
            if (c1) { atomicAdd(blk_count + b0, c1); atomicAdd(sx + ((b0 >> kAdSupShift) - s_first), c1); }
            if (c2) { atomicAdd(blk_count + b0 + 1, c2); atomicAdd(sx + (((b0 + 1) >> kAdSupShift) - s_first), c2); }
        }
    }
    __syncthreads();
    if (t < 2 * kMaskTX && cs[t]) {
        const int x = ox + kMaskH + (t >> 1);
        atomicAdd(sup_count + (int)(xyz(r, x, oy + kMaskH, oz + kMaskH) >> (8 + kAdSupShift)) + (t & 1), cs[t]);
    }
    // store: 32 consecutive z per 32 lanes
    for (int it = 0; it < 32; ++it) {
        const int row = it * 8 + (t >> 5), k = t & 31;
        const int i = kMaskH + row / kMaskTY, j = kMaskH + row % kMaskTY;
        const int x = ox + i, y = oy + j, z = oz + kMaskH + k;
        if (x >= r || y >= r || z >= r) continue;
        C[xyz(r, x, y, z)] = (uint8_t)((ma[i * kMaskRY + j] >> (kMaskH + k)) & 1ull);
    }
    if (!blk_list) return;
    // the search's blocks with a candidate
    const int bs = 1 << shift, nbx = kMaskTX >> shift, nby = kMaskTY >> shift, nbz = kMaskTZ >> shift;
    for (int b0 = 0; b0 < nbx * nby * nbz; b0 += 256) {
        const int b = b0 + t;
        bool has = false;
        int id = 0;
        if (b < nbx * nby * nbz) {
            const int bz = b % nbz, by = (b / nbz) % nby, bx = b / (nbz * nby);
            unsigned long long m = 0;
            for (int di = 0; di < bs; ++di)
                for (int dj = 0; dj < bs; ++dj) m |= ma[(kMaskH + bx * bs + di) * kMaskRY + kMaskH + by * bs + dj];
            has = ((m >> (kMaskH + bz * bs)) & ((1ull << bs) - 1ull)) != 0;
            id = ((((oz + kMaskH) >> shift) + bz) * nbk + (((oy + kMaskH) >> shift) + by)) * nbk + (((ox + kMaskH) >> shift) + bx);
        }
        const unsigned long long bal = __ballot(has);
        const int lane = t & 63, w = t >> 6;
        __syncthreads();                                         // (lbase / wcnt of the round before have been read)
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        if (t == 0) { const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]; lbase = tot ? atomicAdd(n_blocks, tot) : 0; }
        __syncthreads();
        if (has) {
            int at = lbase + __popcll(bal & ((1ull << lane) - 1ull));
            for (int q = 0; q < w; ++q) at += wcnt[q];
            blk_list[at] = id;
        }
    }
}

// compaction in linear ([x][y][z]) order = the reference's nonzero() order.  A workgroup = one block of 256 voxels; its offset =
// the second-level counters before its group of 64 blocks + the block counters before it inside the group, summed by the
// workgroup itself (a few hundred ints from L2 - the pattern of k_outlier_small); workgroup 0 also publishes the level's total.
__global__ __launch_bounds__(256) void k_ad_scatter(const uint8_t *__restrict__ C, int r, const int32_t *__restrict__ blk_count, const int32_t *__restrict__ sup_count,
                                                   int n_sup, int32_t *__restrict__ map, float *__restrict__ pts, int *__restrict__ total)
{
    __shared__ int ws[4], red[2][4];
    const int64_t n = (int64_t)r * r * r;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int my_sup = (int)(blockIdx.x >> kAdSupShift);
    const bool want_all = blockIdx.x == 0;
    int before = 0, all = 0;
    for (int k = threadIdx.x; k < (want_all ? n_sup : my_sup); k += 256) { const int t = sup_count[k]; all += t; if (k < my_sup) before += t; }
    {
        const int k = (my_sup << kAdSupShift) + (int)threadIdx.x;
        if (k < (int)blockIdx.x) before += blk_count[k];
    }
    for (int d = 32; d >= 1; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    const bool c = i < n && C[i];
    const unsigned long long b = __ballot(c);
    if (lane == 0) { ws[w] = __popcll(b); red[0][w] = before; red[1][w] = all; }
    __syncthreads();
    if (want_all && threadIdx.x == 0) *total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (!c) return;
    int k = red[0][0] + red[0][1] + red[0][2] + red[0][3] + __popcll(b & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) k += ws[q];
    const int z = (int)(i % r), y = (int)((i / r) % r), x = (int)(i / ((int64_t)r * r));
    map[k] = (int32_t)(((int64_t)z * r + y) * r + x);
    const f3 p = lattice_world(r, x, y, z);          // == batch_eval's mapping of (coords * stride), bit for bit (quotients of the same rationals)
    pts[3 * (int64_t)k] = p.x; pts[3 * (int64_t)k + 1] = p.y; pts[3 * (int64_t)k + 2] = p.z;
}

// the exact nearest triangle of the CANDIDATES of the listed P x P x P blocks (the other lanes are parked: the packet walks the
// union of its lanes' searches, and a block holds ~15 candidates of 64 points): the packet traversal of k_nearest<lattice>,
// one block per WORKGROUP (a grid that fills the wave slots strides over the list), its walk shared by the NW wavefronts (nearest_shared; NW == 1: one block per wavefront).
// A level's few thousand blocks do not fill the wave slots: with one wavefront per block the launch lasted as long as its
// longest walk (4^3 blocks: 390 us per level; 2^3 blocks, 8 of 64 lanes at work: 270 us; shared 4^3 blocks: ~100 us).
template <int P, int NW>
__global__ __launch_bounds__(NW == 1 ? 256 : NW * 64) void k_ad_nearest(MeshDev m, int r, int nbk, const int32_t *__restrict__ list,
                                                                       const int *__restrict__ n_list, const uint8_t *__restrict__ C, NearRef near, float sdf_clip, ShareDbg dbg)
{
    constexpr int kWaves = NW == 1 ? 4 : NW;
    __shared__ int lds[kWaves * kStackDepth];
    __shared__ __attribute__((aligned(16))) char smem[share_lds_bytes(NW)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = *n_list;
    for (int b = NW == 1 ? blockIdx.x * 4 + wave : blockIdx.x; b < nb; b += NW == 1 ? gridDim.x * 4 : gridDim.x) {
        const int e = list[b];
        const int bx = e % nbk, by = (e / nbk) % nbk, bz = e / (nbk * nbk);
        const bool used = lane < P * P * P;
        const int l = used ? lane : 0;
        const int ix = bx * P + l % P, iy = by * P + (l / P) % P, iz = bz * P + l / (P * P);
        const int cx = min(ix, r - 1), cy = min(iy, r - 1), cz = min(iz, r - 1);
        const bool live = used && ix < r && iy < r && iz < r && C[xyz(r, cx, cy, cz)] != 0;
        const f3 p = lattice_world(r, cx, cy, cz);
        Nearest nr;
        if (NW == 1) nr = nearest_packet(m, p, live, lds + wave * kStackDepth, nullptr, nullptr, INFINITY, nullptr, P == 4 ? 21 : 0);
        else nr = nearest_shared<NW>(m, p, live, lds + wave * kStackDepth, smem, P == 4 ? 21 : 0, dbg);
        if (live && (NW == 1 || wave == 0)) store_near(near, ((int64_t)cz * r + cy) * r + cx, nr, sdf_clip);
    }
}

template <class T>
int grow(T **p, size_t n)
{
    (void)hipFree(*p); *p = nullptr;
    ICON_HIP(hipMalloc((void **)p, n * sizeof(T)));
    return ICON_OK;
}

}  // namespace
}  // namespace icon

using namespace icon;

// ---------------------------------------------------------------------------------------------------------------------------
// C ABI: the whole schedule.  resolutions[n_levels] ascending, odd, each 2 r - 1 of the one before; the standard box
// (b_min = [-1,1,-1], b_max = [1,-1,1]), align_corners=True, identity calibration - the reconEngine of apps/ICON.py:78-90.
// d_out [res_last^3] receives the final volume ([z][y][x]); h_counts (optional, host) is filled AFTER a stream
// synchronisation with the points queried per level and h_counts[n_levels] = 1 when some voxel of the coarsest level
// exceeds 0.5 (else the reference returns None) - pass NULL to stay asynchronous and read them later with
// icon_adaptive_counts.
// ---------------------------------------------------------------------------------------------------------------------------
static int adaptive_eval_impl(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp, int prior_type, float sdf_clip,
                              int cmap_mode, const int *resolutions, int n_levels, float balance, float *d_out, int64_t *h_counts,
                              int search, int precision, icon_work_t *work, void *stream, bool defer_rescue);

// The range safety net of the split-precision MLP (k_rescue_fused behind every fused launch: a 5 us launch that reads one word) is
// DEFERRED when the call synchronises for its counts anyway: the schedule's fused kernels raise one sticky word, the host reads
// it with the counters, and only a raised word - operands beyond the f16 range, which no shipped checkpoint produces - runs the
// schedule a second time the per-launch way.  Asynchronous calls (h_counts == NULL) keep the per-launch rescue.
extern "C" int icon_adaptive_eval(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp, int prior_type, float sdf_clip,
                                  int cmap_mode, const int *resolutions, int n_levels, float balance, float *d_out, int64_t *h_counts,
                                  int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(work != nullptr, "icon_adaptive_eval: null argument");
    const bool defer = h_counts != nullptr && !rescue_always();
    int rc = adaptive_eval_impl(mesh, feat, mlp, prior_type, sdf_clip, cmap_mode, resolutions, n_levels, balance, d_out, h_counts, search, precision,
                                work, stream, defer);
    if (rc == ICON_OK && defer && work->ad && work->ad->h_counters[10] != 0) {
        ++work->range_reruns;
        rc = adaptive_eval_impl(mesh, feat, mlp, prior_type, sdf_clip, cmap_mode, resolutions, n_levels, balance, d_out, h_counts, search, precision,
                                work, stream, false);
    }
    return rc;
}

// how many schedules on this workspace were run a second time because a fused kernel met operands beyond the f16 range
extern "C" int icon_adaptive_reruns(icon_work_t *work, int *n)
{
    ICON_ARG(work != nullptr && n != nullptr, "icon_adaptive_reruns: null argument");
    *n = work->range_reruns;
    return ICON_OK;
}

static int adaptive_eval_impl(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp, int prior_type, float sdf_clip,
                              int cmap_mode, const int *resolutions, int n_levels, float balance, float *d_out, int64_t *h_counts,
                              int search, int precision, icon_work_t *work, void *stream, bool defer_rescue)
{
    ICON_ARG(feat && mlp && work && resolutions && d_out, "icon_adaptive_eval: null argument");
    ICON_ARG(n_levels >= 1 && n_levels <= kAdMaxLevels, "icon_adaptive_eval: 1..8 levels");
    ICON_ARG(prior_type != ICON_PRIOR_ICON || mesh != nullptr, "icon_adaptive_eval: the icon prior needs a mesh");
    if (precision != ICON_PRECISION_F16X3 || search != ICON_SEARCH_BVH)
        return fail(ICON_ERR_UNSUPPORTED, "icon_adaptive_eval: the fused f16x3 path with the BVH search only (other settings: the host-driven schedule)");
    if (work->tie_rule != 0) return fail(ICON_ERR_UNSUPPORTED, "icon_adaptive_eval: diagnostics tie rule set on this workspace");
    for (int l = 0; l < n_levels; ++l) {
        ICON_ARG(resolutions[l] >= 3 && (resolutions[l] & 1), "icon_adaptive_eval: resolutions must be odd and >= 3 (seg3d_lossless.py:84-86)");
        ICON_ARG(l == 0 || resolutions[l] == 2 * resolutions[l - 1] - 1, "icon_adaptive_eval: every level must be 2 r - 1 of the one before");
    }
    ICON_ARG((int64_t)resolutions[n_levels - 1] * resolutions[n_levels - 1] * resolutions[n_levels - 1] < (1ll << 31), "icon_adaptive_eval: lattice too large");
    // (r^3 < 2^31 => r <= 1290: the 16 core rows of one x span 15 r + 32 voxels of the linear order)
    static_assert(15 * 1290 + 32 <= (256 << kAdSupShift), "k_ad_mask: a tile's rows of one x must span at most two second-level counters");
    hipStream_t st = (hipStream_t)stream;
    int rc;

    // ---- level buffers (allocated once per schedule) ----------------------------------------------------------------
    icon_adaptive *a = work->ad;
    bool same = a && a->n_levels == n_levels;
    for (int l = 0; same && l < n_levels; ++l) same = a->res[l] == resolutions[l];
    if (!same) {
        // built in a local and published only when every allocation succeeded: a failure half-way must not leave a record behind
        // whose resolutions match the next call's (it would skip this block and launch on null buffers)
        adaptive_destroy(a);
        work->ad = nullptr;
        a = new icon_adaptive();
        a->n_levels = n_levels;
        for (int l = 0; l < n_levels; ++l) a->res[l] = resolutions[l];
        const int rq = n_levels >= 2 ? resolutions[n_levels - 2] : resolutions[0];      // the largest level that is queried / masked
        a->cap = (int64_t)rq * rq * rq;
        rc = ICON_OK;
        for (int l = 0; !rc && l + 1 < n_levels; ++l) {
            const size_t n = (size_t)resolutions[l] * resolutions[l] * resolutions[l];
            rc = grow(&a->occ[l], n);
            if (!rc && l >= 1) rc = grow(&a->D[l], n);
        }
        const int nbk = (rq + 1) / 2;                            // blocks of the finest granularity used (2^3)
        if (!rc) rc = grow(&a->P, (size_t)a->cap);
        if (!rc) rc = grow(&a->M1, (size_t)a->cap);
        if (!rc) rc = grow(&a->map, (size_t)a->cap);
        if (!rc) rc = grow(&a->pts, (size_t)a->cap * 3);
        if (!rc) rc = grow(&a->blk_count, (size_t)(a->cap + 255) / 256 + (size_t)((a->cap + 255) / 256 >> kAdSupShift) + 2);
        if (!rc) rc = grow(&a->blk_list, (size_t)nbk * nbk * nbk);
        if (!rc) rc = grow(&a->counters_base, (size_t)32);
        if (!rc && hipMemset(a->counters_base, 0, 32 * sizeof(int)) != hipSuccess) rc = fail(ICON_ERR_HIP, "icon_adaptive_eval: hipMemset of the counters failed");
        if (!rc && hipHostMalloc((void **)&a->h_counters, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
            rc = fail(ICON_ERR_HIP, "icon_adaptive_eval: hipHostMalloc of the counter mirror failed");
        if (!rc && hipHostGetDevicePointer((void **)&a->h_counters_dev, a->h_counters, 0) != hipSuccess)
            rc = fail(ICON_ERR_HIP, "icon_adaptive_eval: no device address for the counter mirror");
        a->counters = a->counters_base; a->cur_set = 0; a->other_clean = true;
        if (rc) { adaptive_destroy(a); return rc; }
        work->ad = a;
    }
    a->occ[n_levels - 1] = d_out;
    // this call counts in the set the call before did not use; that set was zeroed by the last upsample launch of the call before
    // (or has never been used) - unless that call ended early
    a->cur_set ^= 1;
    a->counters = a->counters_base + 16 * a->cur_set;
    if (!a->other_clean) ICON_HIP(hipMemsetAsync(a->counters, 0, 16 * sizeof(int), st));
    a->other_clean = false;
    int *const other_set = a->counters_base + 16 * (a->cur_set ^ 1);
    bool mirrored = false;
    // whatever way this function is left, the workspace must not keep pointing at a level's compaction buffers: a later point
    // query on it would read them as ITS map
    struct QMapGuard { icon_work *w; ~QMapGuard() { w->q_map = nullptr; w->q_n_dev = nullptr; w->defer_range_flag = nullptr; } } q_guard{work};
    work->defer_range_flag = defer_rescue ? a->counters + 10 : nullptr;

    // ---- level 0: the coarsest lattice, dense, ONE call ---------------------------------------------------------------
    const int r0 = resolutions[0];
    float *occ0 = a->occ[0];
    work->q_map = nullptr; work->q_n_dev = nullptr;
    if ((rc = icon_grid_eval_slab(mesh, feat, mlp, prior_type, sdf_clip, cmap_mode, r0, 0, r0, occ0, search, precision, work, stream))) return rc;
    if (n_levels == 1) {                                        // no upsample launch to carry the any-positive test
        const int64_t n0 = (int64_t)r0 * r0 * r0;
        hipLaunchKernelGGL(k_ad_pbits, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, st, occ0, r0, balance, (uint8_t *)nullptr, a->counters + 9);
    }

    const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int l = 1; l < n_levels; ++l) {
        const int r = resolutions[l], rp = resolutions[l - 1];
        const int64_t n = (int64_t)r * r * r;
        const unsigned nbv = (unsigned)((n + 255) / 256);
        // the upsample launch also books the source level: its bits P (needed when this level is examined), D of the source level
        // (needed when the source level was a queried one: l - 1 >= 1), level 0's any-positive flag, and the zeroed counters
        const bool examined = l < n_levels - 1;
        const int64_t nsrc = (int64_t)rp * rp * rp;
        const int nb_l = (int)nbv, nsup_l = (nb_l >> kAdSupShift) + 1;
        AdBook bk{};
        bk.P = examined ? a->P : nullptr;
        bk.D = (examined && l >= 2) ? a->D[l - 1] : nullptr;
        bk.C = a->M1;                                           // still the source level's mask (k_ad_mask of THIS level rewrites it later)
        bk.Dprev = l >= 3 ? a->D[l - 2] : nullptr;
        bk.rpp = l >= 2 ? resolutions[l - 2] : 0;
        bk.balance = balance;
        bk.any_pos = l == 1 ? a->counters + 9 : nullptr;
        bk.zero = examined ? a->blk_count : nullptr;
        bk.n_zero = examined ? nb_l + nsup_l : 0;
        if (l == n_levels - 1) {
            bk.zero_set = other_set;                            // (the set of the call before: nobody reads it any more)
            a->other_clean = true;
            if (n_levels >= 3 && h_counts) { bk.mirror_src = a->counters; bk.mirror_dst = a->h_counters_dev; mirrored = true; }
        }
        const int64_t book = (bk.P || bk.any_pos) ? (nsrc + 255) / 256 : 0;
        const int64_t grid_up = std::max<int64_t>(std::max<int64_t>(((int64_t)r * r + 3) / 4, book), (bk.n_zero + 255) / 256);
        hipLaunchKernelGGL(k_ad_up, dim3((unsigned)grid_up), dim3(256), 0, st, a->occ[l - 1], rp, a->occ[l], r,
                           examined ? a->counters + 8 : (int *)nullptr, bk);     // (the last level starts no block list - and the mirror must not see a reset)
        if (!examined) break;                                   // "last step no examine": interpolate only
        // P holds (occ_{l-1} > balance): level 0's from above, later levels' from the end of the previous iteration
        const int rad = l == 1 ? 4 : (l == 2 ? 3 : 1);           // SmoothConv3D 9 / 7 / 3 (seg3d_lossless.py:105-112, 219-226)
        const unsigned nt = (unsigned)((r + kMaskTX - 1) / kMaskTX) * (unsigned)((r + kMaskTY - 1) / kMaskTY) * (unsigned)((r + kMaskTZ - 1) / kMaskTZ);
        uint8_t *C = a->M1;
        const bool icon_prior = prior_type == ICON_PRIOR_ICON;
        const int P = lattice_packet();
        const int shift = P == 4 ? 2 : 1;
        const int nbk = (r + P - 1) / P;
        hipLaunchKernelGGL(k_ad_mask, dim3(nt), dim3(256), 0, st, a->P, rp, (const uint8_t *)(l >= 2 ? a->D[l - 1] : nullptr), C, r, rad,
                           shift, nbk, icon_prior ? a->blk_list : (int32_t *)nullptr, a->counters + 8, a->blk_count, a->blk_count + nb_l);
        hipLaunchKernelGGL(k_ad_scatter, dim3(nbv), dim3(256), 0, st, C, r, a->blk_count, a->blk_count + nb_l, nsup_l, a->map, a->pts, a->counters + l);
        ICON_HIP(hipGetLastError());
        debug_sync("adaptive: upsample + boundary + dilate + compact", st);
        // ---- the level's query: ONE call over the compacted points (count on the device) ---------------------------------
        if ((rc = ensure_work_points(work, std::max<int64_t>(n, 1)))) return rc;      // near arrays by lattice index, codes / signs by call index
        work->q_map = a->map; work->q_n_dev = a->counters + l;
        Calib cal;
        memcpy(cal.m, ident, sizeof(ident)); cal.d = nullptr;
        const bool icon = icon_prior;
        const bool needs_list = icon && cmap_mode == ICON_CMAP_REFERENCE && (feat->dev.smpl_mask & kSmplCmap);
        if (icon) {
            if (mesh->F > kNearLoSlots && work->cap_points_hi < work->cap_points) {
                (void)hipFree(work->d_near_hi); work->d_near_hi = nullptr; work->cap_points_hi = 0;
                ICON_HIP(hipMalloc((void **)&work->d_near_hi, (size_t)work->cap_points));
                work->cap_points_hi = work->cap_points;
            }
            NearRef raw = work_near(work, mesh);
            raw.map = nullptr;                                   // the search writes by lattice index
            int n_cu = 0;
            if ((rc = device_cu_count(&n_cu))) return rc;
            // the number of blocks is known on the device only: a grid that fills the wave slots, every workgroup loops
            const int share_env = share_waves_override();        // diagnostics: 1 = one wave per block
            // wavefronts per shared walk by level: the candidates of a level grow ~4x with the resolution, the walks get shorter -
            // measured per level of the [33 .. 513] schedule (us, NW = 16 / 8 / 4 / 1): 65^3 75 / 97 / 160 / 462, 129^3 183 / 123 /
            // 142 / 311, 257^3 381 / 221 / 150 / 196 (tools/share_probe.sh): few long walks want many waves each, many walks few
            const int by_level = r <= 65 ? kShareWavesFew : (r <= 129 ? kShareWavesMany : 4);
            const int nw = (share_env == 1 || share_env == 4 || share_env == 8 || share_env == 16) ? share_env : by_level;
            ShareDbg dbg{};
            if ((rc = work_share_dbg(work, &dbg))) return rc;
#define ICON_AD_NEAREST(PP, NW) hipLaunchKernelGGL((k_ad_nearest<PP, NW>), dim3((unsigned)(n_cu * 32 / (NW == 1 ? 4 : NW))), dim3(NW == 1 ? 256 : NW * 64), 0, st, \
                                                   mesh->dev, r, nbk, a->blk_list, a->counters + 8, C, raw, sdf_clip, dbg)
            if (P == 4 && nw == 16) ICON_AD_NEAREST(4, 16);
            else if (P == 4 && nw == 8) ICON_AD_NEAREST(4, 8);
            else if (P == 4 && nw == 4) ICON_AD_NEAREST(4, 4);
            else if (P == 4) ICON_AD_NEAREST(4, 1);
            else if (nw == 16) ICON_AD_NEAREST(2, 16);
            else if (nw == 8) ICON_AD_NEAREST(2, 8);
            else ICON_AD_NEAREST(2, 1);
#undef ICON_AD_NEAREST
            ICON_HIP(hipGetLastError());
            debug_sync("adaptive: k_ad_nearest", st);
            if ((rc = launch_sign(mesh, cal, r, 0, a->pts, n, sdf_clip, work, false, st))) return rc;
            if (needs_list && (rc = outlier_list_dev(work, a->counters + l, n, st))) return rc;
        }
        FusedSigns fs{};
        fs.mode = needs_list ? kSignSelf : kSignNone;
        fs.list = work->d_signs; fs.k_dev = work->d_total;
        LatticeMap L{};
        rc = launch_fused_f16x3(mesh, feat, mlp, prior_type, cal, L, 0, 0, a->pts, n, sdf_clip, cmap_mode == ICON_CMAP_LOCAL ? 1 : 0, work, fs,
                                a->occ[l], false, st);
        work->q_map = nullptr; work->q_n_dev = nullptr;
        if (rc) return rc;
        debug_sync("adaptive: sign + list + fused", st);
        // (the bookkeeping for the next level - D, P - rides on the next level's upsample launch: k_ad_up)
        ICON_HIP(hipGetLastError());
    }
    ICON_HIP(hipGetLastError());
    if (!a->other_clean) {                                      // one level only: no upsample launch to carry the zeroing
        ICON_HIP(hipMemsetAsync(other_set, 0, 16 * sizeof(int), st));
        a->other_clean = true;
    }
    if (h_counts) {
        int *host = a->h_counters;                              // host-mapped: written by the last upsample launch; else one direct copy
        if (!mirrored) ICON_HIP(hipMemcpyAsync(host, a->counters, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
        ICON_HIP(hipStreamSynchronize(st));
        for (int l = 0; l < n_levels; ++l) h_counts[l] = host[l];
        h_counts[0] = (int64_t)r0 * r0 * r0;                    // level 0 evaluates every voxel
        h_counts[n_levels] = host[9];
        if ((rc = work_check_err(work))) return rc;             // the stream is idle: the verdict on THIS schedule's shared walks
    }
    return ICON_OK;
}

// the counters of the most recent icon_adaptive_eval on this workspace (synchronises the stream): points queried per level,
// then 1 / 0 for "some voxel of the coarsest level exceeds 0.5"
extern "C" int icon_adaptive_counts(icon_work_t *work, int n_levels, int64_t *h_counts, void *stream)
{
    ICON_ARG(work && work->ad && h_counts && n_levels == work->ad->n_levels, "icon_adaptive_counts: no matching icon_adaptive_eval on this workspace");
    int *host = work->ad->h_counters;
    ICON_HIP(hipMemcpyAsync(host, work->ad->counters, 16 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ICON_HIP(hipStreamSynchronize((hipStream_t)stream));
    for (int l = 0; l < n_levels; ++l) h_counts[l] = host[l];
    h_counts[0] = (int64_t)work->ad->res[0] * work->ad->res[0] * work->ad->res[0];
    h_counts[n_levels] = host[9];
    return work_check_err(work);
}
