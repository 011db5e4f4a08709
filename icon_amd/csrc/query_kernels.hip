// query_kernels.hip - gfx950 kernels for the per-point geometry/feature half of HGPIFuNet.query:
//   nearest triangle (BVH2, per-lane LDS stack, or LDS-tiled brute force), +x ray parity over
//   (y,z) bins, Heidrich barycentric interpolation, outlier clipping, bilinear / trilinear
//   feature gather + front/back select, and assembly of the 16-float MLP input rows.
//
// Reference being replaced (paths relative to the reference root):
//   lib/dataset/mesh_util.py:357-396 (cal_sdf_batch), :319-354 (barycentrics), :266-277 (feat_select)
//   lib/net/HGPIFuNet.py:268-365 (query), lib/net/geometry.py:21-61 (index, orthogonal)
//
// Arithmetic spec: float32, the only fused operations are the explicit fmaf() calls,
// IEEE division / sqrt (hipcc default: correctly rounded).  This file is compiled with
// -ffp-contract=off; the pragma below is a second line of defence.
#pragma clang fp contract(off)

#include "geom_device.h"

namespace icon {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// cal_sdf_batch for explicit points (API parity with the reference function; also the unit-test
// surface for the geometry kernels)
template <bool BRUTE>
__global__ __launch_bounds__(kBlock) void k_sdf_query(MeshDev m, const float *__restrict__ pts, int64_t N,
                                                      float *sdf, float *nrm, float *cm, float *vis,
                                                      int64_t *face, uint8_t *inside_out, const int32_t *__restrict__ perm)
{
    __shared__ int lds[kBruteTile * 24];   // brute: TriPre tile; packet: 4 wave stacks
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < N;
    if (live && perm) i = perm[i];           // Morton order (sort_points.hip): a wave's 64 points are neighbours
    const int64_t ic = live ? i : (N - 1);
    const f3 p = clamp_raw(mk3(pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]));
    Nearest nr;
    bool ins;
    if (BRUTE) { nr = nearest_brute<kBlock>(m, p, reinterpret_cast<float *>(lds)); ins = inside_brute(m, p); }
    else       { nr = nearest_packet(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth); ins = inside_bins(m, p); }
    if (!live) return;
    const SdfOut o = sdf_attrs(m, p, nr, ins);
    sdf[i] = o.sdf;
    nrm[3 * i] = o.nrm.x; nrm[3 * i + 1] = o.nrm.y; nrm[3 * i + 2] = o.nrm.z;
    cm[3 * i] = o.cm.x; cm[3 * i + 1] = o.cm.y; cm[3 * i + 2] = o.cm.z;
    vis[i] = o.vis;
    if (face) face[i] = nr.face;
    if (inside_out) inside_out[i] = ins ? 1 : 0;
}

// point mode, one wavefront per point (see nearest_coop)
__global__ __launch_bounds__(kCoopWaves * 64) void k_nearest_coop(MeshDev m, Calib cal, const float *__restrict__ pts, int64_t N,
                                                                 NearRef near, int cap, float sdf_clip)
{
    extern __shared__ __attribute__((aligned(16))) char coop_smem[];
    const int64_t i = (int64_t)blockIdx.x * kCoopWaves + (threadIdx.x >> 6);
    if (i >= N) return;
    const f3 p = project(resolve_calib(cal), mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    const Nearest nr = nearest_coop(m, p, coop_lds(coop_smem, threadIdx.x >> 6, cap));
    if ((threadIdx.x & 63) == 0) store_near(near, i, nr, sdf_clip);
}

__global__ __launch_bounds__(kCoopWaves * 64) void k_sdf_query_coop(MeshDev m, const float *__restrict__ pts, int64_t N,
                                                                   float *sdf, float *nrm, float *cm, float *vis,
                                                                   int64_t *face, uint8_t *inside_out, int cap)
{
    extern __shared__ __attribute__((aligned(16))) char coop_smem[];
    const int64_t i = (int64_t)blockIdx.x * kCoopWaves + (threadIdx.x >> 6);
    if (i >= N) return;
    const f3 p = clamp_raw(mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    const Nearest nr = nearest_coop(m, p, coop_lds(coop_smem, threadIdx.x >> 6, cap));
    if ((threadIdx.x & 63) != 0) return;
    const bool ins = inside_bins(m, p);
    const SdfOut o = sdf_attrs(m, p, nr, ins);
    sdf[i] = o.sdf;
    nrm[3 * i] = o.nrm.x; nrm[3 * i + 1] = o.nrm.y; nrm[3 * i + 2] = o.nrm.z;
    cm[3 * i] = o.cm.x; cm[3 * i + 1] = o.cm.y; cm[3 * i + 2] = o.cm.z;
    vis[i] = o.vis;
    if (face) face[i] = nr.face;
    if (inside_out) inside_out[i] = ins ? 1 : 0;
}

// Nearest-triangle search as its own launch: the packet traversal needs ~50 VGPRs, so it runs at
// full occupancy (8 waves / SIMD hide the dependent scalar-load chain), which the register-heavier
// attribute / gather code would cap at 5.  Output per point, structure of arrays: the slot of the nearest
// triangle with the "outside the clip band" flag, and - inside the band only - its squared distance (store_near).
// ALT (diagnostics, icon_work_set_tie_rule): the alternative tie rule - highest face index among the faces within
// `tie_ulps` float32 ulps of the minimum d^2 (nearest_packet_alt) - for measuring how much of the output hangs on
// the unpinned tie behaviour of the kaolin leaf (lib/dataset/mesh_util.py:374-390).
template <bool LATTICE, bool ALT = false>
__global__ __launch_bounds__(kBlock) void k_nearest(MeshDev m, Calib cal, LatticeMap L, const float *__restrict__ pts, int64_t N,
                                                    NearRef near, const int32_t *__restrict__ perm,
                                                    float sdf_clip, int tie_ulps, const LatticeFast *lf)
{
    __shared__ int lds[(kBlock / 64) * kStackDepth];
    int64_t i; bool live; f3 p;
    int root = 0;
    if (LATTICE) {
        int cx, cy, cz;
        if (lf) {                                                // 4^3 packets, default block order: the set-up precomputed per call
            const LatticeFast F = lattice_fast_load(lf);
            if ((int)blockIdx.x >= F.nb) return;                 // beyond the trimmed tiling (the host tiled the whole slab)
            live = lattice_point_fast(F, blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, cx, cy, cz);
            p = lattice_world_fast(lf, cx, cy, cz + L.z0);
            root = F.root;
        } else {
            L = lattice_trim(L, m);
            if ((int)blockIdx.x >= L.tx * L.ty * L.tz) return;
            int ix, iy, iz;
            live = lattice_point(L, ix, iy, iz);
            lattice_clamp(L, ix, iy, iz, cx, cy, cz);
            p = lattice_world(L.res, cx, cy, cz + L.z0);
            root = mesh_root(m);
        }
        i = ((int64_t)cz * L.res + cy) * L.res + cx;
    } else {
        i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        live = i < N;
        if (!live) i = N - 1;
        if (perm) i = perm[i];          // Morton order: the wave's 64 points are neighbours (sort_points.hip)
        p = project(resolve_calib(cal), mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        root = mesh_root(m);
    }
    Nearest nr = nearest_packet(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, nullptr, nullptr, INFINITY, nullptr,
                                LATTICE ? packet_center_lane(L) : 21, true, root);
    if (ALT) nr = nearest_packet_alt(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, (uint32_t)__float_as_int(nr.d2), (uint32_t)tie_ulps);
    // (staging the 16 x 4 x 4 block through LDS so that 16 threads store one 64-byte run removes the partial-line
    //  writes but the block-wide barrier costs 0.14 ms; not kept)
    if (live) store_near(near, i, nr, sdf_clip);
}

// Lattices of few packets (33^3 .. 65^3 slabs: the first level of the reference's schedule): one PACKET per workgroup, its walk
// shared by the NW wavefronts (nearest_shared) - such a launch lasts as long as its longest walk, and this divides the walk.
// Workgroup b serves packet b & 3 of tile b >> 2 of k_nearest's tiling.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_nearest_shared(MeshDev m, LatticeMap L, NearRef near, float sdf_clip, ShareDbg dbg)
{
    __shared__ int lds[NW * kStackDepth];
    __shared__ __attribute__((aligned(16))) char smem[share_lds_bytes(NW)];
    L = lattice_trim(L, m);
    if ((int)(blockIdx.x >> 2) >= L.tx * L.ty * L.tz) return;
    int ix, iy, iz, cx, cy, cz;
    const bool live = lattice_point_at(L, (int)(blockIdx.x >> 2), (int)(blockIdx.x & 3), threadIdx.x & 63, ix, iy, iz);
    lattice_clamp(L, ix, iy, iz, cx, cy, cz);
    const f3 p = lattice_world(L.res, cx, cy, cz + L.z0);
    const Nearest nr = nearest_shared<NW>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, smem, packet_center_lane(L), dbg);
    if (live && threadIdx.x < 64) store_near(near, ((int64_t)cz * L.res + cy) * L.res + cx, nr, sdf_clip);
}

// Diagnostics (icon_sdf_query_ties): winner, runner-up and the ulp gap between their squared distances
__global__ __launch_bounds__(kBlock) void k_nearest_ties(MeshDev m, const float *__restrict__ pts, int64_t N, const int32_t *__restrict__ perm,
                                                         int32_t *__restrict__ face, int32_t *__restrict__ face2, uint8_t *__restrict__ ulps)
{
    __shared__ int lds[(kBlock / 64) * kStackDepth];
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < N;
    if (!live) i = N - 1;
    if (perm) i = perm[i];
    const f3 p = clamp_raw(mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    unsigned long long k2 = 0;
    const Nearest nr = nearest_packet<false, true>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, nullptr, nullptr, INFINITY, &k2);
    if (!live) return;
    const uint32_t b1 = (uint32_t)__float_as_int(nr.d2), b2 = (uint32_t)(k2 >> 32);
    const int f2 = (int)(k2 & 0xffffffffu);
    const bool has2 = f2 != 0x7fffffff && b2 >= b1;
    face[i] = nr.face;
    face2[i] = has2 ? f2 : -1;
    ulps[i] = (uint8_t)(has2 ? min(b2 - b1, 255u) : 255u);
}

// Feature assembly: one 16-float row per point,
//   icon : [img(csel) | sdf | cmap r g b | norm x y z | 0.. | code]
//   pamir: [img(C) | vol(Cv) | 0.. | code]      pifu: [img(C) | z | 0.. | code]
// Rows are indexed by the point's linear index (lattice: (z*R + y)*R + x relative to plane z0).
template <int PRIOR, bool LATTICE, bool BRUTE>
__global__ __launch_bounds__(kBlock) void k_features(MeshDev m, FeatDev f, Calib cal, LatticeMap L,
                                                     const float *__restrict__ pts, int64_t N,
                                                     float sdf_clip, int cmap_local,
                                                     const int32_t *__restrict__ row_count, const int32_t *__restrict__ row_slots,
                                                     NearRef near,
                                                     float *__restrict__ X, uint8_t *__restrict__ code8, int skip_shell)
{
    __shared__ int lds[(PRIOR == ICON_PRIOR_ICON && BRUTE) ? kBruteTile * 24 : 1];
    int64_t i; bool live; f3 p;
    if (LATTICE) {
        // L tiles the WHOLE slab here (every point gets a row); skip_shell: the geometry pre-pass left the shell out
        int ix, iy, iz, cx, cy, cz;
        live = lattice_point(L, ix, iy, iz);
        lattice_clamp(L, ix, iy, iz, cx, cy, cz);
        p = lattice_world(L.res, cx, cy, cz + L.z0);
        i = ((int64_t)cz * L.res + cy) * L.res + cx;
        if (skip_shell && !in_cube_bit(p)) {
            // a shell point: multiplied by 0 whatever its row holds (in_cube, HGPIFuNet.py:363) - a zero row and the code
            // byte k_sign wrote (icon) / in_cube = 0, without touching the search results that do not exist for it
            if (live) {
                float z[kXRow];
#pragma unroll
                for (int k = 0; k < kXRow; ++k) z[k] = 0.0f;
                uint32_t c = 0;
                if (PRIOR == ICON_PRIOR_ICON) c = code8[i];
                z[kCodeSlot] = __int_as_float((int)c);
                store_row(X, i, z);
                if (PRIOR != ICON_PRIOR_ICON) code8[i] = (uint8_t)c;
            }
            return;
        }
    } else {
        i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        live = i < N;
        if (!live) i = N - 1;
        p = project(resolve_calib(cal), mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    }
    float row[kXRow];
#pragma unroll
    for (int k = 0; k < kXRow; ++k) row[k] = 0.0f;
    uint32_t code = in_cube_bit(p);
    if (PRIOR == ICON_PRIOR_ICON) {
        Nearest nr;
        bool ins;
        float s;
        f3 cmv;
        SdfOut o;
        if (BRUTE) {
            nr = nearest_brute<kBlock>(m, p, reinterpret_cast<float *>(lds)); ins = inside_brute(m, p);
            o = sdf_attrs(m, p, nr, ins);
            code = sign_code(p, nr.d2, ins, sdf_clip);
        } else {
            // the geometry pre-pass ran on the same stream just before: slot of the nearest triangle, the code byte
            // (outlier / sign / inside / in_cube) and, for points inside the clip band only, d^2
            code = code8[i];
            nr.slot = near_slot_of(near, i); nr.face = 0;
            nr.d2 = (code & kCodeOutlier) ? 0.0f : near_d2(near, i);
            ins = (code & kCodeInside) != 0;
            o = sdf_attrs(m, p, nr, ins);
        }
        s = o.sdf;
        cmv = o.cm;
        if (code & kCodeOutlier) {            // HGPIFuNet.py:298-305
            s = (float)((int)((code >> kCodeSignShift) & 3u) - 1);
            if (cmap_local) cmv = mk3(s, s, s);   // reference mode: patched later from the sign list
        }
        float g[16];
        gather_planes_dyn(f, (f.n_select == 2 && o.vis == 0.0f) ? 1 : 0, p.x, p.y, g);   // feat_select: vis==1 -> front half; no 'vis': all channels
        const int h = f.csel;
        for (int k = 0; k < h; ++k) row[k] = g[k];
        int hh = h;                                       // [img | sdf | cmap (if) | norm (if)], HGPIFuNet.py:301-311
        row[hh++] = s;
        if (f.smpl_mask & kSmplCmap) { row[hh] = cmv.x; row[hh + 1] = cmv.y; row[hh + 2] = cmv.z; hh += 3; }
        if (f.smpl_mask & kSmplNorm) { row[hh] = o.nrm.x; row[hh + 1] = o.nrm.y; row[hh + 2] = o.nrm.z; }
    } else {
        float g[16];
        gather_planes_dyn(f, 0, p.x, p.y, g);
        const int h = f.csel;
        for (int k = 0; k < h; ++k) row[k] = g[k];
        if (PRIOR == ICON_PRIOR_PAMIR) {
            float v[8];
            if (f.vpad == 8) gather_volume<2>(f, p.x, p.y, p.z, v); else gather_volume<1>(f, p.x, p.y, p.z, v);
            for (int k = 0; k < f.Cv; ++k) row[h + k] = v[k];
        } else {
            row[h] = p.z;
        }
    }
    row[kCodeSlot] = __int_as_float((int)code);
    if (live) { store_row(X, i, row); code8[i] = (uint8_t)code; }   // byte copy of the code word: the outlier passes stream 1 B/pt
}

// diagnostics: per-wavefront BVH work of the lattice traversal (DESIGN.md reports visited nodes / point)
__global__ __launch_bounds__(kBlock) void k_traversal_stats(MeshDev m, LatticeMap L, unsigned long long *out /* [4] */, int seeded)
{
    __shared__ int lds[(kBlock / 64) * kStackDepth];
    L = lattice_trim(L, m);
    if ((int)blockIdx.x >= L.tx * L.ty * L.tz) return;
    int ix, iy, iz, cx, cy, cz;
    const bool live = lattice_point(L, ix, iy, iz);
    lattice_clamp(L, ix, iy, iz, cx, cy, cz);
    const f3 p = lattice_world(L.res, cx, cy, cz + L.z0);
    int nn = 0, nt = 0;
    Nearest nr = nearest_packet<true>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, &nn, &nt);
    if (seeded) {       // experiment: the work of a traversal that starts from the final bound (seeded == 1) or from the
                        // bound a coarse pass would give: best distance at the tile's centre lane + tile radius (seeded == 2)
        nn = 0; nt = 0;
        float thr0 = prune_threshold(nr.d2);
        if (seeded == 2) {
            const float dc = sqrtf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(nr.d2), 21)));
            const float h = 2.0f / (float)(L.res - 1);
            const float rad = 2.6f * h;                       // > half diagonal of a 4x4x4 block (sqrt(3) * 1.5 h)
            thr0 = prune_threshold((dc + rad) * (dc + rad));
        }
        if (seeded == 3) {      // the bound a wave walking along x would carry over: this lane's result one packet (4 points) back
            const float h = 2.0f / (float)(L.res - 1);
            int na = 0, nb = 0;
            const Nearest pv = nearest_packet<true>(m, mk3(p.x - 4.0f * h, p.y, p.z), live, lds + (threadIdx.x >> 6) * kStackDepth, &na, &nb);
            const float dn = sqrtf(pv.d2) + 4.0001f * h;
            thr0 = prune_threshold(dn * dn);
        }
        nr = nearest_packet<true>(m, p, live, lds + (threadIdx.x >> 6) * kStackDepth, &nn, &nt, thr0);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], 1ull); atomicAdd(&out[1], (unsigned long long)nn); atomicAdd(&out[2], (unsigned long long)nt);
    }
    if (live && nr.slot < 0) atomicAdd(&out[3], 1ull);
    if ((threadIdx.x & 63) == 0) atomicMax(&out[4], (unsigned long long)(nn + nt));      // the longest walk: what a latency-bound launch waits for
}

// one thread per (y, z) row of the slab: triangles whose (y,z) projection covers the row
__global__ __launch_bounds__(kBlock) void k_row_crossings(MeshDev m, LatticeMap L, int32_t *row_count, int32_t *row_slots, LatticeFast *lf)
{
    if (lf && blockIdx.x == 0 && threadIdx.x == 0) lattice_fast_write(lf, L, m);      // the search's per-packet set-up, once per call
    if (lf) lattice_fast_coords(lf, L.res, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (int64_t)L.nz * L.res) return;
    const int iy = (int)(row % L.res), iz = (int)(row / L.res);
    const f3 p = lattice_world(L.res, 0, iy, iz + L.z0);
    int n = 0;
    const MeshDyn &d = *m.dyn;
    if (d.gy == 0) { row_count[row] = -1; return; }          // no bin lists (kMeshBinOverflow): every point takes inside_bins' brute-force branch
    if (p.y >= d.bin_y0 && p.y <= d.bin_y1 && p.z >= d.bin_z0 && p.z <= d.bin_z1) {
        const int cy = bin_cell(p.y, d.bin_y0, d.bin_inv_y, d.gy);
        const int cz = bin_cell(p.z, d.bin_z0, d.bin_inv_z, d.gz);
        const int cell = cz * d.gy + cy;
        const int beg = m.bin_start[cell], end = m.bin_start[cell + 1];
        for (int k = beg; k < end; ++k) {
            const int slot = m.bin_slots[k];
            f3 a, b, c; int ia, ib, ic;
            load_tri_full(m.tris + slot, a, b, c, ia, ib, ic);
            if (ray_covers(p, a, b, c, ia, ib, ic)) {
                if (n < kRowCap) row_slots[row * kRowCap + n] = slot;
                ++n;
            }
        }
    }
    row_count[row] = (n <= kRowCap) ? n : -1;
}

// ---------------------------------------------------------------------------------------------
// outlier sign list (reference cmap mode): count -> scan -> compact -> patch, all in the linear
// point order of the call.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t row_code(const float *X, int64_t i)
{
    return (uint32_t)__float_as_int(X[i * kXRow + kCodeSlot]);
}

__global__ __launch_bounds__(kScanBlock) void k_outlier_count(const uint8_t *__restrict__ code8, int64_t N, int32_t *block_counts)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const bool o = (i < N) && (code8[i] & kCodeOutlier);
    const unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) s += wsum[w];
        block_counts[blockIdx.x] = s;
    }
}

// The same with 16 lanes per row (slabs of few rows - the coarse lattices: 33^2 = 1,089 threads walking ~100-entry bin lists one
// after the other were a 28 us launch of pure latency): the lanes of a group take every 16th entry of the bin list, the hits are
// appended in list order through a ballot (the same row_slots as the one-thread version, entry for entry).
__global__ __launch_bounds__(kBlock) void k_row_crossings_wide(MeshDev m, LatticeMap L, int32_t *row_count, int32_t *row_slots, LatticeFast *lf)
{
    if (lf && blockIdx.x == 0 && threadIdx.x == 0) lattice_fast_write(lf, L, m);
    if (lf) lattice_fast_coords(lf, L.res, (int64_t)blockIdx.x * kBlock + threadIdx.x);
    const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
    const bool have = row < (int64_t)L.nz * L.res;
    const MeshDyn &d = *m.dyn;
    if (d.gy == 0) { if (have && s == 0) row_count[row] = -1; return; }
    const int iy = have ? (int)(row % L.res) : 0, iz = have ? (int)(row / L.res) : 0;
    const f3 p = lattice_world(L.res, 0, iy, iz + L.z0);
    int beg = 0, end = 0;
    if (have && p.y >= d.bin_y0 && p.y <= d.bin_y1 && p.z >= d.bin_z0 && p.z <= d.bin_z1) {
        const int cy = bin_cell(p.y, d.bin_y0, d.bin_inv_y, d.gy);
        const int cz = bin_cell(p.z, d.bin_z0, d.bin_inv_z, d.gz);
        const int cell = cz * d.gy + cy;
        beg = m.bin_start[cell]; end = m.bin_start[cell + 1];
    }
    int n = 0;
    for (int k = beg + s; __any(k - s < end); k += 16) {
        bool hit = false;
        int slot = 0;
        if (k < end) {
            slot = m.bin_slots[k];
            f3 a, b, c; int ia, ib, ic;
            load_tri_full(m.tris + slot, a, b, c, ia, ib, ic);
            hit = ray_covers(p, a, b, c, ia, ib, ic);
        }
        const unsigned grp = (unsigned)(__ballot(hit) >> (16 * g)) & 0xffffu;
        const int at = n + __popc(grp & ((1u << s) - 1u));
        if (hit && at < kRowCap) row_slots[row * kRowCap + at] = slot;
        n += __popc(grp);
    }
    if (have && s == 0) row_count[row] = (n <= kRowCap) ? n : -1;
}

// Exclusive scan of the per-tile outlier counts (66,308 entries for a 257^3 call; int64 offsets: a 513^3
// lattice has 1.35e8 points) in two coalesced passes: k_scan_local scans 1,024 consecutive counts per
// workgroup (wave shuffles + one LDS hop) and records the chunk total, k_scan_apply adds the sum of the
// preceding chunk totals.  (A single workgroup walking 65 strided entries per thread was latency-bound:
// 0.12 ms - more than k_sign, the count and the compaction together.)
__device__ __forceinline__ int64_t block_exclusive_scan_1024(int64_t v, int64_t *wtot /* LDS [16] */, int64_t *block_total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int64_t base = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int64_t t = wtot[k]; if (k < w) base += t; all += t; }
    *block_total = all;
    return base + incl - v;
}

// n_points_dev (adaptive levels): the call's size lives on the device - n = its number of 256-point blocks
__global__ __launch_bounds__(1024) void k_scan_local(const int32_t *__restrict__ counts, int64_t n, int32_t *__restrict__ local, int64_t *__restrict__ part,
                                                     const int *__restrict__ n_points_dev)
{
    __shared__ int64_t wtot[16];
    if (n_points_dev) n = ((int64_t)*n_points_dev + kScanBlock - 1) / kScanBlock;
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int64_t v = (i < n) ? counts[i] : 0;
    int64_t total;
    const int64_t ex = block_exclusive_scan_1024(v, wtot, &total);
    if (i < n) local[i] = (int32_t)ex;
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_apply(const int32_t *__restrict__ local, const int64_t *__restrict__ part, int64_t nchunks, int64_t n,
                                                     int64_t *__restrict__ offsets, int64_t *__restrict__ total, const int *__restrict__ n_points_dev)
{
    __shared__ int64_t wtot[16];
    if (n_points_dev) n = ((int64_t)*n_points_dev + kScanBlock - 1) / kScanBlock;
    // sum of the chunk totals before this chunk (and of all chunks, for *total): nchunks <= a few hundred
    int64_t before = 0, all = 0;
    for (int64_t k = threadIdx.x; k < nchunks; k += 1024) { const int64_t t = part[k]; all += t; if (k < blockIdx.x) before += t; }
    int64_t sum_all, sum_before;
    (void)block_exclusive_scan_1024(all, wtot, &sum_all);
    __syncthreads();
    (void)block_exclusive_scan_1024(before, wtot, &sum_before);
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < n) offsets[i] = sum_before + local[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = sum_all;
}

__device__ __forceinline__ int64_t outlier_rank(bool o, const int64_t *block_offsets, int *wsum)
{
    // exclusive rank of this thread's outlier among the outliers of the call (linear order)
    const unsigned long long b = __ballot(o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    before += __popcll(b & ((1ull << lane) - 1ull));
    return block_offsets[blockIdx.x] + before;
}

__global__ __launch_bounds__(kScanBlock) void k_outlier_compact(const uint8_t *__restrict__ code8, int64_t N,
                                                                const int64_t *block_offsets, int8_t *signs, const int *__restrict__ n_dev)
{
    __shared__ int wsum[kScanBlock / 64];
    if (n_dev) { N = *n_dev; if ((int64_t)blockIdx.x * kScanBlock >= N) return; }
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t r = outlier_rank(o, block_offsets, wsum);
    if (o) signs[r] = (int8_t)((int)((code >> kCodeSignShift) & 3u) - 1);
}

// scan + compact in ONE launch for calls of few blocks (the levels of the reference's schedule: 20,000 - 60,000 points, 80 - 250
// blocks): every workgroup sums the counts before its own block itself (a few hundred ints from L2) instead of waiting for two
// scan launches - three launches of ~5 us each were the launch floor, not work.  O(blocks^2) reads: the host picks it by size.
__global__ __launch_bounds__(kScanBlock) void k_outlier_small(const int32_t *__restrict__ counts, const uint8_t *__restrict__ code8, int64_t N,
                                                              int64_t *__restrict__ block_offsets, int64_t *__restrict__ total, int8_t *__restrict__ signs,
                                                              const int *__restrict__ n_dev)
{
    __shared__ int wsum[kScanBlock / 64];
    __shared__ int64_t red[2][kScanBlock / 64];
    if (n_dev) N = *n_dev;
    const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nblk == 0) *total = 0;
    if ((int64_t)blockIdx.x >= nblk) return;
    int64_t before = 0, all = 0;
    const bool want_all = blockIdx.x == 0;                       // block 0 also publishes the call's total
    for (int64_t k = threadIdx.x; k < (want_all ? nblk : (int64_t)blockIdx.x); k += kScanBlock) { const int64_t t = counts[k]; all += t; if (k < blockIdx.x) before += t; }
    for (int d = 32; d >= 1; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = before; red[1][w] = all; }
    __syncthreads();
    before = 0; all = 0;
    for (int k = 0; k < kScanBlock / 64; ++k) { before += red[0][k]; all += red[1][k]; }
    if (threadIdx.x == 0) { block_offsets[blockIdx.x] = before; if (want_all) *total = all; }
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const unsigned long long b = __ballot(o);
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int64_t r = before + __popcll(b & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) r += wsum[k];
    if (o) signs[r] = (int8_t)((int)((code >> kCodeSignShift) & 3u) - 1);
}
constexpr int64_t kSmallListBlocks = 16384;      // up to this many blocks (4.2 M points) the one-launch list; worst case 1 GB of L2 reads

// cmap[j][k] = s[(3j + k) mod K]  with j = global outlier rank = rank_offset + local rank
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                              const int64_t *block_offsets, const int8_t *signs_global,
                                                              int64_t k_total, int64_t rank_offset)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = rank_offset + outlier_rank(o, block_offsets, wsum);
    if (o) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;                 // j < K  =>  mm < 3K
            if (mm >= k_total) mm -= k_total;
            if (mm >= k_total) mm -= k_total;
            row[k] = (float)signs_global[mm];
        }
    }
}

// The same patch with the global sign list left where the all_gather put it: rank r's message is
// gathered + r * stride = [int64 count_r][int8 signs_r ...] (multi-GPU path, recon.py).  K, this
// rank's offset and the segment of every index are derived on the device from the headers, so the
// host never has to read the counts (no synchronisation between the exchange and the MLP launch).
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch_seg(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                                  const int64_t *block_offsets, const int8_t *__restrict__ gathered,
                                                                  int64_t stride, int world, int rank)
{
    __shared__ int wsum[kScanBlock / 64];
    __shared__ int64_t off[kMaxWorld + 1];
    if (threadIdx.x == 0) {
        int64_t a = 0;
        for (int r = 0; r < world; ++r) { off[r] = a; a += *reinterpret_cast<const int64_t *>(gathered + (int64_t)r * stride); }
        off[world] = a;
    }
    __syncthreads();
    const int64_t K = off[world];
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = off[rank] + outlier_rank(o, block_offsets, wsum);
    if (o && K > 0) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;                 // j < K  =>  mm < 3K
            if (mm >= K) mm -= K;
            if (mm >= K) mm -= K;
            int r = 0;
            while (mm >= off[r + 1]) ++r;
            const int64_t e = mm - off[r];                      // 2 bits per sign (sign + 1), four to a byte
            const uint32_t byte = (uint8_t)gathered[(int64_t)r * stride + 8 + (e >> 2)];
            row[k] = (float)((int)((byte >> (2 * (int)(e & 3))) & 3u) - 1);
        }
    }
}

// sign message of the multi-GPU exchange: [int64 K][K signs, 2 bits each (sign + 1), four to a byte, zero padded]
__global__ __launch_bounds__(256) void k_pack_signs(const int8_t *__restrict__ signs, const int64_t *__restrict__ k_dev, uint8_t *__restrict__ msg, int64_t cap_bytes)
{
    const int64_t K = *k_dev;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) *reinterpret_cast<int64_t *>(msg) = K;
    if (t >= cap_bytes || 4 * t >= K) return;
    uint32_t b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * t + j < K) b |= (uint32_t)(signs[4 * t + j] + 1) << (2 * j);
    msg[8 + t] = (uint8_t)b;
}

// repack feature planes [C][H][W] -> [n_select][H][W][cpad] (channel-last, zero padded)
__global__ void k_pack_planes(const float *__restrict__ src, int C, int H, int W, int n_select, int csel, int cpad, float *dst)
{
    const int64_t n = (int64_t)n_select * H * W * cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpad);
        const int64_t pix = (i / cpad) % ((int64_t)H * W);
        const int sel = (int)(i / ((int64_t)cpad * H * W));
        dst[i] = (c < csel) ? src[((int64_t)(sel * csel + c)) * H * W + pix] : 0.0f;
    }
}

}  // namespace icon

// =============================================================================================
// C ABI
// =============================================================================================
using namespace icon;


extern "C" int icon_sdf_query(const icon_mesh_t *mesh, const float *d_points, int64_t N,
                              float *d_sdf, float *d_norm, float *d_cmap, float *d_vis,
                              int64_t *d_face, uint8_t *d_inside, int search, void *stream)
{
    ICON_ARG(mesh && d_points && d_sdf && d_norm && d_cmap && d_vis, "icon_sdf_query: null argument");
    ICON_ARG(N >= 0, "icon_sdf_query: negative N");
    if (N == 0) return ICON_OK;
    const int64_t nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb < (1ll << 31), "icon_sdf_query: N too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (search == ICON_SEARCH_BRUTE) {
        hipLaunchKernelGGL(k_sdf_query<true>, dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, d_points, N, d_sdf, d_norm,
                           d_cmap, d_vis, d_face, d_inside, nullptr);
    } else if (N < kPacketMinPoints) {    // sparse / unordered points: one wavefront per point (see nearest_coop)
        const int cap = coop_cap(mesh->depth_bound);
        hipLaunchKernelGGL(k_sdf_query_coop, dim3((unsigned)((N + kCoopWaves - 1) / kCoopWaves)), dim3(kCoopWaves * 64), kCoopWaves * coop_wave_bytes(cap), st,
                           mesh->dev, d_points, N, d_sdf, d_norm, d_cmap, d_vis, d_face, d_inside, cap);
    } else {                              // large batches: packets over the Morton order (scratch freed after a stream sync)
        icon_work tmp;
        static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const int32_t *perm = nullptr;
        int rc = morton_order(&tmp, d_points, ident, nullptr, N, st, &perm);
        if (!rc) {
            hipLaunchKernelGGL(k_sdf_query<false>, dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, d_points, N, d_sdf, d_norm,
                               d_cmap, d_vis, d_face, d_inside, perm);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(ICON_ERR_HIP, "icon_sdf_query: launch failed");
        }
        (void)hipFree(tmp.d_sort_keys); (void)hipFree(tmp.d_sort_idx); (void)hipFree(tmp.d_sort_tmp);
        return rc;
    }
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

extern "C" int icon_feat_create(const float *d_planes, int C, int H, int W, int n_select,
                                const float *d_vol, int Cv, int Dv, int Hv, int Wv,
                                void *stream, icon_feat_t **out)
{
    ICON_ARG(out != nullptr, "icon_feat_create: out is null");
    *out = nullptr;
    ICON_ARG(d_planes && C > 0 && H > 1 && W > 1, "icon_feat_create: bad planes");
    ICON_ARG(n_select == 1 || n_select == 2, "icon_feat_create: n_select must be 1 or 2");
    ICON_ARG(C % n_select == 0, "icon_feat_create: C not divisible by n_select");
    const int csel = C / n_select;
    const int cpad = (csel + 3) & ~3;
    if (cpad > 16) return fail(ICON_ERR_UNSUPPORTED, "icon_feat_create: more than 16 channels per tap");
    if (d_vol) {
        ICON_ARG(Cv > 0 && Dv > 1 && Hv > 1 && Wv > 1, "icon_feat_create: bad volume");
        if (Cv > 8) return fail(ICON_ERR_UNSUPPORTED, "icon_feat_create: more than 8 volume channels");
    }
    hipStream_t st = (hipStream_t)stream;
    icon_feat *f = new icon_feat();
    const size_t n = (size_t)n_select * H * W * cpad;
    hipError_t e = hipMalloc((void **)&f->d_planes, n * sizeof(float));
    if (e != hipSuccess) { delete f; return fail(ICON_ERR_HIP, std::string("hipMalloc planes: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(k_pack_planes, dim3(1024), dim3(256), 0, st, d_planes, C, H, W, n_select, csel, cpad, f->d_planes);
    FeatDev &d = f->dev;
    d.planes = f->d_planes; d.C = C; d.H = H; d.W = W; d.n_select = n_select; d.csel = csel; d.cpad = cpad;
    d.smpl_mask = kSmplCmap | kSmplNorm;
    d.vol = nullptr; d.Cv = 0; d.Dv = d.Hv = d.Wv = 0; d.vpad = 0;
    if (d_vol) {
        const int vpad = (Cv + 3) & ~3;
        const size_t nv = (size_t)Dv * Hv * Wv * vpad;
        e = hipMalloc((void **)&f->d_vol, nv * sizeof(float));
        if (e != hipSuccess) { icon_feat_destroy(f); return fail(ICON_ERR_HIP, std::string("hipMalloc vol: ") + hipGetErrorString(e)); }
        // a volume is a "plane" of H' = D*H rows: same channel-last repack
        hipLaunchKernelGGL(k_pack_planes, dim3(1024), dim3(256), 0, st, d_vol, Cv, Dv * Hv, Wv, 1, Cv, vpad, f->d_vol);
        d.vol = f->d_vol; d.Cv = Cv; d.Dv = Dv; d.Hv = Hv; d.Wv = Wv; d.vpad = vpad;
    }
    e = hipGetLastError();
    if (e != hipSuccess) { icon_feat_destroy(f); return fail(ICON_ERR_HIP, std::string("pack planes: ") + hipGetErrorString(e)); }
    *out = f;
    return ICON_OK;
}

extern "C" int icon_feat_set_smpl_feats(icon_feat_t *f, int has_cmap, int has_norm)
{
    ICON_ARG(f != nullptr, "icon_feat_set_smpl_feats: feat is null");
    f->dev.smpl_mask = (has_cmap ? kSmplCmap : 0) | (has_norm ? kSmplNorm : 0);
    return ICON_OK;
}

extern "C" int icon_feat_destroy(icon_feat_t *f)
{
    if (!f) return ICON_OK;
    (void)hipFree(f->d_planes); (void)hipFree(f->d_vol);
    delete f;
    return ICON_OK;
}

extern "C" int icon_work_create(icon_work_t **out)
{
    ICON_ARG(out != nullptr, "icon_work_create: out is null");
    *out = new icon_work();
    return ICON_OK;
}

extern "C" int icon_work_destroy(icon_work_t *w)
{
    if (!w) return ICON_OK;
    (void)hipFree(w->d_x); (void)hipFree(w->d_grp_mask); (void)hipFree(w->d_block_counts); (void)hipFree(w->d_block_offsets); (void)hipFree(w->d_scan_local); (void)hipFree(w->d_scan_part);
    (void)hipFree(w->d_signs); (void)hipFree(w->d_total); (void)hipFree(w->d_flag); (void)hipFree(w->d_seg); (void)hipFree(w->d_row_count); (void)hipFree(w->d_row_slots); (void)hipFree(w->d_lfast); if (w->h_err) (void)hipHostFree(w->h_err); if (w->h_any) (void)hipHostFree(w->h_any); (void)hipFree(w->d_near16); (void)hipFree(w->d_near_hi); (void)hipFree(w->d_near_d2); (void)hipFree(w->d_code8);
    (void)hipFree(w->d_sort_keys); (void)hipFree(w->d_sort_idx); (void)hipFree(w->d_sort_tmp);
    for (int k = 0; k < 6; ++k) if (w->ev[k]) (void)hipEventDestroy(w->ev[k]);
    (void)hipFree(w->d_clock); (void)hipFree(w->d_steal);
    icon::mc_destroy(w->mc);
    icon::clean_destroy(w->clean);
    icon::adaptive_destroy(w->ad);
    delete w;
    return ICON_OK;
}

// out[0] = the nearest-triangle search kernel alone (ms; 0 when the call had none), out[1] = shader cycles and out[2] = wall
// time (ms) of the fused kernel's workgroup 0 (s_memtime / s_memrealtime stamps), out[3] = out[1] / out[2] as MHz - the
// EFFECTIVE clock under that launch's load.  Synchronises (on the call's last event).
extern "C" int icon_work_profile_detail(icon_work_t *w, double out[4])
{
    ICON_ARG(w != nullptr && out != nullptr, "icon_work_profile_detail: null argument");
    if (!w->prof || !w->ev_valid) return fail(ICON_ERR_STATE, "icon_work_profile_detail: no profiled call on this workspace");
    ICON_HIP(hipEventSynchronize(w->ev[3]));
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (w->ev_search) {
        float ms = 0.0f;
        ICON_HIP(hipEventElapsedTime(&ms, w->ev[4], w->ev[5]));
        out[0] = ms;
    }
    unsigned long long c[4] = {0, 0, 0, 0};
    ICON_HIP(hipMemcpy(c, w->d_clock, sizeof(c), hipMemcpyDeviceToHost));
    int dev = 0, khz = 0;
    ICON_HIP(hipGetDevice(&dev));
    ICON_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (c[2] > c[0] && c[3] > c[1] && khz > 0) {
        out[1] = (double)(c[2] - c[0]);
        out[2] = (double)(c[3] - c[1]) / (double)khz;
        out[3] = out[1] / out[2] * 1e-3;
    }
    return ICON_OK;
}

// The workgroup records of the most recent profiled launch of the fused kernel.  rec[5 * b + ...] for workgroup b:
// [0] XCC id, [1] start (ms after the earliest workgroup's start), [2] span (ms, constant-rate counter), [3] shader cycles,
// [4] tiles evaluated.  *n = workgroups of that launch (at most cap are written).  Synchronises like icon_work_stage_ms.
extern "C" int icon_work_profile_workgroups(icon_work_t *w, double *rec, int cap, int *n)
{
    ICON_ARG(w != nullptr && rec != nullptr && n != nullptr && cap >= 0, "icon_work_profile_workgroups: bad argument");
    if (!w->prof || !w->ev_valid) return fail(ICON_ERR_STATE, "icon_work_profile_workgroups: no profiled call on this workspace");
    ICON_HIP(hipEventSynchronize(w->ev[3]));
    const int g = std::min(w->clock_grid, kMaxProfGrid);
    *n = g;
    if (g <= 0) return ICON_OK;
    std::vector<unsigned long long> c((size_t)kWgRec * g);
    ICON_HIP(hipMemcpy(c.data(), w->d_clock + 4, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int dev = 0, khz = 0;
    ICON_HIP(hipGetDevice(&dev));
    ICON_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ICON_ARG(khz > 0, "icon_work_profile_workgroups: the device reports no wall clock rate");
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < g; ++b) t0 = std::min(t0, c[(size_t)kWgRec * b + 1]);
    for (int b = 0; b < g && b < cap; ++b) {
        const unsigned long long *r = &c[(size_t)kWgRec * b];
        rec[5 * b + 0] = (double)(r[4] >> 32);
        rec[5 * b + 1] = (double)(r[1] - t0) / (double)khz;
        rec[5 * b + 2] = (double)(r[3] - r[1]) / (double)khz;
        rec[5 * b + 3] = (double)(r[2] - r[0]);
        rec[5 * b + 4] = (double)r[5];
    }
    return ICON_OK;
}

// permille of every workgroup's span of tiles that is drawn dynamically (0 = the static partition of rounds 1-5), in contiguous
// groups of `group` tiles (<= 127), own XCD's spans first.  Any setting gives the same volume bit for bit (a tile's result does not depend on who evaluates it).
extern "C" int icon_work_set_steal(icon_work_t *w, int permille, int group)
{
    ICON_ARG(w != nullptr && permille >= 0 && permille <= 1000 && group >= 1 && group <= 127, "icon_work_set_steal: bad argument");
    w->steal_permille = permille; w->steal_grp = group;
    return ICON_OK;
}

namespace icon {
// the stride-(sx, sy, sz) sub-lattice of a [res]^3 volume: one thread per point, one store per wavefront that finds a value above
// the level into a HOST-MAPPED word (visible to the host once the stream has drained: no copy, no second launch)
__global__ __launch_bounds__(256) void k_any_above(const float *__restrict__ occ, int res, int sx, int sy, int sz, int nx, int ny, int nz, float level, int *flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool pos = false;
    if (i < (int64_t)nx * ny * nz) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        pos = occ[((int64_t)z * sz * res + (int64_t)y * sy) * res + (int64_t)x * sx] > level;
    }
    if (__any(pos) && (threadIdx.x & 63) == 0) *flag = 1;
}
}  // namespace icon

// Seg3dLossless._forward_faster's None test (lib/common/seg3d_lossless.py:173-177: nothing above 0.5 on the COARSEST lattice) on a
// dense device volume: *h_any = 1 when some point of the stride-(sx, sy, sz) sub-lattice exceeds `level`.  One kernel, then the
// call waits for the stream (as the reference's `.sum() == 0` does) - four torch operators, a device-to-host copy and ~35 us of
// GPU time before.
extern "C" int icon_volume_any_above(const float *d_occ, int res, int sx, int sy, int sz, float level, icon_work_t *w, void *stream, int *h_any)
{
    ICON_ARG(d_occ && w && h_any && res >= 1 && sx >= 1 && sy >= 1 && sz >= 1, "icon_volume_any_above: bad argument");
    if (!w->h_any) {
        ICON_HIP(hipHostMalloc((void **)&w->h_any, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
    }
    void *dev = nullptr;
    ICON_HIP(hipHostGetDevicePointer(&dev, w->h_any, 0));
    *(volatile int *)w->h_any = 0;
    const int nx = (res - 1) / sx + 1, ny = (res - 1) / sy + 1, nz = (res - 1) / sz + 1;
    const int64_t n = (int64_t)nx * ny * nz;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(icon::k_any_above, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_occ, res, sx, sy, sz, nx, ny, nz, level, (int *)dev);
    ICON_HIP(hipGetLastError());
    ICON_HIP(hipStreamSynchronize(st));
    *h_any = *(volatile int *)w->h_any;
    return ICON_OK;
}

extern "C" int icon_work_set_reserve_cus(icon_work_t *w, int n)
{
    ICON_ARG(w != nullptr && n >= 0, "icon_work_set_reserve_cus: bad argument");
    w->reserve_cus = n;
    return ICON_OK;
}

extern "C" int icon_work_profile(icon_work_t *w, int enable)
{
    ICON_ARG(w != nullptr, "icon_work_profile: work is null");
    if (enable && !w->ev[0])
        for (int k = 0; k < 6; ++k) ICON_HIP(hipEventCreate(&w->ev[k]));
    if (enable && !w->d_clock) {
        const size_t bytes = (size_t)(4 + kWgRec * kMaxProfGrid) * sizeof(unsigned long long);
        ICON_HIP(hipMalloc((void **)&w->d_clock, bytes));
        ICON_HIP(hipMemset(w->d_clock, 0, bytes));
    }
    w->clock_grid = 0;
    w->ev_search = false;
    w->prof = enable != 0;
    w->ev_valid = false;
    return ICON_OK;
}

extern "C" int icon_work_stage_ms(icon_work_t *w, float out_ms[3])
{
    ICON_ARG(w && out_ms, "icon_work_stage_ms: null argument");
    if (!w->prof || !w->ev_valid) return fail(ICON_ERR_STATE, "icon_work_stage_ms: no profiled call on this workspace");
    ICON_HIP(hipEventSynchronize(w->ev[3]));
    for (int k = 0; k < 3; ++k) ICON_HIP(hipEventElapsedTime(&out_ms[k], w->ev[k], w->ev[k + 1]));
    return ICON_OK;
}

namespace icon {
// device-side K: same as k_outlier_patch but K and the list come from this call's own scan
__global__ __launch_bounds__(kScanBlock) void k_outlier_patch_self(float *__restrict__ X, const uint8_t *__restrict__ code8, int64_t N, int cmap_slot,
                                                                   const int64_t *block_offsets, const int8_t *signs,
                                                                   const int64_t *k_total_dev)
{
    __shared__ int wsum[kScanBlock / 64];
    const int64_t K = *k_total_dev;
    const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t code = (i < N) ? code8[i] : 0u;
    const bool o = code & kCodeOutlier;
    const int64_t j = outlier_rank(o, block_offsets, wsum);
    if (o) {
        float *row = X + i * kXRow + cmap_slot;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int64_t mm = 3 * j + k;
            if (mm >= K) mm -= K;
            if (mm >= K) mm -= K;
            row[k] = (float)signs[mm];
        }
    }
}
}  // namespace icon

namespace icon {
// ---- shared walks: error record and test switches ------------------------------------------------------------------------
int g_share_ring = 0, g_share_lose = 0, g_share_spin_log2 = 0;     // icon_debug_set_option; all zero in production
int g_lattice_fast = -1;                                          // -1: read ICON_AMD_LATTICE_FAST once (default on)
int g_share_waves = -2;                                           // -2: read ICON_AMD_SHARE once; -1: by launch size; 1 / 4 / 8 / 16: forced
int share_waves_override()
{
    if (g_share_waves == -2) g_share_waves = getenv("ICON_AMD_SHARE") ? atoi(getenv("ICON_AMD_SHARE")) : -1;
    return g_share_waves;
}

int work_share_dbg(icon_work *w, ShareDbg *out)
{
    if (!w->h_err) {
        ICON_HIP(hipHostMalloc((void **)&w->h_err, 8 * sizeof(int), hipHostMallocMapped | hipHostMallocPortable));   // (portable: a process may drive several devices)
        memset(w->h_err, 0, 8 * sizeof(int));
    }
    void *dev = nullptr;
    ICON_HIP(hipHostGetDevicePointer(&dev, w->h_err, 0));
    out->err = (int *)dev;
    out->ring = g_share_ring; out->lose = g_share_lose; out->spin_log2 = g_share_spin_log2;
    return ICON_OK;
}

int work_check_err(icon_work *w)
{
    if (!w || !w->h_err) return ICON_OK;
    volatile int *e = w->h_err;
    const int code = e[0];
    if (code == 0 || code == kShareErrClaimed) return ICON_OK;     // (claimed: a wave is still writing the details - the next look reports it)
    char buf[320];
    static const char *what[] = {"", "a wave waited in vain for the node of its queue ticket (lost or abandoned push)",
                                 "a wave waited in vain for its ring slot to be cleared (ring a full turn behind)",
                                 "an idle wave outlasted its bound while the workgroup was still active"};
    snprintf(buf, sizeof(buf), "shared-walk search: %s [code %d, workgroup %d, wave %d, ticket %d, head %d, tail %d, avail %d, active %d]; the results of "
             "that launch are not to be trusted", what[code >= 1 && code <= 3 ? code : 0], code, e[1], e[7], e[2], e[3], e[4], e[5], e[6]);
    for (int k = 0; k < 8; ++k) e[k] = 0;
    return fail(ICON_ERR_STATE, buf);
}

}  // namespace icon

namespace {

inline void mark(icon_work *w, int k, hipStream_t st)
{
    if (w->prof) { (void)hipEventRecord(w->ev[k], st); if (k == 3) w->ev_valid = true; }
}

// scratch for n_points: (slot, d^2) + 1-byte codes + scan arrays + sign list; the 64-byte input rows only
// for the paths that still materialise them (need_x: precision f32 and the brute-force search)
int ensure_work(icon_work *w, int64_t n_points, bool need_x)
{
    { const int rc = work_check_err(w); if (rc) return rc; }      // an earlier launch on this workspace reported: no new work on its results
    if (n_points > w->cap_points) {
        (void)hipFree(w->d_near16); (void)hipFree(w->d_near_d2); w->d_near16 = nullptr; w->d_near_d2 = nullptr; w->cap_points = 0;
        ICON_HIP(hipMalloc((void **)&w->d_near16, (size_t)n_points * sizeof(uint16_t)));
        ICON_HIP(hipMalloc((void **)&w->d_near_d2, (size_t)n_points * sizeof(float)));
        (void)hipFree(w->d_code8); w->d_code8 = nullptr;
        ICON_HIP(hipMalloc((void **)&w->d_code8, (size_t)n_points));
        w->cap_points = n_points;
    }
    if (need_x && n_points > w->cap_x) {
        (void)hipFree(w->d_x); w->d_x = nullptr; w->cap_x = 0;
        ICON_HIP(hipMalloc((void **)&w->d_x, (size_t)n_points * kXRow * sizeof(float)));
        w->cap_x = n_points;
    }
    const int64_t nblk = (n_points + kScanBlock - 1) / kScanBlock;
    if (nblk > w->cap_blocks) {
        (void)hipFree(w->d_block_counts); (void)hipFree(w->d_block_offsets); (void)hipFree(w->d_scan_local); (void)hipFree(w->d_scan_part);
        w->d_block_counts = nullptr; w->d_block_offsets = nullptr; w->d_scan_local = nullptr; w->d_scan_part = nullptr; w->cap_blocks = 0;
        (void)hipFree(w->d_grp_mask); w->d_grp_mask = nullptr;
        ICON_HIP(hipMalloc((void **)&w->d_grp_mask, (size_t)nblk * 4 * sizeof(uint64_t)));
        ICON_HIP(hipMalloc((void **)&w->d_block_counts, (size_t)nblk * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&w->d_block_offsets, (size_t)nblk * sizeof(int64_t)));
        ICON_HIP(hipMalloc((void **)&w->d_scan_local, (size_t)nblk * sizeof(int32_t)));
        ICON_HIP(hipMalloc((void **)&w->d_scan_part, (size_t)((nblk + 1023) / 1024) * sizeof(int64_t)));
        w->cap_blocks = nblk;
    }
    if (n_points > w->cap_signs) {
        (void)hipFree(w->d_signs); w->d_signs = nullptr; w->cap_signs = 0;
        ICON_HIP(hipMalloc((void **)&w->d_signs, (size_t)n_points));
        w->cap_signs = n_points;
    }
    if (!w->d_total) ICON_HIP(hipMalloc((void **)&w->d_total, sizeof(int64_t)));
    if (!w->d_flag) ICON_HIP(hipMalloc((void **)&w->d_flag, sizeof(int)));
    if (!w->d_steal) {
        ICON_HIP(hipMalloc((void **)&w->d_steal, 9 * sizeof(unsigned int)));
        ICON_HIP(hipMemset(w->d_steal, 0, 9 * sizeof(unsigned int)));     // (synchronous with respect to the host: ordered before any launch)
    }
    if (!w->d_seg) ICON_HIP(hipMalloc((void **)&w->d_seg, (kMaxWorld + 1) * sizeof(int64_t)));
    return ICON_OK;
}

int check_prior(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior, int *c0)
{
    ICON_ARG(feat != nullptr, "feature handle is null");
    const FeatDev &f = feat->dev;
    if (prior == ICON_PRIOR_ICON) {
        ICON_ARG(mesh != nullptr, "icon prior needs a mesh handle");
        // n_select = 2: smpl_vis picks the front or back half (HGPIFuNet.py:334-336); 1: smpl_feats without 'vis', every channel is an input (:345-346)
        *c0 = f.csel + 1 + ((f.smpl_mask & kSmplCmap) ? 3 : 0) + ((f.smpl_mask & kSmplNorm) ? 3 : 0);
    } else if (prior == ICON_PRIOR_PAMIR) {
        ICON_ARG(f.n_select == 1 && f.vol != nullptr, "pamir prior needs n_select = 1 and a volume");
        *c0 = f.csel + f.Cv;
    } else if (prior == ICON_PRIOR_PIFU) {
        ICON_ARG(f.n_select == 1, "pifu prior needs n_select = 1");
        *c0 = f.csel + 1;
    } else {
        return fail(ICON_ERR_ARG, "unknown prior_type");
    }
    if (*c0 > kCodeSlot) return fail(ICON_ERR_UNSUPPORTED, "more than 15 MLP input channels");
    return ICON_OK;
}

// geometry pre-pass of the icon prior (BVH search): per-row crossing lists (lattice), nearest triangle -> near
template <bool LATTICE>
int launch_nearest(const icon_mesh_t *mesh, const Calib &cal, const LatticeMap &L, const float *d_points, int64_t N, float sdf_clip,
                   icon_work *work, hipStream_t st)
{
    LatticeFast *lf = nullptr;
    if (LATTICE) {
        const int64_t rows = (int64_t)L.nz * L.res;
        if (rows > work->cap_rows) {
            (void)hipFree(work->d_row_count); (void)hipFree(work->d_row_slots);
            work->d_row_count = nullptr; work->d_row_slots = nullptr; work->cap_rows = 0;
            ICON_HIP(hipMalloc((void **)&work->d_row_count, (size_t)rows * sizeof(int32_t)));
            ICON_HIP(hipMalloc((void **)&work->d_row_slots, (size_t)rows * kRowCap * sizeof(int32_t)));
            work->cap_rows = rows;
        }
        // the record of the search's per-packet set-up (LatticeFast): 4^3 packets in the default block order only
        if (g_lattice_fast < 0) g_lattice_fast = getenv("ICON_AMD_LATTICE_FAST") ? (atoi(getenv("ICON_AMD_LATTICE_FAST")) != 0) : 1;   // 0: every packet derives its own (A/B runs)
        if (g_lattice_fast && (L.pk == 0 || L.pk == 4) && L.remap == 0 && L.res <= kLatticeFastMaxRes) {
            if (!work->d_lfast) ICON_HIP(hipMalloc((void **)&work->d_lfast, kLatticeFastBytes));
            lf = work->d_lfast;
        }
        if (rows <= 20000)                       // up to 129^2 rows: 16 lanes per row
            hipLaunchKernelGGL(k_row_crossings_wide, dim3((unsigned)((rows * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, mesh->dev, L,
                               work->d_row_count, work->d_row_slots, lf);
        else
            hipLaunchKernelGGL(k_row_crossings, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, mesh->dev, L,
                               work->d_row_count, work->d_row_slots, lf);
    }
    int64_t nb;
    if (LATTICE) nb = (int64_t)L.tx * L.ty * L.tz; else nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb >= 0 && nb < (1ll << 31), "too many workgroups for one launch");
    if (mesh->F > kNearLoSlots && work->cap_points_hi < work->cap_points) {      // big meshes: the byte of higher slot bits
        (void)hipFree(work->d_near_hi); work->d_near_hi = nullptr; work->cap_points_hi = 0;
        ICON_HIP(hipMalloc((void **)&work->d_near_hi, (size_t)work->cap_points));
        work->cap_points_hi = work->cap_points;
    }
    const NearRef near = work_near(work, mesh);
    // point mode: sparse batches walk the tree one wavefront per point; a batch dense enough for a wave's 64
    // Morton neighbours to be close together goes through the packet kernel
    const int32_t *perm = nullptr;
    static const int mode = getenv("ICON_AMD_POINT_SEARCH") ? atoi(getenv("ICON_AMD_POINT_SEARCH")) : 0;   // 0 auto, 2 coop, 3 packets
    const bool alt = work->tie_rule != 0;        // diagnostics: the alternative tie rule lives in the packet kernel only
    if (!LATTICE && !alt && mode != 3 && (N < kPacketMinPoints || mode == 2)) {
        const int cap = coop_cap(mesh->depth_bound);
        hipLaunchKernelGGL(k_nearest_coop, dim3((unsigned)((N + kCoopWaves - 1) / kCoopWaves)), dim3(kCoopWaves * 64),
                           kCoopWaves * coop_wave_bytes(cap), st, mesh->dev, cal, d_points, N, near, cap, sdf_clip);
    } else if (nb > 0) {                         // nb == 0: a slab that is all shell (nothing to search)
        if (!LATTICE) {
            const int rc = morton_order(work, d_points, cal.m, cal.d, N, st, &perm);
            if (rc) return rc;
        }
        const int share_env = share_waves_override();            // diagnostics: 1 = never, 8 / 16
        const int nw = (!LATTICE || alt) ? 1 : ((share_env == 1 || share_env == 8 || share_env == 16) ? share_env : share_waves(nb * 4));
        ShareDbg dbg{};
        if (nw > 1) { const int rc = work_share_dbg(work, &dbg); if (rc) return rc; }
        if (work->prof) (void)hipEventRecord(work->ev[4], st);
        if (nw == 16) hipLaunchKernelGGL(k_nearest_shared<16>, dim3((unsigned)nb * 4), dim3(16 * 64), 0, st, mesh->dev, L, near, sdf_clip, dbg);
        else if (nw == 8) hipLaunchKernelGGL(k_nearest_shared<8>, dim3((unsigned)nb * 4), dim3(8 * 64), 0, st, mesh->dev, L, near, sdf_clip, dbg);
        else if (alt) hipLaunchKernelGGL((k_nearest<LATTICE, true>), dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, cal, L, d_points, N, near, perm, sdf_clip, work->tie_ulps, lf);
        else hipLaunchKernelGGL((k_nearest<LATTICE, false>), dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, cal, L, d_points, N, near, perm, sdf_clip, 0, lf);
        if (work->prof) { (void)hipEventRecord(work->ev[5], st); work->ev_search = true; }
    }
    ICON_HIP(hipGetLastError());
    debug_sync(LATTICE ? "k_row_crossings + k_nearest<lattice>" : "nearest (points)", st);
    const int rc_sign = launch_sign(mesh, cal, L.res, L.z0, d_points, N, sdf_clip, work, LATTICE, st);
    debug_sync("k_sign", st);
    return rc_sign;
}

// X rows + codes (the materialising path): k_features, reading `near` unless the search is brute force
template <bool LATTICE>
int launch_features(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior, float sdf_clip, int cmap_mode,
                    const Calib &cal, const LatticeMap &L, const float *d_points, int64_t N, int search,
                    icon_work *work, hipStream_t st)
{
    // every point of the slab gets a row: tile the whole slab, whatever region the geometry pre-pass covered
    LatticeMap Lf = L;
    const int skip_shell = L.off;
    if (LATTICE) {
        Lf.sx0 = Lf.sy0 = Lf.sz0 = 0; Lf.sx1 = Lf.sy1 = L.res; Lf.sz1 = L.nz;
        Lf.tx = (L.res + 15) / 16; Lf.ty = (L.res + 3) / 4; Lf.tz = (L.nz + 3) / 4; Lf.pk = 4; Lf.trim = 0;
    }
    int64_t nb;
    if (LATTICE) nb = (int64_t)Lf.tx * Lf.ty * Lf.tz; else nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb > 0 && nb < (1ll << 31), "too many workgroups for one launch");
    const dim3 grid((unsigned)nb), block(kBlock);
    const MeshDev md = mesh ? mesh->dev : MeshDev{};
    const int local = (cmap_mode == ICON_CMAP_LOCAL) ? 1 : 0;
    const bool brute = (search == ICON_SEARCH_BRUTE);
#define ICON_LAUNCH(P, B) hipLaunchKernelGGL((k_features<P, LATTICE, B>), grid, block, 0, st, md, feat->dev, cal, Lf, d_points, N, sdf_clip, local, work->d_row_count, work->d_row_slots, work_near(work, mesh), work->d_x, work->d_code8, skip_shell)
    if (prior == ICON_PRIOR_ICON) { if (brute) ICON_LAUNCH(ICON_PRIOR_ICON, true); else ICON_LAUNCH(ICON_PRIOR_ICON, false); }
    else if (prior == ICON_PRIOR_PAMIR) ICON_LAUNCH(ICON_PRIOR_PAMIR, false);
    else ICON_LAUNCH(ICON_PRIOR_PIFU, false);
#undef ICON_LAUNCH
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

// count + scan + compact over the codes of points [0, N): fills w->d_block_offsets, w->d_total, and `signs`
int outlier_list(icon_work *w, int64_t N, int8_t *signs, bool counted, hipStream_t st)
{
    const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
    // `counted`: k_sign already left the outlier count of every 256-point block in d_block_counts
    if (!counted) hipLaunchKernelGGL(k_outlier_count, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_code8, N, w->d_block_counts);
    if (nblk <= 1024) {                           // a small call (host-known size): one launch
        hipLaunchKernelGGL(k_outlier_small, dim3((unsigned)std::max<int64_t>(nblk, 1)), dim3(kScanBlock), 0, st, w->d_block_counts, w->d_code8, N, w->d_block_offsets, w->d_total,
                           signs, (const int *)nullptr);
        ICON_HIP(hipGetLastError());
        debug_sync("outlier list (one launch)", st);
        return ICON_OK;
    }
    const int64_t nchunks = (nblk + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nchunks), dim3(1024), 0, st, w->d_block_counts, nblk, w->d_scan_local, w->d_scan_part, (const int *)nullptr);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nchunks), dim3(1024), 0, st, w->d_scan_local, w->d_scan_part, nchunks, nblk, w->d_block_offsets, w->d_total, (const int *)nullptr);
    hipLaunchKernelGGL(k_outlier_compact, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_code8, N, w->d_block_offsets, signs, (const int *)nullptr);
    ICON_HIP(hipGetLastError());
    debug_sync("outlier scan + compact", st);
    return ICON_OK;
}

}  // namespace

namespace icon {
// the call's outlier sign list when its size is known on the device only (k_sign left the block counts): w->d_signs, w->d_total
int outlier_list_dev(icon_work *w, const int *n_dev, int64_t n_max, hipStream_t st)
{
    const int64_t nblk = (n_max + kScanBlock - 1) / kScanBlock;
    if (nblk <= kSmallListBlocks) {               // (the size on the device is a few percent of n_max: see k_outlier_small)
        hipLaunchKernelGGL(k_outlier_small, dim3((unsigned)std::max<int64_t>(nblk, 1)), dim3(kScanBlock), 0, st, w->d_block_counts, w->d_code8, n_max, w->d_block_offsets,
                           w->d_total, w->d_signs, n_dev);
        ICON_HIP(hipGetLastError());
        return ICON_OK;
    }
    const int64_t nchunks = (nblk + 1023) / 1024;
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nchunks), dim3(1024), 0, st, w->d_block_counts, nblk, w->d_scan_local, w->d_scan_part, n_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nchunks), dim3(1024), 0, st, w->d_scan_local, w->d_scan_part, nchunks, nblk, w->d_block_offsets, w->d_total, n_dev);
    hipLaunchKernelGGL(k_outlier_compact, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_code8, n_max, w->d_block_offsets, w->d_signs, n_dev);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}
int ensure_work_points(icon_work *w, int64_t n_points) { return ensure_work(w, n_points, false); }
}  // namespace icon

namespace {

int patch_self(icon_work *w, int64_t N, int cmap_slot, hipStream_t st)
{
    const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(icon::k_outlier_patch_self, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, w->d_x, w->d_code8, N, cmap_slot,
                       w->d_block_offsets, w->d_signs, w->d_total);
    ICON_HIP(hipGetLastError());
    return ICON_OK;
}

// The f16x3 precision (the default) with the BVH search runs FUSED: no input rows in HBM (fused_f16x3.hip).
// ICON_AMD_UNFUSED=1 forces the materialising path (tests compare the two bit for bit).
int g_unfused = -1;      // -1: read ICON_AMD_UNFUSED once; 0 / 1: set by icon_debug_set_unfused
int g_shell_skip = -1;   // -1: read ICON_AMD_SHELL_SKIP once (default on); 0 / 1: set by icon_debug_set_shell_skip
inline bool want_fused(int precision, int search)
{
    if (g_unfused < 0) g_unfused = (getenv("ICON_AMD_UNFUSED") && atoi(getenv("ICON_AMD_UNFUSED")) != 0) ? 1 : 0;
    return !g_unfused && precision == ICON_PRECISION_F16X3 && search != ICON_SEARCH_BRUTE;
}

// Phase 1 of every query: what can be done before the outlier sign list of the whole call is known.
//   icon prior, BVH: nearest search + k_sign (+ the call's own sign list in reference cmap mode); no rows yet
//   icon prior, brute force: k_features (rows + codes) + sign list
//   pamir / pifu: nothing
// Records in `work` everything phase 2 needs (the handles must stay alive until then).
template <bool LATTICE>
int phase1(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior, float sdf_clip, int cmap_mode, const Calib &cal,
           const LatticeMap &L, const float *d_points, int64_t N, int search, int8_t *d_signs_out, icon_work *work, hipStream_t st)
{
    int rc;
    work->q_mesh = mesh; work->q_feat = feat; work->q_prior = prior; work->q_sdf_clip = sdf_clip; work->q_cmap_mode = cmap_mode;
    work->q_cal = cal; work->q_L = L; work->q_points = d_points; work->q_N = N; work->q_search = search; work->q_lattice = LATTICE;
    work->q_rows_ready = false; work->slab_patched = false;
    work->slab_needs_patch = (prior == ICON_PRIOR_ICON && cmap_mode == ICON_CMAP_REFERENCE && (feat->dev.smpl_mask & kSmplCmap));
    work->slab_cmap_slot = feat->dev.csel + 1;
    if (prior != ICON_PRIOR_ICON) return ICON_OK;
    if (search == ICON_SEARCH_BRUTE) {
        if ((rc = ensure_work(work, N, true))) return rc;
        if ((rc = launch_features<LATTICE>(mesh, feat, prior, sdf_clip, cmap_mode, cal, L, d_points, N, search, work, st))) return rc;
        work->q_rows_ready = true;
    } else {
        if ((rc = launch_nearest<LATTICE>(mesh, cal, L, d_points, N, sdf_clip, work, st))) return rc;
    }
    if (work->slab_needs_patch) {
        int8_t *signs = d_signs_out ? d_signs_out : work->d_signs;
        if ((rc = outlier_list(work, N, signs, search != ICON_SEARCH_BRUTE, st))) return rc;
    }
    return ICON_OK;
}

// The MLP input rows of the prepared call in work->d_x ([N, kXRow], reference channel order, slot kCodeSlot = code word),
// the reference-mode outlier cmap patched in from wherever the call's sign list is.  Idempotent per prepared call.
int materialise_rows(icon_work *work, const FusedSigns &fs, hipStream_t st)
{
    int rc;
    const int prior = work->q_prior;
    const int64_t N = work->q_N;
    if (!work->q_rows_ready) {
        if ((rc = ensure_work(work, N, true))) return rc;
        rc = work->q_lattice ? launch_features<true>(work->q_mesh, work->q_feat, prior, work->q_sdf_clip, work->q_cmap_mode, work->q_cal, work->q_L,
                                                     work->q_points, N, work->q_search, work, st)
                             : launch_features<false>(work->q_mesh, work->q_feat, prior, work->q_sdf_clip, work->q_cmap_mode, work->q_cal, work->q_L,
                                                      work->q_points, N, work->q_search, work, st);
        if (rc) return rc;
        work->q_rows_ready = true;
    }
    if (work->slab_needs_patch && !work->slab_patched) {
        const int64_t nblk = (N + kScanBlock - 1) / kScanBlock;
        if (fs.mode == kSignSelf) {
            if ((rc = patch_self(work, N, work->slab_cmap_slot, st))) return rc;
        } else if (fs.mode == kSignGlobal) {
            if (fs.k_host > 0)
                hipLaunchKernelGGL(k_outlier_patch, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, work->d_x, work->d_code8, N, work->slab_cmap_slot,
                                   work->d_block_offsets, fs.list, fs.k_host, fs.rank_offset);
        } else if (fs.mode == kSignSeg) {
            hipLaunchKernelGGL(k_outlier_patch_seg, dim3((unsigned)nblk), dim3(kScanBlock), 0, st, work->d_x, work->d_code8, N, work->slab_cmap_slot,
                               work->d_block_offsets, fs.gathered, fs.stride, fs.world, fs.rank);
        }
        ICON_HIP(hipGetLastError());
        work->slab_patched = true;
    }
    return ICON_OK;
}

// Phase 2: the MLP input of every point and the MLP itself, given where the call's outlier signs are.
// Lattice calls evaluate the planes [za, zb) of the prepared slab into the SLAB's buffer d_occ (several calls may
// cover a slab piece by piece: the multi-GPU driver gathers the first half while the second is computed).
int phase2(const icon_mlp_t *mlp, const FusedSigns &fs, float *d_occ, int precision, icon_work *work, hipStream_t st, int za = 0, int zb = 0)
{
    int rc;
    const int prior = work->q_prior;
    const int64_t N = work->q_N;
    const bool fused = want_fused(precision, work->q_search);
    const int local = (work->q_cmap_mode == ICON_CMAP_LOCAL) ? 1 : 0;
    const LatticeMap &L = work->q_L;
    const bool first = !work->q_lattice || za == L.z0;
    if (fused) {
        if (first) mark(work, 2, st);
        rc = launch_fused_f16x3(work->q_mesh, work->q_feat, mlp, prior, work->q_cal, L, za, zb, work->q_points, N,
                                work->q_sdf_clip, local, work, fs, d_occ, work->q_lattice, st);
        mark(work, 3, st);
        return rc;
    }
    if ((rc = materialise_rows(work, fs, st))) return rc;
    if (first) mark(work, 2, st);
    if (work->q_lattice) {
        const int64_t o = (int64_t)(za - L.z0) * L.res * L.res, n = (int64_t)(zb - za) * L.res * L.res;
        rc = mlp_launch(mlp, work->d_x + o * kXRow, n, d_occ + o, precision, st);
    } else {
        rc = mlp_launch(mlp, work->d_x, N, d_occ, precision, st);
    }
    mark(work, 3, st);
    return rc;
}

FusedSigns self_signs(const icon_work *work)
{
    FusedSigns fs{};
    fs.mode = work->slab_needs_patch ? kSignSelf : kSignNone;
    fs.list = work->d_signs; fs.k_dev = work->d_total;
    return fs;
}

}  // namespace

extern "C" int icon_debug_set_unfused(int on)
{
    g_unfused = on ? 1 : 0;
    return ICON_OK;
}

static int query_points_impl(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                             int prior_type, float sdf_clip, int cmap_mode, const float *h_calib, const float *d_calib,
                             const float *d_points, int64_t N, float *d_occ,
                             int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_points && d_occ, "icon_query_points: null argument");
    ICON_ARG(N >= 0, "icon_query_points: negative N");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    ICON_ARG(c0 == mlp->c0, "icon_query_points: MLP input width does not match the feature layout");
    if (N == 0) return ICON_OK;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_work(work, N, false))) return rc;
    Calib cal;
    static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(cal.m, h_calib ? h_calib : ident, sizeof(cal.m));
    cal.d = d_calib;
    LatticeMap L{};
    work->slab_ready = false;
    mark(work, 0, st);
    if ((rc = phase1<false>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, d_points, N, search, nullptr, work, st))) return rc;
    mark(work, 1, st);
    return phase2(mlp, self_signs(work), d_occ, precision, work, st);
}

extern "C" int icon_query_points(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                                 int prior_type, float sdf_clip, int cmap_mode, const float *h_calib,
                                 const float *d_points, int64_t N, float *d_occ,
                                 int search, int precision, icon_work_t *work, void *stream)
{
    return query_points_impl(mesh, feat, mlp, prior_type, sdf_clip, cmap_mode, h_calib, nullptr, d_points, N, d_occ, search,
                             precision, work, stream);
}

extern "C" int icon_query_points_dcalib(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                                        int prior_type, float sdf_clip, int cmap_mode, const float *d_calib,
                                        const float *d_points, int64_t N, float *d_occ,
                                        int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(d_calib != nullptr, "icon_query_points_dcalib: d_calib is null");
    return query_points_impl(mesh, feat, mlp, prior_type, sdf_clip, cmap_mode, nullptr, d_calib, d_points, N, d_occ, search,
                             precision, work, stream);
}

// Lattice geometry of a slab.  `shell_ok`: the caller allows the shell skip (everything but the brute-force search,
// which computes its own rows for every point).  The MLP then never sees the shell (in_cube alone makes it 0); the
// SEARCH leaves out every face of the cube that is provably all "outside" outliers without looking at the mesh -
// farther from the body's bounding box than the clip band is wide, k_sign's own far test - and keeps the others: a
// shell point is an entry of the call's outlier sign list and its code byte must be right.
static int lattice_map(int res, int z0, int z1, const icon_mesh_t *mesh, float sdf_clip, bool shell_ok, LatticeMap *L)
{
    ICON_ARG(res >= 3 && (res & 1) == 1, "lattice resolution must be odd and >= 3 (seg3d_lossless.py:84-86)");
    ICON_ARG(z0 >= 0 && z1 > z0 && z1 <= res, "bad z range");
    L->res = res; L->z0 = z0; L->nz = z1 - z0;
    if (g_shell_skip < 0) g_shell_skip = getenv("ICON_AMD_SHELL_SKIP") ? (atoi(getenv("ICON_AMD_SHELL_SKIP")) != 0) : 1;   // 0: evaluate and mask the shell (A/B runs)
    const int off = (shell_ok && g_shell_skip) ? 1 : 0;
    L->off = off;
    L->zs = std::max(z0, off); L->nzi = std::min(z1, res - off) - L->zs;
    if (L->nzi < 0) L->nzi = 0;
    // search region: the whole slab here - the kernel itself leaves out the far faces (lattice_trim: the body's box is
    // known on the device only) and workgroups beyond the trimmed tiling exit at once
    L->sx0 = 0; L->sx1 = res; L->sy0 = 0; L->sy1 = res; L->sz0 = 0; L->sz1 = z1 - z0;
    // points per wavefront of the search: 4^3 blocks (common.h: lattice_packet)
    L->pk = lattice_packet();
    L->tx = (L->sx1 - L->sx0 + 4 * L->pk - 1) / (4 * L->pk); L->ty = (L->sy1 - L->sy0 + L->pk - 1) / L->pk; L->tz = (L->sz1 - L->sz0 + L->pk - 1) / L->pk;
    L->trim = (off && mesh) ? 1 : 0;
    L->trim_need = std::sqrt(far_box_dist2(sdf_clip)) * 1.001f + 1e-5f;
    const char *e = getenv("ICON_AMD_XCD_REMAP");
    // 0: single blocks alternate over the XCDs (default, fastest); 2: whole x-rows of blocks per XCD - 14 % fewer HBM
    // write bytes (partial lines meet in one L2) but 1.5 % slower; 1: contiguous XCD bands, 1.6x slower
    L->remap = e ? atoi(e) : 0;
    return ICON_OK;
}

static int slab_features_impl(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior_type, float sdf_clip, int cmap_mode,
                              int res, int z0, int z1, int8_t *d_signs_local, int64_t *d_count_local, void *d_msg, int64_t msg_bytes,
                              int search, icon_work_t *work, void *stream)
{
    ICON_ARG(work != nullptr, "icon_grid_slab_features: work is null");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    LatticeMap L;
    // the brute-force path computes rows for every point itself; everything else may skip the shell
    if ((rc = lattice_map(res, z0, z1, prior_type == ICON_PRIOR_ICON ? mesh : nullptr, sdf_clip, search != ICON_SEARCH_BRUTE, &L))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)L.nz * res * res;
    ICON_ARG(N < (1ll << 31), "icon_grid_slab_features: more than 2^31 points in one slab");
    if ((rc = ensure_work(work, N, false))) return rc;
    Calib cal{};
    work->slab_ready = false;
    mark(work, 0, st);
    if ((rc = phase1<true>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, nullptr, N, search, d_signs_local, work, st))) return rc;
    work->slab_c0 = c0;
    if (d_count_local) {
        if (work->slab_needs_patch) ICON_HIP(hipMemcpyAsync(d_count_local, work->d_total, sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        else ICON_HIP(hipMemsetAsync(d_count_local, 0, sizeof(int64_t), st));
    }
    if (d_msg) {
        const int64_t cap = msg_bytes - 8;                 // packed bytes available
        ICON_ARG(msg_bytes >= 8 && cap * 4 >= N, "icon_grid_slab_features_msg: message buffer too small (8 + ceil(points / 4) bytes)");
        if (work->slab_needs_patch) {
            const int64_t nb = (std::max<int64_t>(cap, 1) + 255) / 256;
            hipLaunchKernelGGL(k_pack_signs, dim3((unsigned)nb), dim3(256), 0, st, d_signs_local ? d_signs_local : work->d_signs, work->d_total,
                               (uint8_t *)d_msg, cap);
            ICON_HIP(hipGetLastError());
        } else {
            ICON_HIP(hipMemsetAsync(d_msg, 0, sizeof(int64_t), st));
        }
    }
    mark(work, 1, st);
    work->slab_res = res; work->slab_z0 = z0; work->slab_z1 = z1; work->slab_ready = true;
    return ICON_OK;
}

extern "C" int icon_grid_slab_features(const icon_mesh_t *mesh, const icon_feat_t *feat,
                                       int prior_type, float sdf_clip, int cmap_mode,
                                       int res, int z0, int z1, int8_t *d_signs_local, int64_t *d_count_local,
                                       int search, icon_work_t *work, void *stream)
{
    return slab_features_impl(mesh, feat, prior_type, sdf_clip, cmap_mode, res, z0, z1, d_signs_local, d_count_local, nullptr, 0, search, work, stream);
}

extern "C" int icon_grid_slab_features_msg(const icon_mesh_t *mesh, const icon_feat_t *feat,
                                           int prior_type, float sdf_clip, int cmap_mode,
                                           int res, int z0, int z1, void *d_msg, int64_t msg_bytes,
                                           int search, icon_work_t *work, void *stream)
{
    ICON_ARG(d_msg != nullptr, "icon_grid_slab_features_msg: message buffer is null");
    return slab_features_impl(mesh, feat, prior_type, sdf_clip, cmap_mode, res, z0, z1, nullptr, nullptr, d_msg, msg_bytes, search, work, stream);
}

extern "C" int icon_grid_slab_finish(const icon_mlp_t *mlp, int res, int z0, int z1,
                                     const int8_t *d_signs_global, int64_t k_total, int64_t rank_offset,
                                     float *d_occ, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_slab_finish: null argument");
    if (!work->slab_ready || work->slab_res != res || work->slab_z0 != z0 || work->slab_z1 != z1)
        return fail(ICON_ERR_STATE, "icon_grid_slab_finish: no matching icon_grid_slab_features call on this workspace");
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_slab_finish: MLP input width does not match the feature layout");
    FusedSigns fs{};
    if (work->slab_needs_patch) {
        ICON_ARG(k_total == 0 || d_signs_global != nullptr, "icon_grid_slab_finish: sign list is null");
        fs.mode = kSignGlobal; fs.list = d_signs_global; fs.k_host = k_total; fs.rank_offset = rank_offset;
    }
    work->slab_ready = false;
    return phase2(mlp, fs, d_occ, precision, work, (hipStream_t)stream, z0, z1);
}

extern "C" int icon_grid_slab_finish_gathered(const icon_mlp_t *mlp, int res, int z0, int z1, int za, int zb,
                                              const void *d_gathered, int64_t stride, int world, int rank,
                                              float *d_occ, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_slab_finish_gathered: null argument");
    if (!work->slab_ready || work->slab_res != res || work->slab_z0 != z0 || work->slab_z1 != z1)
        return fail(ICON_ERR_STATE, "icon_grid_slab_finish_gathered: no matching icon_grid_slab_features call on this workspace");
    ICON_ARG(za >= z0 && zb > za && zb <= z1, "icon_grid_slab_finish_gathered: planes [za, zb) must lie inside the slab");
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_slab_finish_gathered: MLP input width does not match the feature layout");
    FusedSigns fs{};
    if (work->slab_needs_patch) {
        if (d_gathered) {
            ICON_ARG(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, "icon_grid_slab_finish_gathered: bad world / rank");
            ICON_ARG(stride >= 8 && (stride & 7) == 0, "icon_grid_slab_finish_gathered: stride must be a multiple of 8 (int64 header)");
            fs.mode = kSignSeg; fs.gathered = (const int8_t *)d_gathered; fs.stride = stride; fs.world = world; fs.rank = rank;
        } else {
            fs = self_signs(work);              // no exchange: the slab's own list is the whole list
        }
    }
    // the slab stays prepared: its planes may be finished piece by piece (the next icon_grid_slab_features replaces it)
    return phase2(mlp, fs, d_occ, precision, work, (hipStream_t)stream, za, zb);
}

// ---- the MLP input rows of a call, handed out (regressors whose normalisation runs over the points of the call) ----
static int copy_rows_out(icon_work *work, float *d_rows, hipStream_t st)
{
    int rc = materialise_rows(work, self_signs(work), st);
    if (rc) return rc;
    ICON_HIP(hipMemcpyAsync(d_rows, work->d_x, (size_t)work->q_N * kXRow * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ICON_OK;
}

extern "C" int icon_query_rows(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior_type, float sdf_clip, int cmap_mode,
                               const float *h_calib, const float *d_calib, const float *d_points, int64_t N, float *d_rows,
                               int search, icon_work_t *work, void *stream)
{
    ICON_ARG(work && d_points && d_rows, "icon_query_rows: null argument");
    ICON_ARG(N >= 0, "icon_query_rows: negative N");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    if (N == 0) return ICON_OK;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_work(work, N, false))) return rc;
    Calib cal;
    static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(cal.m, h_calib ? h_calib : ident, sizeof(cal.m));
    cal.d = d_calib;
    LatticeMap L{};
    work->slab_ready = false;
    if ((rc = phase1<false>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, d_points, N, search, nullptr, work, st))) return rc;
    return copy_rows_out(work, d_rows, st);
}

extern "C" int icon_grid_rows(const icon_mesh_t *mesh, const icon_feat_t *feat, int prior_type, float sdf_clip, int cmap_mode,
                              int res, int z0, int z1, float *d_rows, int search, icon_work_t *work, void *stream)
{
    ICON_ARG(work && d_rows, "icon_grid_rows: null argument");
    int c0 = 0;
    int rc = check_prior(mesh, feat, prior_type, &c0);
    if (rc) return rc;
    LatticeMap L;
    // no shell skip: the rows of the shell are inputs of the call's statistics like any other (HGPIFuNet.py:361-363 masks afterwards)
    if ((rc = lattice_map(res, z0, z1, nullptr, sdf_clip, false, &L))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = (int64_t)L.nz * res * res;
    ICON_ARG(N < (1ll << 31), "icon_grid_rows: more than 2^31 points in one slab");
    if ((rc = ensure_work(work, N, false))) return rc;
    Calib cal{};
    work->slab_ready = false;
    if ((rc = phase1<true>(mesh, feat, prior_type, sdf_clip, cmap_mode, cal, L, nullptr, N, search, nullptr, work, st))) return rc;
    return copy_rows_out(work, d_rows, st);
}

extern "C" int icon_debug_traversal_stats(const icon_mesh_t *mesh, int res, int z0, int z1, uint64_t out[4])
{
    ICON_ARG(mesh && out, "icon_debug_traversal_stats: null argument");
    LatticeMap L;
    int rc = lattice_map(res, z0, z1, mesh, 0.05f, true, &L);
    if (rc) return rc;
    unsigned long long *d = nullptr;
    ICON_HIP(hipMalloc((void **)&d, 8 * sizeof(unsigned long long)));
    ICON_HIP(hipMemset(d, 0, 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_traversal_stats, dim3((unsigned)(L.tx * L.ty * L.tz)), dim3(kBlock), 0, 0, mesh->dev, L, d,
                       getenv("ICON_AMD_STATS_SEEDED") ? atoi(getenv("ICON_AMD_STATS_SEEDED")) : 0);
    unsigned long long h[8];
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ICON_ERR_HIP, std::string("traversal stats: ") + hipGetErrorString(e));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[4];
    return ICON_OK;
}

extern "C" int icon_grid_eval_slab(const icon_mesh_t *mesh, const icon_feat_t *feat, const icon_mlp_t *mlp,
                                   int prior_type, float sdf_clip, int cmap_mode,
                                   int res, int z0, int z1, float *d_occ,
                                   int search, int precision, icon_work_t *work, void *stream)
{
    ICON_ARG(mlp && work && d_occ, "icon_grid_eval_slab: null argument");
    int rc = icon_grid_slab_features(mesh, feat, prior_type, sdf_clip, cmap_mode, res, z0, z1, nullptr, nullptr, search, work, stream);
    if (rc) return rc;
    ICON_ARG(work->slab_c0 == mlp->c0, "icon_grid_eval_slab: MLP input width does not match the feature layout");
    // the slab's own sign list is the whole list (single call == whole lattice or caller's choice)
    work->slab_ready = false;
    return phase2(mlp, self_signs(work), d_occ, precision, work, (hipStream_t)stream, z0, z1);
}

// the shared walks' error record (see work_check_err): ICON_ERR_STATE once per reported error, then clear again.  Does not
// synchronise - a caller that wants the verdict on a particular launch synchronises its stream first.
extern "C" int icon_work_status(icon_work_t *work)
{
    ICON_ARG(work != nullptr, "icon_work_status: work is null");
    return work_check_err(work);
}

// test / A-B switches by name (process-wide): "lattice_fast" 0 / 1; "share_waves" wavefronts per shared walk (-1 = by launch size;
// 4: the adaptive schedule's searches only); "share_ring" forced ring size of the shared walks (0 = 64),
// "share_lose_push" the ticket of the push that is announced but never stored (0 = none), "share_spin_log2" wait bound 2^n polls
// (0 = 2^18).  Production leaves all of them alone.
extern "C" int icon_debug_set_option(const char *key, int value)
{
    ICON_ARG(key != nullptr, "icon_debug_set_option: key is null");
    const std::string k(key);
    if (k == "lattice_fast") g_lattice_fast = value ? 1 : 0;
    else if (k == "share_waves") { ICON_ARG(value == -1 || value == 1 || value == 4 || value == 8 || value == 16, "share_waves: -1 (by launch size), 1, 4, 8 or 16"); g_share_waves = value; }
    else if (k == "share_ring") { ICON_ARG(value == 0 || (value >= 2 && value <= 64 && (value & (value - 1)) == 0), "share_ring: 0 or a power of two in 2..64"); g_share_ring = value; }
    else if (k == "share_lose_push") { ICON_ARG(value >= 0, "share_lose_push: a ticket >= 1, or 0"); g_share_lose = value; }
    else if (k == "share_spin_log2") { ICON_ARG(value >= 0 && value < 30, "share_spin_log2: 0..29"); g_share_spin_log2 = value; }
    else return fail(ICON_ERR_ARG, "icon_debug_set_option: unknown key " + k);
    return ICON_OK;
}

extern "C" int icon_debug_set_shell_skip(int on)
{
    g_shell_skip = on ? 1 : 0;
    return ICON_OK;
}

extern "C" int icon_work_set_tie_rule(icon_work_t *work, int rule, int ulps)
{
    ICON_ARG(work != nullptr, "icon_work_set_tie_rule: work is null");
    ICON_ARG(rule == 0 || rule == 1, "icon_work_set_tie_rule: rule must be 0 (definition) or 1 (highest index within ulps)");
    ICON_ARG(ulps >= 0 && ulps <= 255, "icon_work_set_tie_rule: ulps must be 0..255");
    work->tie_rule = rule; work->tie_ulps = ulps;
    return ICON_OK;
}

extern "C" int icon_sdf_query_ties(const icon_mesh_t *mesh, const float *d_points, int64_t N,
                                   int32_t *d_face, int32_t *d_face2, uint8_t *d_ulps, void *stream)
{
    ICON_ARG(mesh && d_points && d_face && d_face2 && d_ulps, "icon_sdf_query_ties: null argument");
    ICON_ARG(N >= 0, "icon_sdf_query_ties: negative N");
    if (N == 0) return ICON_OK;
    const int64_t nb = (N + kBlock - 1) / kBlock;
    ICON_ARG(nb < (1ll << 31), "icon_sdf_query_ties: N too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    icon_work tmp;
    static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const int32_t *perm = nullptr;
    int rc = morton_order(&tmp, d_points, ident, nullptr, N, st, &perm);
    if (!rc) {
        hipLaunchKernelGGL(k_nearest_ties, dim3((unsigned)nb), dim3(kBlock), 0, st, mesh->dev, d_points, N, perm, d_face, d_face2, d_ulps);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = fail(ICON_ERR_HIP, "icon_sdf_query_ties: launch failed");
    }
    (void)hipFree(tmp.d_sort_keys); (void)hipFree(tmp.d_sort_idx); (void)hipFree(tmp.d_sort_tmp);
    return rc;
}
