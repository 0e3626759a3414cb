// Mesh ray casting for the RayCaster sensor (reference: isaaclab/utils/warp/ops.py:24-127,
// isaaclab/utils/warp/kernels.py:13-75, isaaclab/sensors/ray_caster/ray_caster.py:220-260).
//
// Acceleration structure: a 4-wide BVH built on the host at ilx_mesh_create.
//   * top-down object-median splits (longest centroid axis, twice per node -> four children); ranges of
//     <= 4 triangles become leaves;
//   * nodes are numbered breadth first, so the top of the tree is one contiguous prefix that every block
//     stages into LDS once; each node is one 128-byte line: child boxes as SoA float4s + 4 child references;
//   * leaf triangles are stored in leaf order as (v0, e1, e2, original face id) in three float4s.
// Traversal: one lane per ray, per-lane stack of child references in LDS ([slot][thread] -> conflict free),
// children pushed far-to-near.  Box culling is conservative (boxes are inflated at build time and the slab test
// carries slack), so the result equals testing every triangle: closest t, ties to the lowest face index.
//
// Column grid: rays parallel to the z axis (the height scanner's pattern) need no hierarchy at all.  The mesh's
// xy footprint is cut into square cells and every cell knows the triangles whose (inflated) xy projection overlaps
// it; such a ray tests exactly the triangles of the one cell under its origin.  A cell is one 32-bit word: most
// cells see one to three triangles that are neighbours in the record array and name that run directly, the others
// point at a shared candidate list (cells with the same candidates share one list).  Cell words and triangle
// records are both stored in tiles of 16 x 16 cells, so the rays of one wavefront (a patch of one env's scan
// pattern) touch few cache lines whatever the sensor's yaw.  A caller that knows its rays are z-parallel passes
// ILX_RAYS_Z_PARALLEL and gets the column kernel (no LDS, full occupancy); rays that turn out not to be z-parallel
// are left to a tree-walk launch that follows, so the hint only affects speed, never results.
#include "common.h"
#include "lattice.h"
#include "raytri.h"
#include "scene_update_dev.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <unordered_map>
#include <vector>

namespace ilx {

constexpr int kLeafSize = 4;
constexpr int kStackSlots = 40;       // 3 pushes net per level, depth <= 12, plus slack
constexpr int kMaxDepth = 12;
constexpr int kCachedNodes = 85;      // levels 0..3 of the 4-wide tree
constexpr int kRayBlock = 256;
constexpr int kEmptyRef = INT32_MIN;
constexpr double kGridCellsPerTriangle = 8.0;  // column grid resolution over the mesh's bounding rectangle (measured on the
                                               // 2.56 M-triangle terrain: 2 -> 164 us, 4 -> 140, 8 -> 116, 16 -> 117)
constexpr int kGridTile = 16;                  // cell words, column triangles and vertices are stored in tiles of 16 x 16 cells
constexpr double kGridPhase = 0.5;             // the grid starts this fraction of a cell before the mesh: triangle edges
                                               // of a regular height field then run through cells, not along cell borders
// Cell word (32 bits).  Triangle records are stored tile by tile, so the candidates of a cell sit in a narrow window of
// record positions around the first record of the cell's tile (tile_base, one word per tile, read next to the cell
// word); positions inside that window fit a few bits and a word can name up to three unrelated candidates itself:
//   0xFFFFFFFF          empty
//   kind 0  RUN         bits 13..0 = first record - wide window base, bits 17..14 = length - 1  (records first .. first + len - 1;
//                       a run may contain a record that is no candidate of the cell: testing it changes nothing)
//   kind 1  PAIR        bits 13..0 and 27..14 = two records - wide window base
//   kind 2  TRIPLE      bits 8..0, 17..9, 26..18 = three records - narrow window base
//   kind 3..5  FAR RUN  bits 28..0 = absolute position of the first of kind - 2 records (candidates outside the window,
//                       e.g. one huge triangle under many tiles)
//   kind 7  LIST        bits 28..0 = p: the 16-byte word cell_lists[p] = (count, r0, r1, r2), followed by (r3, r4, r5, r6) ...
//                       with absolute record positions, for everything else; cells with equal candidates share a list
// wide window base = tile_base - min(tile_base, 4096), narrow window base = tile_base - min(tile_base, 64).
constexpr int kFlagSlots = 64;  // hinted casts in flight on one mesh (on different streams) that keep their own flag
constexpr uint32_t kCellEmpty = 0xFFFFFFFFu;
constexpr uint32_t kCellIndexMask = 0x1FFFFFFFu;
constexpr int kCellKindShift = 29;
constexpr uint32_t kCellRun = 0, kCellPair = 1, kCellTriple = 2, kCellFar1 = 3, kCellList = 7;  // far runs: kinds 3, 4, 5
constexpr uint32_t kMaxRun = 16;     // longest run a cell word can name
constexpr uint32_t kLooseRun = 3;    // a candidate set spanning at most this many records is named as the whole span
constexpr uint32_t kWideBits = 14, kWideBack = 4096, kNarrowBits = 9, kNarrowBack = 64;

struct alignas(16) Node {
    float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4];
    int32_t ref[4];   // >= 0: node index;  < 0 (and != kEmptyRef): leaf -(1 + (first << 2 | (count - 1)))
    int32_t pad[4];
};
static_assert(sizeof(Node) == 128, "one node per 128-byte line");

struct MeshDev {
    const Node* nodes;
    const float4* tris;  // 3 float4 per triangle, leaf order
    int32_t num_nodes;
    int32_t num_tris;
    // column grid over the xy footprint
    const uint32_t* cells;       // one word per cell, tile order (cell_slot)
    const uint32_t* tile_base;   // per tile: position of its first triangle record
    const uint4* cell_lists;     // candidate lists of the cells whose candidates are not one short run
    const uint4* col_tris;       // per triangle, column order: three positions in col_verts, original face id
    const float4* col_verts;     // vertices (x, y, z, -) in tile order of their xy: neighbouring triangles share them
    float gx0, gy0, gx1, gy1, inv_cell;
    int32_t nx, ny, ntx;         // cells per axis, tiles per row
    // height-field patches (lattice.h), or num_patches == 0
    const LatticePatch* patches;
    const float* lat_z;
    const float* lat_lines;
    const uint32_t* lat_mask;
    int32_t num_patches;
    // patch finder: kPatchBins x kPatchBins bins over the union of the patch extents, up to four patches that overlap a
    // bin (-1: none; -2 in the first entry: more than four, search them all)
    const short4* patch_bins;
    float pb_x0, pb_y0, pb_inv_x, pb_inv_y;
    unsigned long long* lat_served;  // diagnostics: rays served from the height grid are counted here (null: not counted)
};

// Slot of cell (cx, cy) in `cells`: tiles row-major; inside a tile, blocks of 8 x 4 cells (one 128-byte line each,
// so that a line covers a compact patch whatever direction the rays of a wavefront are lined up in), cells row-major
// inside a block.
__host__ __device__ inline int64_t cell_slot(int cx, int cy, int ntx) {
    const int tx = cx % kGridTile, ty = cy % kGridTile;
    const int block = (ty / 4) * (kGridTile / 8) + tx / 8;
    return ((static_cast<int64_t>(cy / kGridTile) * ntx + cx / kGridTile) * (kGridTile * kGridTile)) + block * 32 + (ty % 4) * 8 +
           tx % 8;
}

// Cell coordinate of x along one axis.  The same expression places triangles at build time and rays at cast
// time; it is monotone in x, so a ray inside a triangle's box lands in a cell that lists the triangle.
__host__ __device__ inline int cell_coord(float x, float g0, float inv_cell, int n) {
    const int c = static_cast<int>(floorf((x - g0) * inv_cell));
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

struct Hit {
    float t;
    int32_t face;
    f3 normal_raw;  // e1 x e2 of the hit face (not normalised)
};

// Closed-interval slab test with explicit handling of zero direction components (a ray that runs inside a
// slab's boundary plane counts as overlapping).  Returns the entry distance, or +inf when the box is missed.
ILX_DEV float ray_box(f3 o, f3 inv, bool zx, bool zy, bool zz, float bminx, float bminy, float bminz, float bmaxx,
                      float bmaxy, float bmaxz, float t_limit) {
    const float inf = __builtin_inff();
    float t0 = 0.0f, t1 = t_limit;
    bool ok = true;
    if (zx) {
        ok = ok && (o.x >= bminx && o.x <= bmaxx);
    } else {
        const float a = (bminx - o.x) * inv.x, b = (bmaxx - o.x) * inv.x;
        t0 = fmaxf(t0, fminf(a, b));
        t1 = fminf(t1, fmaxf(a, b));
    }
    if (zy) {
        ok = ok && (o.y >= bminy && o.y <= bmaxy);
    } else {
        const float a = (bminy - o.y) * inv.y, b = (bmaxy - o.y) * inv.y;
        t0 = fmaxf(t0, fminf(a, b));
        t1 = fminf(t1, fmaxf(a, b));
    }
    if (zz) {
        ok = ok && (o.z >= bminz && o.z <= bmaxz);
    } else {
        const float a = (bminz - o.z) * inv.z, b = (bmaxz - o.z) * inv.z;
        t0 = fmaxf(t0, fminf(a, b));
        t1 = fminf(t1, fmaxf(a, b));
    }
    // slack: rounding in the slab arithmetic must never cull a box whose triangle the exact formula hits
    ok = ok && (t0 <= t1 * (1.0f + 4.0e-6f) + 1.0e-6f);
    return ok ? t0 : inf;
}

// `stack` points at this thread's column of an LDS array laid out [slot][blockDim.x].
// `cache` holds the first `ncache` nodes (LDS).
template <bool WANT_NORMAL>
ILX_DEV Hit trace(const MeshDev& m, const Node* cache, int ncache, int32_t* stack, int stride, f3 o, f3 d,
                  float max_dist, int32_t* counters = nullptr) {
    int n_nodes = 0, n_tris = 0;
    const float inf = __builtin_inff();
    Hit best{inf, -1, {0.0f, 0.0f, 0.0f}};
    float limit = max_dist;  // candidates must satisfy t < max_dist; afterwards t <= best.t
    const bool zx = d.x == 0.0f, zy = d.y == 0.0f, zz = d.z == 0.0f;
    const f3 inv = {zx ? 0.0f : div_(1.0f, d.x), zy ? 0.0f : div_(1.0f, d.y), zz ? 0.0f : div_(1.0f, d.z)};
    int sp = 0;
    stack[0] = 0;
    sp = 1;
    while (sp > 0) {
        const int32_t ref = stack[(--sp) * stride];
        if (ref >= 0) {
            ++n_nodes;
            const Node* n = (ref < ncache) ? (cache + ref) : (m.nodes + ref);
            const float4 mnx = *reinterpret_cast<const float4*>(n->minx), mny = *reinterpret_cast<const float4*>(n->miny),
                         mnz = *reinterpret_cast<const float4*>(n->minz), mxx = *reinterpret_cast<const float4*>(n->maxx),
                         mxy = *reinterpret_cast<const float4*>(n->maxy), mxz = *reinterpret_cast<const float4*>(n->maxz);
            const int4 refs = *reinterpret_cast<const int4*>(n->ref);
            const float cut = limit * (1.0f + 1.0e-5f) + 1.0e-6f;
            float tn[4];
            int32_t rf[4] = {refs.x, refs.y, refs.z, refs.w};
            tn[0] = ray_box(o, inv, zx, zy, zz, mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, cut);
            tn[1] = ray_box(o, inv, zx, zy, zz, mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, cut);
            tn[2] = ray_box(o, inv, zx, zy, zz, mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, cut);
            tn[3] = ray_box(o, inv, zx, zy, zz, mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, cut);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (rf[c] == kEmptyRef) tn[c] = inf;
            // sort descending by entry distance (5-comparator network), so the nearest child is popped first
#define ILX_CSWAP(i, j)                    \
    if (tn[i] < tn[j]) {                   \
        const float tf = tn[i];            \
        tn[i] = tn[j];                     \
        tn[j] = tf;                        \
        const int32_t ti = rf[i];          \
        rf[i] = rf[j];                     \
        rf[j] = ti;                        \
    }
            ILX_CSWAP(0, 1) ILX_CSWAP(2, 3) ILX_CSWAP(0, 2) ILX_CSWAP(1, 3) ILX_CSWAP(1, 2)
#undef ILX_CSWAP
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (tn[c] < inf && sp < kStackSlots) {
                    stack[sp * stride] = rf[c];
                    ++sp;
                }
            }
        } else {
            const int32_t code = -(ref + 1);
            const int32_t first = code >> 2, count = (code & 3) + 1;
            n_tris += count;
            for (int k = 0; k < count; ++k) {
                const float4* tp = m.tris + 3 * static_cast<int64_t>(first + k);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const f3 v0 = {a.x, a.y, a.z}, e1 = {a.w, b.x, b.y}, e2 = {b.z, b.w, c.x};
                const int32_t face = __float_as_int(c.y);
                float t;
                if (ray_triangle(o, d, v0, e1, e2, max_dist, t)) {
                    if (t < best.t || (t == best.t && face < best.face)) {
                        best.t = t;
                        best.face = face;
                        limit = t;
                        if (WANT_NORMAL) best.normal_raw = cross3(e1, e2);
                    }
                }
            }
        }
    }
    if (counters) {
        counters[0] = n_nodes;
        counters[1] = n_tris;
    }
    return best;
}

// A candidate of a z-parallel ray, given by its three vertices.  The edges are the same float subtractions the tree's
// triangle records were built with, so both paths evaluate ray_triangle on identical operands.
template <bool WANT_NORMAL>
ILX_DEV void test_column_vertices(f3 v0, f3 v1, f3 v2, int32_t face, f3 o, float dz, float max_dist, Hit& best) {
    const f3 e1 = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, e2 = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z};
    ColumnHit h;
    if (!ray_triangle_z_inside(o.x, o.y, dz, v0.x, v0.y, e1.x, e1.y, e2.x, e2.y, h)) return;
    float t;
    if (!ray_triangle_z_distance(h, o.z, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, max_dist, t)) return;
    if (t < best.t || (t == best.t && face < best.face)) {
        best.t = t;
        best.face = face;
        if (WANT_NORMAL) best.normal_raw = cross3(e1, e2);
    }
}

ILX_DEV f3 column_vertex(const MeshDev& m, uint32_t i) {
    const float4 v = m.col_verts[i];
    return {v.x, v.y, v.z};
}

template <bool WANT_NORMAL>
ILX_DEV void test_column_triangle(const MeshDev& m, uint32_t pos, f3 o, float dz, float max_dist, Hit& best) {
    const uint4 t = m.col_tris[pos];
    test_column_vertices<WANT_NORMAL>(column_vertex(m, t.x), column_vertex(m, t.y), column_vertex(m, t.z), static_cast<int32_t>(t.w),
                                      o, dz, max_dist, best);
}

// z-parallel ray: every candidate lives in the cell under the origin.  The kernel's time is fabric traffic, so the
// structures are small (4 bytes per cell, 16 per triangle, 16 per shared vertex); a wavefront's chain of dependent
// loads is cell word -> (list header) -> vertex indices of the first three candidates -> their vertices, each step
// with all its loads in flight together.  Only cells with more than three candidates go round a loop.
template <bool WANT_NORMAL>
ILX_DEV Hit trace_column(const MeshDev& m, f3 o, f3 d, float max_dist) {
    Hit best{__builtin_inff(), -1, {0.0f, 0.0f, 0.0f}};
    if (!(o.x >= m.gx0 && o.x <= m.gx1 && o.y >= m.gy0 && o.y <= m.gy1)) return best;
    const int cx = cell_coord(o.x, m.gx0, m.inv_cell, m.nx), cy = cell_coord(o.y, m.gy0, m.inv_cell, m.ny);
    const uint32_t word = m.cells[cell_slot(cx, cy, m.ntx)];
    const uint32_t base = m.tile_base[(cy / kGridTile) * m.ntx + cx / kGridTile];
    if (word == kCellEmpty) return best;
    const uint32_t kind = word >> kCellKindShift, p = word & kCellIndexMask;
    const bool listed = kind == kCellList;
    const uint32_t wide = base - min(base, kWideBack), narrow = base - min(base, kNarrowBack);
    uint32_t n, p0, p1, p2;
    if (listed) {
        const uint4 head = m.cell_lists[p];
        n = head.x, p0 = head.y, p1 = head.z, p2 = head.w;
    } else if (kind == kCellRun) {
        p0 = wide + (word & ((1u << kWideBits) - 1)), p1 = p0 + 1, p2 = p0 + 2;
        n = ((word >> kWideBits) & 15u) + 1;
    } else if (kind == kCellPair) {
        p0 = wide + (word & ((1u << kWideBits) - 1)), p1 = wide + ((word >> kWideBits) & ((1u << kWideBits) - 1)), p2 = p0;
        n = 2;
    } else if (kind == kCellTriple) {
        p0 = narrow + (word & 511u), p1 = narrow + ((word >> kNarrowBits) & 511u), p2 = narrow + ((word >> (2 * kNarrowBits)) & 511u);
        n = 3;
    } else {
        p0 = p, p1 = p + 1, p2 = p + 2;
        n = kind - kCellFar1 + 1;
    }
    p1 = n > 1 ? p1 : p0;  // absent candidates repeat the first one: the loads below need no branch
    p2 = n > 2 ? p2 : p0;
    const uint4 t0 = m.col_tris[p0], t1 = m.col_tris[p1], t2 = m.col_tris[p2];
    const f3 a0 = column_vertex(m, t0.x), a1 = column_vertex(m, t0.y), a2 = column_vertex(m, t0.z);
    const f3 b0 = column_vertex(m, t1.x), b1 = column_vertex(m, t1.y), b2 = column_vertex(m, t1.z);
    const f3 c0 = column_vertex(m, t2.x), c1 = column_vertex(m, t2.y), c2 = column_vertex(m, t2.z);
    test_column_vertices<WANT_NORMAL>(a0, a1, a2, static_cast<int32_t>(t0.w), o, d.z, max_dist, best);
    if (n > 1) test_column_vertices<WANT_NORMAL>(b0, b1, b2, static_cast<int32_t>(t1.w), o, d.z, max_dist, best);
    if (n > 2) test_column_vertices<WANT_NORMAL>(c0, c1, c2, static_cast<int32_t>(t2.w), o, d.z, max_dist, best);
    if (listed) {
        for (uint32_t k = 3; k < n; k += 4) {
            const uint4 more = m.cell_lists[p + 1 + (k - 3) / 4];
            test_column_triangle<WANT_NORMAL>(m, more.x, o, d.z, max_dist, best);
            if (k + 1 < n) test_column_triangle<WANT_NORMAL>(m, more.y, o, d.z, max_dist, best);
            if (k + 2 < n) test_column_triangle<WANT_NORMAL>(m, more.z, o, d.z, max_dist, best);
            if (k + 3 < n) test_column_triangle<WANT_NORMAL>(m, more.w, o, d.z, max_dist, best);
        }
    } else {
        for (uint32_t k = 3; k < n; ++k) test_column_triangle<WANT_NORMAL>(m, p0 + k, o, d.z, max_dist, best);
    }
    return best;
}

ILX_DEV bool z_parallel(f3 d) { return d.x == 0.0f && d.y == 0.0f; }

ILX_DEV void stage_nodes(const MeshDev& m, Node* cache, int ncache) {
    // 128-byte nodes copied as float4s by the whole block
    const float4* src = reinterpret_cast<const float4*>(m.nodes);
    float4* dst = reinterpret_cast<float4*>(cache);
    for (int i = threadIdx.x; i < ncache * 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

struct CastArgs {
    MeshDev mesh;
    const float* starts;  // (n,3)
    const float* dirs;    // (n,3)
    float max_dist;
    float* hits;          // (n,3)
    float* dist;          // (n) or null
    float* normal;        // (n,3) or null
    int32_t* face;        // (n) or null
    int64_t n;
    int32_t* counters;    // (n,2) or null: nodes visited / triangles tested per ray (diagnostics)
    int32_t ticket;       // identifies this call among the calls that share the mesh's flag slots (never 0)
    int32_t* off_grid;    // column kernel: set to `ticket` when a ray was not z-parallel; tree kernel: null = all rays,
                          // else only the rays the column kernel left out (and nothing at all while the flag is 0)
};

ILX_DEV void write_hit(const Hit& h, f3 o, f3 d, float* hits, float* dist, float* normal, int32_t* face, int64_t i) {
    const float inf = __builtin_inff();
    const bool ok = h.face >= 0;
    // kernels.py:69: start + t * dir (no fused multiply-add)
    const f3 p = ok ? f3{o.x + h.t * d.x, o.y + h.t * d.y, o.z + h.t * d.z} : f3{inf, inf, inf};
    if (hits) store3(hits + 3 * i, p);
    if (dist) dist[i] = ok ? h.t : inf;
    if (face) face[i] = ok ? h.face : -1;
    if (normal) store3(normal + 3 * i, ok ? normalize3(h.normal_raw) : f3{inf, inf, inf});
}

__global__ void __launch_bounds__(kRayBlock) raycast_kernel(CastArgs a) {
    __shared__ Node cache[kCachedNodes];
    __shared__ int32_t stack[kStackSlots * kRayBlock];
    if (a.off_grid && *a.off_grid != a.ticket) return;  // mop-up launch with nothing to mop up
    const int ncache = min(a.mesh.num_nodes, kCachedNodes);
    stage_nodes(a.mesh, cache, ncache);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kRayBlock + threadIdx.x; i < a.n;
         i += static_cast<int64_t>(gridDim.x) * kRayBlock) {
        const f3 o = load3(a.starts + 3 * i), d = load3(a.dirs + 3 * i);
        if (a.off_grid && z_parallel(d)) continue;
        int32_t* cnt = a.counters ? a.counters + 2 * i : nullptr;
        const Hit h = a.normal ? trace<true>(a.mesh, cache, ncache, stack + threadIdx.x, kRayBlock, o, d, a.max_dist, cnt)
                               : trace<false>(a.mesh, cache, ncache, stack + threadIdx.x, kRayBlock, o, d, a.max_dist, cnt);
        write_hit(h, o, d, a.hits, a.dist, a.normal, a.face, i);
    }
}

__global__ void __launch_bounds__(kRayBlock) raycast_column_kernel(CastArgs a) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kRayBlock + threadIdx.x;
    if (i >= a.n) return;
    const f3 o = load3(a.starts + 3 * i), d = load3(a.dirs + 3 * i);
    if (!z_parallel(d)) {  // left to the tree kernel that follows
        *a.off_grid = a.ticket;
        return;
    }
    const Hit h = a.normal ? trace_column<true>(a.mesh, o, d, a.max_dist) : trace_column<false>(a.mesh, o, d, a.max_dist);
    write_hit(h, o, d, a.hits, a.dist, a.normal, a.face, i);
}

// RayCaster._update_buffers_impl fused with the cast: sensor pose -> world rays -> hits (+ height-scan term).
struct ScanArgs {
    MeshDev mesh;
    const float* pos_w;       // sensor parent position, row stride pos_stride floats
    const float* quat_w;      // wxyz, row stride quat_stride floats
    int64_t pos_stride, quat_stride;
    const float* drift;       // (n,3) added to the position, or null
    float* pos_out;           // (n,3) / (n,4): RayCasterData.pos_w / quat_w of the refreshed envs, or null
    float* quat_out;
    const float* ray_starts;  // (r,3) local
    const float* ray_dirs;    // (r,3) local
    const uint8_t* env_mask;  // (n) or null: only flagged envs are refreshed
    int yaw_only;
    float max_dist;
    float* ray_hits_w;        // (n,r,3)
    float* height;            // (n,r) or null: pos_z - hit_z - offset  (observations.py:165-173)
    float height_offset;
    int64_t n;
    int r;
    int32_t ticket;           // as in CastArgs
    int32_t* off_grid;
    // SensorBase clocks of the refreshed envs (sensor_base.py:284-293), or null: timestamp_last_update <- timestamp and
    // is_outdated <- 0 where is_outdated was set.  Done by the launch that follows the column kernel.
    const float* clock_now;
    float* clock_last;
    uint8_t* outdated;
};

// What the column kernel leaves in ray_hits_w[..., 0] of a ray it could not serve (a quiet NaN no cast produces: hits
// are finite or +inf): the tree launch that follows picks exactly these rays up, without reading the env mask again.
constexpr uint32_t kLeftOverBits = 0x7FC0DEADu;

// Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8), each with its own L2.  This maps the
// hardware block index to a logical one such that consecutive logical blocks -- neighbouring rays, which read the
// same cells and triangles -- share an XCD.
ILX_DEV unsigned xcd_block(unsigned b, unsigned nblocks) {
    constexpr unsigned kXcds = 8;
    const unsigned per = nblocks / kXcds, full = per * kXcds;
    if (b >= full) return b;  // the ragged tail keeps its place
    return (b % kXcds) * per + b / kXcds;
}

// World ray i = (env, ray) of the sensor; false when the env is masked out.
// `yaw` (optional): this env's yaw-only quaternion, precomputed once per env of the block.
// `ignore_mask`: the caller already knows that the ray belongs to a refreshed env.
ILX_DEV bool sensor_ray(const ScanArgs& a, int64_t i, f3& o, f3& d, float& pos_z, const f4* yaw = nullptr,
                        bool ignore_mask = false) {
    const int64_t env = i / a.r;
    const int ray = static_cast<int>(i - env * a.r);
    if (!ignore_mask && a.env_mask && !a.env_mask[env]) return false;
    f3 pos = load3(a.pos_w + a.pos_stride * env);
    const f4 q = load4u(a.quat_w + a.quat_stride * env);
    if (a.drift) {  // ray_caster.py:233-240
        const f3 dr = load3(a.drift + 3 * env);
        pos = {pos.x + dr.x, pos.y + dr.y, pos.z + dr.z};
    }
    if (a.pos_out && ray == 0) {
        store3(a.pos_out + 3 * env, pos);
        float* qo = a.quat_out + 4 * env;
        qo[0] = q.w, qo[1] = q.x, qo[2] = q.y, qo[3] = q.z;
    }
    const f3 ls = load3(a.ray_starts + 3 * ray), ld = load3(a.ray_dirs + 3 * ray);
    if (a.yaw_only) {  // ray_caster.py:243-247: starts rotated by yaw only, directions unrotated
        o = yaw ? quat_apply(*yaw, ls) : quat_apply_yaw(q, ls);
        d = ld;
    } else {  // ray_caster.py:248-252
        o = quat_apply(q, ls);
        d = quat_apply(q, ld);
    }
    o = {o.x + pos.x, o.y + pos.y, o.z + pos.z};
    pos_z = pos.z;
    return true;
}

ILX_DEV void write_scan(const ScanArgs& a, const Hit& h, f3 o, f3 d, float pos_z, int64_t i) {
    write_hit(h, o, d, a.ray_hits_w, nullptr, nullptr, nullptr, i);
    if (a.height) {
        const float hz = (h.face >= 0) ? (o.z + h.t * d.z) : __builtin_inff();
        a.height[i] = pos_z - hz - a.height_offset;
    }
}

__global__ void __launch_bounds__(kRayBlock) raycaster_update_kernel(ScanArgs a) {
    __shared__ Node cache[kCachedNodes];
    __shared__ int32_t stack[kStackSlots * kRayBlock];
    const bool mop_up = a.off_grid != nullptr;
    if (mop_up && a.outdated) {  // the column kernel has read the mask: the refreshed envs record the time
        for (int64_t e = static_cast<int64_t>(blockIdx.x) * kRayBlock + threadIdx.x; e < a.n;
             e += static_cast<int64_t>(gridDim.x) * kRayBlock) {
            if (a.outdated[e]) {
                a.clock_last[e] = a.clock_now[e];
                a.outdated[e] = 0;
            }
        }
    }
    if (mop_up && *a.off_grid != a.ticket) return;  // nothing to mop up
    const int ncache = min(a.mesh.num_nodes, kCachedNodes);
    stage_nodes(a.mesh, cache, ncache);
    const int64_t total = a.n * a.r;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kRayBlock + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kRayBlock) {
        if (mop_up && __float_as_uint(a.ray_hits_w[3 * i]) != kLeftOverBits) continue;
        f3 o, d;
        float pos_z;
        if (!sensor_ray(a, i, o, d, pos_z, nullptr, mop_up)) continue;
        const Hit h = trace<false>(a.mesh, cache, ncache, stack + threadIdx.x, kRayBlock, o, d, a.max_dist);
        write_scan(a, h, o, d, pos_z, i);
    }
}

__global__ void __launch_bounds__(kRayBlock) raycaster_update_column_kernel(ScanArgs a) {
    // the yaw-only rotation (atan2, sin, cos, normalise) is per env, not per ray: the first threads of the block
    // compute it for the envs the block's rays belong to (at most one env per thread)
    __shared__ f4 yaw_q[kRayBlock];
    const int64_t first = static_cast<int64_t>(xcd_block(blockIdx.x, gridDim.x)) * kRayBlock;
    const int64_t total = a.n * a.r;
    const int64_t env0 = first / a.r;
    if (a.yaw_only) {
        const int64_t last = (first + kRayBlock - 1 < total ? first + kRayBlock - 1 : total - 1) / a.r;
        const int64_t e = env0 + threadIdx.x;
        if (e <= last) yaw_q[threadIdx.x] = yaw_quat(load4u(a.quat_w + a.quat_stride * e));
        __syncthreads();
    }
    const int64_t i = first + threadIdx.x;
    if (i >= total) return;
    f3 o, d;
    float pos_z;
    if (!sensor_ray(a, i, o, d, pos_z, a.yaw_only ? &yaw_q[i / a.r - env0] : nullptr)) return;
    if (!z_parallel(d)) {  // left to the tree kernel that follows
        *a.off_grid = a.ticket;
        a.ray_hits_w[3 * i] = __uint_as_float(kLeftOverBits);
        return;
    }
    const Hit h = trace_column<false>(a.mesh, o, d, a.max_dist);
    write_scan(a, h, o, d, pos_z, i);
}

// ------------------------------------------------------------------------------------------------
// Sensor refresh over height-field patches (lattice.h): a preparation launch, then one wavefront per env.
//
// An env's scan pattern covers a patch of a couple of metres, i.e. a few hundred lattice vertices.  The wavefront
// finds the bounding rectangle of its ray starts, copies the z values, the lattice lines and the simple-cell bits
// under it into LDS with row-contiguous loads, and every ray then intersects the two faces of its cell out of LDS.
// Per env this reads about 25 x 25 floats of mesh data instead of ~190 cell words, ~400 index records and ~1000
// vertices scattered over partly used 128-byte lines.  Rays that are not strictly inside a simple cell, envs whose
// rectangle is not inside one patch or exceeds the window, and meshes without patches use the general structures.
// ------------------------------------------------------------------------------------------------
constexpr int kLatWin = 32;         // lattice lines per axis that an env's staged window can hold
constexpr int kLatMaskWords = 2;    // 31 cells starting anywhere inside a 32-bit word touch at most two words
// Wavefronts (= envs) per workgroup of the cast.  One: the wavefronts' lives differ (3.9 to 11 us at 16 384 envs: where the
// window's lines come from, how many wavefronts share the SIMD), and a slot that waits for the slowest of a workgroup's
// wavefronts is a slot the second round cannot start in (4 per workgroup: 18.5 us per launch, 1: 18.0).
constexpr int kLatWaves = 1;
constexpr int kLatKeep = 3;         // rays per lane whose pattern entry stays in registers (patterns up to 192 rays)
constexpr int kLatMaxRays = 1024;   // longest scan pattern served per env; longer ones use the per-ray kernels
constexpr float kLatMargin = 1.0f / 32768.0f;   // of the cell width; the barycentric slack is 1e-6 of a face
constexpr float kLatDiagonal = 1.0f / 16384.0f; // of the cell area: closer to the diagonal than this, both faces are tested
#ifndef ILX_CAST_ATTR
#define ILX_CAST_ATTR __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(80)))
#endif
constexpr int kLatNotDue = -2;      // LatEnvRec::sel of an env that this refresh leaves alone

// Per env, written by the preparation launch and read (as scalars) by the env's wavefront in the cast launch.  The
// scalar work of an env -- mask, drift, clocks, yaw angle -> quaternion (atan2, sin, cos), the patch search, the window
// of lattice lines -- is done once by one thread there, instead of as whole wavefront instruction streams here, and the
// wavefront's first instructions can already be the loads of its window.
struct alignas(16) LatEnvRec {
    float px, py, pz;        // sensor position (+ drift)
    float rw, rx, ry, rz;    // yaw part of the sensor orientation
    int32_t sel;             // patch under the env's rays, -1: none, kLatNotDue: env not refreshed
    int32_t i_lo, j_lo, wi, wj;  // staged window: lines i_lo .. i_lo + wi - 1 (x), j_lo .. j_lo + wj - 1 (y)
    float x0, y0, inv_dx, inv_dy;            // the fields of the patch that the cast needs
    int32_t nx, ny, mask_stride, face_base;
    uint32_t z_off, line_off, mask_off, pad;
};
static_assert(sizeof(LatEnvRec) == 96, "six 16-byte words");
// ilx_raycaster_refresh_static: the most envs its built-in buffer serves (at up to kLatMaxRays rays each), and the
// buffer's size: the pattern (16 bytes per ray, padded to whole records) and one record per env = 3,162,144 bytes.
constexpr int kStaticEnvs = 32768;
constexpr size_t kStaticScratchBytes =
    (static_cast<size_t>(kLatMaxRays) * 16 + sizeof(LatEnvRec) - 1) / sizeof(LatEnvRec) * sizeof(LatEnvRec) +
    static_cast<size_t>(kStaticEnvs) * sizeof(LatEnvRec);

// What one wavefront keeps in LDS for its env.
// (Lines first: a ray's two line pairs and its cell's four heights are then all within reach of the 8-bit dword offsets of a
// two-word LDS read from the index alone.)
struct LatStage {
    float lines[2 * kLatWin];  // x lines, then y lines
    float z[kLatWin * kLatWin];
    uint32_t mask[kLatWin];    // per window row: bit cj = the simple-cell bit of the row's cell cj (a window row has at most 31)
};

ILX_DEV float wave_min(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
ILX_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

#ifdef ILX_RAY_STAMPS  // diagnostics build only: wall-clock stamps (10 ns ticks) of the first wavefront of the first blocks
__device__ unsigned long long g_ray_stamps[8 * 4096];
__device__ unsigned long long g_prep_stamps[8 * 256];
#define ILX_PREP_STAMP(k) \
    if (threadIdx.x == 0 && blockIdx.x < 256) g_prep_stamps[blockIdx.x * 8 + (k)] = wall_clock64();
#define ILX_RAY_STAMP(k) \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_ray_stamps[blockIdx.x * 8 + (k)] = wall_clock64();
#else
#define ILX_RAY_STAMP(k)
#define ILX_PREP_STAMP(k)
#endif

// Preparation launch of the height-field refresh: one thread per env (yaw-only sensors, ray_caster.py:243-247).
//   pos_w / quat_w of RayCasterData, the sensor clocks (sensor_base.py:284-293), the yaw quaternion, the rectangle
//   that the rotated scan pattern covers, the patch that holds it and the window of lattice lines -> LatEnvRec.
// The pattern's xy box is reduced by every workgroup for itself (a few loads per thread).
constexpr int kPatchBins = 64;

// The per-env part of the preparation: the record of env `env` from its pose (already loaded: `pos`, `q`, drift `dr`)
// and the pattern's box; for a due env also RayCasterData.pos_w / quat_w and the refresh time.  `now` (or null): the
// env's clock value when the caller has it in a register (else it is read from clock_now).  `sel_before`: the `sel` that the
// env's record held (any value: it is only a guess at the patch).
ILX_DEV void prepare_env(const ScanArgs& a, int64_t env, bool due, f3 pos, f4 q, f3 dr, float bx0, float bx1, float by0, float by1,
                         LatEnvRec* __restrict__ recs, int32_t sel_before, const float* now = nullptr) {
    const float inf = __builtin_inff();
    LatEnvRec le{};
    le.sel = kLatNotDue;
    if (due) {
        le.sel = -1;
        if (a.drift) pos = {pos.x + dr.x, pos.y + dr.y, pos.z + dr.z};  // ray_caster.py:233-240
        if (a.pos_out) {
            store3(a.pos_out + 3 * env, pos);
            float* qo = a.quat_out + 4 * env;
            qo[0] = q.w, qo[1] = q.x, qo[2] = q.y, qo[3] = q.z;
        }
        if (a.outdated) {  // the refreshed env records the time
            a.clock_last[env] = now ? *now : a.clock_now[env];
            a.outdated[env] = 0;
        }
        const f4 rot = yaw_quat(q);
        le.px = pos.x, le.py = pos.y, le.pz = pos.z;
        le.rw = rot.w, le.rx = rot.x, le.ry = rot.y, le.rz = rot.z;
        if (bx0 <= bx1) {
            // The rotation is about z: every ray start lies in the rotated box of the pattern, up to rounding.  The
            // rectangle only decides which lines are staged -- a ray outside them is simply not served from LDS -- so a
            // small pad for the rounding is enough.
            float mnx = inf, mxx = -inf, mny = inf, mxy = -inf;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f3 o = quat_apply(rot, f3{(c & 1) ? bx1 : bx0, (c & 2) ? by1 : by0, 0.0f});
                mnx = fminf(mnx, o.x), mxx = fmaxf(mxx, o.x), mny = fminf(mny, o.y), mxy = fmaxf(mxy, o.y);
            }
            const float pad = 1.0e-5f * ((mxx - mnx) + (mxy - mny)) + 4.0e-7f * (fabsf(pos.x) + fabsf(pos.y));
            mnx = mnx + pos.x - pad, mxx = mxx + pos.x + pad;
            mny = mny + pos.y - pad, mxy = mxy + pos.y + pad;
            // the patch whose extent holds the rectangle (any such patch will do: where patches overlap, the cells are
            // not simple)
            ILX_PREP_STAMP(2)
            int sel = -1;
            LatticePatch P{};
            auto holds = [&](const LatticePatch& c) { return mnx >= c.x0 && mxx <= c.x1 && mny >= c.y0 && mxy <= c.y1; };
            // a patch that holds the rectangle overlaps the bin of its centre: at most four candidates, whose whole
            // descriptors are fetched together (one trip to memory for the test and for the fields of the chosen one)
            const int bx = max(0, min(kPatchBins - 1, static_cast<int>((0.5f * (mnx + mxx) - a.mesh.pb_x0) * a.mesh.pb_inv_x)));
            const int by = max(0, min(kPatchBins - 1, static_cast<int>((0.5f * (mny + mxy) - a.mesh.pb_y0) * a.mesh.pb_inv_y)));
            // The patch of the env's last refresh is asked for together with the bin: an env rarely leaves its patch from one
            // refresh to the next, and when the old patch still holds the rectangle the candidates' descriptors -- a third trip
            // to memory behind the pose and the bin -- are not needed.  (Which of several holding patches is taken makes no
            // difference to the hits, see above.)
            const int before = max(0, min(sel_before, a.mesh.num_patches - 1));
            LatticePatch old_patch = a.mesh.patches[before];
            int2 cand_words = *reinterpret_cast<const int2*>(a.mesh.patch_bins + (by * kPatchBins + bx));
            // no instruction: both are in registers here, whole.  (Otherwise the compiler reads the old patch one field at a
            // time, each behind the comparison that decides whether the next is needed, and the bin behind all of them.)
            asm volatile("" : "+v"(old_patch.x0), "+v"(old_patch.y0), "+v"(old_patch.x1), "+v"(old_patch.y1), "+v"(old_patch.inv_dx),
                         "+v"(old_patch.inv_dy), "+v"(old_patch.nx), "+v"(old_patch.ny), "+v"(old_patch.z_off), "+v"(old_patch.line_off),
                         "+v"(old_patch.mask_off), "+v"(old_patch.mask_stride), "+v"(old_patch.face_base), "+v"(cand_words.x), "+v"(cand_words.y));
            const short4 cand = {static_cast<short>(cand_words.x), static_cast<short>(cand_words.x >> 16), static_cast<short>(cand_words.y),
                                 static_cast<short>(cand_words.y >> 16)};
            if (holds(old_patch)) {
                sel = before, P = old_patch;
            } else if (cand.x == -2) {
                for (int p = 0; p < a.mesh.num_patches && sel < 0; ++p) {
                    P = a.mesh.patches[p];
                    if (holds(P)) sel = p;
                }
            } else {
                const int ids[4] = {cand.x, cand.y, cand.z, cand.w};
                LatticePatch c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = a.mesh.patches[max(0, ids[k])];
#pragma unroll
                for (int k = 3; k >= 0; --k)   // the first candidate that holds it wins
                    if (ids[k] >= 0 && holds(c[k])) sel = ids[k], P = c[k];
            }
            ILX_PREP_STAMP(3)
            if (sel >= 0) {
                const int gx0 = lattice_guess(mnx, P.x0, P.inv_dx), gx1 = lattice_guess(mxx, P.x0, P.inv_dx);
                const int gy0 = lattice_guess(mny, P.y0, P.inv_dy), gy1 = lattice_guess(mxy, P.y0, P.inv_dy);
                le.i_lo = max(0, min(gx0 - 1, P.nx - 2));
                le.j_lo = max(0, min(gy0 - 1, P.ny - 2));
                le.wi = min(P.nx - 1, max(gx1 + 2, le.i_lo + 1)) - le.i_lo + 1;
                le.wj = min(P.ny - 1, max(gy1 + 2, le.j_lo + 1)) - le.j_lo + 1;
                if (le.wi <= kLatWin && le.wj <= kLatWin) le.sel = sel;
                le.x0 = P.x0, le.y0 = P.y0, le.inv_dx = P.inv_dx, le.inv_dy = P.inv_dy;
                le.nx = P.nx, le.ny = P.ny, le.mask_stride = P.mask_stride, le.face_base = P.face_base;
                le.z_off = P.z_off, le.line_off = P.line_off, le.mask_off = P.mask_off;
            }
        }
    }
    recs[env] = le;
}

// `box_cache` (or null): four floats that keep the pattern's box; `reuse`: they and `pattern` are those of this pattern.
__global__ void __launch_bounds__(256) raycaster_prepare_kernel(ScanArgs a, LatEnvRec* __restrict__ recs, float4* __restrict__ pattern,
                                                                float* __restrict__ box_cache, int reuse) {
    __shared__ float box_s[4][4];
    const float inf = __builtin_inff();
    ILX_PREP_STAMP(0)
    // the env's inputs go out first: they travel while the block reduces the pattern's box
    const int64_t env = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const bool due = env < a.n && (!a.env_mask || a.env_mask[env]);
    f3 pos{0.0f, 0.0f, 0.0f}, dr{0.0f, 0.0f, 0.0f};
    f4 q{1.0f, 0.0f, 0.0f, 0.0f};
    int32_t sel_before = -1;
    if (due) {
        sel_before = recs[env].sel;
        pos = load3(a.pos_w + a.pos_stride * env);
        q = load4u(a.quat_w + a.quat_stride * env);
        if (a.drift) dr = load3(a.drift + 3 * env);
    }
    float bx0 = inf, bx1 = -inf, by0 = inf, by1 = -inf;  // xy box of the z-parallel pattern entries
    if (reuse) {
        bx0 = box_cache[0], bx1 = box_cache[1], by0 = box_cache[2], by1 = box_cache[3];
    } else {
        for (int ray = threadIdx.x; ray < a.r; ray += 256) {
            const f3 l = load3(a.ray_starts + 3 * ray), d = load3(a.ray_dirs + 3 * ray);
            const bool zp = z_parallel(d);
            if (zp) {
                bx0 = fminf(bx0, l.x), bx1 = fmaxf(bx1, l.x);
                by0 = fminf(by0, l.y), by1 = fmaxf(by1, l.y);
            }
            // the pattern as the cast wants it: one 16-byte word per ray, local start and z of the direction -- NaN when
            // the direction is not z-parallel (such a ray never takes the height-grid path)
            if (blockIdx.x == 0) pattern[ray] = make_float4(l.x, l.y, l.z, zp ? d.z : __builtin_nanf(""));
        }
        bx0 = wave_min(bx0), bx1 = wave_max(bx1), by0 = wave_min(by0), by1 = wave_max(by1);
        if ((threadIdx.x & 63) == 0) {
            float* b = box_s[threadIdx.x >> 6];
            b[0] = bx0, b[1] = bx1, b[2] = by0, b[3] = by1;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            bx0 = fminf(bx0, box_s[w][0]), bx1 = fmaxf(bx1, box_s[w][1]);
            by0 = fminf(by0, box_s[w][2]), by1 = fmaxf(by1, box_s[w][3]);
        }
        if (box_cache && blockIdx.x == 0 && threadIdx.x == 0) box_cache[0] = bx0, box_cache[1] = bx1, box_cache[2] = by0, box_cache[3] = by1;
    }
    ILX_PREP_STAMP(1)
    if (env >= a.n) return;
    prepare_env(a, env, due, pos, q, dr, bx0, bx1, by0, by1, recs, sel_before);
    ILX_PREP_STAMP(4)
}

// InteractiveScene.update and the preparation launch of the height scanner's refresh as one launch (the last scene
// update before the scanner is read).  The two share no data but the scanner's clock, so they are dealt to different
// workgroups of one grid: the first `prep_blocks` workgroups own the scanner's clock -- one thread per env advances it
// and flags the env (sensor_base.py:197-205), then, knowing whether the env is due, turns the sensor pose into the
// env's record as raycaster_prepare_kernel does; the other workgroups are the scene update without that clock.  (The
// preparation as extra work of the update's own clock threads -- a few lanes per wavefront -- took 16 us: every
// wavefront of the update then runs the preparation's instruction stream.)  The pattern's box is the cached one.
__global__ void __launch_bounds__(256) scene_update_prepare_kernel(SceneArgs s, ScanArgs a, LatEnvRec* __restrict__ recs,
                                                                   const float* __restrict__ box_cache, unsigned prep_blocks) {
    request_kernel_arguments<kernel_argument_bytes<SceneArgs, ScanArgs, LatEnvRec*, const float*, unsigned>()>();
    if (blockIdx.x >= prep_blocks) {
        scene_update_thread(s, blockIdx.x - prep_blocks, false);
        return;
    }
    const int64_t env = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (env >= a.n) return;
    // every load of the thread goes out at once: the clock, the pose, the pattern's box
    const Clock r = read_clock(s.ray_ts, s.ray_ts_last, s.ray_outdated, env, 1, s.c.dt, s.ray_period);
    const f3 pos = load3(a.pos_w + a.pos_stride * env);
    const f4 q = load4u(a.quat_w + a.quat_stride * env);
    const f3 dr = a.drift ? load3(a.drift + 3 * env) : f3{0.0f, 0.0f, 0.0f};
    const float bx0 = box_cache[0], bx1 = box_cache[1], by0 = box_cache[2], by1 = box_cache[3];
    const int32_t sel_before = recs[env].sel;
    write_clock(s.ray_ts, s.ray_ts_last, s.ray_outdated, env, r, true);
    prepare_env(a, env, r.refresh, pos, q, dr, bx0, bx1, by0, by1, recs, sel_before, &r.ts_new);
}

// The staged lattice cell that the index estimate names for coordinate x (staged line k is line lo_abs + k of the patch),
// and whether x lies strictly inside a cell [lo, hi] by the margin.  On an evenly spaced lattice the estimate
// is the cell for every such x; where it is not (uneven lines, the window's edge, NaN) the ray is not served here.
// (The estimate only picks the cell that the test below then accepts or refuses: floor and conversion as one instruction,
// saturating like the conversion it replaces; the clamp to [0, n - 2], n >= 2, as a median.)
ILX_DEV int lattice_index(int n, float x, float x0, float inv, int lo_abs) {
    int guess, c;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(guess) : "v"((x - x0) * inv));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(c) : "v"(guess - lo_abs), "s"(n - 2));
    return c;
}
ILX_DEV bool lattice_inside(float x, float lo, float hi) {
    const float m = (hi - lo) * kLatMargin;
    return (x - lo) > m && (hi - x) > m;
}

// Cast launch of the height-field refresh: one wavefront per env, no barrier.  The wavefront reads its env's record
// (uniform: scalar registers), copies the window of the height grid into LDS and intersects its rays there; a
// wavefront's own LDS traffic is ordered.  Rays that are not strictly inside a simple cell use the column grid (or
// are left to the tree launch that follows).
// Registers: a SIMD admits floor(800 / (scalar registers rounded up to 16, + 16)) wavefronts, eight at most.  Left alone
// the compiler takes 98 scalar and 70 vector registers: six wavefronts per SIMD, 2.7 rounds of wavefronts for 16 384
// envs.  Held to 80 scalar registers (the surplus lives in lanes of a vector register) and eight wavefronts per SIMD
// (64 vector registers), eight fit -- and their LDS, 32 x 4.6 KB per CU: two rounds.
__global__ void __launch_bounds__(64 * kLatWaves) ILX_CAST_ATTR
raycaster_update_lattice_kernel(ScanArgs a, const LatEnvRec* __restrict__ recs, const float4* __restrict__ pattern) {
    __shared__ LatStage stage_s[kLatWaves];
    request_kernel_arguments<kernel_argument_bytes<ScanArgs, const LatEnvRec*, const float4*>()>();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float inf = __builtin_inff();
    const int64_t env = static_cast<int64_t>(xcd_block(blockIdx.x, gridDim.x)) * kLatWaves + wave;
    if (env >= a.n) return;
    ILX_RAY_STAMP(0)
    const LatEnvRec E = recs[env];

    // pattern entries of the lane: local start, z of the direction (NaN: not z-parallel)
    float4 kp[kLatKeep];
#pragma unroll
    for (int r = 0; r < kLatKeep; ++r) {
        const int ray = lane + 64 * r;
        kp[r] = ray < a.r ? pattern[ray] : make_float4(0.0f, 0.0f, 0.0f, __builtin_nanf(""));
    }
    const int sel = E.sel, i_lo = E.i_lo, j_lo = E.j_lo, wi = E.wi, wj = E.wj;
    if (sel == kLatNotDue) return;
    LatStage& st = stage_s[wave];
    const int w0 = j_lo >> 5;
    ILX_RAY_STAMP(1)
    if (sel >= 0) {
        // Every load of the window leaves at once -- one trip to memory -- and without a branch: a 16-byte load per lane
        // takes eight window rows (lane = row * 8 + quad of columns), four of them the whole 32 x 32 window.  A row past the
        // window's last repeats that one; the words of a quad past the window's last column belong to the same patch or to
        // the padding behind the last one (ilx_mesh_create); neither is ever read back.
        const int half = lane >> 5, col = lane & 31;
        const int qrow = lane >> 3, qcol = (lane & 7) * 4;
        const uint32_t ny = static_cast<uint32_t>(E.ny);
        struct __attribute__((packed, aligned(4))) Quad {
            float v[4];
        };
        const float* zg = a.mesh.lat_z + E.z_off + static_cast<uint32_t>(i_lo) * ny + j_lo + qcol;
        Quad v[kLatWin / 8];
#pragma unroll
        for (int k = 0; k < kLatWin / 8; ++k)
            v[k] = *reinterpret_cast<const Quad*>(zg + static_cast<uint32_t>(min(8 * k + qrow, wi - 1)) * ny);
        // lattice lines: x lines by lanes 0-31, y lines by lanes 32-63 (past the window: its last line again)
        const uint32_t line_at = E.line_off + (half ? static_cast<uint32_t>(E.nx + j_lo) : static_cast<uint32_t>(i_lo)) +
                                 static_cast<uint32_t>(min(col, (half ? wj : wi) - 1));
        const float line = a.mesh.lat_lines[line_at];
        // the cells' bits: kLatMaskWords words per window row (past the patch's words: the last one again, bits of cells
        // the window does not hold)
        const int mrow = min(lane / kLatMaskWords, max(wi - 2, 0)), mw = min(w0 + (lane & (kLatMaskWords - 1)), E.mask_stride - 1);
        const uint32_t bits = a.mesh.lat_mask[E.mask_off + static_cast<uint32_t>(i_lo + mrow) * E.mask_stride + mw];
#pragma unroll
        for (int k = 0; k < kLatWin / 8; ++k)
            *reinterpret_cast<float4*>(&st.z[(8 * k + qrow) * kLatWin + qcol]) = make_float4(v[k].v[0], v[k].v[1], v[k].v[2], v[k].v[3]);
        st.lines[lane] = line;
        // the row's two words, funnel-shifted so that the window's first cell is bit 0: the even lane of each pair has the low
        // word, its neighbour the high one (a quad permute: lanes 0 and 2 of a quad read lanes 1 and 3)
        const uint32_t high = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(bits), 0xF5, 0xF, 0xF, true));
        if ((lane & 1) == 0) st.mask[lane >> 1] = __builtin_amdgcn_alignbit(high, bits, static_cast<uint32_t>(j_lo & 31));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    ILX_RAY_STAMP(2)

    const f3 pos = {E.px, E.py, E.pz};
    const f4 rot = {E.rw, E.rx, E.ry, E.rz};
    float* const hits_env = a.ray_hits_w ? a.ray_hits_w + 3 * (env * a.r) : nullptr;   // null: height scan only (ILX_RAYS_HEIGHT_ONLY)
    float* const height_env = a.height ? a.height + env * a.r : nullptr;
    // World start of a pattern entry: quat_apply(rot, l) + pos (ray_caster.py:245-246) for a yaw quaternion.  rot.x and
    // rot.y are exact zeros (yaw_quat builds (cos, 0, 0, sin) and divides by the norm), so the products with them in
    // quat_apply's two cross products are exact zeros and drop out; what is left is the same sequence of roundings:
    //   t = 2 (u x l) = (-(z ly) 2, (z lx) 2, 0),  u x t = (-(z ty), z tx, 0),  o = l + w t + u x t
    auto world_start = [&](f3 l) {
        const float tx = -(rot.z * l.y) * 2.0f, ty = (rot.z * l.x) * 2.0f;
        return f3{(l.x + rot.w * tx + -(rot.z * ty)) + pos.x, (l.y + rot.w * ty + rot.z * tx) + pos.y, l.z + pos.z};
    };
    // A ray strictly inside a simple cell is served from LDS.  Its side of the cell diagonal picks the face.  Away from
    // the diagonal the ray is inside that face by margins (of the cell: kLatMargin from the borders, kLatDiagonal from
    // the diagonal) dozens of times the barycentric slack and the rounding of the test, so the inside test of
    // ray_triangle_z_inside is known to pass and that of the other face to fail: only the distance t is evaluated,
    // with the operations of ray_triangle_z_inside / ray_triangle_z_distance that feed it.  Near the diagonal both
    // faces go through the full test.
    auto serve = [&](f3 o, float dz, int ray) -> bool {
        const int ci = lattice_index(wi, o.x, E.x0, E.inv_dx, i_lo), cj = lattice_index(wj, o.y, E.y0, E.inv_dy, j_lo);
        const int jabs = j_lo + cj;
        // the cell's lines, its bit and its four heights are all asked for at once (their addresses need the cell's indices,
        // not each other's values): one LDS latency for the ray instead of four
        int mask_row = ci;
        asm("" : "+v"(mask_row));   // no instruction: keeps the word's address a shift and an add of its own (derived from the heights' address it was a 64-bit multiply-add)
        const float xlo = st.lines[ci], xhi = st.lines[ci + 1], ylo = st.lines[kLatWin + cj], yhi = st.lines[kLatWin + cj + 1];
        const uint32_t word = st.mask[mask_row];
        const float* zc = st.z + ci * kLatWin + cj;
        const float z00 = zc[0], z01 = zc[1], z10 = zc[kLatWin], z11 = zc[kLatWin + 1];
        // no instruction: the nine values are in registers here.  Without it the compiler moves each read behind the test
        // that decides whether its value is used -- lines, lines, bit, heights: four LDS latencies one after the other
        asm volatile("" ::"v"(xlo), "v"(xhi), "v"(ylo), "v"(yhi), "v"(word), "v"(z00), "v"(z01), "v"(z10), "v"(z11));
        const bool in_x = lattice_inside(o.x, xlo, xhi), in_y = lattice_inside(o.y, ylo, yhi);
        if (!(in_x && in_y && ((word >> cj) & 1u) && dz != 0.0f)) return false;
        const float wx = xhi - xlo, wy = yhi - ylo;
        const float sx = o.x - xlo, sy = o.y - ylo, sz = o.z - z00;  // ray start relative to v0 = v00
        const float side = sy * wx - sx * wy;  // > 0: the half that holds v01, face (ind0, ind3, ind1); else (ind0, ind2, ind3)
        const bool upper = side > 0.0f;
        float t;
        bool ok;
        if (fabsf(side) > wx * wy * kLatDiagonal) {
            // v1 - v0 and v2 - v0 of the face: (v11, v01) or (v10, v11)
            const float e1x = wx, e1y = upper ? wy : ylo - ylo, e1z = (upper ? z11 : z10) - z00;
            const float e2x = upper ? xlo - xlo : wx, e2y = wy, e2z = (upper ? z01 : z11) - z00;
            const float px = -(dz * e2y), py = dz * e2x;
            const float inv = div_(1.0f, e1x * px + e1y * py);
            const float qz = fmaf_(sx, e1y, -(sy * e1x));
            const float qx = fmaf_(sy, e1z, -(sz * e1y)), qy = fmaf_(sz, e1x, -(sx * e1z));
            t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
            ok = t >= 0.0f && t < a.max_dist;
        } else {
            const f3 v00 = {xlo, ylo, z00}, v01 = {xlo, yhi, z01}, v10 = {xhi, ylo, z10}, v11 = {xhi, yhi, z11};
            const int32_t face = E.face_base + 2 * ((i_lo + ci) * (E.ny - 1) + jabs);
            Hit h{inf, -1, {0.0f, 0.0f, 0.0f}};
            test_column_vertices<false>(v00, v11, v01, face, o, dz, a.max_dist, h);
            test_column_vertices<false>(v00, v10, v11, face + 1, o, dz, a.max_dist, h);
            t = h.t, ok = h.face >= 0;
        }
        const float hz = o.z + t * dz;  // kernels.py:69: start + t * dir; x and y are exact for a z-parallel ray
        // streaming stores for the hit records: nothing on the step reads them again (the managers read the height
        // scan, which is stored the usual way and waits for them in the caches), and written the usual way they would
        // push the height grid out of the L2 that the other envs' windows are read from
        if (hits_env) {
            const f3 hit = ok ? f3{o.x, o.y, hz} : f3{inf, inf, inf};
            float* const hp = hits_env + 3 * ray;
            __builtin_nontemporal_store(hit.x, hp);
            __builtin_nontemporal_store(hit.y, hp + 1);
            __builtin_nontemporal_store(hit.z, hp + 2);
        }
        if (height_env) height_env[ray] = pos.z - (ok ? hz : inf) - a.height_offset;
        if (a.mesh.lat_served) atomicAdd(a.mesh.lat_served, 1ull);
        return true;
    };
    uint32_t pending = 0;  // one bit per round of the ray loop: rays left to the general structures
#pragma unroll
    for (int r = 0; r < kLatKeep; ++r) {
        const int ray = lane + 64 * r;
        const bool zp = kp[r].w == kp[r].w;
        if (ray < a.r && !(sel >= 0 && zp && serve(world_start(f3{kp[r].x, kp[r].y, kp[r].z}), kp[r].w, ray))) pending |= 1u << r;
        if (r == 0) { ILX_RAY_STAMP(3) }
    }
    ILX_RAY_STAMP(4)
    for (int ray = lane + 64 * kLatKeep, round = kLatKeep; ray < a.r; ray += 64, ++round) {
        const float4 e = pattern[ray];
        if (!(sel >= 0 && e.w == e.w && serve(world_start(f3{e.x, e.y, e.z}), e.w, ray))) pending |= 1u << round;
    }
    static_assert(kLatMaxRays <= 32 * 64, "one pending bit per round of the ray loop");
    for (int ray = lane, round = 0; ray < a.r && (pending >> round) != 0u; ray += 64, ++round) {
        if (!((pending >> round) & 1u)) continue;
        const f3 l = load3(a.ray_starts + 3 * ray), d = load3(a.ray_dirs + 3 * ray);
        f3 o = quat_apply(rot, l);  // as sensor_ray
        o = {o.x + pos.x, o.y + pos.y, o.z + pos.z};
        const int64_t i = env * a.r + ray;
        if (!z_parallel(d) || !a.mesh.cells) {  // left to the tree kernel that follows
            *a.off_grid = a.ticket;
            if (a.ray_hits_w) a.ray_hits_w[3 * i] = __uint_as_float(kLeftOverBits);
            else a.height[i] = __uint_as_float(kLeftOverBits);   // height scan only: no launch follows (the caller's promise was broken)
            continue;
        }
        const Hit h = trace_column<false>(a.mesh, o, d, a.max_dist);
        write_scan(a, h, o, d, pos.z, i);
    }
    ILX_RAY_STAMP(5)
}

// RayCasterCamera._update_buffers_impl (ray_caster_camera.py:257-318): one thread per (env, pixel) of the refreshed
// cameras.  Camera pose = view pose composed with the offset (:415-430); rays carry the full orientation (:269-272);
// the cast has no range limit (1e6, :282) because the clipping happens on the two distance images afterwards.
struct CameraArgs {
    MeshDev mesh;
    const float* view_pos;   // strided rows
    const float* view_quat;
    int64_t pos_stride, quat_stride;
    const float* offset_pos;   // (n,3)
    const float* offset_quat;  // (n,4)
    const float* ray_starts;   // (n,r,3): per camera, the intrinsics may differ
    const float* ray_dirs;     // (n,r,3)
    const uint8_t* env_mask;   // (n) or null
    float* pos_w;              // (n,3)
    float* quat_w_world;       // (n,4)
    int64_t* frame;            // (n) or null
    float* ray_hits_w;         // (n,r,3)
    float* image_plane;        // (n,r) or null
    float* to_camera;          // (n,r) or null
    float* normals;            // (n,r,3) or null
    float max_distance;
    int clipping;              // ILX_DEPTH_CLIP_*
    int64_t n;
    int r;
};

__global__ void __launch_bounds__(kRayBlock) raycaster_camera_kernel(CameraArgs a) {
    __shared__ Node cache[kCachedNodes];
    __shared__ int32_t stack[kStackSlots * kRayBlock];
    const int ncache = min(a.mesh.num_nodes, kCachedNodes);
    stage_nodes(a.mesh, cache, ncache);
    const int64_t total = a.n * a.r;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kRayBlock + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kRayBlock) {
        const int64_t env = i / a.r;
        if (a.env_mask && !a.env_mask[env]) continue;
        const f4 qv = load4u(a.view_quat + a.quat_stride * env);
        const f3 pv = load3(a.view_pos + a.pos_stride * env);
        const f3 off = quat_apply(qv, load3(a.offset_pos + 3 * env));
        const f3 pos = {pv.x + off.x, pv.y + off.y, pv.z + off.z};
        const f4 q = quat_mul(qv, load4u(a.offset_quat + 4 * env));
        if (i == env * a.r) {  // first pixel of the camera
            store3(a.pos_w + 3 * env, pos);
            float* qo = a.quat_w_world + 4 * env;
            qo[0] = q.w, qo[1] = q.x, qo[2] = q.y, qo[3] = q.z;
            if (a.frame) a.frame[env] += 1;
        }
        f3 o = quat_apply(q, load3(a.ray_starts + 3 * i));
        o = {o.x + pos.x, o.y + pos.y, o.z + pos.z};
        const f3 d = quat_apply(q, load3(a.ray_dirs + 3 * i));
        const Hit h = a.normals ? trace<true>(a.mesh, cache, ncache, stack + threadIdx.x, kRayBlock, o, d, 1.0e6f)
                                : trace<false>(a.mesh, cache, ncache, stack + threadIdx.x, kRayBlock, o, d, 1.0e6f);
        write_hit(h, o, d, a.ray_hits_w, nullptr, a.normals, nullptr, i);
        const float inf = __builtin_inff();
        const float depth = h.face >= 0 ? h.t : inf;
        if (a.image_plane) {  // :294-308: x of the hit offset expressed in the camera frame, then the range rule
            const f3 rel = quat_apply(quat_inv(q), f3{depth * d.x, depth * d.y, depth * d.z});
            float v = rel.x;
            if (a.clipping == ILX_DEPTH_CLIP_MAX) {
                v = (v != v) ? a.max_distance : fminf(v, a.max_distance);
            } else if (a.clipping == ILX_DEPTH_CLIP_ZERO) {
                v = (v != v || v > a.max_distance) ? 0.0f : v;
            }
            a.image_plane[i] = v;
        }
        if (a.to_camera) {  // :312-316
            float v = depth;
            if (a.clipping == ILX_DEPTH_CLIP_MAX) v = fminf(v, a.max_distance);
            else if (a.clipping == ILX_DEPTH_CLIP_ZERO) v = v > a.max_distance ? 0.0f : v;
            a.to_camera[i] = v;
        }
    }
}

// sensor_base.py:284-293 after the refresh: the refreshed envs record the time and are no longer outdated.
__global__ void __launch_bounds__(256)
sensor_refreshed_kernel(const float* __restrict__ ts, float* __restrict__ ts_last, uint8_t* __restrict__ outdated, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n || !outdated[i]) return;
    ts_last[i] = ts[i];
    outdated[i] = 0;
}

// ------------------------------------------------------------------------------------------------
// host-side build
// ------------------------------------------------------------------------------------------------
struct TmpNode {
    Box box[4];
    int32_t ref[4];   // >= 0: index into tmp nodes; leaf encoding otherwise
};

// Top-down object-median build: a node's range is cut at the median along the longest axis of its centroid
// bounds, and each half is cut again along its own longest axis, giving four children with compact, barely
// overlapping boxes (a vertical ray over a terrain then enters about one child per level).
struct Builder {
    std::vector<Box>& tri_box;      // permuted in place together with `order`
    std::vector<float>& cen;        // centroid (3 floats) per ORIGINAL triangle id
    std::vector<int>& order;        // order[i] = original triangle id at sorted position i
    std::vector<TmpNode> nodes;
    int max_depth = 0;

    int split(int lo, int hi) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = cen[3 * static_cast<size_t>(order[i]) + a];
                mn[a] = std::min(mn[a], c), mx[a] = std::max(mx[a], c);
            }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
        const int mid = lo + (hi - lo) / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int x, int y) {
            const float cx = cen[3 * static_cast<size_t>(x) + axis], cy = cen[3 * static_cast<size_t>(y) + axis];
            return cx != cy ? cx < cy : x < y;
        });
        return mid;
    }

    // builds the node for [lo, hi) (hi - lo > kLeafSize) and returns its index
    int build(int lo, int hi, int depth, Box& out) {
        max_depth = std::max(max_depth, depth);
        const int idx = static_cast<int>(nodes.size());
        nodes.emplace_back();
        int cut[5];
        cut[0] = lo, cut[4] = hi;
        cut[2] = split(lo, hi);
        cut[1] = (cut[2] - lo > 1) ? split(lo, cut[2]) : cut[2];
        cut[3] = (hi - cut[2] > 1) ? split(cut[2], hi) : hi;
        out.reset();
        for (int c = 0; c < 4; ++c) {
            const int a = cut[c], b = cut[c + 1];
            Box bx;
            bx.reset();
            int32_t ref = kEmptyRef;
            if (b > a) {
                if (b - a <= kLeafSize) {
                    for (int t = a; t < b; ++t) bx.grow(tri_box[order[t]]);
                    ref = -(1 + ((a << 2) | (b - a - 1)));
                } else {
                    ref = build(a, b, depth + 1, bx);
                }
            }
            nodes[idx].box[c] = bx;
            nodes[idx].ref[c] = ref;
            if (b > a) out.grow(bx);
        }
        return idx;
    }
};

// How the cells of the column grid encode their candidates (diagnostics).
struct GridStats {
    int64_t cells, empty, run[3], pairs, triples, list_cells, lists, list_words, candidates;
};

struct MeshHost {
    MeshDev dev{};
    Node* d_nodes = nullptr;
    float4* d_tris = nullptr;
    uint32_t* d_cells = nullptr;
    uint32_t* d_cell_lists = nullptr;  // read by the kernels as 16-byte words
    uint32_t* d_tile_base = nullptr;
    uint4* d_col_tris = nullptr;
    float4* d_col_verts = nullptr;
    GridStats grid{};
    LatticePatch* d_patches = nullptr;
    short4* d_patch_bins = nullptr;
    float* d_lat_z = nullptr;
    float* d_lat_lines = nullptr;
    uint32_t* d_lat_mask = nullptr;
    int64_t lat_cells = 0, lat_simple = 0, lat_faces = 0;
    unsigned long long* d_lat_served = nullptr;
    // Workspace (scan pattern + env records) between the launches of the height-field refresh: one buffer per stream (launches on one stream
    // are ordered, so they can share it), grown on demand; outgrown buffers are kept until the mesh is destroyed because
    // work that uses them may still be queued.
    std::mutex ws_lock;
    struct Workspace {
        void* first = nullptr;   // device memory
        size_t second = 0;       // its size; the last 16 bytes hold the bounding box of the pattern named below
        const void* starts = nullptr;  // the scan pattern that the last refresh on this stream packed into it
        const void* dirs = nullptr;
        int rays = 0;
    };
    std::unordered_map<void*, Workspace> ws_by_stream;
    std::vector<void*> ws_retired;
    // buffers set aside by ilx_raycaster_reserve: a stream that needs a (larger) workspace takes one of these before it
    // turns to hipMalloc
    std::vector<std::pair<void*, size_t>> ws_spare;
    int32_t* d_off_grid = nullptr;  // kFlagSlots flags, one raised by a column kernel for the tree kernel launched after it
    std::atomic<uint32_t> calls{0}; // hinted casts so far: call c uses slot c % kFlagSlots and the ticket c + 1
    int depth = 0;
    int64_t num_verts = 0;
};

template <class T>
static int upload(T** dst, const std::vector<T>& src, const char* what) {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    int rc = check_hip(hipMalloc(dst, bytes), "ilx_mesh_create: hipMalloc");
    if (rc == ILX_OK && !src.empty())
        rc = check_hip(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice), what);
    return rc;
}

// Per-triangle boxes, inflated (the intersection formula may accept points a few ulps outside the exact triangle),
// and box centres.  False when a vertex coordinate is not finite.
static bool padded_boxes(const float* verts_host, const int32_t* tris_host, int nt, std::vector<Box>& box, std::vector<float>& cen) {
    box.resize(nt);
    cen.resize(3 * static_cast<size_t>(nt));
    for (int t = 0; t < nt; ++t) {
        Box b;
        b.reset();
        for (int k = 0; k < 3; ++k) {
            const float* v = verts_host + 3 * static_cast<int64_t>(tris_host[3 * t + k]);
            for (int a = 0; a < 3; ++a) {
                if (!std::isfinite(v[a])) return false;
                b.mn[a] = std::min(b.mn[a], v[a]), b.mx[a] = std::max(b.mx[a], v[a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            cen[3 * static_cast<size_t>(t) + a] = 0.5f * (b.mn[a] + b.mx[a]);
            const float ext = b.mx[a] - b.mn[a];
            const float pad = 1.0e-4f * std::max(ext, 1.0e-3f) + 4.0e-6f * std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a]));
            b.mn[a] -= pad, b.mx[a] += pad;
        }
        box[t] = b;
    }
    return true;
}

}  // namespace ilx

using namespace ilx;

struct ilx_mesh_s {
    MeshHost h;
};

static int static_scratch(void** out);   // the built-in buffer of ilx_raycaster_refresh_static on the current device

extern "C" int ilx_mesh_create(const float* verts_host, int64_t num_verts, const int32_t* tris_host, int64_t num_tris,
                               ilx_mesh_t* out) {
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_mesh_create: out is NULL");
    *out = nullptr;
    ILX_REQUIRE(verts_host && tris_host, ILX_EINVAL, "ilx_mesh_create: NULL vertex or index array");
    ILX_REQUIRE(num_verts > 0 && num_tris > 0, ILX_EINVAL, "ilx_mesh_create: empty mesh (%lld vertices, %lld triangles)",
                (long long)num_verts, (long long)num_tris);
    ILX_REQUIRE(num_tris < (1 << 28), ILX_ESHAPE, "ilx_mesh_create: %lld triangles (max 2^28)", (long long)num_tris);
    const int nt = static_cast<int>(num_tris);
    for (int64_t i = 0; i < 3 * num_tris; ++i)
        ILX_REQUIRE(tris_host[i] >= 0 && tris_host[i] < num_verts, ILX_EINVAL,
                    "ilx_mesh_create: triangle index %d outside [0, %lld)", tris_host[i], (long long)num_verts);

    // per-triangle boxes (inflated) and centroids
    std::vector<Box> box;
    std::vector<float> cen;
    if (!padded_boxes(verts_host, tris_host, nt, box, cen)) return fail(ILX_EINVAL, "ilx_mesh_create: non-finite vertex coordinate");
    std::vector<int> order(nt);
    std::iota(order.begin(), order.end(), 0);
    Builder bld{box, cen, order};
    Box root_box;
    if (nt <= kLeafSize) {
        bld.nodes.emplace_back();
        TmpNode& r = bld.nodes[0];
        for (int c = 0; c < 4; ++c) r.box[c].reset(), r.ref[c] = kEmptyRef;
        for (int t = 0; t < nt; ++t) r.box[0].grow(box[t]);
        r.ref[0] = -(1 + ((0 << 2) | (nt - 1)));
    } else {
        bld.nodes.reserve(static_cast<size_t>(nt) / 2 + 16);
        bld.build(0, nt, 0, root_box);
    }
    ILX_REQUIRE(bld.max_depth <= kMaxDepth, ILX_ESHAPE, "ilx_mesh_create: tree depth %d exceeds %d", bld.max_depth,
                kMaxDepth);

    // breadth-first renumbering
    const int nn = static_cast<int>(bld.nodes.size());
    std::vector<int> bfs;
    bfs.reserve(nn);
    std::vector<int> new_index(nn, -1);
    bfs.push_back(0);
    for (size_t head = 0; head < bfs.size(); ++head) {
        new_index[bfs[head]] = static_cast<int>(head);
        for (int c = 0; c < 4; ++c)
            if (bld.nodes[bfs[head]].ref[c] >= 0) bfs.push_back(bld.nodes[bfs[head]].ref[c]);
    }
    std::vector<Node> nodes(nn);
    for (int i = 0; i < nn; ++i) {
        const TmpNode& s = bld.nodes[bfs[i]];
        Node& d = nodes[i];
        std::memset(&d, 0, sizeof(Node));
        for (int c = 0; c < 4; ++c) {
            d.minx[c] = s.box[c].mn[0], d.miny[c] = s.box[c].mn[1], d.minz[c] = s.box[c].mn[2];
            d.maxx[c] = s.box[c].mx[0], d.maxy[c] = s.box[c].mx[1], d.maxz[c] = s.box[c].mx[2];
            d.ref[c] = s.ref[c] >= 0 ? new_index[s.ref[c]] : s.ref[c];
        }
    }
    std::vector<float4> tris(3 * static_cast<size_t>(nt));
    for (int i = 0; i < nt; ++i) {
        const int t = order[i];
        const float* v0 = verts_host + 3 * static_cast<int64_t>(tris_host[3 * t]);
        const float* v1 = verts_host + 3 * static_cast<int64_t>(tris_host[3 * t + 1]);
        const float* v2 = verts_host + 3 * static_cast<int64_t>(tris_host[3 * t + 2]);
        const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
        const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
        float idbits;
        std::memcpy(&idbits, &t, sizeof(float));
        tris[3 * i] = make_float4(v0[0], v0[1], v0[2], e1[0]);
        tris[3 * i + 1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
        tris[3 * i + 2] = make_float4(e2[2], idbits, 0.0f, 0.0f);
    }

    // column grid: about eight cells per triangle over the xy footprint, bounded in cells and in references
    // (a mesh whose triangles are huge next to the cell size would blow the lists up; then the cells are coarsened,
    // and a mesh that still does not fit gets no grid -- its z-parallel rays walk the tree).
    // A cell lists a triangle when the triangle's xy projection, grown by the same pad as its box, overlaps the cell
    // grown by the rounding slack of cell_coord: box range first, then the three edge half-planes (separating axes).
    float m0[2] = {INFINITY, INFINITY}, g0[2], g1[2] = {-INFINITY, -INFINITY};
    for (int t = 0; t < nt; ++t)
        for (int a = 0; a < 2; ++a) m0[a] = std::min(m0[a], box[t].mn[a]), g1[a] = std::max(g1[a], box[t].mx[a]);
    std::vector<uint32_t> cell_start, cell_refs;
    int gnx = 0, gny = 0;
    float inv_cell = 0.0f;
    {
        const double wx = std::max(double(g1[0]) - m0[0], 1e-6), wy = std::max(double(g1[1]) - m0[1], 1e-6);
        const double span = std::max(std::max(std::fabs(double(m0[0])), std::fabs(double(g1[0]))),
                                     std::max(std::fabs(double(m0[1])), std::fabs(double(g1[1])))) + std::max(wx, wy);
        double cell = std::sqrt(wx * wy / (kGridCellsPerTriangle * nt));
        cell = std::max(cell, std::max(wx, wy) / 8192.0);
        const int64_t max_refs = std::min<int64_t>(std::max<int64_t>(int64_t(64) * nt, 1 << 16), kCellIndexMask - 8);
        static_assert((1u << 28) <= kCellIndexMask, "a cell word must be able to name every triangle");
        for (int attempt = 0; attempt < 12; ++attempt, cell *= 2.0) {
            inv_cell = static_cast<float>(1.0 / cell);
            for (int a = 0; a < 2; ++a) g0[a] = static_cast<float>(double(m0[a]) - kGridPhase / double(inv_cell));
            gnx = std::max(1, cell_coord(g1[0], g0[0], inv_cell, 1 << 20) + 1);
            gny = std::max(1, cell_coord(g1[1], g0[1], inv_cell, 1 << 20) + 1);
            if (static_cast<int64_t>(gnx) * gny > (int64_t(1) << 26)) continue;
            const double cw = 1.0 / double(inv_cell);
            const double half = 0.5 * cw + 4e-3 * cw + 2e-5 * span;  // half cell + slack of the float cell mapping
            // visits the cells of sorted triangle i; fn(cell index)
            auto for_cells = [&](int i, auto&& fn) {
                const int t = order[i];
                const Box& b = box[t];
                const int x0 = cell_coord(b.mn[0], g0[0], inv_cell, gnx), x1 = cell_coord(b.mx[0], g0[0], inv_cell, gnx);
                const int y0 = cell_coord(b.mn[1], g0[1], inv_cell, gny), y1 = cell_coord(b.mx[1], g0[1], inv_cell, gny);
                if (x1 - x0 < 2 && y1 - y0 < 2) {  // nothing to gain from the edge tests
                    for (int y = y0; y <= y1; ++y)
                        for (int x = x0; x <= x1; ++x) fn(static_cast<size_t>(y) * gnx + x);
                    return;
                }
                double P[3][2];
                for (int k = 0; k < 3; ++k)
                    for (int a = 0; a < 2; ++a) P[k][a] = verts_host[3 * static_cast<int64_t>(tris_host[3 * t + k]) + a];
                // pad of the box (the larger of x and y), recovered from the inflated box
                double pad = 0.0;
                for (int a = 0; a < 2; ++a) {
                    const double lo = std::min(P[0][a], std::min(P[1][a], P[2][a]));
                    pad = std::max(pad, lo - double(b.mn[a]));
                }
                double nx[3], ny[3], off[3], reach[3];
                bool use[3];
                for (int k = 0; k < 3; ++k) {
                    const double* A = P[k];
                    const double* Bv = P[(k + 1) % 3];
                    const double* C = P[(k + 2) % 3];
                    double ex = -(Bv[1] - A[1]), ey = Bv[0] - A[0];
                    const double side = ex * (C[0] - A[0]) + ey * (C[1] - A[1]);
                    const double len = std::sqrt(ex * ex + ey * ey);
                    use[k] = len > 0.0 && std::fabs(side) > 1e-12 * len * len;  // degenerate projection: no culling
                    if (side < 0.0) ex = -ex, ey = -ey;
                    nx[k] = ex, ny[k] = ey;
                    off[k] = ex * A[0] + ey * A[1];
                    reach[k] = half * (std::fabs(ex) + std::fabs(ey)) + pad * len;
                }
                for (int y = y0; y <= y1; ++y) {
                    const double cy = double(g0[1]) + (y + 0.5) * cw;
                    for (int x = x0; x <= x1; ++x) {
                        const double cx = double(g0[0]) + (x + 0.5) * cw;
                        bool hit = true;
                        for (int k = 0; k < 3 && hit; ++k)
                            if (use[k]) hit = (nx[k] * cx + ny[k] * cy - off[k]) + reach[k] >= 0.0;
                        if (hit) fn(static_cast<size_t>(y) * gnx + x);
                    }
                }
            };
            cell_start.assign(static_cast<size_t>(gnx) * gny + 1, 0u);
            int64_t total = 0;
            for (int i = 0; i < nt && total <= max_refs; ++i) for_cells(i, [&](size_t c) { ++cell_start[c + 1], ++total; });
            if (total > max_refs) {
                cell_start.clear();
                continue;
            }
            for (size_t c = 1; c < cell_start.size(); ++c) cell_start[c] += cell_start[c - 1];
            cell_refs.assign(static_cast<size_t>(total) + 4, 0u);  // 4 entries of padding for the 4-wide reads
            std::vector<uint32_t> fill(cell_start.begin(), cell_start.end() - 1);
            for (int i = 0; i < nt; ++i) for_cells(i, [&](size_t c) { cell_refs[fill[c]++] = static_cast<uint32_t>(i); });
            break;
        }
    }
    // Column triangles (three vertex positions + face id), ordered by the tile of the cell under their centroid and
    // by face id inside a tile: triangles that follow each other in the caller's mesh (the two halves of a quad,
    // neighbours along a row) stay next to each other, so most cells can name their candidates as one short run.
    // Column vertices are ordered by the tile of their own cell.
    std::vector<uint4> col_tris(cell_start.empty() ? 0 : static_cast<size_t>(nt));
    std::vector<float4> col_verts(cell_start.empty() ? 0 : static_cast<size_t>(num_verts));
    std::vector<uint32_t> cells, cell_lists, tile_base;
    int ntx = 0;
    GridStats gstats{};
    if (!cell_start.empty()) {
        const int T = kGridTile;
        ntx = (gnx + T - 1) / T;
        const int nty = (gny + T - 1) / T;
        auto tile_of = [&](float x, float y) {
            const int cx = cell_coord(x, g0[0], inv_cell, gnx), cy = cell_coord(y, g0[1], inv_cell, gny);
            return static_cast<uint64_t>(cy / T) * ntx + cx / T;
        };
        // vertices
        std::vector<uint64_t> vkey(num_verts);
        for (int64_t v = 0; v < num_verts; ++v) vkey[v] = tile_of(verts_host[3 * v], verts_host[3 * v + 1]);
        std::vector<uint32_t> vorder(num_verts), vpos(num_verts);
        std::iota(vorder.begin(), vorder.end(), 0u);
        std::stable_sort(vorder.begin(), vorder.end(), [&](uint32_t a, uint32_t b) { return vkey[a] < vkey[b]; });
        for (int64_t r = 0; r < num_verts; ++r) {
            vpos[vorder[r]] = static_cast<uint32_t>(r);
            const float* v = verts_host + 3 * static_cast<int64_t>(vorder[r]);
            col_verts[r] = make_float4(v[0], v[1], v[2], 0.0f);
        }
        // triangles
        std::vector<uint64_t> key(nt);
        for (int i = 0; i < nt; ++i) {
            const int t = order[i];
            double c[2] = {0.0, 0.0};
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 2; ++a) c[a] += verts_host[3 * static_cast<int64_t>(tris_host[3 * t + k]) + a];
            key[i] = tile_of(static_cast<float>(c[0] / 3.0), static_cast<float>(c[1] / 3.0));
        }
        // Order inside a tile: triangles that are candidates of the same cell should be neighbours, so that the cell can
        // name them as a run.  Every pair of same-tile triangles that shares a small cell (2..4 candidates) is an edge,
        // weighted by the number of such cells; edges are taken greedily by weight into vertex-disjoint paths (each
        // triangle gets at most two neighbours, no cycles), and a tile's paths are laid out one after the other.  On a
        // height field this keeps the two halves of a quad and one of their edge neighbours adjacent.
        std::vector<uint32_t> rank_in_tile(nt);
        {
            std::vector<uint64_t> pairs;
            pairs.reserve(static_cast<size_t>(nt) * 4);
            for (size_t c = 0; c + 1 < cell_start.size(); ++c) {
                const uint32_t b = cell_start[c], e = cell_start[c + 1];
                if (e - b < 2 || e - b > 4) continue;
                for (uint32_t x = b; x < e; ++x)
                    for (uint32_t y = x + 1; y < e; ++y) {
                        uint32_t p = cell_refs[x], q = cell_refs[y];
                        if (key[p] != key[q]) continue;
                        if (p > q) std::swap(p, q);
                        pairs.push_back(static_cast<uint64_t>(p) << 32 | q);
                    }
            }
            std::sort(pairs.begin(), pairs.end());
            struct Edge {
                uint32_t w, a, b;
            };
            std::vector<Edge> edges;
            for (size_t i = 0; i < pairs.size();) {
                size_t j = i;
                while (j < pairs.size() && pairs[j] == pairs[i]) ++j;
                edges.push_back({static_cast<uint32_t>(j - i), static_cast<uint32_t>(pairs[i] >> 32), static_cast<uint32_t>(pairs[i])});
                i = j;
            }
            pairs.clear();
            pairs.shrink_to_fit();
            std::stable_sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.w > y.w; });
            std::vector<uint32_t> comp(nt), nb0(nt, UINT32_MAX), nb1(nt, UINT32_MAX);
            std::iota(comp.begin(), comp.end(), 0u);
            auto find = [&](uint32_t v) {
                while (comp[v] != v) v = comp[v] = comp[comp[v]];
                return v;
            };
            for (const Edge& ed : edges) {
                if (nb1[ed.a] != UINT32_MAX || nb1[ed.b] != UINT32_MAX) continue;  // already two neighbours
                const uint32_t ra = find(ed.a), rb = find(ed.b);
                if (ra == rb) continue;  // would close a cycle
                comp[ra] = rb;
                (nb0[ed.a] == UINT32_MAX ? nb0[ed.a] : nb1[ed.a]) = ed.b;
                (nb0[ed.b] == UINT32_MAX ? nb0[ed.b] : nb1[ed.b]) = ed.a;
            }
            // walk every path from its end with the lower face id; paths of a tile in the order of that id
            std::vector<uint32_t> by_face(nt);
            std::iota(by_face.begin(), by_face.end(), 0u);
            std::sort(by_face.begin(), by_face.end(), [&](uint32_t a, uint32_t b) {
                return key[a] != key[b] ? key[a] < key[b] : order[a] < order[b];
            });
            std::vector<uint8_t> placed(nt, 0);
            uint32_t next = 0;
            for (uint32_t start : by_face) {
                if (placed[start] || nb1[start] != UINT32_MAX) continue;  // interior vertices are reached from an end
                uint32_t prev = UINT32_MAX, v = start;
                while (v != UINT32_MAX && !placed[v]) {
                    placed[v] = 1;
                    rank_in_tile[v] = next++;
                    const uint32_t to = nb0[v] != prev ? nb0[v] : nb1[v];
                    prev = v;
                    v = to;
                }
            }
            for (uint32_t v : by_face)  // nothing is left (paths have two ends), but never leave a triangle unplaced
                if (!placed[v]) rank_in_tile[v] = next++;
        }
        std::vector<uint32_t> by_key(nt), pos(nt);
        std::iota(by_key.begin(), by_key.end(), 0u);
        std::sort(by_key.begin(), by_key.end(), [&](uint32_t a, uint32_t b) {
            return key[a] != key[b] ? key[a] < key[b] : rank_in_tile[a] < rank_in_tile[b];
        });
        for (int r = 0; r < nt; ++r) {
            pos[by_key[r]] = static_cast<uint32_t>(r);
            const int t = order[by_key[r]];
            col_tris[r] = make_uint4(vpos[tris_host[3 * t]], vpos[tris_host[3 * t + 1]], vpos[tris_host[3 * t + 2]],
                                     static_cast<uint32_t>(t));
        }
        // first record of every tile (records are sorted by tile)
        tile_base.assign(static_cast<size_t>(ntx) * nty + 1, 0u);
        for (int i = 0; i < nt; ++i) ++tile_base[key[i] + 1];
        for (size_t t = 1; t < tile_base.size(); ++t) tile_base[t] += tile_base[t - 1];
        // cell words; candidate lists are shared between cells with the same candidates
        cells.assign(static_cast<size_t>(ntx) * nty * T * T, kCellEmpty);
        cell_lists.reserve(1 << 16);
        std::unordered_map<uint64_t, std::vector<uint32_t>> known;  // hash of a candidate list -> offsets of lists with it
        std::vector<uint32_t> cand;
        // cells are visited tile by tile, so the lists of a tile's cells are neighbours in memory as well
        for (int64_t slot = 0; slot < static_cast<int64_t>(cells.size()); ++slot) {
            {
                const int64_t tile = slot / (T * T);  // the same numbering as the triangles' tile key
                const int in_tile = static_cast<int>(slot % (T * T));
                const int cx = static_cast<int>(tile % ntx) * T + in_tile % T, cy = static_cast<int>(tile / ntx) * T + in_tile / T;
                if (cx >= gnx || cy >= gny) continue;
                const size_t c = static_cast<size_t>(cy) * gnx + cx;
                const uint32_t b = cell_start[c], e = cell_start[c + 1];
                ++gstats.cells;
                if (b == e) {
                    ++gstats.empty;
                    continue;
                }
                cand.clear();
                for (uint32_t k = b; k < e; ++k) cand.push_back(pos[cell_refs[k]]);
                std::sort(cand.begin(), cand.end());
                uint32_t& word = cells[cell_slot(cx, cy, ntx)];
                const uint32_t n = static_cast<uint32_t>(cand.size());
                gstats.candidates += n;
                const uint32_t span = cand[n - 1] - cand[0] + 1;
                const uint32_t tb = tile_base[tile], wide = tb - std::min(tb, kWideBack), narrow = tb - std::min(tb, kNarrowBack);
                auto in_window = [&](uint32_t lo, uint32_t bits) {
                    return cand[0] >= lo && cand[n - 1] - lo < (1u << bits);
                };
                if ((span <= kLooseRun || (span == n && n <= kMaxRun)) && in_window(wide, kWideBits)) {
                    word = (kCellRun << kCellKindShift) | ((span - 1) << kWideBits) | (cand[0] - wide);
                    ++gstats.run[span <= 3 ? span - 1 : 2];
                    continue;
                }
                if (span <= kLooseRun) {  // out of the window
                    word = ((kCellFar1 + span - 1) << kCellKindShift) | cand[0];
                    ++gstats.run[span - 1];
                    continue;
                }
                if (n == 2 && in_window(wide, kWideBits)) {
                    word = (kCellPair << kCellKindShift) | ((cand[1] - wide) << kWideBits) | (cand[0] - wide);
                    ++gstats.pairs;
                    continue;
                }
                if (n == 3 && in_window(narrow, kNarrowBits)) {
                    word = (kCellTriple << kCellKindShift) | ((cand[2] - narrow) << (2 * kNarrowBits)) |
                           ((cand[1] - narrow) << kNarrowBits) | (cand[0] - narrow);
                    ++gstats.triples;
                    continue;
                }
                uint64_t hsh = 1469598103934665603ull;
                for (uint32_t v : cand) hsh = (hsh ^ v) * 1099511628211ull;
                std::vector<uint32_t>& same = known[hsh];
                uint32_t at = kCellEmpty;
                for (uint32_t off : same)
                    if (cell_lists[off] == n && std::equal(cand.begin(), cand.end(), cell_lists.begin() + off + 1)) at = off;
                if (at == kCellEmpty) {
                    at = static_cast<uint32_t>(cell_lists.size());  // a multiple of 4: lists are whole 16-byte words
                    cell_lists.push_back(n);
                    cell_lists.insert(cell_lists.end(), cand.begin(), cand.end());
                    while (cell_lists.size() % 4 != 0) cell_lists.push_back(cand[0]);
                    same.push_back(at);
                    ++gstats.lists;
                }
                word = (kCellList << kCellKindShift) | (at / 4);
                ++gstats.list_cells;
            }
        }
        gstats.list_words = static_cast<int64_t>(cell_lists.size());
    }

    // height-field patches (lattice.h); `box` still holds the padded triangle boxes
    LatticeBuild lat;
    detect_lattices(verts_host, num_verts, tris_host, nt, box, lat);
    if (lat.z.size() >= (size_t(1) << 31)) lat = LatticeBuild{};  // 32-bit offsets in the kernels

    auto* obj = new ilx_mesh_s();
    MeshHost& h = obj->h;
    int rc = check_hip(hipMalloc(&h.d_nodes, nodes.size() * sizeof(Node)), "ilx_mesh_create: hipMalloc(nodes)");
    if (rc == ILX_OK) rc = check_hip(hipMalloc(&h.d_off_grid, kFlagSlots * sizeof(int32_t)), "ilx_mesh_create: hipMalloc(flag)");
    if (rc == ILX_OK) rc = check_hip(hipMemset(h.d_off_grid, 0, kFlagSlots * sizeof(int32_t)), "ilx_mesh_create: hipMemset(flag)");
    if (rc == ILX_OK && !cells.empty()) {
        rc = upload(&h.d_cells, cells, "cells");
        if (rc == ILX_OK) rc = upload(&h.d_tile_base, tile_base, "tile bases");
        if (rc == ILX_OK) rc = upload(&h.d_cell_lists, cell_lists, "cell lists");
        if (rc == ILX_OK) rc = upload(&h.d_col_tris, col_tris, "column triangles");
        if (rc == ILX_OK) rc = upload(&h.d_col_verts, col_verts, "column vertices");
    }
    // patch finder of the sensor refresh (see MeshDev::patch_bins)
    std::vector<short4> patch_bins;
    float pb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!lat.patches.empty() && lat.patches.size() < 32768) {
        float ux0 = lat.patches[0].x0, uy0 = lat.patches[0].y0, ux1 = lat.patches[0].x1, uy1 = lat.patches[0].y1;
        for (const LatticePatch& P : lat.patches)
            ux0 = std::min(ux0, P.x0), uy0 = std::min(uy0, P.y0), ux1 = std::max(ux1, P.x1), uy1 = std::max(uy1, P.y1);
        pb[0] = ux0, pb[1] = uy0;
        pb[2] = ux1 > ux0 ? kPatchBins / (ux1 - ux0) : 0.0f, pb[3] = uy1 > uy0 ? kPatchBins / (uy1 - uy0) : 0.0f;
        patch_bins.assign(kPatchBins * kPatchBins, short4{-1, -1, -1, -1});
        std::vector<int> fill(kPatchBins * kPatchBins, 0);
        // the kernel's own bin arithmetic, widened by one bin for its rounding
        auto bin_of = [&](float v, int axis) {
            return std::max(0, std::min(kPatchBins - 1, static_cast<int>((v - pb[axis]) * pb[2 + axis])));
        };
        for (size_t p = 0; p < lat.patches.size(); ++p) {
            const LatticePatch& P = lat.patches[p];
            for (int by = std::max(0, bin_of(P.y0, 1) - 1); by <= std::min(kPatchBins - 1, bin_of(P.y1, 1) + 1); ++by)
                for (int bx = std::max(0, bin_of(P.x0, 0) - 1); bx <= std::min(kPatchBins - 1, bin_of(P.x1, 0) + 1); ++bx) {
                    short4& b = patch_bins[by * kPatchBins + bx];
                    int& n = fill[by * kPatchBins + bx];
                    const short id = static_cast<short>(p);
                    if (n == 0) b.x = id;
                    else if (n == 1) b.y = id;
                    else if (n == 2) b.z = id;
                    else if (n == 3) b.w = id;
                    ++n;
                }
        }
        for (int i = 0; i < kPatchBins * kPatchBins; ++i)
            if (fill[i] > 4) patch_bins[i] = short4{-2, -1, -1, -1};
    } else if (!lat.patches.empty()) {
        patch_bins.assign(kPatchBins * kPatchBins, short4{-2, -1, -1, -1});
    }
    if (rc == ILX_OK && !lat.patches.empty()) {
        rc = upload(&h.d_patches, lat.patches, "lattice patches");
        if (rc == ILX_OK) rc = upload(&h.d_patch_bins, patch_bins, "lattice patch finder");
        lat.z.resize(lat.z.size() + kLatWin + 4, 0.0f);   // the cast reads whole quads of a 32-wide window wherever it starts
        if (rc == ILX_OK) rc = upload(&h.d_lat_z, lat.z, "lattice heights");
        if (rc == ILX_OK) rc = upload(&h.d_lat_lines, lat.lines, "lattice lines");
        if (rc == ILX_OK) rc = upload(&h.d_lat_mask, lat.mask, "lattice cell bits");
    }
    if (rc == ILX_OK) rc = check_hip(hipMalloc(&h.d_tris, tris.size() * sizeof(float4)), "ilx_mesh_create: hipMalloc(tris)");
    if (rc == ILX_OK)
        rc = check_hip(hipMemcpy(h.d_nodes, nodes.data(), nodes.size() * sizeof(Node), hipMemcpyHostToDevice),
                       "ilx_mesh_create: copy nodes");
    if (rc == ILX_OK)
        rc = check_hip(hipMemcpy(h.d_tris, tris.data(), tris.size() * sizeof(float4), hipMemcpyHostToDevice),
                       "ilx_mesh_create: copy triangles");
    if (rc != ILX_OK) {
        if (h.d_nodes) (void)hipFree(h.d_nodes);
        if (h.d_tris) (void)hipFree(h.d_tris);
        if (h.d_cells) (void)hipFree(h.d_cells);
        if (h.d_cell_lists) (void)hipFree(h.d_cell_lists);
        if (h.d_tile_base) (void)hipFree(h.d_tile_base);
        if (h.d_off_grid) (void)hipFree(h.d_off_grid);
        if (h.d_col_tris) (void)hipFree(h.d_col_tris);
        if (h.d_col_verts) (void)hipFree(h.d_col_verts);
        if (h.d_patches) (void)hipFree(h.d_patches);
        if (h.d_patch_bins) (void)hipFree(h.d_patch_bins);
        if (h.d_lat_z) (void)hipFree(h.d_lat_z);
        if (h.d_lat_lines) (void)hipFree(h.d_lat_lines);
        if (h.d_lat_mask) (void)hipFree(h.d_lat_mask);
        delete obj;
        return rc;
    }
    h.dev.nodes = h.d_nodes, h.dev.tris = h.d_tris, h.dev.num_nodes = nn, h.dev.num_tris = nt;
    h.dev.tile_base = h.d_tile_base;
    h.dev.cells = h.d_cells, h.dev.cell_lists = reinterpret_cast<const uint4*>(h.d_cell_lists), h.dev.col_tris = h.d_col_tris, h.dev.col_verts = h.d_col_verts;
    h.dev.gx0 = g0[0], h.dev.gy0 = g0[1], h.dev.gx1 = g1[0], h.dev.gy1 = g1[1], h.dev.inv_cell = inv_cell;
    h.dev.nx = gnx, h.dev.ny = gny, h.dev.ntx = ntx;
    h.dev.patches = h.d_patches, h.dev.lat_z = h.d_lat_z, h.dev.lat_lines = h.d_lat_lines, h.dev.lat_mask = h.d_lat_mask;
    h.dev.num_patches = h.d_patches ? static_cast<int32_t>(lat.patches.size()) : 0;
    h.dev.patch_bins = h.d_patch_bins, h.dev.pb_x0 = pb[0], h.dev.pb_y0 = pb[1], h.dev.pb_inv_x = pb[2], h.dev.pb_inv_y = pb[3];
    h.lat_cells = lat.cells, h.lat_simple = lat.simple, h.lat_faces = lat.faces;
    h.grid = gstats;
    h.depth = bld.max_depth, h.num_verts = num_verts;
    {   // The address of ilx_raycaster_refresh_static's built-in buffer on this device is looked up now: the runtime loads
        // the code object and registers the symbol on that first request, which a stream under capture may not survive.
        void* unused = nullptr;
        const int status = static_scratch(&unused);
        if (status != ILX_OK) {
            ilx_mesh_destroy(obj);
            return status;
        }
    }
    *out = obj;
    return ILX_OK;
}

extern "C" void ilx_mesh_destroy(ilx_mesh_t mesh) {
    if (!mesh) return;
    if (mesh->h.d_nodes) (void)hipFree(mesh->h.d_nodes);
    if (mesh->h.d_tris) (void)hipFree(mesh->h.d_tris);
    if (mesh->h.d_cells) (void)hipFree(mesh->h.d_cells);
    if (mesh->h.d_cell_lists) (void)hipFree(mesh->h.d_cell_lists);
    if (mesh->h.d_tile_base) (void)hipFree(mesh->h.d_tile_base);
    if (mesh->h.d_off_grid) (void)hipFree(mesh->h.d_off_grid);
    if (mesh->h.d_col_tris) (void)hipFree(mesh->h.d_col_tris);
    if (mesh->h.d_col_verts) (void)hipFree(mesh->h.d_col_verts);
    if (mesh->h.d_patches) (void)hipFree(mesh->h.d_patches);
    if (mesh->h.d_patch_bins) (void)hipFree(mesh->h.d_patch_bins);
    if (mesh->h.d_lat_z) (void)hipFree(mesh->h.d_lat_z);
    if (mesh->h.d_lat_lines) (void)hipFree(mesh->h.d_lat_lines);
    if (mesh->h.d_lat_mask) (void)hipFree(mesh->h.d_lat_mask);
    if (mesh->h.d_lat_served) (void)hipFree(mesh->h.d_lat_served);
    for (auto& kv : mesh->h.ws_by_stream) (void)hipFree(kv.second.first);
    for (void* p : mesh->h.ws_retired) (void)hipFree(p);
    for (auto& sp : mesh->h.ws_spare) (void)hipFree(sp.first);
    delete mesh;
}

extern "C" int ilx_mesh_info(ilx_mesh_t mesh, int64_t* num_tris, int64_t* num_nodes, int32_t* depth) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_mesh_info: mesh is NULL");
    if (num_tris) *num_tris = mesh->h.dev.num_tris;
    if (num_nodes) *num_nodes = mesh->h.dev.num_nodes;
    if (depth) *depth = mesh->h.depth;
    return ILX_OK;
}

extern "C" int ilx_mesh_grid_stats(ilx_mesh_t mesh, int64_t out[12]) {
    ILX_REQUIRE(mesh && out, ILX_EINVAL, "ilx_mesh_grid_stats: NULL argument");
    const GridStats& g = mesh->h.grid;
    const int64_t v[12] = {g.cells, g.empty, g.run[0], g.run[1], g.run[2], g.list_cells, g.lists, g.list_words, g.candidates,
                           mesh->h.dev.nx, g.pairs, g.triples};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
    return ILX_OK;
}

extern "C" int ilx_mesh_lattice_probe(const float* verts_host, int64_t num_verts, const int32_t* tris_host, int64_t num_tris,
                                      int64_t out[4]) {
    ILX_REQUIRE(verts_host && tris_host && out, ILX_EINVAL, "ilx_mesh_lattice_probe: NULL argument");
    ILX_REQUIRE(num_verts > 0 && num_tris > 0 && num_tris < (1 << 28), ILX_EINVAL, "ilx_mesh_lattice_probe: bad mesh size");
    for (int64_t i = 0; i < 3 * num_tris; ++i)
        ILX_REQUIRE(tris_host[i] >= 0 && tris_host[i] < num_verts, ILX_EINVAL,
                    "ilx_mesh_lattice_probe: triangle index %d outside [0, %lld)", tris_host[i], (long long)num_verts);
    std::vector<Box> box;
    std::vector<float> cen;
    ILX_REQUIRE(padded_boxes(verts_host, tris_host, static_cast<int>(num_tris), box, cen), ILX_EINVAL,
                "ilx_mesh_lattice_probe: non-finite vertex coordinate");
    LatticeBuild lat;
    detect_lattices(verts_host, num_verts, tris_host, static_cast<int>(num_tris), box, lat);
    out[0] = static_cast<int64_t>(lat.patches.size()), out[1] = lat.faces, out[2] = lat.cells, out[3] = lat.simple;
    return ILX_OK;
}

extern "C" int ilx_mesh_lattice_stats(ilx_mesh_t mesh, int64_t out[4]) {
    ILX_REQUIRE(mesh && out, ILX_EINVAL, "ilx_mesh_lattice_stats: NULL argument");
    out[0] = mesh->h.dev.num_patches, out[1] = mesh->h.lat_faces, out[2] = mesh->h.lat_cells, out[3] = mesh->h.lat_simple;
    return ILX_OK;
}

extern "C" int ilx_mesh_lattice_counter(ilx_mesh_t mesh, int enable, int64_t* served) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_mesh_lattice_counter: mesh is NULL");
    MeshHost& h = mesh->h;
    if (served) {
        *served = 0;
        if (h.d_lat_served) {
            unsigned long long v = 0;
            const int rc = check_hip(hipMemcpy(&v, h.d_lat_served, sizeof(v), hipMemcpyDeviceToHost), "ilx_mesh_lattice_counter: read");
            if (rc != ILX_OK) return rc;
            *served = static_cast<int64_t>(v);
        }
    }
    if (enable) {  // (re)start the count
        int rc = ILX_OK;
        if (!h.d_lat_served) rc = check_hip(hipMalloc(&h.d_lat_served, sizeof(unsigned long long)), "ilx_mesh_lattice_counter: hipMalloc");
        if (rc == ILX_OK) rc = check_hip(hipMemset(h.d_lat_served, 0, sizeof(unsigned long long)), "ilx_mesh_lattice_counter: clear");
        if (rc != ILX_OK) return rc;
    }
    h.dev.lat_served = enable ? h.d_lat_served : nullptr;
    return ILX_OK;
}

static unsigned ray_grid(int64_t rays) {
    // enough blocks to fill 256 CUs a few times over; the kernels grid-stride over the rest
    const int64_t want = (rays + kRayBlock - 1) / kRayBlock;
    return static_cast<unsigned>(std::min<int64_t>(want, 256 * 12));
}

// `fixed_ws` (optional): a workspace of `fixed_bytes` that the launches use instead of the mesh's per-stream one.
static int launch_scan(ilx_mesh_t mesh, ScanArgs a, int hints, void* stream, void* fixed_ws = nullptr, size_t fixed_bytes = 0);

// Bytes between the two launches of the height-field refresh: the scan pattern (16 bytes per ray, padded to whole env
// records), then one record per env.
static size_t lattice_workspace_bytes(int64_t n, int r) {
    const size_t pat = (static_cast<size_t>(r) * sizeof(float4) + sizeof(LatEnvRec) - 1) / sizeof(LatEnvRec) * sizeof(LatEnvRec);
    return pat + static_cast<size_t>(n) * sizeof(LatEnvRec);
}

// The three launches of the height-field refresh on a workspace of lattice_workspace_bytes(a.n, a.r): preparation (one
// thread per env; it also advances the sensor clocks), cast (one wavefront per env), and the tree launch for rays that
// broke the z-parallel hint -- it only looks at the flag unless there are such rays, and it is left out altogether when
// the caller has checked the pattern (ILX_RAYS_Z_PARALLEL_CHECKED) and the mesh has a column grid for the rays that
// leave the height-grid path.
// `box` (or null) / `reuse`: see lattice_workspace.
static void launch_lattice(ilx_mesh_t mesh, ScanArgs a, int hints, void* ws, float* box, bool reuse, void* stream) {
    float4* pattern = static_cast<float4*>(ws);
    LatEnvRec* recs = reinterpret_cast<LatEnvRec*>(static_cast<char*>(ws) + lattice_workspace_bytes(0, a.r));
    if (!(hints & ILX_RAYS_PREPARED))
        ILX_LAUNCH(raycaster_prepare_kernel, dim3(blocks_for(a.n)), dim3(256), 0, as_stream(stream), a, recs, pattern, box,
                           (box && reuse) ? 1 : 0);
    const bool last_launch = (hints & ILX_RAYS_Z_PARALLEL_CHECKED) && mesh->h.d_cells;
    ScanArgs cast = a;
    if ((hints & ILX_RAYS_HEIGHT_ONLY) && last_launch && a.height) cast.ray_hits_w = nullptr;   // nothing reads the hit records as a work list
    ILX_LAUNCH(raycaster_update_lattice_kernel, dim3(static_cast<unsigned>((a.n + kLatWaves - 1) / kLatWaves)),
                       dim3(64 * kLatWaves), 0, as_stream(stream), cast, recs, pattern);
    if (last_launch) return;
    a.outdated = nullptr;
    ILX_LAUNCH(raycaster_update_kernel, dim3(std::min(ray_grid(a.n * a.r), 256u)), dim3(kRayBlock), 0, as_stream(stream), a);
}

// The workspace of `stream` with room for `bytes`, and the four floats behind it that keep the scan pattern's bounding
// box from one refresh to the next.  `reuse`: the workspace still holds the packed pattern and the box of exactly this
// pattern (same arrays, same length) from the previous refresh on the stream.
static size_t workspace_capacity(size_t bytes) { return (std::max<size_t>(bytes + bytes / 4, size_t(1) << 16) + 31) & ~size_t(15); }

// `packs`: the call's preparation launch (re)writes the packed pattern unless `reuse` comes back true; a call that
// launches no preparation (ILX_RAYS_PREPARED) leaves the record of what the workspace holds as it is.
static int lattice_workspace(ilx_mesh_t mesh, void* stream, size_t bytes, const ScanArgs& a, bool packs, void** out, float** box,
                             bool* reuse) {
    MeshHost& h = mesh->h;
    std::lock_guard<std::mutex> guard(h.ws_lock);
    auto& slot = h.ws_by_stream[stream];
    if (slot.second < bytes + 16) {
        void* fresh = nullptr;
        size_t cap = 0;
        for (size_t i = 0; i < h.ws_spare.size(); ++i)   // a buffer set aside by ilx_raycaster_reserve, if one is large enough
            if (h.ws_spare[i].second >= bytes + 16) {
                fresh = h.ws_spare[i].first, cap = h.ws_spare[i].second;
                h.ws_spare.erase(h.ws_spare.begin() + static_cast<std::ptrdiff_t>(i));
                break;
            }
        if (!fresh) {
            cap = workspace_capacity(bytes);
            const int rc = check_hip(hipMalloc(&fresh, cap), "ilx_raycaster: hipMalloc(workspace)");
            if (rc != ILX_OK) return rc;
        }
        if (slot.first) h.ws_retired.push_back(slot.first);
        slot = MeshHost::Workspace{fresh, cap, nullptr, nullptr, 0};
    }
    *out = slot.first;
    *box = reinterpret_cast<float*>(static_cast<char*>(slot.first) + slot.second - 16);
    *reuse = slot.starts == a.ray_starts && slot.dirs == a.ray_dirs && slot.rays == a.r;
    if (packs) slot.starts = a.ray_starts, slot.dirs = a.ray_dirs, slot.rays = a.r;
    return ILX_OK;
}

extern "C" int ilx_raycaster_reserve(ilx_mesh_t mesh, int64_t num_envs, int32_t num_rays, int32_t streams) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_raycaster_reserve: mesh is NULL");
    ILX_REQUIRE(num_envs >= 0 && num_rays > 0 && streams >= 0 && streams <= 64, ILX_EINVAL,
                "ilx_raycaster_reserve: bad arguments (%lld envs, %d rays, %d streams)", (long long)num_envs, num_rays, streams);
    if (num_envs == 0 || mesh->h.dev.num_patches == 0 || num_rays > kLatMaxRays) return ILX_OK;   // such refreshes keep nothing between launches
    const size_t bytes = lattice_workspace_bytes(num_envs, num_rays);
    MeshHost& h = mesh->h;
    std::lock_guard<std::mutex> guard(h.ws_lock);
    int have = 0;
    for (auto& sp : h.ws_spare) have += sp.second >= bytes + 16;
    for (; have < streams; ++have) {
        void* fresh = nullptr;
        const size_t cap = workspace_capacity(bytes);
        const int rc = check_hip(hipMalloc(&fresh, cap), "ilx_raycaster_reserve: hipMalloc");
        if (rc != ILX_OK) return rc;
        h.ws_spare.emplace_back(fresh, cap);
    }
    return ILX_OK;
}

extern "C" int ilx_mesh_forget_patterns(ilx_mesh_t mesh) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_mesh_forget_patterns: mesh is NULL");
    std::lock_guard<std::mutex> guard(mesh->h.ws_lock);
    for (auto& kv : mesh->h.ws_by_stream) kv.second.starts = kv.second.dirs = nullptr, kv.second.rays = 0;
    return ILX_OK;
}

// A flag slot and a non-zero ticket for one hinted cast.  A slot is reused every kFlagSlots calls; its old value can
// never equal the new ticket, so nothing has to be cleared between calls.
static void take_ticket(ilx_mesh_t mesh, int32_t& ticket, int32_t*& flag) {
    const uint32_t c = mesh->h.calls.fetch_add(1, std::memory_order_relaxed);
    ticket = static_cast<int32_t>((c % 0x7FFFFFFEu) + 1u);
    flag = mesh->h.d_off_grid + c % kFlagSlots;
}
static unsigned column_grid(int64_t rays) { return static_cast<unsigned>((rays + kRayBlock - 1) / kRayBlock); }

extern "C" int ilx_raycast_mesh(ilx_mesh_t mesh, const float* ray_starts, const float* ray_directions, int64_t num_rays,
                                float max_dist, float* ray_hits, float* ray_distance, float* ray_normal,
                                int32_t* ray_face_id, int hints, void* stream) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_raycast_mesh: mesh is NULL");
    ILX_REQUIRE(num_rays >= 0, ILX_EINVAL, "ilx_raycast_mesh: negative ray count");
    if (num_rays == 0) return ILX_OK;
    ILX_REQUIRE(ray_starts && ray_directions && ray_hits, ILX_EINVAL, "ilx_raycast_mesh: NULL operand");
    ILX_REQUIRE(num_rays < (int64_t(1) << 31) * kRayBlock, ILX_ESHAPE, "ilx_raycast_mesh: too many rays");
    CastArgs a{mesh->h.dev, ray_starts, ray_directions, max_dist, ray_hits, ray_distance, ray_normal, ray_face_id, num_rays,
               nullptr, 0, nullptr};
    if ((hints & ILX_RAYS_Z_PARALLEL) && mesh->h.d_cells) {
        // column kernel, then the tree kernel for whatever rays were not z-parallel after all (it leaves at once
        // when the column kernel did not raise this call's flag)
        take_ticket(mesh, a.ticket, a.off_grid);
        ILX_LAUNCH(raycast_column_kernel, dim3(column_grid(num_rays)), dim3(kRayBlock), 0, as_stream(stream), a);
    }
    ILX_LAUNCH(raycast_kernel, dim3(ray_grid(num_rays)), dim3(kRayBlock), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_raycast_mesh");
    return ILX_OK;
}

extern "C" int ilx_raycast_mesh_stats(ilx_mesh_t mesh, const float* ray_starts, const float* ray_directions,
                                      int64_t num_rays, float max_dist, float* ray_hits, int32_t* counters, void* stream) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_raycast_mesh_stats: mesh is NULL");
    if (num_rays <= 0) return ILX_OK;
    ILX_REQUIRE(ray_starts && ray_directions && ray_hits && counters, ILX_EINVAL, "ilx_raycast_mesh_stats: NULL operand");
    CastArgs a{mesh->h.dev, ray_starts, ray_directions, max_dist, ray_hits, nullptr, nullptr, nullptr, num_rays, counters,
               0, nullptr};
    ILX_LAUNCH(raycast_kernel, dim3(ray_grid(num_rays)), dim3(kRayBlock), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_raycast_mesh_stats");
    return ILX_OK;
}

extern "C" int ilx_raycaster_update(ilx_mesh_t mesh, const float* pos_w, const float* quat_w, const float* ray_starts_local,
                                    const float* ray_directions_local, const uint8_t* env_mask, int attach_yaw_only,
                                    float max_dist, float* ray_hits_w, float* height_scan, float height_offset,
                                    int64_t n, int num_rays, int hints, void* stream) {
    ILX_REQUIRE(mesh, ILX_EINVAL, "ilx_raycaster_update: mesh is NULL");
    ILX_REQUIRE(n >= 0 && num_rays > 0, ILX_EINVAL, "ilx_raycaster_update: bad shape (%lld, %d)", (long long)n, num_rays);
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(pos_w && quat_w && ray_starts_local && ray_directions_local && ray_hits_w, ILX_EINVAL,
                "ilx_raycaster_update: NULL operand");
    ScanArgs a{mesh->h.dev, pos_w, quat_w, 3, 4, nullptr, nullptr, nullptr, ray_starts_local, ray_directions_local, env_mask,
               attach_yaw_only, max_dist, ray_hits_w, height_scan, height_offset, n, num_rays, 0, nullptr, nullptr, nullptr, nullptr};
    const int rc = launch_scan(mesh, a, hints, stream);
    if (rc != ILX_OK) return rc;
    ILX_LAUNCH_CHECK("ilx_raycaster_update");
    return ILX_OK;
}

#ifdef ILX_RAY_STAMPS
extern "C" int ilx_debug_prepare_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_prep_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
extern "C" int ilx_debug_ray_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_ray_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int ilx_raycaster_refresh(const ilx_raycaster* rc, void* stream) {
    ILX_REQUIRE(rc, ILX_EINVAL, "ilx_raycaster_refresh: descriptor is NULL");
    const ilx_raycaster& d = *rc;
    ILX_REQUIRE(d.mesh, ILX_EINVAL, "ilx_raycaster_refresh: mesh is NULL");
    ILX_REQUIRE(d.num_envs >= 0 && d.num_rays > 0, ILX_EINVAL, "ilx_raycaster_refresh: bad shape (%lld, %d)",
                (long long)d.num_envs, d.num_rays);
    if (d.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(d.pose_pos && d.pose_quat && d.ray_starts_local && d.ray_directions_local && d.pos_w && d.quat_w &&
                    d.ray_hits_w && d.timestamp && d.timestamp_last_update && d.is_outdated,
                ILX_EINVAL, "ilx_raycaster_refresh: NULL operand");
    ILX_REQUIRE(d.pose_pos_stride >= 3 && d.pose_quat_stride >= 4, ILX_EINVAL, "ilx_raycaster_refresh: bad pose strides");
    ScanArgs a{d.mesh->h.dev, d.pose_pos, d.pose_quat, d.pose_pos_stride, d.pose_quat_stride, d.drift, d.pos_w, d.quat_w,
               d.ray_starts_local, d.ray_directions_local, d.is_outdated, d.attach_yaw_only, d.max_distance, d.ray_hits_w,
               d.height_scan, d.height_offset, d.num_envs, d.num_rays, 0, nullptr, d.timestamp, d.timestamp_last_update,
               d.is_outdated};
    const int status = launch_scan(d.mesh, a, d.hints, stream);
    if (status != ILX_OK) return status;
    ILX_LAUNCH_CHECK("ilx_raycaster_refresh");
    return ILX_OK;
}

// Whether a refresh described by `a` goes over the height-field patches (preparation + cast launches).
static bool lattice_refresh(ilx_mesh_t mesh, const ScanArgs& a, int hints) {
    return (hints & ILX_RAYS_Z_PARALLEL) && a.yaw_only && mesh->h.dev.num_patches > 0 && a.r <= kLatMaxRays && a.n < (int64_t(1) << 31);
}

extern "C" int ilx_scene_update_prepare_raycaster(const ilx_scene* scene, const ilx_raycaster* rc, void* stream) {
    ILX_REQUIRE(scene && rc, ILX_EINVAL, "ilx_scene_update_prepare_raycaster: NULL descriptor");
    const ilx_raycaster& d = *rc;
    ILX_REQUIRE(d.mesh, ILX_EINVAL, "ilx_scene_update_prepare_raycaster: mesh is NULL");
    ILX_REQUIRE(d.num_envs == scene->num_envs && d.num_rays > 0, ILX_EINVAL,
                "ilx_scene_update_prepare_raycaster: scene and sensor differ in their env count (%lld, %lld) or the sensor has no rays",
                (long long)scene->num_envs, (long long)d.num_envs);
    ILX_REQUIRE(d.pose_pos && d.pose_quat && d.ray_starts_local && d.ray_directions_local && d.pos_w && d.quat_w &&
                    d.ray_hits_w && d.timestamp && d.timestamp_last_update && d.is_outdated,
                ILX_EINVAL, "ilx_scene_update_prepare_raycaster: NULL operand");
    ILX_REQUIRE(d.pose_pos_stride >= 3 && d.pose_quat_stride >= 4, ILX_EINVAL, "ilx_scene_update_prepare_raycaster: bad pose strides");
    ILX_REQUIRE(scene->ray_timestamp == d.timestamp && scene->ray_timestamp_last_update == d.timestamp_last_update &&
                    scene->ray_is_outdated == d.is_outdated,
                ILX_EINVAL, "ilx_scene_update_prepare_raycaster: the scene's ray-caster clock is not this sensor's");
    ILX_REQUIRE(!(d.hints & ILX_RAYS_PREPARED), ILX_EINVAL, "ilx_scene_update_prepare_raycaster: ILX_RAYS_PREPARED is for the refresh that follows");
    if (d.num_envs == 0) return ILX_OK;
    ScanArgs a{d.mesh->h.dev, d.pose_pos, d.pose_quat, d.pose_pos_stride, d.pose_quat_stride, d.drift, d.pos_w, d.quat_w,
               d.ray_starts_local, d.ray_directions_local, d.is_outdated, d.attach_yaw_only, d.max_distance, d.ray_hits_w,
               d.height_scan, d.height_offset, d.num_envs, d.num_rays, 0, nullptr, d.timestamp, d.timestamp_last_update,
               d.is_outdated};
    if (lattice_refresh(d.mesh, a, d.hints)) {
        void* ws = nullptr;
        float* box = nullptr;
        bool same_pattern = false;
        int status = lattice_workspace(d.mesh, stream, lattice_workspace_bytes(a.n, a.r), a, true, &ws, &box, &same_pattern);
        if (status != ILX_OK) return status;
        LatEnvRec* recs = reinterpret_cast<LatEnvRec*>(static_cast<char*>(ws) + lattice_workspace_bytes(0, a.r));
        if (same_pattern && (d.hints & ILX_RAYS_PATTERN_STATIC)) {   // the packed pattern and its box are in the workspace: one launch
            SceneArgs s{};
            unsigned blocks = 0;
            status = scene_args_from(*scene, s, blocks);
            if (status != ILX_OK) return status;
            const unsigned prep_blocks = blocks_for(a.n);
            ILX_LAUNCH(scene_update_prepare_kernel, dim3(prep_blocks + blocks), dim3(256), 0, as_stream(stream), s, a, recs,
                               box, prep_blocks);
            ILX_LAUNCH_CHECK("ilx_scene_update_prepare_raycaster");
            return ILX_OK;
        }
        // first use of this pattern on this stream: the update, then the preparation launch proper (it packs the pattern)
        status = ilx_scene_update(scene, stream);
        if (status != ILX_OK) return status;
        ILX_LAUNCH(raycaster_prepare_kernel, dim3(blocks_for(a.n)), dim3(256), 0, as_stream(stream), a, recs,
                           static_cast<float4*>(ws), box, 0);
        ILX_LAUNCH_CHECK("ilx_scene_update_prepare_raycaster");
        return ILX_OK;
    }
    // no preparation launch in this refresh: the update alone; the refresh that follows does everything
    return ilx_scene_update(scene, stream);
}

// ilx_raycaster_refresh for callers that must not allocate.  What ilx_raycaster_refresh keeps between its launches in
// a workspace grown with hipMalloc lives here in one array of the code object, per GPU: every call on a GPU shares it.
// Nothing in it has to start at zero: the preparation launch writes the pattern and every env record before the cast
// reads them, and records beyond the call's envs are never read.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// The array's address on the current device (in host code `scratch` itself is only a handle for the runtime, not a
// device address), looked up once per device: when the device's first mesh is created.
static int static_scratch(void** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex lock;
    static void* address[kMaxDevices] = {};
    int device = -1;
    int rc = check_hip(hipGetDevice(&device), "ilx_raycaster_refresh_static: hipGetDevice");
    if (rc != ILX_OK) return rc;
    ILX_REQUIRE(device >= 0 && device < kMaxDevices, ILX_EINVAL, "ilx_raycaster_refresh_static: device %d out of range", device);
    std::lock_guard<std::mutex> guard(lock);
    if (!address[device]) {
        void* ptr = nullptr;
        rc = check_hip(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "ilx_raycaster_refresh_static: hipGetSymbolAddress");
        if (rc != ILX_OK) return rc;
        ILX_REQUIRE(ptr, ILX_EHIP, "ilx_raycaster_refresh_static: no device address for the built-in buffer");
        address[device] = ptr;
    }
    *out = address[device];
    return ILX_OK;
}

extern "C" int ilx_raycaster_refresh_static(const ilx_raycaster* rc, void* stream) {
    ILX_REQUIRE(rc, ILX_EINVAL, "ilx_raycaster_refresh_static: descriptor is NULL");
    const ilx_raycaster& d = *rc;
    ILX_REQUIRE(d.mesh, ILX_EINVAL, "ilx_raycaster_refresh_static: mesh is NULL");
    ILX_REQUIRE(d.num_envs >= 0 && d.num_rays > 0, ILX_EINVAL, "ilx_raycaster_refresh_static: bad shape (%lld, %d)",
                (long long)d.num_envs, d.num_rays);
    if (d.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(d.pose_pos && d.pose_quat && d.ray_starts_local && d.ray_directions_local && d.pos_w && d.quat_w &&
                    d.ray_hits_w && d.timestamp && d.timestamp_last_update && d.is_outdated,
                ILX_EINVAL, "ilx_raycaster_refresh_static: NULL operand");
    ILX_REQUIRE(d.pose_pos_stride >= 3 && d.pose_quat_stride >= 4, ILX_EINVAL, "ilx_raycaster_refresh_static: bad pose strides");
    // ilx_scene_update_prepare_raycaster writes the mesh's per-stream workspace, never the built-in buffer: no
    // preparation can have been issued for this entry point
    ILX_REQUIRE(!(d.hints & ILX_RAYS_PREPARED), ILX_EINVAL,
                "ilx_raycaster_refresh_static: ILX_RAYS_PREPARED names a preparation in the per-stream workspace; this entry "
                "point prepares in the built-in buffer itself");
    void* ws = nullptr;
    int status = static_scratch(&ws);
    if (status != ILX_OK) return status;
    ScanArgs a{d.mesh->h.dev, d.pose_pos, d.pose_quat, d.pose_pos_stride, d.pose_quat_stride, d.drift, d.pos_w, d.quat_w,
               d.ray_starts_local, d.ray_directions_local, d.is_outdated, d.attach_yaw_only, d.max_distance, d.ray_hits_w,
               d.height_scan, d.height_offset, d.num_envs, d.num_rays, 0, nullptr, d.timestamp, d.timestamp_last_update,
               d.is_outdated};
    status = launch_scan(d.mesh, a, d.hints, stream, ws, kStaticScratchBytes);
    if (status != ILX_OK) return status;
    ILX_LAUNCH_CHECK("ilx_raycaster_refresh_static");
    return ILX_OK;
}

// Column kernel + tree kernel for the rays it left out (that launch also advances the sensor clocks, and is otherwise a
// no-op when there are none), or the tree kernel alone followed by the clock update.
static int launch_scan(ilx_mesh_t mesh, ScanArgs a, int hints, void* stream, void* fixed_ws, size_t fixed_bytes) {
    const int64_t rays = a.n * a.r;
    if (lattice_refresh(mesh, a, hints)) {
        // height-field patches: a preparation launch (one thread per env; it also advances the sensor clocks), the cast
        // (one wavefront per env), and the tree launch, which only looks at the flag: it has work when rays broke the hint
        void* ws = fixed_ws;
        float* box = nullptr;   // the built-in buffer keeps nothing from call to call
        bool same_pattern = false;
        if (fixed_ws) {
            ILX_REQUIRE(lattice_workspace_bytes(a.n, a.r) <= fixed_bytes, ILX_ESHAPE,
                        "ilx_raycaster_refresh_static: %lld envs x %d rays need %zu bytes between the launches, the built-in "
                        "buffer holds %zu (at most %d envs at %d rays); use ilx_raycaster_refresh",
                        (long long)a.n, a.r, lattice_workspace_bytes(a.n, a.r), fixed_bytes, kStaticEnvs, kLatMaxRays);
        } else {
            const int rc = lattice_workspace(mesh, stream, lattice_workspace_bytes(a.n, a.r), a, !(hints & ILX_RAYS_PREPARED), &ws, &box, &same_pattern);
            if (rc != ILX_OK) return rc;
        }
        take_ticket(mesh, a.ticket, a.off_grid);
        launch_lattice(mesh, a, hints, ws, box, same_pattern && (hints & ILX_RAYS_PATTERN_STATIC), stream);
        return ILX_OK;
    }
    if ((hints & ILX_RAYS_Z_PARALLEL) && mesh->h.d_cells) {
        take_ticket(mesh, a.ticket, a.off_grid);
        ILX_LAUNCH(raycaster_update_column_kernel, dim3(column_grid(rays)), dim3(kRayBlock), 0, as_stream(stream), a);
        ILX_LAUNCH(raycaster_update_kernel, dim3(ray_grid(rays)), dim3(kRayBlock), 0, as_stream(stream), a);
        return ILX_OK;
    }
    ILX_LAUNCH(raycaster_update_kernel, dim3(ray_grid(rays)), dim3(kRayBlock), 0, as_stream(stream), a);
    if (a.outdated)
        ILX_LAUNCH(sensor_refreshed_kernel, dim3(blocks_for(a.n)), dim3(256), 0, as_stream(stream), a.clock_now, a.clock_last,
                           a.outdated, a.n);
    return ILX_OK;
}

extern "C" int ilx_raycaster_camera_refresh(const ilx_raycaster_camera* rc, void* stream) {
    ILX_REQUIRE(rc, ILX_EINVAL, "ilx_raycaster_camera_refresh: descriptor is NULL");
    const ilx_raycaster_camera& d = *rc;
    ILX_REQUIRE(d.mesh, ILX_EINVAL, "ilx_raycaster_camera_refresh: mesh is NULL");
    ILX_REQUIRE(d.num_envs >= 0 && d.num_rays > 0, ILX_EINVAL, "ilx_raycaster_camera_refresh: bad shape (%lld, %d)",
                (long long)d.num_envs, d.num_rays);
    ILX_REQUIRE(d.depth_clipping >= ILX_DEPTH_CLIP_NONE && d.depth_clipping <= ILX_DEPTH_CLIP_ZERO, ILX_EINVAL,
                "ilx_raycaster_camera_refresh: unknown depth clipping behaviour %d", d.depth_clipping);
    if (d.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(d.view_pos && d.view_quat && d.offset_pos && d.offset_quat && d.ray_starts && d.ray_directions && d.pos_w &&
                    d.quat_w_world && d.ray_hits_w,
                ILX_EINVAL, "ilx_raycaster_camera_refresh: NULL operand");
    ILX_REQUIRE(d.view_pos_stride >= 3 && d.view_quat_stride >= 4, ILX_EINVAL, "ilx_raycaster_camera_refresh: bad pose strides");
    ILX_REQUIRE(!d.is_outdated || (d.timestamp && d.timestamp_last_update), ILX_EINVAL,
                "ilx_raycaster_camera_refresh: is_outdated comes with the two sensor clocks");
    CameraArgs a{d.mesh->h.dev, d.view_pos, d.view_quat, d.view_pos_stride, d.view_quat_stride, d.offset_pos, d.offset_quat,
                 d.ray_starts, d.ray_directions, d.is_outdated, d.pos_w, d.quat_w_world, d.frame, d.ray_hits_w,
                 d.distance_to_image_plane, d.distance_to_camera, d.normals, d.max_distance, d.depth_clipping, d.num_envs, d.num_rays};
    ILX_LAUNCH(raycaster_camera_kernel, dim3(ray_grid(d.num_envs * d.num_rays)), dim3(kRayBlock), 0, as_stream(stream), a);
    if (d.is_outdated)
        ILX_LAUNCH(sensor_refreshed_kernel, dim3(blocks_for(d.num_envs)), dim3(256), 0, as_stream(stream), d.timestamp,
                           d.timestamp_last_update, d.is_outdated, d.num_envs);
    ILX_LAUNCH_CHECK("ilx_raycaster_camera_refresh");
    return ILX_OK;
}
