// Height-field patches of a triangle mesh, recognised at ilx_mesh_create.
//
// The reference's terrains are mostly height fields triangulated by convert_height_field_to_mesh
// (isaaclab/terrains/height_field/utils.py:119-173): vertices on an x/y lattice in row-major order (x is the slow
// index, y the fast one), two faces per cell in the order (ind0, ind3, ind1), (ind0, ind2, ind3).  For such a run of
// faces a z-parallel ray needs neither a tree nor a cell list: the cell under the ray is found from the lattice lines,
// its two faces are implied, and their vertices are (xs[i], ys[j], z[i][j]) -- 4 bytes per vertex instead of a 4-byte
// cell word + 16-byte index records + 16-byte vertices.  The kernels rebuild v0, v1, v2 from those three arrays; the
// values are the mesh's own floats, so the intersection arithmetic sees the same operands as every other path.
//
// Not every cell of a recognised run qualifies.  A cell is SIMPLE when
//   * its four corners sit exactly on the lattice lines (the slope correction of utils.py:128-150 moves vertices), and
//   * no other triangle of the mesh -- one with a moved vertex, one of another patch, one outside any patch -- overlaps
//     the cell's rectangle in xy (padded boxes, then separating axes).
// Only rays strictly inside a simple cell (by a margin far wider than the barycentric slack of the intersection test)
// take the lattice path; every other ray goes through the general structures, so the lattice never changes a result.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace ilx {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; ++k) mn[k] = std::numeric_limits<float>::infinity(), mx[k] = -mn[k];
    }
    void grow(const Box& o) {
        for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], o.mn[k]), mx[k] = std::max(mx[k], o.mx[k]);
    }
};

// One recognised run (device-visible, 64 bytes).  Vertex (i, j), 0 <= i < nx, 0 <= j < ny, has x = lines[line_off + i],
// y = lines[line_off + nx + j], z = z[z_off + i * ny + j].  Cell (i, j) holds faces face_base + 2 * (i * (ny - 1) + j)
// = (v(i,j), v(i+1,j+1), v(i,j+1)) and the next one = (v(i,j), v(i+1,j), v(i+1,j+1)).  Bit (j & 31) of
// mask[mask_off + i * mask_stride + (j >> 5)] says that cell (i, j) is simple.
struct LatticePatch {
    float x0, y0, x1, y1;   // first and last lattice line per axis
    float inv_dx, inv_dy;   // (lines - 1) / extent: floor((x - x0) * inv_dx) is within one of the cell index
    int32_t nx, ny;
    uint32_t z_off, line_off, mask_off;
    int32_t mask_stride;
    int32_t face_base;
    int32_t pad[3];
};
static_assert(sizeof(LatticePatch) == 64, "two descriptors per 128-byte line");

constexpr int kLatticeMinLines = 5;  // shortest run kept, in lattice lines per axis (the kernels read five lines at once)

// The cell-index estimate shared by the build-time check and the kernels.
__host__ __device__ inline int lattice_guess(float x, float x0, float inv) {
    return static_cast<int>(floorf((x - x0) * inv));
}

struct LatticeBuild {
    std::vector<LatticePatch> patches;
    std::vector<float> z, lines;
    std::vector<uint32_t> mask;
    int64_t cells = 0, simple = 0, faces = 0;
};

namespace lattice_detail {

struct Run {
    int face0;    // first face of the run
    int vert0;    // vertex (0, 0)
    int nx, ny;   // lattice lines
};

// Faces t, t + 1 as the two halves of cell k of a row that starts at vertex `row0` in a lattice with ny columns?
inline bool cell_matches(const int32_t* tris, int t, int row0, int k, int ny) {
    const int a = row0 + k, b = a + 1, c = a + ny, d = c + 1;
    const int32_t* f0 = tris + 3 * static_cast<int64_t>(t);
    const int32_t* f1 = f0 + 3;
    return f0[0] == a && f0[1] == d && f0[2] == b && f1[0] == a && f1[1] == c && f1[2] == d;
}

// Longest run of whole rows that starts at face t (0 rows when the faces there do not open a lattice).
inline bool run_at(const int32_t* tris, int nt, int64_t nv, int t, Run& out) {
    if (t + 1 >= nt) return false;
    const int32_t* f0 = tris + 3 * static_cast<int64_t>(t);
    const int a = f0[0], ny = f0[1] - 1 - a;  // f0 = (a, a + ny + 1, a + 1)
    if (ny < kLatticeMinLines || f0[2] != a + 1) return false;
    const int cells_per_row = ny - 1;
    int rows = 0;
    for (;; ++rows) {
        const int64_t first = static_cast<int64_t>(t) + 2ll * rows * cells_per_row;
        if (first + 2ll * cells_per_row > nt) break;
        const int64_t row0 = static_cast<int64_t>(a) + static_cast<int64_t>(rows) * ny;
        if (row0 + 2ll * ny > nv) break;  // the row's far vertices must exist
        bool ok = true;
        for (int k = 0; k < cells_per_row && ok; ++k)
            ok = cell_matches(tris, static_cast<int>(first) + 2 * k, static_cast<int>(row0), k, ny);
        if (!ok) break;
    }
    if (rows + 1 < kLatticeMinLines) return false;
    out = Run{t, a, rows + 1, ny};
    return true;
}

inline float median_of(std::vector<float>& v) {
    const size_t mid = v.size() / 2;
    std::nth_element(v.begin(), v.begin() + mid, v.end());
    return v[mid];
}

}  // namespace lattice_detail

// `box`: the padded per-triangle boxes of the mesh (the pad covers the slack of the intersection test).
inline void detect_lattices(const float* verts, int64_t nv, const int32_t* tris, int nt, const std::vector<Box>& box,
                            LatticeBuild& out) {
    using namespace lattice_detail;
    std::vector<Run> runs;
    for (int t = 0; t + 1 < nt;) {
        Run r;
        if (run_at(tris, nt, nv, t, r)) {
            runs.push_back(r);
            t += 2 * (r.nx - 1) * (r.ny - 1);
        } else {
            ++t;
        }
    }
    // lattice lines; runs whose lines are not strictly increasing, or too uneven for the index estimate, are dropped
    struct Cand {
        Run run;
        std::vector<float> xs, ys;
        float inv_dx, inv_dy;
    };
    std::vector<Cand> cands;
    std::vector<float> tmp;
    // The line of row i is the coordinate most of its vertices share -- unless the slope correction moved most of the
    // row, so: take the row medians, estimate a uniform lattice from them (median spacing, median offset), and let
    // every row pick, among its own values, the one nearest to the estimate.  fetch(i, k) = k-th value of row i.
    // False when the rows do not advance.
    auto lines_of = [&](int rows, int per_row, auto&& fetch, std::vector<float>& lines) {
        std::vector<double> med(rows);
        for (int i = 0; i < rows; ++i) {
            tmp.resize(per_row);
            for (int k = 0; k < per_row; ++k) tmp[k] = fetch(i, k);
            med[i] = median_of(tmp);
        }
        std::vector<float> d(rows - 1);
        for (int i = 0; i + 1 < rows; ++i) d[i] = static_cast<float>(med[i + 1] - med[i]);
        const double step = median_of(d);
        if (!(step > 0.0)) return false;
        std::vector<float> o(rows);
        for (int i = 0; i < rows; ++i) o[i] = static_cast<float>(med[i] - i * step);
        const double first = median_of(o);
        for (int i = 0; i < rows; ++i) {
            const double want = first + i * step;
            float best = fetch(i, 0);
            for (int k = 1; k < per_row; ++k)
                if (std::fabs(fetch(i, k) - want) < std::fabs(best - want)) best = fetch(i, k);
            // a row moved as a whole has no vertex on its line: the estimate stands in (no cell next to it is simple,
            // the line only has to keep the order)
            lines[i] = std::fabs(best - want) < 0.25 * step ? best : static_cast<float>(want);
        }
        return true;
    };
    for (const Run& r : runs) {
        Cand c{r, std::vector<float>(r.nx), std::vector<float>(r.ny), 0.0f, 0.0f};
        auto vx = [&](int i, int j) { return verts[3 * (static_cast<int64_t>(r.vert0) + static_cast<int64_t>(i) * r.ny + j)]; };
        auto vy = [&](int j, int i) { return verts[3 * (static_cast<int64_t>(r.vert0) + static_cast<int64_t>(i) * r.ny + j) + 1]; };
        if (!lines_of(r.nx, r.ny, vx, c.xs) || !lines_of(r.ny, r.nx, vy, c.ys)) continue;
        auto usable = [](const std::vector<float>& l, float& inv) {
            for (size_t k = 1; k < l.size(); ++k)
                if (!(l[k] > l[k - 1])) return false;
            inv = static_cast<float>((l.size() - 1) / (double(l.back()) - double(l.front())));
            if (!std::isfinite(inv) || !(inv > 0.0f)) return false;
            for (size_t k = 0; k < l.size(); ++k) {
                const int g = lattice_guess(l[k], l.front(), inv);
                if (g < static_cast<int>(k) - 1 || g > static_cast<int>(k) + 1) return false;
            }
            return true;
        };
        if (usable(c.xs, c.inv_dx) && usable(c.ys, c.inv_dy)) cands.push_back(std::move(c));
    }
    if (cands.empty()) return;

    // which patch (if any) a face belongs to: runs are disjoint and sorted by face0
    auto patch_of_face = [&](int t) {
        int lo = 0, hi = static_cast<int>(cands.size()) - 1, at = -1;
        while (lo <= hi) {
            const int mid = (lo + hi) / 2;
            if (cands[mid].run.face0 <= t) at = mid, lo = mid + 1;
            else hi = mid - 1;
        }
        if (at < 0) return -1;
        const Run& r = cands[at].run;
        return t < r.face0 + 2 * (r.nx - 1) * (r.ny - 1) ? at : -1;
    };

    // per patch: one byte per cell while building
    std::vector<std::vector<uint8_t>> simple(cands.size());
    std::vector<std::vector<uint8_t>> on_lattice(cands.size());
    for (size_t p = 0; p < cands.size(); ++p) {
        const Cand& c = cands[p];
        const Run& r = c.run;
        std::vector<uint8_t>& on = on_lattice[p];
        on.resize(static_cast<size_t>(r.nx) * r.ny);
        for (int i = 0; i < r.nx; ++i)
            for (int j = 0; j < r.ny; ++j) {
                const float* v = verts + 3 * (static_cast<int64_t>(r.vert0) + static_cast<int64_t>(i) * r.ny + j);
                on[static_cast<size_t>(i) * r.ny + j] = v[0] == c.xs[i] && v[1] == c.ys[j];
            }
        std::vector<uint8_t>& s = simple[p];
        s.resize(static_cast<size_t>(r.nx - 1) * (r.ny - 1));
        for (int i = 0; i + 1 < r.nx; ++i)
            for (int j = 0; j + 1 < r.ny; ++j)
                s[static_cast<size_t>(i) * (r.ny - 1) + j] = on[static_cast<size_t>(i) * r.ny + j] && on[static_cast<size_t>(i) * r.ny + j + 1] &&
                                                             on[static_cast<size_t>(i + 1) * r.ny + j] &&
                                                             on[static_cast<size_t>(i + 1) * r.ny + j + 1];
    }

    // A triangle that is not the regular content of a cell clears every cell of patch p that it may overlap.
    auto clear_under = [&](size_t p, int t) {
        const Cand& c = cands[p];
        const Run& r = c.run;
        const Box& b = box[t];
        if (b.mx[0] < c.xs.front() || b.mn[0] > c.xs.back() || b.mx[1] < c.ys.front() || b.mn[1] > c.ys.back()) return;
        // cells [i0, i1] x [j0, j1] whose closed rectangle meets the padded box
        auto range = [](const std::vector<float>& l, float lo, float hi, int& k0, int& k1) {
            const int n = static_cast<int>(l.size());
            k0 = static_cast<int>(std::lower_bound(l.begin() + 1, l.end(), lo) - (l.begin() + 1));  // first cell with upper line >= lo
            k1 = static_cast<int>(std::upper_bound(l.begin(), l.end(), hi) - l.begin()) - 1;         // last cell with lower line <= hi
            k0 = std::max(0, std::min(k0, n - 2));
            k1 = std::max(0, std::min(k1, n - 2));
        };
        int i0, i1, j0, j1;
        range(c.xs, b.mn[0], b.mx[0], i0, i1);
        range(c.ys, b.mn[1], b.mx[1], j0, j1);
        std::vector<uint8_t>& s = simple[p];
        if (i1 - i0 < 2 && j1 - j0 < 2) {
            for (int i = i0; i <= i1; ++i)
                for (int j = j0; j <= j1; ++j) s[static_cast<size_t>(i) * (r.ny - 1) + j] = 0;
            return;
        }
        // a wide box: cull with the three edge half-planes of the projected triangle, grown by the box's pad
        double P[3][2];
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 2; ++a) P[k][a] = verts[3 * static_cast<int64_t>(tris[3 * static_cast<int64_t>(t) + k]) + a];
        double pad = 0.0;
        for (int a = 0; a < 2; ++a) pad = std::max(pad, std::min(P[0][a], std::min(P[1][a], P[2][a])) - double(b.mn[a]));
        double ex[3], ey[3], off[3], grow[3];
        bool use[3];
        for (int k = 0; k < 3; ++k) {
            const double* A = P[k];
            const double* B = P[(k + 1) % 3];
            const double* C = P[(k + 2) % 3];
            double nx_ = -(B[1] - A[1]), ny_ = B[0] - A[0];
            const double side = nx_ * (C[0] - A[0]) + ny_ * (C[1] - A[1]);
            const double len = std::sqrt(nx_ * nx_ + ny_ * ny_);
            use[k] = len > 0.0 && std::fabs(side) > 1e-12 * len * len;  // degenerate projection: no culling
            if (side < 0.0) nx_ = -nx_, ny_ = -ny_;
            ex[k] = nx_, ey[k] = ny_, off[k] = nx_ * A[0] + ny_ * A[1], grow[k] = pad * len;
        }
        for (int i = i0; i <= i1; ++i) {
            const double cx = 0.5 * (double(c.xs[i]) + c.xs[i + 1]), hx = 0.5 * (double(c.xs[i + 1]) - c.xs[i]) * (1.0 + 1e-6);
            for (int j = j0; j <= j1; ++j) {
                const double cy = 0.5 * (double(c.ys[j]) + c.ys[j + 1]), hy = 0.5 * (double(c.ys[j + 1]) - c.ys[j]) * (1.0 + 1e-6);
                bool hit = true;
                for (int k = 0; k < 3 && hit; ++k)
                    if (use[k]) hit = (ex[k] * cx + ey[k] * cy - off[k]) + (hx * std::fabs(ex[k]) + hy * std::fabs(ey[k])) + grow[k] >= 0.0;
                if (hit) s[static_cast<size_t>(i) * (r.ny - 1) + j] = 0;
            }
        }
    };

    // coarse bins over the union of the patch extents: which patches a triangle has to be held against
    float u0[2] = {INFINITY, INFINITY}, u1[2] = {-INFINITY, -INFINITY};
    for (const Cand& c : cands) {
        u0[0] = std::min(u0[0], c.xs.front()), u1[0] = std::max(u1[0], c.xs.back());
        u0[1] = std::min(u0[1], c.ys.front()), u1[1] = std::max(u1[1], c.ys.back());
    }
    constexpr int G = 64;
    const double bw[2] = {std::max((double(u1[0]) - u0[0]) / G, 1e-9), std::max((double(u1[1]) - u0[1]) / G, 1e-9)};
    auto bin = [&](float v, int a) {
        const int k = static_cast<int>(std::floor((double(v) - u0[a]) / bw[a]));
        return std::max(0, std::min(G - 1, k));
    };
    std::vector<std::vector<int>> bins(G * G);
    for (size_t p = 0; p < cands.size(); ++p) {
        const Cand& c = cands[p];
        for (int by = bin(c.ys.front(), 1); by <= bin(c.ys.back(), 1); ++by)
            for (int bx = bin(c.xs.front(), 0); bx <= bin(c.xs.back(), 0); ++bx) bins[by * G + bx].push_back(static_cast<int>(p));
    }
    std::vector<int> seen(cands.size(), -1);
    for (int t = 0; t < nt; ++t) {
        const int own = patch_of_face(t);
        bool regular = false;
        if (own >= 0) {  // regular content of its own cell when all three vertices sit on the lattice
            const Run& r = cands[own].run;
            regular = true;
            for (int k = 0; k < 3; ++k) regular = regular && on_lattice[own][static_cast<size_t>(tris[3 * static_cast<int64_t>(t) + k] - r.vert0)];
        }
        const Box& b = box[t];
        if (b.mx[0] < u0[0] || b.mn[0] > u1[0] || b.mx[1] < u0[1] || b.mn[1] > u1[1]) continue;
        for (int by = bin(b.mn[1], 1); by <= bin(b.mx[1], 1); ++by)
            for (int bx = bin(b.mn[0], 0); bx <= bin(b.mx[0], 0); ++bx)
                for (int p : bins[by * G + bx]) {
                    if (seen[p] == t) continue;
                    seen[p] = t;
                    if (p == own && regular) continue;
                    clear_under(static_cast<size_t>(p), t);
                }
    }

    // pack
    for (size_t p = 0; p < cands.size(); ++p) {
        const Cand& c = cands[p];
        const Run& r = c.run;
        LatticePatch d{};
        d.x0 = c.xs.front(), d.x1 = c.xs.back(), d.y0 = c.ys.front(), d.y1 = c.ys.back();
        d.inv_dx = c.inv_dx, d.inv_dy = c.inv_dy;
        d.nx = r.nx, d.ny = r.ny;
        d.z_off = static_cast<uint32_t>(out.z.size());
        d.line_off = static_cast<uint32_t>(out.lines.size());
        d.mask_off = static_cast<uint32_t>(out.mask.size());
        d.mask_stride = (r.ny - 1 + 31) / 32;
        d.face_base = r.face0;
        for (int i = 0; i < r.nx; ++i)
            for (int j = 0; j < r.ny; ++j) out.z.push_back(verts[3 * (static_cast<int64_t>(r.vert0) + static_cast<int64_t>(i) * r.ny + j) + 2]);
        out.lines.insert(out.lines.end(), c.xs.begin(), c.xs.end());
        out.lines.insert(out.lines.end(), c.ys.begin(), c.ys.end());
        out.mask.resize(out.mask.size() + static_cast<size_t>(r.nx - 1) * d.mask_stride, 0u);
        for (int i = 0; i + 1 < r.nx; ++i)
            for (int j = 0; j + 1 < r.ny; ++j)
                if (simple[p][static_cast<size_t>(i) * (r.ny - 1) + j]) {
                    out.mask[d.mask_off + static_cast<size_t>(i) * d.mask_stride + (j >> 5)] |= 1u << (j & 31);
                    ++out.simple;
                }
        out.cells += static_cast<int64_t>(r.nx - 1) * (r.ny - 1);
        out.faces += 2ll * (r.nx - 1) * (r.ny - 1);
        out.patches.push_back(d);
    }
}

}  // namespace ilx
