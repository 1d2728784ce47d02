// smpl_amd/csrc/field.hip -- distance-field construction on the GPU: the step immediately before the
// hot path.  What it stands behind: sbpl::OccupancyGrid::addPointsToField / removePointsFromField / updatePointsInField
// with its reference counts (smpl/src/occupancy_grid.cpp:357-422, initRefCounts :424-441) over
// DistanceMap<EuclidDistance>::addPointsToMap / removePointsFromMap / updatePointsInMap
// (smpl/include/smpl/distance_map/detail/distance_map.hpp:306-435) and its propagation (:627-839).
//
// Specification.  The reference's propagation is a bucketed brushfire whose values depend on the pop order of its
// bucket stacks where distances tie, and it needs Eigen, so it can neither be reproduced bit
// for bit nor be compiled here.  SURVEY 8(c) allows the construction to be any EXACT Euclidean transform with the
// reference's cap and border rule, which is what this is:
//     d2(c) = min( dmax^2 , min over occupied or border cells o of |c - o|^2 )        (cells, integer arithmetic)
// dmax = ceil(max_dist / res) (distance_map.hpp:126); the cells of the one-cell border layer around the grid count as
// obstacles (:560-606).  Checked against a brute-force nearest-obstacle search and against the host builder of the
// test scenes (tests/test_gpu_field.py).  Parity with the reference's own propagation: unpinned.
//
// Method: the squared Euclidean distance separates by axes,
//     d2(x,y,z) = min_x' (x-x')^2 + [ min_y' (y-y')^2 + [ min_z' over occupied (x',y',z') of (z-z')^2 ] ],
// and with a cap every inner minimum only needs a window of dmax cells.  Three passes (z, y, x), one thread per cell,
// each scanning outwards until the offset alone exceeds the best value so far.  Reads run along z (the fastest axis)
// in every pass, so a wavefront's 64 loads share cache lines; nothing is atomic, nothing iterates to convergence.
// HBM traffic: occupancy 1 B + two 2-byte intermediates + the 2-byte brick-tiled result per cell.
//
// An edit is INCREMENTAL (what the reference's add/remove propagation is for): a changed cell can only move values within
// dmax cells of it in every pass, and both intermediate passes stay resident, so the three passes are re-run on the
// bounding box of the changed cells grown by dmax -- 81^3 cells for one point at dmax = 40, whatever the grid's size.
// An edit whose window covers more than half of the grid recomputes everything.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/smpl_amd.h"
#include "grid_handle.h"

namespace {

#define FIELD_BLOCK 256

// a box of cells, inclusive
struct Window { int x0, y0, z0, x1, y1, z1; };
__host__ __device__ inline long long window_cells(const Window& w) { return (long long)(w.x1 - w.x0 + 1) * (w.y1 - w.y0 + 1) * (w.z1 - w.z0 + 1); }
__device__ __forceinline__ void window_cell(const Window& w, long long i, int& x, int& y, int& z)
{
    const int wz = w.z1 - w.z0 + 1, wy = w.y1 - w.y0 + 1;
    z = w.z0 + (int)(i % wz);
    y = w.y0 + (int)(i / wz % wy);
    x = w.x0 + (int)(i / ((long long)wz * wy));
}

__global__ void __launch_bounds__(FIELD_BLOCK)
k_occ_boxes(unsigned char* __restrict__ occ, int* __restrict__ counts, int nx, int ny, int nz, const int* __restrict__ boxes, int nboxes, Window w)
{
    // boxes: inclusive cell ranges {x0, y0, z0, x1, y1, z1} inside the window w; every cell inside one of them becomes
    // occupied (with reference counts: once per box that covers it)
    const long long total = window_cells(w);
    for (long long k = (long long)blockIdx.x * FIELD_BLOCK + threadIdx.x; k < total; k += (long long)gridDim.x * FIELD_BLOCK) {
        int x, y, z;
        window_cell(w, k, x, y, z);
        const size_t i = ((size_t)x * ny + y) * nz + z;
        int in = 0;
        for (int b = 0; b < nboxes; ++b) {
            const int* r = boxes + 6 * b;
            in += (x >= r[0] && y >= r[1] && z >= r[2] && x <= r[3] && y <= r[4] && z <= r[5]) ? 1 : 0;
        }
        if (in) { occ[i] = 1; if (counts) counts[i] += in; }
    }
}

// OccupancyGrid::addPointsToField / removePointsFromField for a list of cells (occupancy_grid.cpp:357-406).  Without
// reference counts a cell simply becomes occupied / free.  With them every point counts (a cell listed twice is counted
// twice, as the reference's sequential loop does): the cell becomes occupied when its count leaves 0 and free when it
// returns to 0; a removal from a cell with count 0 does nothing.
__global__ void __launch_bounds__(FIELD_BLOCK)
k_occ_points(unsigned char* __restrict__ occ, int* __restrict__ counts, int nx, int ny, int nz, const int* __restrict__ cells, int n, int add)
{
    const int i = blockIdx.x * FIELD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int x = cells[3 * i], y = cells[3 * i + 1], z = cells[3 * i + 2];
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return;   // out of bounds: skipped (:366, :390; distance_map.hpp:312-316)
    const size_t c = ((size_t)x * ny + y) * nz + z;
    if (!counts) { occ[c] = (unsigned char)(add ? 1 : 0); return; }
    if (add) {
        if (atomicAdd(&counts[c], 1) == 0) occ[c] = 1;
    } else {
        int cur = counts[c];
        while (cur > 0) {
            const int seen = atomicCAS(&counts[c], cur, cur - 1);
            if (seen == cur) { if (cur == 1) occ[c] = 0; break; }
            cur = seen;
        }
    }
}

// OccupancyGrid::initRefCounts (occupancy_grid.cpp:424-441): 1 where the cell is an obstacle, else 0
__global__ void __launch_bounds__(FIELD_BLOCK)
k_counts_init(const unsigned char* __restrict__ occ, int* __restrict__ counts, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * FIELD_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * FIELD_BLOCK) counts[i] = occ[i] ? 1 : 0;
}

// pass 1: squared distance along z to the nearest occupied cell of the same column, or to the border cells z = -1 / nz
__global__ void __launch_bounds__(FIELD_BLOCK)
k_edt_z(const unsigned char* __restrict__ occ, unsigned short* __restrict__ g1, int nx, int ny, int nz, int dmax, Window w)
{
    const long long total = window_cells(w);
    for (long long k = (long long)blockIdx.x * FIELD_BLOCK + threadIdx.x; k < total; k += (long long)gridDim.x * FIELD_BLOCK) {
        int x, y, z;
        window_cell(w, k, x, y, z);
        const size_t i = ((size_t)x * ny + y) * nz + z;
        const unsigned char* col = occ + (i - z);
        int best = z + 1 < nz - z ? z + 1 : nz - z;       // the nearer border cell
        if (best > dmax) best = dmax;
        if (col[z]) best = 0;
        for (int d = 1; d < best; ++d) {
            const bool lo = z - d >= 0 && col[z - d], hi = z + d < nz && col[z + d];
            if (lo || hi) { best = d; break; }
        }
        g1[i] = (unsigned short)(best * best);
    }
}

// passes 2 and 3: lower envelope along one more axis.  in(c +- d along the axis) + d^2, the border cells beyond the
// ends of the axis being obstacles themselves (value 0 there).  stride = cells between neighbours along the axis.
template <bool Tiled>
__global__ void __launch_bounds__(FIELD_BLOCK)
k_edt_axis(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int nx, int ny, int nz, int axis, int dmax,
           int bricks_y, int bricks_z, Window w)
{
    const long long total = window_cells(w);
    const int cap = dmax * dmax;
    for (long long k = (long long)blockIdx.x * FIELD_BLOCK + threadIdx.x; k < total; k += (long long)gridDim.x * FIELD_BLOCK) {
        int x, y, z;
        window_cell(w, k, x, y, z);
        const size_t i = ((size_t)x * ny + y) * nz + z;
        const int pos = axis == 1 ? y : x, len = axis == 1 ? ny : nx;
        const size_t stride = axis == 1 ? (size_t)nz : (size_t)nz * ny;
        int best = in[i];
        if (best > cap) best = cap;
        for (int d = 1; d * d < best; ++d) {
            const int dd = d * d;
            // towards the low end: cell pos - d, or the border cell at -1
            int v;
            if (pos - d >= 0) v = dd + in[i - d * stride];
            else if (pos - d == -1) v = dd;
            else v = best;
            if (v < best) best = v;
            if (pos + d < len) v = dd + in[i + d * stride];
            else if (pos + d == len) v = dd;
            else v = best;
            if (v < best) best = v;
        }
        if (Tiled) {
            const size_t brick = ((size_t)(x >> 2) * bricks_y + (y >> 2)) * bricks_z + (z >> 2);
            out[brick * 64 + ((x & 3) << 4) + ((y & 3) << 2) + (z & 3)] = (unsigned short)best;
        } else {
            out[i] = (unsigned short)best;
        }
    }
}

__global__ void __launch_bounds__(FIELD_BLOCK)
k_untile(const unsigned short* __restrict__ tiled, int* __restrict__ out, int nx, int ny, int nz, int bricks_y, int bricks_z)
{
    const size_t total = (size_t)nx * ny * nz;
    for (size_t i = (size_t)blockIdx.x * FIELD_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * FIELD_BLOCK) {
        const int z = (int)(i % nz), y = (int)(i / nz % ny), x = (int)(i / ((size_t)nz * ny));
        const size_t brick = ((size_t)(x >> 2) * bricks_y + (y >> 2)) * bricks_z + (z >> 2);
        out[i] = (int)tiled[brick * 64 + ((x & 3) << 4) + ((y & 3) << 2) + (z & 3)];
    }
}

inline int blocks_of(size_t n) { const size_t b = (n + FIELD_BLOCK - 1) / FIELD_BLOCK; return (int)(b < 65536 ? b : 65536); }

}  // namespace

// engine.hip owns smplx_last_error(); this lets field.hip set its text
extern "C" int smplx_internal_set_error(int code, const char* msg);

#define FIELD_TRY(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return smplx_internal_set_error(SMPLX_E_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

namespace {

// the three passes on a window of cells (the whole grid: what a new field needs)
int field_update(smplx_grid* g, const Window& w)
{
    const int nx = g->n[0], ny = g->n[1], nz = g->n[2];
    const size_t total = (size_t)nx * ny * nz;
    unsigned short* t1 = g->d_tmp;
    unsigned short* t2 = g->d_tmp + total;
    const int nb = blocks_of((size_t)window_cells(w));
    hipLaunchKernelGGL(k_edt_z, dim3(nb), dim3(FIELD_BLOCK), 0, 0, g->d_occ, t1, nx, ny, nz, g->dmax_int, w);
    hipLaunchKernelGGL(k_edt_axis<false>, dim3(nb), dim3(FIELD_BLOCK), 0, 0, t1, t2, nx, ny, nz, 1, g->dmax_int, g->dev.bricks[1], g->dev.bricks[2], w);
    hipLaunchKernelGGL(k_edt_axis<true>, dim3(nb), dim3(FIELD_BLOCK), 0, 0, t2, (unsigned short*)g->d_d2, nx, ny, nz, 0, g->dmax_int, g->dev.bricks[1],
                       g->dev.bricks[2], w);
    FIELD_TRY(hipGetLastError());
    FIELD_TRY(hipDeviceSynchronize());
    g->last_window_cells = window_cells(w);
    ++g->epoch;
    return SMPLX_OK;
}

Window whole_grid(const smplx_grid* g) { return Window{0, 0, 0, g->n[0] - 1, g->n[1] - 1, g->n[2] - 1}; }

// the cells an edit of the cells inside `box` can change: the box grown by dmax along every axis (a value of pass k moves
// only where a value of pass k - 1 moved within dmax cells along that pass's axis); the whole grid when that is most of it
Window edit_window(const smplx_grid* g, const Window& box)
{
    const int d = g->dmax_int;
    Window w;
    w.x0 = std::max(0, box.x0 - d); w.y0 = std::max(0, box.y0 - d); w.z0 = std::max(0, box.z0 - d);
    w.x1 = std::min(g->n[0] - 1, box.x1 + d); w.y1 = std::min(g->n[1] - 1, box.y1 + d); w.z1 = std::min(g->n[2] - 1, box.z1 + d);
    if (2 * window_cells(w) > window_cells(whole_grid(g))) return whole_grid(g);
    return w;
}

// OccupancyGrid::worldToGrid (distance_map.hpp:520-527)
inline int world_to_cell(const smplx_grid* g, double w, int a) { return (int)(g->dev.inv_res * (w - g->dev.origin_minus_res[a]) + 0.5) - 1; }

inline bool finite_coords(const double* v, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(v[i] > -1.0e6 && v[i] < 1.0e6)) return false;      // also refuses NaN: the cast to a cell index would be undefined
    return true;
}

}  // namespace

extern "C" {

int smplx_grid_create_empty(const double origin[3], int nx, int ny, int nz, double res, double max_dist, smplx_grid** out)
{
    if (!origin || !out || nx <= 0 || ny <= 0 || nz <= 0 || !(res > 0.0)) return smplx_internal_set_error(SMPLX_E_ARG, "bad grid arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return smplx_internal_set_error(SMPLX_E_HIP, "no HIP device: the engine has no CPU path");
    smplx_grid* g = new smplx_grid;
    const double inv_res = 1.0 / res;
    g->dmax_int = (int)std::ceil(max_dist * inv_res);   // distance_map.hpp:126
    g->dmax_sqrd = g->dmax_int * g->dmax_int;
    if (g->dmax_sqrd > 65535) { delete g; return smplx_internal_set_error(SMPLX_E_LIMIT, "max_dist/res exceeds 255 cells (16-bit squared distances)"); }
    g->res = res; g->max_dist = max_dist;
    g->n[0] = nx; g->n[1] = ny; g->n[2] = nz;
    const int bx = (nx + 3) / 4, by = (ny + 3) / 4, bz = (nz + 3) / 4;
    const size_t total = (size_t)nx * ny * nz, tiled = (size_t)bx * by * bz * 64;
    hipError_t e = hipMalloc((void**)&g->d_d2, tiled * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemset(g->d_d2, 0, tiled * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_occ, total);
    if (e == hipSuccess) e = hipMemset(g->d_occ, 0, total);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_tmp, 2 * total * sizeof(uint16_t));
    if (e != hipSuccess) {
        const std::string msg = std::string("grid allocation: ") + hipGetErrorString(e);
        smplx_grid_destroy(g);
        return smplx_internal_set_error(SMPLX_E_HIP, msg.c_str());
    }
    for (int a = 0; a < 3; ++a) { g->origin[a] = origin[a]; g->dev.origin_minus_res[a] = origin[a] - res; g->dev.n[a] = g->n[a]; }
    g->dev.res = res; g->dev.inv_res = inv_res;
    g->dev.bricks[0] = bx; g->dev.bricks[1] = by; g->dev.bricks[2] = bz;
    g->dev.dmax_sqrd = g->dmax_sqrd; g->dev.pad = 0;
    g->dev.d2 = g->d_d2;
    if (int rc = field_update(g, whole_grid(g))) { smplx_grid_destroy(g); return rc; }   // the empty field: distance to the border cells
    *out = g;
    return SMPLX_OK;
}

int smplx_grid_set_ref_counted(smplx_grid* g, int on)
{
    if (!g || !g->d_occ) return smplx_internal_set_error(SMPLX_E_STATE, "the grid was created from a finished field (smplx_grid_create), not on the GPU");
    if (!on) {
        if (g->d_counts) (void)hipFree(g->d_counts);
        g->d_counts = nullptr;
        g->ref_counted = false;
        return SMPLX_OK;
    }
    const size_t total = (size_t)g->n[0] * g->n[1] * g->n[2];
    if (!g->d_counts) FIELD_TRY(hipMalloc((void**)&g->d_counts, sizeof(int32_t) * total));
    hipLaunchKernelGGL(k_counts_init, dim3(blocks_of(total)), dim3(FIELD_BLOCK), 0, 0, g->d_occ, g->d_counts, total);   // initRefCounts
    FIELD_TRY(hipGetLastError());
    FIELD_TRY(hipDeviceSynchronize());
    g->ref_counted = true;
    return SMPLX_OK;
}

static int change_cells(smplx_grid* g, const int* host, int n, int width, int add, bool counted)
{
    if (!g || !g->d_occ) return smplx_internal_set_error(SMPLX_E_STATE, "the grid was created from a finished field (smplx_grid_create), not on the GPU");
    if (n <= 0) return SMPLX_OK;
    // the cells that can change: inside the bounding box of the listed cells / boxes
    Window box{1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
    for (int i = 0; i < n; ++i) {
        const int* c = host + (size_t)i * width;
        const int hi[3] = {width == 6 ? c[3] : c[0], width == 6 ? c[4] : c[1], width == 6 ? c[5] : c[2]};
        if (width == 3 && (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= g->n[0] || c[1] >= g->n[1] || c[2] >= g->n[2])) continue;
        box.x0 = std::min(box.x0, c[0]); box.y0 = std::min(box.y0, c[1]); box.z0 = std::min(box.z0, c[2]);
        box.x1 = std::max(box.x1, hi[0]); box.y1 = std::max(box.y1, hi[1]); box.z1 = std::max(box.z1, hi[2]);
    }
    if (box.x1 < 0) return SMPLX_OK;      // nothing inside the grid
    int* d = nullptr;
    FIELD_TRY(hipMalloc((void**)&d, sizeof(int) * (size_t)n * width));
    hipError_t e = hipMemcpy(d, host, sizeof(int) * (size_t)n * width, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t total = (size_t)g->n[0] * g->n[1] * g->n[2];
        int* counts = counted && g->ref_counted ? g->d_counts : nullptr;
        (void)total;
        if (width == 6) hipLaunchKernelGGL(k_occ_boxes, dim3(blocks_of((size_t)window_cells(box))), dim3(FIELD_BLOCK), 0, 0, g->d_occ, counts, g->n[0], g->n[1], g->n[2], d, n, box);
        else hipLaunchKernelGGL(k_occ_points, dim3(blocks_of((size_t)n)), dim3(FIELD_BLOCK), 0, 0, g->d_occ, counts, g->n[0], g->n[1], g->n[2], d, n, add);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d);
    if (e != hipSuccess) return smplx_internal_set_error(SMPLX_E_HIP, (std::string("occupancy update: ") + hipGetErrorString(e)).c_str());
    return field_update(g, edit_window(g, box));
}

int smplx_grid_add_boxes(smplx_grid* g, const double* boxes, int n)
{
    if (!g || (!boxes && n > 0) || n < 0) return smplx_internal_set_error(SMPLX_E_ARG, "bad argument");
    if (n > 0 && !finite_coords(boxes, (size_t)n * 6)) return smplx_internal_set_error(SMPLX_E_ARG, "box coordinates must be finite (|x| < 1e6)");
    // an axis-aligned box occupies the cells from the cell of its low corner to the cell of its high corner (the rule the
    // test scenes are built with, smpl_amd/scenes.py box_cells; the reference voxelises a triangle mesh of the box,
    // smpl/src/geometry/voxelize.cpp:673-735, which needs Eigen: parity unpinned)
    std::vector<int> r((size_t)n * 6);
    int kept = 0;
    for (int b = 0; b < n; ++b) {
        const double* c = boxes + 6 * (size_t)b;
        int lo[3], hi[3];
        bool any = true;
        for (int a = 0; a < 3; ++a) {
            lo[a] = world_to_cell(g, c[a] - 0.5 * c[3 + a], a);
            hi[a] = world_to_cell(g, c[a] + 0.5 * c[3 + a], a);
            if (lo[a] < 0) lo[a] = 0;
            if (hi[a] > g->n[a] - 1) hi[a] = g->n[a] - 1;
            any = any && hi[a] >= lo[a];
        }
        if (!any) continue;
        for (int a = 0; a < 3; ++a) { r[6 * (size_t)kept + a] = lo[a]; r[6 * (size_t)kept + 3 + a] = hi[a]; }
        ++kept;
    }
    return change_cells(g, r.data(), kept, 6, 1, true);
}

static int points_to_cells(smplx_grid* g, const double* xyz, int n, std::vector<int>& c)
{
    if (!g || (!xyz && n > 0) || n < 0) return smplx_internal_set_error(SMPLX_E_ARG, "bad argument");
    if (n > 0 && !finite_coords(xyz, (size_t)n * 3)) return smplx_internal_set_error(SMPLX_E_ARG, "point coordinates must be finite (|x| < 1e6)");
    c.resize((size_t)n * 3);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) c[3 * (size_t)i + a] = world_to_cell(g, xyz[3 * (size_t)i + a], a);
    return SMPLX_OK;
}

int smplx_grid_add_points(smplx_grid* g, const double* xyz, int n)
{
    std::vector<int> c;
    if (int e = points_to_cells(g, xyz, n, c)) return e;
    return change_cells(g, c.data(), n, 3, 1, true);
}

int smplx_grid_remove_points(smplx_grid* g, const double* xyz, int n)
{
    std::vector<int> c;
    if (int e = points_to_cells(g, xyz, n, c)) return e;
    return change_cells(g, c.data(), n, 3, 0, true);
}

int smplx_grid_update_points(smplx_grid* g, const double* old_xyz, int n_old, const double* new_xyz, int n_new)
{
    // DistanceMap::updatePointsInMap (distance_map.hpp:367-435): as SETS of cells, remove old \ new, then add new \ old.
    // OccupancyGrid::updatePointsInField passes the points through without touching its reference counts ("TODO: ref
    // counting", occupancy_grid.cpp:408-415): neither does this.
    std::vector<int> co, cn;
    if (int e = points_to_cells(g, old_xyz, n_old, co)) return e;
    if (int e = points_to_cells(g, new_xyz, n_new, cn)) return e;
    auto as_set = [&](const std::vector<int>& c) {
        std::set<std::array<int, 3>> s;
        for (size_t i = 0; i + 2 < c.size(); i += 3) {
            if (c[i] < 0 || c[i + 1] < 0 || c[i + 2] < 0 || c[i] >= g->n[0] || c[i + 1] >= g->n[1] || c[i + 2] >= g->n[2]) continue;   // isCellValid
            s.insert({c[i], c[i + 1], c[i + 2]});
        }
        return s;
    };
    const std::set<std::array<int, 3>> so = as_set(co), sn = as_set(cn);
    std::vector<int> rem, add;
    for (const auto& c : so) if (!sn.count(c)) rem.insert(rem.end(), c.begin(), c.end());
    for (const auto& c : sn) if (!so.count(c)) add.insert(add.end(), c.begin(), c.end());
    if (int e = change_cells(g, rem.data(), (int)(rem.size() / 3), 3, 0, false)) return e;
    return change_cells(g, add.data(), (int)(add.size() / 3), 3, 1, false);
}

int smplx_grid_copy_counts(const smplx_grid* g, int32_t* counts)
{
    if (!g || !counts) return smplx_internal_set_error(SMPLX_E_ARG, "null argument");
    if (!g->ref_counted || !g->d_counts) return smplx_internal_set_error(SMPLX_E_STATE, "the grid keeps no reference counts");
    FIELD_TRY(hipMemcpy(counts, g->d_counts, sizeof(int32_t) * (size_t)g->n[0] * g->n[1] * g->n[2], hipMemcpyDeviceToHost));
    return SMPLX_OK;
}

long long smplx_grid_last_edit_cells(const smplx_grid* g) { return g ? g->last_window_cells : 0; }

int smplx_grid_copy_d2(const smplx_grid* g, int32_t* d2)
{
    if (!g || !d2) return smplx_internal_set_error(SMPLX_E_ARG, "null argument");
    const size_t total = (size_t)g->n[0] * g->n[1] * g->n[2];
    int* d = nullptr;
    FIELD_TRY(hipMalloc((void**)&d, sizeof(int) * total));
    hipLaunchKernelGGL(k_untile, dim3(blocks_of(total)), dim3(FIELD_BLOCK), 0, 0, g->d_d2, d, g->n[0], g->n[1], g->n[2], g->dev.bricks[1], g->dev.bricks[2]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(d2, d, sizeof(int) * total, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return smplx_internal_set_error(SMPLX_E_HIP, (std::string("field download: ") + hipGetErrorString(e)).c_str());
    return SMPLX_OK;
}

}  // extern "C"
