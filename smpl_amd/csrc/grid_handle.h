// smpl_amd/csrc/grid_handle.h -- the smplx_grid handle (include/smpl_amd.h), shared by engine.hip (lookup side) and
// field.hip (construction side)
#pragma once

#include <stdint.h>

#include "device_types.h"

struct smplx_grid {
    SmplxGridDev dev;
    uint16_t* d_d2 = nullptr;        // brick-tiled squared cell distances (what the kernels read)
    unsigned char* d_occ = nullptr;  // occupancy, x-major / z fastest (only grids built on the GPU: field.hip)
    uint16_t* d_tmp = nullptr;       // two intermediate passes of the distance transform (kept: an edit recomputes a window of them)
    int32_t* d_counts = nullptr;     // reference counts per cell (OccupancyGrid::m_counts), only while ref_counted
    bool ref_counted = false;
    uint64_t epoch = 0;              // bumped by every edit: spaces remember the epoch their caches belong to
    long long last_window_cells = 0; // cells the last edit recomputed (diagnostics)
    double origin[3];
    double res, max_dist;
    int n[3];
    int dmax_int, dmax_sqrd;
};
