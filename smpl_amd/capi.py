"""ctypes binding of smpl_amd/libsmpl_amd.so -- the C-ABI declared in include/smpl_amd.h.

Thin by design: numpy arrays in, numpy arrays out, one Python method per C entry point.  The
library is the product; this file only marshals pointers.  There is no CPU fallback: if the
shared object is missing or no GPU is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)

# every symbol include/smpl_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "smplx_last_error", "smplx_device_count", "smplx_shard_range", "smplx_grid_create", "smplx_grid_destroy", "smplx_model_create",
    "smplx_model_destroy", "smplx_model_counts", "smplx_model_joints", "smplx_model_nodes", "smplx_model_pairs",
    "smplx_space_create", "smplx_space_destroy", "smplx_space_num_vars", "smplx_space_num_prims",
    "smplx_space_discretization", "smplx_cc_state_valid_batch", "smplx_cc_edge_valid_batch", "smplx_cc_interpolate",
    "smplx_cc_sphere_positions", "smplx_set_goal_joint", "smplx_set_goal_xyz", "smplx_goal_pose",
    "smplx_heuristic_batch", "smplx_bfs_size", "smplx_bfs_copy", "smplx_bfs_levels", "smplx_expand_batch",
    "smplx_expand_work_bytes", "smplx_expand_batch_device", "smplx_set_start", "smplx_start_id", "smplx_goal_id",
    "smplx_get_succs", "smplx_hint_frontier", "smplx_get_goal_heuristic", "smplx_num_states", "smplx_get_state",
    "smplx_plan", "smplx_expansion_log_size", "smplx_expansion_log", "smplx_extract_path", "smplx_post_process_path", "smplx_space_specialized", "smplx_model_const_header", "smplx_profile_begin",
    "smplx_profile_end", "smplx_counters_bytes", "smplx_counters_read", "smplx_plan_multi",
    "smplx_bfs_metric_goal_distance", "smplx_bfs_metric_start_distance", "smplx_space_status", "smplx_space_clear_status",
    "smplx_check_joint_limits", "smplx_cc_state_valid_batch_device", "smplx_space_counters",
    "smplx_table_sync", "smplx_compact_rec_b_bytes", "smplx_compact_blocks", "smplx_expand_batch_k5_device", "smplx_expand_batch_k5",
    "smplx_compact_totals_len", "smplx_compact_capacity",
    "smplx_grid_create_empty", "smplx_grid_add_boxes", "smplx_grid_add_points", "smplx_grid_remove_points", "smplx_grid_copy_d2",
    "smplx_search_counters", "smplx_grid_set_ref_counted", "smplx_grid_update_points", "smplx_grid_copy_counts", "smplx_grid_last_edit_cells",
]


class Params(C.Structure):
    _fields_ = [("resolutions", C.c_double * 16), ("bfs_inflation_radius", C.c_double), ("cost_per_cell", C.c_int32),
                ("use_short_dist_mprims", C.c_int32), ("short_dist_mprims_thresh", C.c_double),
                ("use_xyzrpy_snap_mprim", C.c_int32), ("xyzrpy_snap_dist_thresh", C.c_double),
                ("xy_rotate_by_var3", C.c_int32), ("use_long_and_short", C.c_int32), ("padding", C.c_double),
                ("batch_states", C.c_int32), ("flags", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [("initial_eps", C.c_double), ("final_eps", C.c_double), ("delta_eps", C.c_double),
                ("improve", C.c_int32), ("bounded", C.c_int32), ("max_expansions_init", C.c_int32),
                ("max_expansions", C.c_int32)]


class SearchStats(C.Structure):
    _fields_ = [("solved", C.c_int32), ("path_len", C.c_int32), ("cost", C.c_int32), ("expansions", C.c_int32),
                ("expansions_init", C.c_int32), ("satisfied_eps", C.c_double), ("seconds", C.c_double),
                ("gpu_succ_evals", C.c_int64), ("committed_succ_evals", C.c_int64), ("gpu_batches", C.c_int64),
                ("cache_hits", C.c_int64), ("cache_misses", C.c_int64), ("grid_lookups", C.c_int64)]


_lib = None


def lib():
    """Load (building in-tree if needed) the shared library."""
    global _lib
    if _lib is None:
        path = os.environ.get("SMPL_AMD_LIB", _build.LIB)   # override: kernel experiments (tools/)
        if not os.path.exists(path):
            _build.build()
        L = C.CDLL(path)
        L.smplx_last_error.restype = C.c_char_p
        L.smplx_bfs_size.restype = C.c_int64
        L.smplx_expand_work_bytes.restype = C.c_size_t
        L.smplx_grid_create.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _ip, C.POINTER(C.c_void_p)]
        L.smplx_model_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.smplx_space_create.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(Params), C.POINTER(C.c_void_p)]
        for name in ["smplx_grid_destroy", "smplx_model_destroy", "smplx_space_destroy"]:
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.smplx_expand_batch_device.argtypes = [C.c_void_p] + [C.c_void_p, C.c_int] + [C.c_void_p] * 9
        L.smplx_expand_work_bytes.argtypes = [C.c_void_p, C.c_int]
        L.smplx_counters_bytes.restype = C.c_size_t
        L.smplx_counters_bytes.argtypes = [C.c_void_p, C.c_int]
        L.smplx_counters_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _u64p]
        L.smplx_bfs_metric_goal_distance.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.smplx_bfs_metric_start_distance.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.smplx_space_clear_status.argtypes = [C.c_void_p]
        L.smplx_space_clear_status.restype = None
        _lib = L
    return _lib


class SmplxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"smplx error {code}: {msg}")
        self.code = code


def _chk(code):
    if code != 0:
        raise SmplxError(code, lib().smplx_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(t)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Grid:
    """OccupancyGrid, lookup side, resident in HBM (16-bit squared distances in 4x4x4 bricks)."""

    def __init__(self, origin, dims, res, max_dist, d2):
        self.h = C.c_void_p()
        o = _f64(origin)
        d2 = np.ascontiguousarray(d2, dtype=np.int32)
        assert d2.shape == tuple(dims)
        _chk(lib().smplx_grid_create(_p(o, _dp), dims[0], dims[1], dims[2], res, max_dist, _p(d2, _ip), C.byref(self.h)))
        self.dims = tuple(dims)

    @classmethod
    def empty(cls, origin, dims, res, max_dist):
        """A grid built and kept on the GPU (row N1): starts empty (only the border cells count as obstacles)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        o = _f64(origin)
        lib().smplx_grid_create_empty.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
        _chk(lib().smplx_grid_create_empty(_p(o, _dp), dims[0], dims[1], dims[2], res, max_dist, C.byref(self.h)))
        self.dims = tuple(dims)
        return self

    @classmethod
    def from_boxes(cls, origin, dims, res, max_dist, boxes):
        self = cls.empty(origin, dims, res, max_dist)
        self.add_boxes(boxes)
        return self

    def add_boxes(self, boxes):
        b = _f64([list(c) + list(s) for (c, s) in boxes]).reshape(-1, 6)
        lib().smplx_grid_add_boxes.argtypes = [C.c_void_p, _dp, C.c_int]
        _chk(lib().smplx_grid_add_boxes(self.h, _p(b, _dp), b.shape[0]))

    def add_points(self, xyz):
        p = _f64(xyz).reshape(-1, 3)
        lib().smplx_grid_add_points.argtypes = [C.c_void_p, _dp, C.c_int]
        _chk(lib().smplx_grid_add_points(self.h, _p(p, _dp), p.shape[0]))

    def remove_points(self, xyz):
        p = _f64(xyz).reshape(-1, 3)
        lib().smplx_grid_remove_points.argtypes = [C.c_void_p, _dp, C.c_int]
        _chk(lib().smplx_grid_remove_points(self.h, _p(p, _dp), p.shape[0]))

    def set_ref_counted(self, on=True):
        lib().smplx_grid_set_ref_counted.argtypes = [C.c_void_p, C.c_int]
        _chk(lib().smplx_grid_set_ref_counted(self.h, int(on)))

    def update_points(self, old_xyz, new_xyz):
        a = _f64(old_xyz).reshape(-1, 3); b = _f64(new_xyz).reshape(-1, 3)
        lib().smplx_grid_update_points.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_int]
        _chk(lib().smplx_grid_update_points(self.h, _p(a, _dp), a.shape[0], _p(b, _dp), b.shape[0]))

    def counts(self):
        out = np.zeros(self.dims, np.int32)
        lib().smplx_grid_copy_counts.argtypes = [C.c_void_p, _ip]
        _chk(lib().smplx_grid_copy_counts(self.h, _p(out, _ip)))
        return out

    def last_edit_cells(self):
        lib().smplx_grid_last_edit_cells.restype = C.c_longlong
        lib().smplx_grid_last_edit_cells.argtypes = [C.c_void_p]
        return int(lib().smplx_grid_last_edit_cells(self.h))

    def d2(self):
        out = np.zeros(self.dims, np.int32)
        lib().smplx_grid_copy_d2.argtypes = [C.c_void_p, _ip]
        _chk(lib().smplx_grid_copy_d2(self.h, _p(out, _ip)))
        return out

    def close(self):
        if self.h:
            lib().smplx_grid_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Compiled robot model (host-side compile; no GPU needed)."""

    def __init__(self, robot_text: str):
        self.h = C.c_void_p()
        _chk(lib().smplx_model_create(robot_text.encode(), C.byref(self.h)))
        c = [C.c_int() for _ in range(6)]
        _chk(lib().smplx_model_counts(self.h, *[C.byref(x) for x in c]))
        self.njoints, self.nvars, self.ntrees, self.nnodes, self.npairs, self.nslots = [x.value for x in c]

    def arrays(self):
        L = lib()
        origins = np.zeros((self.njoints, 12)); k = np.zeros(self.njoints); fi = np.zeros(self.njoints, np.int32)
        _chk(L.smplx_model_joints(self.h, _p(origins, _dp), _p(k, _dp), _p(fi, _ip)))
        xyzr = np.zeros((self.nnodes, 4)); left = np.zeros(self.nnodes, np.int32); right = np.zeros(self.nnodes, np.int32)
        first = np.zeros(self.ntrees + 1, np.int32)
        _chk(L.smplx_model_nodes(self.h, _p(xyzr, _dp), _p(left, _ip), _p(right, _ip), _p(first, _ip)))
        pairs = np.zeros((self.npairs, 2), np.int32)
        if self.npairs:
            _chk(L.smplx_model_pairs(self.h, _p(pairs, _ip)))
        return dict(origins=origins, k=k, file_index=fi, xyzr=xyzr, left=left, right=right, tree_first=first, pairs=pairs)

    def close(self):
        if self.h:
            lib().smplx_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Space:
    """ManipLattice + BfsHeuristic + CollisionSpace for one query, on one GPU."""

    def __init__(self, model: Model, grid: Grid, mprim_text: str, params, batch_states: int = 0, fused: bool = False,
                 tiny_work_list: bool = False, no_small_kernel: bool = False, generic_kernels: bool = False,
                 padding: float = 0.0):
        self.model, self.grid = model, grid
        P = Params()
        for i, r in enumerate(params.resolutions):
            P.resolutions[i] = r
        P.bfs_inflation_radius = params.bfs_radius
        P.cost_per_cell = params.cost_per_cell
        P.use_short_dist_mprims = int(params.use_short)
        P.short_dist_mprims_thresh = params.short_thresh
        P.use_xyzrpy_snap_mprim = int(params.use_xyzrpy_snap)
        P.xyzrpy_snap_dist_thresh = params.xyzrpy_thresh
        P.xy_rotate_by_var3 = int(params.xy_rotate_by_var3)
        P.use_long_and_short = int(params.use_long_and_short)
        P.padding = padding
        P.batch_states = batch_states
        P.flags = (1 if fused else 0) | (4 if no_small_kernel else 0) | (8 if generic_kernels else 0)
        self.h = C.c_void_p()
        _chk(lib().smplx_space_create(model.h, grid.h, mprim_text.encode(), C.byref(P), C.byref(self.h)))
        if tiny_work_list:   # test hook (smpl_amd/csrc/test_hooks.h; not part of include/smpl_amd.h)
            lib().smplx_test_set_work_list_items.argtypes = [C.c_void_p, C.c_int]
            _chk(lib().smplx_test_set_work_list_items(self.h, 8 * 16))
        self.N = lib().smplx_space_num_vars(self.h)
        self.M = lib().smplx_space_num_prims(self.h)

    @classmethod
    def from_config(cls, cfg, batch_states: int = 0, xy_rotate=None, fused: bool = False, tiny_work_list: bool = False,
                    no_small_kernel: bool = False, generic_kernels: bool = False, padding: float = 0.0):
        g = Grid(cfg.grid.origin, cfg.grid.dims, cfg.grid.res, cfg.grid.max_dist, cfg.grid.d2)
        m = Model(cfg.robot_text)
        p = cfg.params
        if xy_rotate is not None:
            import copy
            p = copy.copy(p)
            p.xy_rotate_by_var3 = xy_rotate
        return cls(m, g, cfg.mprim, p, batch_states, fused, tiny_work_list, no_small_kernel, generic_kernels, padding)

    def specialized(self):
        """(True/False, note): whether the space runs the per-robot kernel build (smplx_space_specialized)."""
        buf = C.create_string_buffer(4096)
        lib().smplx_space_specialized.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        r = lib().smplx_space_specialized(self.h, buf, 4096)
        return bool(r), buf.value.decode(errors="replace")

    def close(self):
        if self.h:
            lib().smplx_space_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def discretization(self):
        v = np.zeros(self.N, np.int32); d = np.zeros(self.N)
        _chk(lib().smplx_space_discretization(self.h, _p(v, _ip), _p(d, _dp)))
        return v, d

    def check_joint_limits(self, q):
        q = _f64(q).reshape(-1, self.N); out = np.zeros(q.shape[0], np.uint8)
        lib().smplx_check_joint_limits.argtypes = [C.c_void_p, _dp, C.c_int, _up]
        _chk(lib().smplx_check_joint_limits(self.h, _p(q, _dp), q.shape[0], _p(out, _up)))
        return out

    # ---- CollisionChecker ----
    def state_valid_batch(self, q):
        q = _f64(q).reshape(-1, self.N); n = q.shape[0]
        out = np.zeros(n, np.uint8); lk = np.zeros(n, np.int32)
        _chk(lib().smplx_cc_state_valid_batch(self.h, _p(q, _dp), n, _p(out, _up), _p(lk, _ip)))
        return out, lk

    def state_valid_batch_device(self, d_q, n, d_valid, d_lookups, stream):
        """Raw device pointers (ints); launches on `stream`, does not synchronise."""
        lib().smplx_cc_state_valid_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(lib().smplx_cc_state_valid_batch_device(self.h, d_q, n, d_valid, d_lookups, stream))

    def edge_valid_batch(self, a, b):
        a = _f64(a).reshape(-1, self.N); b = _f64(b).reshape(-1, self.N); n = a.shape[0]
        out = np.zeros(n, np.uint8); lk = np.zeros(n, np.int32); w = np.zeros(n, np.int32)
        _chk(lib().smplx_cc_edge_valid_batch(self.h, _p(a, _dp), _p(b, _dp), n, _p(out, _up), _p(lk, _ip), _p(w, _ip)))
        return out, lk, w

    def interpolate(self, a, b, cap=4096):
        a = _f64(a); b = _f64(b); out = np.zeros((cap, self.N)); n = C.c_int()
        _chk(lib().smplx_cc_interpolate(self.h, _p(a, _dp), _p(b, _dp), _p(out, _dp), cap, C.byref(n)))
        return out[:min(n.value, cap)].copy(), n.value

    def sphere_positions(self, q):
        q = _f64(q).reshape(-1, self.N); n = q.shape[0]
        out = np.zeros((n, self.model.nnodes, 3))
        _chk(lib().smplx_cc_sphere_positions(self.h, _p(q, _dp), n, _p(out, _dp)))
        return out

    # ---- heuristic ----
    def set_goal_joint(self, angles, tol):
        a = _f64(angles); t = _f64(tol)
        _chk(lib().smplx_set_goal_joint(self.h, _p(a, _dp), _p(t, _dp)))

    def set_goal_xyz(self, xyz, tol):
        a = _f64(xyz); t = _f64(tol)
        _chk(lib().smplx_set_goal_xyz(self.h, _p(a, _dp), _p(t, _dp)))

    def goal_pose(self):
        x = np.zeros(3)
        _chk(lib().smplx_goal_pose(self.h, _p(x, _dp)))
        return x

    def heuristic_batch(self, q):
        q = _f64(q).reshape(-1, self.N); n = q.shape[0]
        h = np.zeros(n, np.int32); xyz = np.zeros((n, 3))
        _chk(lib().smplx_heuristic_batch(self.h, _p(q, _dp), n, _p(h, _ip), _p(xyz, _dp)))
        return h, xyz

    def metric_goal_distance(self, xyz):
        xyz = _f64(xyz).reshape(-1, 3); out = np.zeros(xyz.shape[0])
        _chk(lib().smplx_bfs_metric_goal_distance(self.h, _p(xyz, _dp), xyz.shape[0], _p(out, _dp)))
        return out

    def metric_start_distance(self, xyz):
        xyz = _f64(xyz).reshape(-1, 3); out = np.zeros(xyz.shape[0])
        _chk(lib().smplx_bfs_metric_start_distance(self.h, _p(xyz, _dp), xyz.shape[0], _p(out, _dp)))
        return out

    def status(self):
        buf = C.create_string_buffer(1024)
        lib().smplx_space_status.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        return lib().smplx_space_status(self.h, buf, 1024), buf.value.decode(errors="replace")

    def bfs_grid(self):
        n = lib().smplx_bfs_size(self.h)
        out = np.zeros(n, np.int32)
        _chk(lib().smplx_bfs_copy(self.h, _p(out, _ip)))
        g = self.grid.dims
        return out.reshape(g[2] + 2, g[1] + 2, g[0] + 2)

    def bfs_levels(self):
        return lib().smplx_bfs_levels(self.h)

    # ---- lattice ----
    def expand_batch(self, q):
        q = _f64(q).reshape(-1, self.N); B = q.shape[0]; M, N = self.M, self.N
        flags = np.zeros((B, M), np.uint8); coord = np.zeros((B, M, N), np.int32); sq = np.zeros((B, M, N))
        h = np.zeros((B, M), np.int32); cost = np.zeros((B, M), np.int32); lk = np.zeros((B, M), np.int32)
        _chk(lib().smplx_expand_batch(self.h, _p(q, _dp), B, _p(flags, _up), _p(coord, _ip), _p(sq, _dp), _p(h, _ip),
                                      _p(cost, _ip), _p(lk, _ip)))
        return dict(flags=flags, coord=coord, q=sq, h=h, cost=cost, lookups=lk)

    def table_sync(self):
        _chk(lib().smplx_table_sync(self.h))

    def compact_rec_b_bytes(self):
        lib().smplx_compact_rec_b_bytes.restype = C.c_size_t
        return lib().smplx_compact_rec_b_bytes(self.h)

    def compact_blocks(self, B):
        return lib().smplx_compact_blocks(self.h, B)

    def expand_batch_k5(self, q, cap_a=None, cap_b=None):
        """Dense outputs + device-table ids + the compacted successor stream (decoded into per-state lists)."""
        q = _f64(q).reshape(-1, self.N); B = q.shape[0]; M, N = self.M, self.N
        cap_a = lib().smplx_compact_capacity(self.h, B) if cap_a is None else cap_a
        cap_b = lib().smplx_compact_capacity(self.h, B) if cap_b is None else cap_b
        flags = np.zeros((B, M), np.uint8); coord = np.zeros((B, M, N), np.int32); sq = np.zeros((B, M, N))
        h = np.zeros((B, M), np.int32); sid = np.zeros((B, M), np.int32)
        rb = self.compact_rec_b_bytes(); nb = self.compact_blocks(B)
        rec_a = np.zeros((cap_a, 2), np.int32); rec_b = np.zeros((cap_b, rb), np.uint8)
        bt = np.zeros((nb, 4), np.int32); tot = np.zeros(3, np.int32)
        lib().smplx_expand_batch_k5.argtypes = [C.c_void_p, _dp, C.c_int, _up, _ip, _dp, _ip, _ip, _ip, C.c_int, C.c_void_p, C.c_int,
                                                _ip, _ip]
        _chk(lib().smplx_expand_batch_k5(self.h, _p(q, _dp), B, _p(flags, _up), _p(coord, _ip), _p(sq, _dp), _p(h, _ip), _p(sid, _ip),
                                         _p(rec_a, _ip), cap_a, rec_b.ctypes.data_as(C.c_void_p), cap_b, _p(bt, _ip), _p(tot, _ip)))
        return dict(flags=flags, coord=coord, q=sq, h=h, succ_id=sid, rec_a=rec_a, rec_b=rec_b, block_tab=bt, totals=tot)

    def expand_batch_k5_device(self, d_q, B, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups, d_id, d_rec_a, cap_a, d_rec_b, cap_b,
                               d_block_tab, d_totals, d_work, d_counters, stream):
        lib().smplx_expand_batch_k5_device.argtypes = ([C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_int] +
                                                       [C.c_void_p] * 5)
        _chk(lib().smplx_expand_batch_k5_device(self.h, d_q, B, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups, d_id, d_rec_a, cap_a,
                                                d_rec_b, cap_b, d_block_tab, d_totals, d_work, d_counters, stream))

    def expand_work_bytes(self, B):
        return lib().smplx_expand_work_bytes(self.h, B)

    def counters_bytes(self, B):
        return lib().smplx_counters_bytes(self.h, B)

    def counters_read(self, d_counters, B):
        out = np.zeros(6, np.uint64)
        _chk(lib().smplx_counters_read(self.h, d_counters, B, _p(out, _u64p)))
        return [int(x) for x in out]

    def expand_batch_device(self, d_q, B, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups, d_work, d_counters, stream):
        """All arguments are raw device pointers (ints); launches on `stream`, does not synchronise."""
        _chk(lib().smplx_expand_batch_device(self.h, d_q, B, d_flags, d_coord, d_sq, d_h, d_cost, d_lookups, d_work,
                                             d_counters, stream))

    def profile_begin(self, max_launches):
        _chk(lib().smplx_profile_begin(self.h, max_launches))

    def profile_end(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        _chk(lib().smplx_profile_end(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def set_start(self, q):
        q = _f64(q); i = C.c_int()
        _chk(lib().smplx_set_start(self.h, _p(q, _dp), C.byref(i)))
        return i.value

    def get_succs(self, i):
        s = np.zeros(self.M, np.int32); k = np.zeros(self.M, np.int32); n = C.c_int()
        _chk(lib().smplx_get_succs(self.h, i, _p(s, _ip), _p(k, _ip), self.M, C.byref(n)))
        return s[:n.value].copy(), k[:n.value].copy()

    def hint_frontier(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        _chk(lib().smplx_hint_frontier(self.h, _p(ids, _ip), ids.shape[0]))

    def goal_heuristic(self, i):
        h = C.c_int32()
        _chk(lib().smplx_get_goal_heuristic(self.h, i, C.byref(h)))
        return h.value

    def num_states(self):
        return lib().smplx_num_states(self.h)

    def get_state(self, i):
        q = np.zeros(self.N); c = np.zeros(self.N, np.int32)
        _chk(lib().smplx_get_state(self.h, i, _p(q, _dp), _p(c, _ip)))
        return q, c

    def search_counters(self):
        """Diagnostics of the device-resident search (smplx_search_counters)."""
        out = (C.c_int64 * 16)()
        lib().smplx_search_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _chk(lib().smplx_search_counters(self.h, out))
        names = ["searches", "grows", "dup_pushes", "t_idle", "t_select_pop", "t_evaluate", "t_commit", "t_relax", "t_reorder", "t_launch_io",
                 "device_states", "heap_cache_entries", "spec_rounds", "spec_hits"]
        return {n: int(out[i]) for i, n in enumerate(names)}

    def set_search_helper(self, on):
        """Test hook (csrc/test_hooks.h): run the device search with (default) or without its helper wave."""
        lib().smplx_test_set_search_helper.argtypes = [C.c_void_p, C.c_int]
        _chk(lib().smplx_test_set_search_helper(self.h, 1 if on else 0))

    def set_search_capacity(self, states):
        """Test hook (csrc/test_hooks.h): first capacity of the device search's buffers."""
        lib().smplx_test_set_search_capacity.argtypes = [C.c_void_p, C.c_int]
        _chk(lib().smplx_test_set_search_capacity(self.h, int(states)))

    # ---- search ----
    def plan(self, eps0, eps_final, eps_delta, improve=True, bounded=False, max_init=0, max_rep=0, cap=100000):
        P = SearchParams(eps0, eps_final, eps_delta, int(improve), int(bounded), max_init, max_rep)
        S = SearchStats()
        ids = np.zeros(cap, np.int32)
        _chk(lib().smplx_plan(self.h, C.byref(P), _p(ids, _ip), cap, C.byref(S)))
        n = lib().smplx_expansion_log_size(self.h)
        log = np.zeros(n, np.int32)
        if n:
            _chk(lib().smplx_expansion_log(self.h, _p(log, _ip)))
        out = {f: getattr(S, f) for f, _ in SearchStats._fields_}
        out["path"] = ids[:S.path_len].copy()
        out["expansion_log"] = log
        return out

    @staticmethod
    def plan_multi(spaces, eps0, eps_final, eps_delta, improve=True, bounded=False, max_init=0, max_rep=0, cap=4096,
                   host_threads=1):
        """Interleaved ARA* over independent queries on one GPU (smplx_plan_multi)."""
        nq = len(spaces)
        P = SearchParams(eps0, eps_final, eps_delta, int(improve), int(bounded), max_init, max_rep)
        St = (SearchStats * nq)()
        H = (C.c_void_p * nq)(*[sp.h for sp in spaces])
        ids = np.zeros((nq, cap), np.int32)
        wall = C.c_double()
        _chk(lib().smplx_plan_multi(H, nq, C.byref(P), _p(ids, _ip), cap, St, C.byref(wall), int(host_threads)))
        out = []
        for q, sp in enumerate(spaces):
            d = {f: getattr(St[q], f) for f, _ in SearchStats._fields_}
            d["path"] = ids[q, :St[q].path_len].copy()
            n = lib().smplx_expansion_log_size(sp.h)
            log = np.zeros(n, np.int32)
            if n:
                _chk(lib().smplx_expansion_log(sp.h, _p(log, _ip)))
            d["expansion_log"] = log
            out.append(d)
        return out, wall.value

    def post_process_path(self, path, shortcut=True, interpolate=True, upstream_limits=False):
        """PlannerInterface::postProcessPath (planner_interface.cpp:2651-2697).  Returns (path, stats)."""
        path = np.ascontiguousarray(path, np.float64).reshape(-1, self.N)
        flags = (1 if shortcut else 0) | (2 if interpolate else 0) | (4 if upstream_limits else 0)
        n = C.c_int(); st = (C.c_int64 * 2)()
        lib().smplx_post_process_path.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, _ip, C.POINTER(C.c_int64)]
        _chk(lib().smplx_post_process_path(self.h, _p(path, _dp), path.shape[0], flags, None, 0, C.byref(n), st))
        out = np.zeros((max(n.value, 1), self.N))
        _chk(lib().smplx_post_process_path(self.h, _p(path, _dp), path.shape[0], flags, _p(out, _dp), out.shape[0], C.byref(n), st))
        return out[:n.value], dict(edge_batches=st[0], configs=st[1])

    def extract_path(self, ids):
        ids = np.ascontiguousarray(ids, np.int32); q = np.zeros((ids.shape[0], self.N))
        _chk(lib().smplx_extract_path(self.h, _p(ids, _ip), ids.shape[0], _p(q, _dp)))
        return q
