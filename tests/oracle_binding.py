"""ctypes binding of oracle/liboracle.so (the CPU checker).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build_oracle():
    if not (os.path.exists(os.path.join(ORACLE_DIR, "liboracle.so")) and
            os.path.getmtime(os.path.join(ORACLE_DIR, "liboracle.so")) >=
            max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("smpl_oracle.hpp", "oracle_capi.cpp"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so", "liboracle_libm.so"],
                              stdout=subprocess.DEVNULL)
        subprocess.call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_ubyte)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """One planning context of the CPU restatement (robot + grid + heuristic + lattice + ARA*)."""

    def __init__(self, cfg, libm: bool = False, xy_rotate=None):
        build_oracle()
        self.lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle_libm.so" if libm else "liboracle.so"))
        L = self.lib
        L.orc_create.restype = C.c_void_p
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_grid_sqdist.restype = C.c_double
        L.orc_grid_sqdist.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_metric_goal_distance.restype = C.c_double
        L.orc_metric_goal_distance.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.orc_normalize_angle.restype = C.c_double
        L.orc_normalize_angle.argtypes = [C.c_double]
        L.orc_bfs_size.restype = C.c_long
        L.orc_total_lookups.restype = C.c_long
        L.orc_world_to_grid.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, _ip]
        L.orc_sincos.argtypes = [C.c_double, _dp, _dp]
        L.orc_search_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        g, p = cfg.grid, cfg.params
        self.cfg = cfg
        d2 = np.ascontiguousarray(g.d2, dtype=np.int32)
        origin = np.asarray(g.origin, dtype=np.float64)
        res = np.asarray(p.resolutions, dtype=np.float64)
        xyrot = p.xy_rotate_by_var3 if xy_rotate is None else xy_rotate
        L.orc_create.argtypes = [C.c_char_p, C.c_char_p, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _ip,
                                 _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                 C.c_int]
        self.h = C.c_void_p(L.orc_create(cfg.robot_text.encode(), cfg.mprim.encode(), _p(origin, _dp),
                                         g.dims[0], g.dims[1], g.dims[2], g.res, g.max_dist, _p(d2, _ip),
                                         _p(res, _dp), len(res), p.bfs_radius, p.cost_per_cell, int(p.use_short),
                                         p.short_thresh, int(p.use_xyzrpy_snap), p.xyzrpy_thresh, int(xyrot),
                                         int(p.use_long_and_short)))
        err = L.orc_last_error(self.h)
        if err:
            raise RuntimeError("oracle: " + err.decode())
        self.N = L.orc_num_vars(self.h)
        self.M = L.orc_num_prims(self.h)
        L.orc_search_params(self.h, p.eps0, p.eps_final, p.eps_delta, 1, 0, 0, 0)

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    # --- model ---
    def model(self):
        L = self.lib
        nj, nt, nn, npairs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.orc_model_counts(self.h, C.byref(nj), C.byref(nt), C.byref(nn), C.byref(npairs))
        origins = np.zeros((nj.value, 12)); k = np.zeros(nj.value)
        L.orc_model_joints(self.h, _p(origins, _dp), _p(k, _dp))
        xyzr = np.zeros((nn.value, 4)); left = np.zeros(nn.value, np.int32); right = np.zeros(nn.value, np.int32)
        link = np.zeros(nn.value, np.int32); first = np.zeros(nt.value + 1, np.int32)
        L.orc_model_nodes(self.h, _p(xyzr, _dp), _p(left, _ip), _p(right, _ip), _p(link, _ip), _p(first, _ip))
        pairs = np.zeros((npairs.value, 2), np.int32)
        L.orc_model_pairs(self.h, _p(pairs, _ip))
        vals = np.zeros(self.N, np.int32); deltas = np.zeros(self.N)
        L.orc_discretization(self.h, _p(vals, _ip), _p(deltas, _dp))
        return dict(origins=origins, k=k, xyzr=xyzr, left=left, right=right, link=link, tree_first=first,
                    pairs=pairs, coord_vals=vals, coord_deltas=deltas)

    def set_padding(self, padding: float):
        self.lib.orc_set_padding.argtypes = [C.c_void_p, C.c_double]
        self.lib.orc_set_padding(self.h, padding)

    def use_aos_cells(self, on: bool = True):
        """Timing variant: the reference's 48-byte AoS distance cells over the padded grid (same values)."""
        self.lib.orc_use_aos_cells.argtypes = [C.c_void_p, C.c_int]
        self.lib.orc_use_aos_cells(self.h, int(on))

    def set_order(self, chain: bool):
        self.lib.orc_set_traversal_order(self.h, 1 if chain else 0)

    # --- primitives ---
    def sincos(self, x):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos(x, C.byref(s), C.byref(c))
        return s.value, c.value

    def state_to_coord(self, q):
        q = np.ascontiguousarray(q, np.float64); c = np.zeros(self.N, np.int32)
        self.lib.orc_state_to_coord(self.h, _p(q, _dp), _p(c, _ip))
        return c

    def check_joint_limits(self, q):
        q = np.ascontiguousarray(q, np.float64)
        return bool(self.lib.orc_check_joint_limits(self.h, _p(q, _dp)))

    def planning_fk(self, q):
        q = np.ascontiguousarray(q, np.float64); x = np.zeros(3)
        self.lib.orc_planning_fk(self.h, _p(q, _dp), _p(x, _dp))
        return x

    def sphere_positions(self, q, nnodes):
        q = np.ascontiguousarray(q, np.float64); x = np.zeros((nnodes, 3))
        self.lib.orc_sphere_positions(self.h, _p(q, _dp), _p(x, _dp))
        return x

    def grid_sqdist(self, x, y, z):
        return self.lib.orc_grid_sqdist(self.h, x, y, z)

    def world_to_grid(self, x, y, z):
        c = np.zeros(3, np.int32)
        self.lib.orc_world_to_grid(self.h, x, y, z, _p(c, _ip))
        return c

    def state_valid(self, q):
        q = np.ascontiguousarray(q, np.float64); l = C.c_int()
        ok = self.lib.orc_state_valid(self.h, _p(q, _dp), C.byref(l))
        return bool(ok), l.value

    def state_valid_batch_timed(self, Q):
        Q = np.ascontiguousarray(Q, np.float64).reshape(-1, self.N)
        n = Q.shape[0]; out = np.zeros(n, np.uint8); lk = np.zeros(n, np.int32); sec = C.c_double()
        self.lib.orc_state_valid_batch_timed.argtypes = [C.c_void_p, _dp, C.c_int, _up, _ip, _dp]
        self.lib.orc_state_valid_batch_timed(self.h, _p(Q, _dp), n, _p(out, _up), _p(lk, _ip), C.byref(sec))
        return out, lk, sec.value

    def waypoint_count(self, a, b):
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
        return self.lib.orc_waypoint_count(self.h, _p(a, _dp), _p(b, _dp))

    def edge_valid(self, a, b):
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64); l = C.c_int()
        ok = self.lib.orc_edge_valid(self.h, _p(a, _dp), _p(b, _dp), C.byref(l))
        return bool(ok), l.value

    def edge_valid_batch(self, A, B):
        A = np.ascontiguousarray(A, np.float64); B = np.ascontiguousarray(B, np.float64)
        n = A.shape[0]; out = np.zeros(n, np.uint8); lk = np.zeros(n, np.int32)
        self.lib.orc_edge_valid_batch(self.h, _p(A, _dp), _p(B, _dp), n, _p(out, _up), _p(lk, _ip))
        return out, lk

    # --- heuristic ---
    def set_goal_joint(self, angles, tol):
        a = np.ascontiguousarray(angles, np.float64); t = np.ascontiguousarray(tol, np.float64)
        return self.lib.orc_set_goal_joint(self.h, _p(a, _dp), _p(t, _dp))

    def set_goal_xyz(self, xyz, tol):
        a = np.ascontiguousarray(xyz, np.float64); t = np.ascontiguousarray(tol, np.float64)
        return self.lib.orc_set_goal_xyz(self.h, _p(a, _dp), _p(t, _dp))

    def goal_pose(self):
        x = np.zeros(3)
        self.lib.orc_goal_pose(self.h, _p(x, _dp))
        return x

    def bfs_grid(self):
        n = self.lib.orc_bfs_size(self.h)
        out = np.zeros(n, np.int32)
        self.lib.orc_bfs_copy(self.h, _p(out, _ip))
        g = self.cfg.grid.dims
        return out.reshape(g[2] + 2, g[1] + 2, g[0] + 2)   # [z][y][x], bfs3d.h:213-220

    def heuristic_q(self, q):
        q = np.ascontiguousarray(q, np.float64)
        return self.lib.orc_heuristic_q(self.h, _p(q, _dp))

    def metric_goal_distance(self, x, y, z):
        return self.lib.orc_metric_goal_distance(self.h, x, y, z)

    # --- lattice ---
    def set_start(self, q):
        q = np.ascontiguousarray(q, np.float64)
        return self.lib.orc_set_start(self.h, _p(q, _dp))

    def num_states(self):
        return self.lib.orc_num_states(self.h)

    def get_state(self, i):
        q = np.zeros(self.N); c = np.zeros(self.N, np.int32)
        self.lib.orc_get_state(self.h, i, _p(q, _dp), _p(c, _ip))
        return q, c

    def get_succs(self, i):
        s = np.zeros(self.M, np.int32); k = np.zeros(self.M, np.int32)
        n = self.lib.orc_get_succs(self.h, i, _p(s, _ip), _p(k, _ip), self.M)
        return s[:n].copy(), k[:n].copy()

    def eval_state(self, q):
        q = np.ascontiguousarray(q, np.float64)
        M, N = self.M, self.N
        flags = np.zeros(M, np.uint8); coord = np.zeros((M, N), np.int32); sq = np.zeros((M, N))
        h = np.zeros(M, np.int32); cost = np.zeros(M, np.int32); lk = np.zeros(M, np.int32)
        self.lib.orc_eval_state(self.h, _p(q, _dp), _p(flags, _up), _p(coord, _ip), _p(sq, _dp), _p(h, _ip),
                                _p(cost, _ip), _p(lk, _ip))
        return dict(flags=flags, coord=coord, q=sq, h=h, cost=cost, lookups=lk)

    def eval_batch_timed(self, Q, min_seconds):
        Q = np.ascontiguousarray(Q, np.float64).reshape(-1, self.N)
        ev, va = C.c_long(), C.c_long(); sec = C.c_double(); ps = C.c_int()
        self.lib.orc_eval_batch_timed.argtypes = [C.c_void_p, _dp, C.c_int, C.c_double, C.POINTER(C.c_long),
                                                  C.POINTER(C.c_long), _dp, _ip]
        self.lib.orc_eval_batch_timed(self.h, _p(Q, _dp), Q.shape[0], float(min_seconds), C.byref(ev), C.byref(va),
                                      C.byref(sec), C.byref(ps))
        return dict(evals=ev.value, valid=va.value, seconds=sec.value, passes=ps.value)

    # --- search ---
    def search_params(self, eps0, eps_final, eps_delta, improve=True, bounded=False, max_init=0, max_rep=0):
        self.lib.orc_search_params(self.h, eps0, eps_final, eps_delta, int(improve), int(bounded), max_init, max_rep)

    def plan(self, cap=100000):
        ids = np.zeros(cap, np.int32)
        n, cost, exp = C.c_int(), C.c_int(), C.c_int()
        ev = C.c_long(); eps = C.c_double(); sec = C.c_double()
        ok = self.lib.orc_plan(self.h, _p(ids, _ip), cap, C.byref(n), C.byref(cost), C.byref(exp), C.byref(ev),
                               C.byref(eps), C.byref(sec))
        ne = self.lib.orc_expansion_log_size(self.h)
        log = np.zeros(ne, np.int32)
        if ne:
            self.lib.orc_expansion_log(self.h, _p(log, _ip))
        return dict(ok=ok, path=ids[:n.value].copy(), cost=cost.value, expansions=exp.value, succ_evals=ev.value,
                    eps=eps.value, seconds=sec.value, expansion_log=log)

    def post_process(self, path, shortcut=True, interpolate=True, upstream_limits=False, cap=20000):
        path = np.ascontiguousarray(path, np.float64).reshape(-1, self.N)
        out = np.zeros((cap, self.N)); ec, sc = C.c_long(), C.c_long()
        mode = (1 if shortcut else 0) | (2 if interpolate else 0) | (4 if upstream_limits else 0)
        self.lib.orc_post_process.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        n = self.lib.orc_post_process(self.h, _p(path, _dp), path.shape[0], mode, _p(out, _dp), cap, C.byref(ec), C.byref(sc))
        return out[:n].copy(), ec.value, sc.value

    def total_lookups(self):
        return self.lib.orc_total_lookups(self.h)

    def heap_run(self, ops):
        ops = np.ascontiguousarray(ops, np.int32)
        out = np.zeros(ops.shape[0], np.int32)
        self.lib.orc_heap_run(_p(ops, _ip), ops.shape[0], _p(out, _ip))
        return out
