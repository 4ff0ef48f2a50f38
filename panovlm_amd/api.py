"""ctypes binding of include/pvlm.h.  Thin: argument marshalling and error translation only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POINT2PLANE_METER, POINT2PLANE_ANGLE, POINT2LINE_METER, POINT2LINE_ANGLE, PLANE2PLANE_GLOBAL, PLANE_IOU = range(6)
ERR_CAPACITY = -5            # pvlm_status PVLM_ERR_CAPACITY
FLAG_NORMALIZE_DISTANCE = 1
FLAG_ASSOC_KEEP_INDICES = 0x100   # pvlm_assoc_point2plane: keep query / neighbour indices for pvlm_assoc_point2plane_debug
FLAG_ASSOC_EXACT_FIT = 0x200      # pvlm_assoc_point2plane: the reference's QR for every query (bit-identical records)
LOSS_NONE, LOSS_HUBER = 0, 1
PAIR_BLOCK = 121
STRIDE = {0: 7, 1: 7, 2: 9, 3: 9, 4: 10, 5: 12}

# every symbol include/pvlm.h declares (tests/test_abi.py checks the header and the .so against it)
ABI_SYMBOLS = [
    "pvlm_create", "pvlm_destroy", "pvlm_last_error", "pvlm_version", "pvlm_set_stream", "pvlm_use_own_stream", "pvlm_synchronize",
    "pvlm_timer_start", "pvlm_timer_stop", "pvlm_device_info", "pvlm_profile_enable", "pvlm_profile_read", "pvlm_set_poses", "pvlm_set_poses_dev",
    "pvlm_resset_upload", "pvlm_resset_destroy", "pvlm_resset_info", "pvlm_resset_download", "pvlm_eval",
    "pvlm_eval_dev", "pvlm_eval_pair_blocks", "pvlm_eval_pair_blocks_dev", "pvlm_neq_create", "pvlm_neq_destroy",
    "pvlm_neq_size", "pvlm_neq_accumulate_dev", "pvlm_neq_accumulate", "pvlm_neq_accumulate_async", "pvlm_neq_accumulate_sets", "pvlm_resset_set_pose_ids", "pvlm_comm_unique_id", "pvlm_comm_create",
    "pvlm_comm_destroy", "pvlm_allreduce_sum_f64", "pvlm_scan_upload", "pvlm_scan_upload_batch", "pvlm_scan_destroy",
    "pvlm_knn", "pvlm_centre_orders", "pvlm_assoc_point2plane", "pvlm_assoc_point2plane_debug", "pvlm_line2line_votes",
    "pvlm_cam_to_image_f32", "pvlm_cam_to_image_f64", "pvlm_image_to_cam_f32", "pvlm_image_to_cam_f64",
    "pvlm_cam_lidar_votes", "pvlm_line2line_votes_batch", "pvlm_line2line_best_batch", "pvlm_cam_lidar_votes_batch", "pvlm_cam_lidar_votes_batch_sparse",
    "pvlm_cam_to_image_f32_dev", "pvlm_image_to_cam_f32_dev", "pvlm_project_lidar_depth", "pvlm_spd_solve", "pvlm_spd_solve_blocks", "pvlm_mvs_init_conf_map", "pvlm_mvs_filter_depth", "pvlm_mvs_filter_depth_refine", "pvlm_mvs_propagate", "pvlm_mvs_propagate_sequential", "pvlm_mvs_views_estimate_sequential", "pvlm_mvs_views_estimate_sequential_batch", "pvlm_mvs_views_create", "pvlm_mvs_views_destroy", "pvlm_mvs_views_upload", "pvlm_mvs_views_download",
    "pvlm_mvs_views_snapshot_depth", "pvlm_mvs_views_estimate", "pvlm_mvs_views_filter_refine",
    "pvlm_ba_create", "pvlm_ba_destroy", "pvlm_ba_structure", "pvlm_ba_packed_size", "pvlm_ba_get_points", "pvlm_ba_set_points", "pvlm_ba_set_constant",
    "pvlm_ba_eval", "pvlm_ba_reduce", "pvlm_ba_step", "pvlm_ba_cost", "pvlm_ba_accept",
    "pvlm_reserve", "pvlm_reserve_staging", "pvlm_preload", "pvlm_trim", "pvlm_mem_info", "pvlm_graph_begin", "pvlm_graph_end", "pvlm_graph_launch", "pvlm_graph_destroy",
    "pvlm_allreduce_sum_f64_host", "pvlm_host_alloc", "pvlm_host_free", "pvlm_eval_host_async", "pvlm_eval_wrench_host_async", "pvlm_eval_force_host_async", "pvlm_line2line_residuals", "pvlm_mvs_init_depth_normal", "pvlm_mvs_remove_small_segments", "pvlm_mvs_depth_to_cloud", "pvlm_mvs_views_depth_to_cloud",
    "pvlm_spd_plan_info", "pvlm_spd_plan_schedule", "pvlm_spd_plan_tail", "pvlm_spd_one_launch", "pvlm_spd_plan_prefetch", "pvlm_spd_plan_prefetch_hits", "pvlm_line_grow_batch", "pvlm_line_grow_begin", "pvlm_line_grow_finish", "pvlm_line_grow_scan", "pvlm_line_grow_destroy", "pvlm_ring_extract_batch", "pvlm_ring_extract_batch_picks", "pvlm_ring_debug_sort", "pvlm_undistort_batch", "pvlm_assoc_point2plane_stats", "pvlm_assoc_point2plane_stats2", "pvlm_scan_transform_batch", "pvlm_scan_set_pose", "pvlm_scan_cloud_info", "pvlm_scan_cloud_fetch", "pvlm_ring_batch_scan", "pvlm_ring_batch_fetch", "pvlm_ring_batch_timing", "pvlm_ring_batch_destroy",
]


class PvlmError(RuntimeError):
    pass


def lib_path():
    # PVLM_LIB selects another build of the same library (measured kernel variants); never a different implementation
    return os.environ.get("PVLM_LIB") or os.path.join(_HERE, "libpvlm.so")


_LIB = None


def load_library():
    """Loads libpvlm.so.  Raises if it has not been built (python -m panovlm_amd.build): the HIP
    extension IS the product, nothing else can stand in for it."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise PvlmError("libpvlm.so is missing (%s): build it with `python -m panovlm_amd.build` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    try:
        # PyTorch ships its own libamdhip64 with the same SONAME; when it is in the process it must
        # be loaded first so that one HIP runtime serves both.
        import sys
        if "torch" in sys.modules or os.environ.get("PVLM_PRELOAD_TORCH", "0") == "1":
            import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.pvlm_last_error.restype = C.c_char_p
    lib.pvlm_version.restype = C.c_char_p
    lib.pvlm_neq_size.restype = C.c_int64
    lib.pvlm_ba_packed_size.restype = C.c_int64
    _LIB = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Graph:
    """A captured step (pvlm_graph_*): launch() replays it on the context's stream."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle

    def launch(self):
        self.ctx._check(self.ctx.lib.pvlm_graph_launch(self.ctx._h, self._h), "pvlm_graph_launch")

    def close(self):
        if self._h and self.ctx._h:
            self.ctx.lib.pvlm_graph_destroy(self.ctx._h, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        self.lib = load_library()
        self._h = C.c_void_p()
        rc = self.lib.pvlm_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise PvlmError("pvlm_create(device=%d) failed with status %d: no usable HIP device "
                            "(this package has no CPU path)" % (device, rc))
        self.device = device

    def _check(self, rc, what):
        if rc != 0:
            raise PvlmError("%s failed (%d): %s" % (what, rc, self.lib.pvlm_last_error(self._h).decode()))

    def close(self):
        if self._h:
            self.lib.pvlm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        """Issue all kernels on this hipStream_t (0 / None = HIP's default stream, e.g. torch's default stream)."""
        self._check(self.lib.pvlm_set_stream(self._h, C.c_void_p(stream_handle or 0)), "pvlm_set_stream")

    def use_own_stream(self):
        self._check(self.lib.pvlm_use_own_stream(self._h), "pvlm_use_own_stream")

    def synchronize(self):
        self._check(self.lib.pvlm_synchronize(self._h), "pvlm_synchronize")

    def reserve(self, nbytes):
        """One free range of `nbytes` in the context's device pool (hipMalloc is 40-70 ms per GB: pay it at set-up)."""
        self._check(self.lib.pvlm_reserve(self._h, C.c_int64(int(nbytes))), "pvlm_reserve")

    def trim(self):
        self._check(self.lib.pvlm_trim(self._h), "pvlm_trim")

    def mem_info(self):
        r, u, p, n = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.pvlm_mem_info(self._h, C.byref(r), C.byref(u), C.byref(p), C.byref(n)), "pvlm_mem_info")
        return dict(reserved=r.value, in_use=u.value, peak=p.value, device_allocs=n.value)

    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 creates it, the launcher carries it to the other ranks)."""
        buf = (C.c_ubyte * 128)()
        self._check(self.lib.pvlm_comm_unique_id(self._h, buf), "pvlm_comm_unique_id")
        return bytes(buf)

    def host_alloc(self, nbytes):
        """Page-locked host memory (pvlm_host_alloc) as a numpy float64 array; free with host_free(array)."""
        p = C.c_void_p()
        self._check(self.lib.pvlm_host_alloc(self._h, C.c_int64(int(nbytes)), C.byref(p)), "pvlm_host_alloc")
        n = max(int(nbytes) // 8, 1)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, array):
        p = getattr(self, "_pinned", {}).pop(array.ctypes.data, None)
        if p is not None:
            self._check(self.lib.pvlm_host_free(self._h, p), "pvlm_host_free")

    def graph_begin(self):
        self._check(self.lib.pvlm_graph_begin(self._h), "pvlm_graph_begin")

    def graph_end(self):
        g = C.c_void_p()
        self._check(self.lib.pvlm_graph_end(self._h, C.byref(g)), "pvlm_graph_end")
        return Graph(self, g)

    def timer_start(self):
        self._check(self.lib.pvlm_timer_start(self._h), "pvlm_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.pvlm_timer_stop(self._h, C.byref(ms)), "pvlm_timer_stop")
        return float(ms.value)

    def device_info(self):
        cu = C.c_int(); hbm = C.c_int64(); name = C.create_string_buffer(64)
        self._check(self.lib.pvlm_device_info(self._h, C.byref(cu), C.byref(hbm), name, C.c_int(64)), "pvlm_device_info")
        return dict(cu_count=cu.value, hbm_bytes=hbm.value, arch=name.value.decode())

    def profile_enable(self, on=True):
        self._check(self.lib.pvlm_profile_enable(self._h, C.c_int(1 if on else 0)), "pvlm_profile_enable")

    def profile_read(self, which=0):
        ms = C.c_double(); n = C.c_int64()
        self._check(self.lib.pvlm_profile_read(self._h, C.c_int(which), C.byref(ms), C.byref(n)), "pvlm_profile_read")
        return float(ms.value), int(n.value)

    def set_poses(self, angle_axis, translation):
        aa = _f64(angle_axis).reshape(-1, 3); t = _f64(translation).reshape(-1, 3)
        assert aa.shape == t.shape
        self._check(self.lib.pvlm_set_poses(self._h, C.c_int(aa.shape[0]), _p(aa, C.c_double), _p(t, C.c_double)), "pvlm_set_poses")

    def set_poses_dev(self, n, d_aa_ptr, d_t_ptr):
        self._check(self.lib.pvlm_set_poses_dev(self._h, C.c_int(n), C.c_void_p(d_aa_ptr), C.c_void_p(d_t_ptr)), "pvlm_set_poses_dev")

    # --- equirectangular -------------------------------------------------------------------------
    def cam_to_image(self, rows, cols, cam):
        cam = np.ascontiguousarray(cam)
        n = cam.shape[0]
        if cam.dtype == np.float32:
            px = np.empty((n, 2), np.float32)
            self._check(self.lib.pvlm_cam_to_image_f32(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), _p(cam, C.c_float), _p(px, C.c_float)), "pvlm_cam_to_image_f32")
        else:
            cam = _f64(cam); px = np.empty((n, 2), np.float64)
            self._check(self.lib.pvlm_cam_to_image_f64(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), _p(cam, C.c_double), _p(px, C.c_double)), "pvlm_cam_to_image_f64")
        return px

    def image_to_cam(self, rows, cols, px, r=1.0):
        px = np.ascontiguousarray(px)
        n = px.shape[0]
        if px.dtype == np.float32:
            cam = np.empty((n, 3), np.float32)
            self._check(self.lib.pvlm_image_to_cam_f32(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), _p(px, C.c_float), C.c_float(r), _p(cam, C.c_float)), "pvlm_image_to_cam_f32")
        else:
            px = _f64(px); cam = np.empty((n, 3), np.float64)
            self._check(self.lib.pvlm_image_to_cam_f64(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), _p(px, C.c_double), C.c_double(r), _p(cam, C.c_double)), "pvlm_image_to_cam_f64")
        return cam

    def cam_lidar_votes(self, rows, cols, lines, lidar_scan, T_cl):
        lines = _f32(lines).reshape(-1, 4); T = _f64(T_cl).reshape(16)
        votes = np.zeros((lines.shape[0], max(1, lidar_scan.n_segments)), np.int32)
        self._check(self.lib.pvlm_cam_lidar_votes(self._h, C.c_int(rows), C.c_int(cols), _p(lines, C.c_float), C.c_int(lines.shape[0]),
                                                  lidar_scan._h, _p(T, C.c_double), _p(votes, C.c_int32)), "pvlm_cam_lidar_votes")
        return votes[:, :lidar_scan.n_segments]

    def cam_to_image_f32_dev(self, rows, cols, n, d_cam_ptr, d_px_ptr):
        self._check(self.lib.pvlm_cam_to_image_f32_dev(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), C.c_void_p(d_cam_ptr), C.c_void_p(d_px_ptr)),
                    "pvlm_cam_to_image_f32_dev")

    def image_to_cam_f32_dev(self, rows, cols, n, d_px_ptr, r, d_cam_ptr):
        self._check(self.lib.pvlm_image_to_cam_f32_dev(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(n), C.c_void_p(d_px_ptr), C.c_float(r),
                                                       C.c_void_p(d_cam_ptr)), "pvlm_image_to_cam_f32_dev")

    def spd_solve(self, A, B):
        """Dense SPD solve on the GPU (blocked Cholesky kernels of libpvlm.so): returns (X, info)."""
        A = _f64(A); n = A.shape[0]
        B2 = np.asfortranarray(np.asarray(B, np.float64).reshape(n, -1))
        info = C.c_int()
        self._check(self.lib.pvlm_spd_solve(self._h, C.c_int(n), C.c_int(B2.shape[1]), _p(A, C.c_double),
                                            B2.ctypes.data_as(C.POINTER(C.c_double)), C.byref(info)), "pvlm_spd_solve")
        return np.ascontiguousarray(B2).reshape(np.shape(B)), info.value

    def spd_solve_blocks(self, n, row_idx, col_idx, mirror, blocks, scale, diag_add, rhs):
        row_idx = _i32(row_idx).reshape(-1, 6); col_idx = _i32(col_idx).reshape(-1, 6); blocks = _f64(blocks).reshape(-1, 36); mirror = _i32(mirror)
        x = _f64(rhs).copy(); info = C.c_int()
        self._check(self.lib.pvlm_spd_solve_blocks(self._h, C.c_int(n), C.c_int(len(blocks)), _p(row_idx, C.c_int), _p(col_idx, C.c_int), _p(mirror, C.c_int),
                                                   _p(blocks, C.c_double), _p(_f64(scale), C.c_double), _p(_f64(diag_add), C.c_double),
                                                   _p(x, C.c_double), C.byref(info)), "pvlm_spd_solve_blocks")
        return x, info.value

    def line_grow_batch(self, clouds, two_halves=False):
        """pvlm_line_grow_batch (K27): the line segments upstream's walk keeps for every edge cloud (n x >= 3 float32 each).  Returns one dict per cloud:
        status, seg_task, seg_offset, members, coeffs (n_segments x 6); the last call's kernel_ms / tasks_run under those keys of the first dict.
        two_halves: through pvlm_line_grow_begin / _finish."""
        class _Cloud(C.Structure):
            _fields_ = [("xyz", C.POINTER(C.c_float)), ("n", C.c_int), ("stride_floats", C.c_int)]

        class _Result(C.Structure):
            _fields_ = [("status", C.c_int), ("n_points", C.c_int), ("n_segments", C.c_int), ("seg_task", C.POINTER(C.c_int)), ("seg_offset", C.POINTER(C.c_int)),
                        ("members", C.POINTER(C.c_int)), ("coeffs", C.POINTER(C.c_double)), ("kernel_ms", C.c_double), ("tasks_run", C.c_longlong)]
        arrs = [np.ascontiguousarray(c, np.float32).reshape(len(c), -1) if len(c) else np.zeros((0, 3), np.float32) for c in clouds]
        descs = (_Cloud * max(len(arrs), 1))()
        for k, a in enumerate(arrs):
            descs[k].xyz = a.ctypes.data_as(C.POINTER(C.c_float)); descs[k].n = len(a); descs[k].stride_floats = a.shape[1] if len(a) else 3
        h = C.c_void_p()
        if two_halves:
            self._check(self.lib.pvlm_line_grow_begin(self._h, C.c_int(len(arrs)), descs, C.byref(h)), "pvlm_line_grow_begin")
            st = self.lib.pvlm_line_grow_finish(self._h, h)
            if st:
                self.lib.pvlm_line_grow_destroy(self._h, h)
                self._check(st, "pvlm_line_grow_finish")
        else:
            self._check(self.lib.pvlm_line_grow_batch(self._h, C.c_int(len(arrs)), descs, C.byref(h)), "pvlm_line_grow_batch")
        out = []
        try:
            for k in range(len(arrs)):
                r = _Result()
                self._check(self.lib.pvlm_line_grow_scan(h, C.c_int(k), C.byref(r)), "pvlm_line_grow_scan")
                ns = r.n_segments
                off = np.ctypeslib.as_array(r.seg_offset, (ns + 1,)).copy() if ns else np.zeros(1, np.int32)
                mem = np.ctypeslib.as_array(r.members, (int(off[-1]),))[int(off[0]):].copy() if ns else np.zeros(0, np.int32)
                out.append(dict(status=r.status, seg_task=np.ctypeslib.as_array(r.seg_task, (ns,)).copy() if ns else np.zeros(0, np.int32), seg_offset=off - off[0], members=mem,
                                coeffs=np.ctypeslib.as_array(r.coeffs, (ns, 6)).copy() if ns else np.zeros((0, 6)), kernel_ms=r.kernel_ms, tasks_run=r.tasks_run))
        finally:
            self.lib.pvlm_line_grow_destroy(self._h, h)
        return out

    def spd_plan_prefetch(self, n, row_idx, col_idx, mirror):
        """pvlm_spd_plan_prefetch: the host half of the plan of this structure starts on a thread of the library; the next spd_solve_blocks with exactly these lists takes it."""
        row_idx = _i32(row_idx).reshape(-1, 6); col_idx = _i32(col_idx).reshape(-1, 6); mirror = _i32(mirror)
        self._check(self.lib.pvlm_spd_plan_prefetch(self._h, C.c_int(n), C.c_int(len(mirror)), _p(row_idx, C.c_int), _p(col_idx, C.c_int), _p(mirror, C.c_int)), "pvlm_spd_plan_prefetch")

    def spd_plan_prefetch_hits(self):
        h = C.c_longlong()
        self._check(self.lib.pvlm_spd_plan_prefetch_hits(self._h, C.byref(h)), "pvlm_spd_plan_prefetch_hits")
        return h.value

    def centre_orders(self, xyz):
        """pvlm_centre_orders: for every centre the positions of all centres ordered by (float squared distance, position) — (n, n) uint16."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3); n = len(xyz)
        out = np.zeros((n, n), np.uint16)
        self._check(self.lib.pvlm_centre_orders(self._h, C.c_int(n), _p(xyz, C.c_float), _p(out, C.c_uint16)), "pvlm_centre_orders")
        return out

    def spd_one_launch(self, enable=None):
        """pvlm_spd_one_launch: enable / disable the one-launch form of the tile-sparse factorisation (None: query only); returns the number of solves this context has
        redone with the level launches."""
        fb = C.c_longlong()
        self._check(self.lib.pvlm_spd_one_launch(self._h, C.c_int(-1 if enable is None else (int(enable) if isinstance(enable, int) and not isinstance(enable, bool) else int(bool(enable)))), C.byref(fb)),
                    "pvlm_spd_one_launch")
        return fb.value

    def spd_plan(self):
        """How the last spd_solve_blocks structure is factorised: tile_sparse, update_fraction (pvlm_spd_plan_info) and the schedule (pvlm_spd_plan_schedule):
        levels (0 = column by column), block_columns, padded_rows."""
        ts, lv, bc, pr = C.c_int(), C.c_int(), C.c_int(), C.c_int(); fr = C.c_double()
        self._check(self.lib.pvlm_spd_plan_info(self._h, C.byref(ts), C.byref(fr)), "pvlm_spd_plan_info")
        self._check(self.lib.pvlm_spd_plan_schedule(self._h, C.byref(lv), C.byref(bc), C.byref(pr)), "pvlm_spd_plan_schedule")
        tc, ll = C.c_int(), C.c_int()
        self._check(self.lib.pvlm_spd_plan_tail(self._h, C.byref(tc), C.byref(ll)), "pvlm_spd_plan_tail")
        return dict(tile_sparse=bool(ts.value), update_fraction=fr.value, levels=lv.value, block_columns=bc.value, padded_rows=pr.value, tail_block_columns=tc.value,
                    launched_levels=ll.value)

    def mvs_init_conf_map(self, ref_gray, nei_grays, R_nr, t_nr, depth, normal, half_window=3, step=1, conf=None, nei_depths=None):
        """MVS::InitPatchMap + InitConfMap(use_geometry=False) on the GPU: returns (conf, depth, normal) copies."""
        ref = np.ascontiguousarray(ref_gray, np.uint8); rows, cols = ref.shape
        neis = [np.ascontiguousarray(g, np.uint8) for g in nei_grays]
        ptrs = (C.POINTER(C.c_ubyte) * max(len(neis), 1))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
        R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
        d = np.array(depth, np.float32, copy=True); nrm = np.array(normal, np.float32, copy=True)
        c = np.zeros((rows, cols), np.float32) if conf is None else np.array(conf, np.float32, copy=True)
        dptrs = None
        if nei_depths is not None:
            nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
            dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
        self._check(self.lib.pvlm_mvs_init_conf_map(self._h, C.c_int(rows), C.c_int(cols), C.c_int(half_window), C.c_int(step), _p(ref, C.c_ubyte),
                                                    C.c_int(len(neis)), ptrs, _p(R, C.c_float), _p(t, C.c_float), _p(d, C.c_float), _p(nrm, C.c_float),
                                                    _p(c, C.c_float), dptrs), "pvlm_mvs_init_conf_map")
        return c, d, nrm

    def mvs_init_depth_normal(self, rows, cols, lidar_depth16=None, mask=None, min_depth=0.1, max_depth=20.0, keep_lidar_constant=True, seed=1):
        """MVS::InitDepthNormal on the GPU: returns (depth, normal, depth_constant uint8)."""
        l16 = None if lidar_depth16 is None else np.ascontiguousarray(lidar_depth16, np.uint16)
        m = None if mask is None else np.ascontiguousarray(mask, np.float32)
        d = np.zeros((rows, cols), np.float32); n = np.zeros((rows, cols, 3), np.float32); c = np.zeros((rows, cols), np.uint8)
        self._check(self.lib.pvlm_mvs_init_depth_normal(self._h, C.c_int(rows), C.c_int(cols), _p(l16, C.c_uint16), _p(m, C.c_float), C.c_float(min_depth),
                                                        C.c_float(max_depth), C.c_int(1 if keep_lidar_constant else 0), C.c_ulonglong(seed), _p(d, C.c_float),
                                                        _p(n, C.c_float), _p(c, C.c_ubyte)), "pvlm_mvs_init_depth_normal")
        return d, n, c

    def mvs_remove_small_segments(self, depth, normal, conf, depth_diff_threshold=0.01, min_segment=100):
        """MVS::RemoveSmallSegments (host, sequential by construction): returns (depth, normal, conf, removed)."""
        d = np.array(depth, np.float32, copy=True); n = np.array(normal, np.float32, copy=True); c = np.array(conf, np.float32, copy=True)
        rows, cols = d.shape
        k = C.c_int64()
        self._check(self.lib.pvlm_mvs_remove_small_segments(self._h, C.c_int(rows), C.c_int(cols), C.c_float(depth_diff_threshold), C.c_int(min_segment),
                                                            _p(d, C.c_float), _p(n, C.c_float), _p(c, C.c_float), C.byref(k)), "pvlm_mvs_remove_small_segments")
        return d, n, c, int(k.value)

    def mvs_propagate(self, ref_gray, nei_grays, R_nr, t_nr, depth, normal, conf, half_window=3, step=1, nei_depths=None, depth_constant=None, min_depth=0.1,
                      max_depth=20.0, seed=1, max_iter=1, conf_threshold=-1.0, sequential=False):
        """MVS::EstimateDepthMapSingle on the GPU — checkerboard PatchMatch, or (sequential=True) the raster-order sweep the Room /
        Floor configs select, run anti-diagonal by anti-diagonal: returns (depth, normal, conf) copies."""
        ref = np.ascontiguousarray(ref_gray, np.uint8); rows, cols = ref.shape
        neis = [np.ascontiguousarray(g, np.uint8) for g in nei_grays]
        ptrs = (C.POINTER(C.c_ubyte) * max(len(neis), 1))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
        R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
        d = np.array(depth, np.float32, copy=True); nrm = np.array(normal, np.float32, copy=True); c = np.array(conf, np.float32, copy=True)
        dptrs = None
        if nei_depths is not None:
            nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
            dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
        dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
        fn = self.lib.pvlm_mvs_propagate_sequential if sequential else self.lib.pvlm_mvs_propagate
        self._check(fn(self._h, C.c_int(rows), C.c_int(cols), C.c_int(half_window), C.c_int(step), _p(ref, C.c_ubyte), C.c_int(len(neis)),
                       ptrs, _p(R, C.c_float), _p(t, C.c_float), _p(d, C.c_float), _p(nrm, C.c_float), _p(c, C.c_float), dptrs,
                       _p(dc, C.c_ubyte), C.c_float(min_depth), C.c_float(max_depth), C.c_ulonglong(seed), C.c_int(max_iter),
                       C.c_float(conf_threshold)), "pvlm_mvs_propagate_sequential" if sequential else "pvlm_mvs_propagate")
        return d, nrm, c

    def mvs_filter_depth(self, nei_depths, R_nr, t_nr, depth, conf=None, depth_constant=None, thr=0.01):
        """MVS::FilterDepthImage on the GPU: returns (depth_filter, conf_filter)."""
        d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
        nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
        dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
        R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
        cf = None if conf is None else np.ascontiguousarray(conf, np.float32)
        dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
        out_d = np.zeros((rows, cols), np.float32); out_c = np.zeros((rows, cols), np.float32)
        self._check(self.lib.pvlm_mvs_filter_depth(self._h, C.c_int(rows), C.c_int(cols), C.c_int(len(nd)), dptrs, _p(R, C.c_float), _p(t, C.c_float),
                                                   _p(d, C.c_float), _p(cf, C.c_float), _p(dc, C.c_ubyte), C.c_float(thr), _p(out_d, C.c_float),
                                                   _p(out_c, C.c_float) if cf is not None else None), "pvlm_mvs_filter_depth")
        return out_d, out_c

    def mvs_depth_to_cloud(self, depth, bgr, T_wc, max_depth=20.0, filter_sky=True, normal=None):
        """MVS::DepthImageToCloud (normal None) / DepthNormalToCloud (normal given; pass filter_sky=False for the reference's behaviour) on the GPU:
        (xyz n x 3 float32, rgb n x 3 uint8[, normal n x 3 float32]) in raster order."""
        d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
        c = np.ascontiguousarray(bgr, np.uint8); assert c.shape == (rows, cols, 3)
        T = np.ascontiguousarray(np.asarray(T_wc, np.float64).reshape(-1)[:12])
        nin = None if normal is None else np.ascontiguousarray(normal, np.float32)
        assert nin is None or nin.shape == (rows, cols, 3)
        xyz = np.empty((rows * cols, 3), np.float32); rgb = np.empty((rows * cols, 3), np.uint8)
        nout = None if nin is None else np.empty((rows * cols, 3), np.float32)
        n = C.c_longlong(0)
        self._check(self.lib.pvlm_mvs_depth_to_cloud(self._h, C.c_int(rows), C.c_int(cols), _p(d, C.c_float), _p(c, C.c_ubyte), _p(nin, C.c_float), _p(T, C.c_double),
                                                     C.c_float(max_depth), C.c_int(1 if filter_sky else 0), _p(xyz, C.c_float), _p(rgb, C.c_ubyte), _p(nout, C.c_float),
                                                     C.byref(n)), "pvlm_mvs_depth_to_cloud")
        k = n.value
        return (xyz[:k].copy(), rgb[:k].copy()) if nout is None else (xyz[:k].copy(), rgb[:k].copy(), nout[:k].copy())

    def mvs_filter_depth_refine(self, nei_depths, nei_confs, R_nr, t_nr, depth, conf, depth_constant=None, thr=0.01, min_depth=0.1, max_depth=20.0):
        """MVS::FilterDepthImageRefine on the GPU: returns (depth_filter, conf_filter, conf) — conf = the reference frame's
        conf_map after the call (zeroed where depth <= 0)."""
        d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
        nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
        nc = [np.ascontiguousarray(x, np.float32) for x in nei_confs]
        assert len(nd) == len(nc)
        dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
        cptrs = (C.POINTER(C.c_float) * max(len(nc), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nc])
        R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
        cf = np.array(conf, np.float32, copy=True, order="C")
        dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
        out_d = np.zeros((rows, cols), np.float32); out_c = np.zeros((rows, cols), np.float32)
        self._check(self.lib.pvlm_mvs_filter_depth_refine(self._h, C.c_int(rows), C.c_int(cols), C.c_int(len(nd)), dptrs, cptrs, _p(R, C.c_float),
                                                          _p(t, C.c_float), _p(d, C.c_float), _p(cf, C.c_float), _p(dc, C.c_ubyte), C.c_float(thr),
                                                          C.c_float(min_depth), C.c_float(max_depth), _p(out_d, C.c_float), _p(out_c, C.c_float)),
                    "pvlm_mvs_filter_depth_refine")
        return out_d, out_c, cf

    def project_lidar_depth(self, rows, cols, xyz, T_cl, size=3):
        xyz = _f32(xyz).reshape(-1, 3); T = _f64(T_cl).reshape(16)
        out = np.zeros((rows, cols), np.uint16)
        self._check(self.lib.pvlm_project_lidar_depth(self._h, C.c_int(rows), C.c_int(cols), C.c_int64(xyz.shape[0]), _p(xyz, C.c_float),
                                                      _p(T, C.c_double), C.c_uint(size), _p(out, C.c_uint16)), "pvlm_project_lidar_depth")
        return out

    def cam_lidar_votes_batch(self, rows, cols, lines_list, lidar_scans, T_cl_list):
        """One launch for many (image lines, LiDAR-local scan, T_cl) triples; returns the list of vote matrices."""
        n = len(lidar_scans)
        assert len(lines_list) == n and len(T_cl_list) == n
        ls = [_f32(l).reshape(-1, 4) for l in lines_list]
        off = _i64(np.concatenate([[0], np.cumsum([len(l) for l in ls])]))
        flat = _f32(np.concatenate(ls) if n and off[-1] else np.zeros((0, 4)))
        T = _f64(np.array([np.asarray(t, np.float64).reshape(16) for t in T_cl_list]).reshape(-1))
        hs = (C.c_void_p * max(n, 1))(*[s._h for s in lidar_scans])
        voff = np.zeros(n + 1, np.int64)
        args = (self._h, C.c_int(n), C.c_int(rows), C.c_int(cols), _p(off, C.c_int64), _p(flat, C.c_float), hs, _p(T, C.c_double), _p(voff, C.c_int64))
        self._check(self.lib.pvlm_cam_lidar_votes_batch(*args, None, C.c_int64(0)), "pvlm_cam_lidar_votes_batch")
        votes = np.zeros(max(int(voff[-1]), 1), np.int32)
        self._check(self.lib.pvlm_cam_lidar_votes_batch(*args, _p(votes, C.c_int32), C.c_int64(len(votes))), "pvlm_cam_lidar_votes_batch")
        return [votes[voff[p]:voff[p + 1]].reshape(len(ls[p]), lidar_scans[p].n_segments) for p in range(n)]

    def cam_lidar_votes_batch_sparse(self, rows, cols, lines_list, lidar_scans, T_cl_list):
        """The same launch, the votes returned sparse: (vote_offsets [n + 1], nz_index, nz_count) — the non-zero counters of the dense
        layout in ascending position (pvlm_cam_lidar_votes_batch_sparse)."""
        n = len(lidar_scans)
        assert len(lines_list) == n and len(T_cl_list) == n
        ls = [_f32(l).reshape(-1, 4) for l in lines_list]
        off = _i64(np.concatenate([[0], np.cumsum([len(l) for l in ls])]))
        flat = _f32(np.concatenate(ls) if n and off[-1] else np.zeros((0, 4)))
        T = _f64(np.array([np.asarray(t, np.float64).reshape(16) for t in T_cl_list]).reshape(-1))
        hs = (C.c_void_p * max(n, 1))(*[s._h for s in lidar_scans])
        voff = np.zeros(n + 1, np.int64)
        nnz = C.c_int64(0)
        cap = max(1, int(sum(len(l) for l in ls)) * 4)          # a line rarely collects votes for more than a few segments
        import time as _time
        while True:
            idx = np.empty(cap, np.int64); cnt = np.empty(cap, np.int32)
            t0 = _time.perf_counter()
            rc = self.lib.pvlm_cam_lidar_votes_batch_sparse(self._h, C.c_int(n), C.c_int(rows), C.c_int(cols), _p(off, C.c_int64), _p(flat, C.c_float), hs,
                                                             _p(T, C.c_double), _p(voff, C.c_int64), _p(idx, C.c_int64), _p(cnt, C.c_int32), C.c_int64(cap), C.byref(nnz))
            self.last_call_s = _time.perf_counter() - t0      # the C ABI call alone (the lists above are this wrapper's)
            if rc == ERR_CAPACITY and nnz.value > cap:
                cap = int(nnz.value)
                continue
            self._check(rc, "pvlm_cam_lidar_votes_batch_sparse")
            return voff, idx[:nnz.value], cnt[:nnz.value]

    # --- association -------------------------------------------------------------------------------
    def knn(self, scan, queries, k, max_dist, which=0):
        q = _f32(queries).reshape(-1, 3)
        idx = np.empty((q.shape[0], k), np.int32); sqd = np.empty((q.shape[0], k), np.float32)
        self._check(self.lib.pvlm_knn(self._h, scan._h, C.c_int(which), _p(q, C.c_float), C.c_int(q.shape[0]), C.c_int(k),
                                      C.c_float(max_dist), _p(idx, C.c_int32), _p(sqd, C.c_float)), "pvlm_knn")
        return idx, sqd

    def assoc_point2plane(self, refs, neis, plane_tolerance, dist_threshold, kind=POINT2PLANE_ANGLE,
                          flags=FLAG_NORMALIZE_DISTANCE, weight=1.0):
        n = len(refs)
        assert n == len(neis)
        ra = (C.c_void_p * max(n, 1))(*[s._h for s in refs])
        na = (C.c_void_p * max(n, 1))(*[s._h for s in neis])
        h = C.c_void_p()
        self._check(self.lib.pvlm_assoc_point2plane(self._h, C.c_int(n), ra, na, C.c_double(plane_tolerance), C.c_float(dist_threshold),
                                                    C.c_int(kind), C.c_uint(flags), C.c_double(weight), C.byref(h)), "pvlm_assoc_point2plane")
        return ResidualSet(self, h)

    def line2line_votes(self, ref, nei, dist_threshold):
        votes = np.zeros((max(1, nei.n_segments), max(1, ref.n_segments)), np.int32)
        self._check(self.lib.pvlm_line2line_votes(self._h, ref._h, nei._h, C.c_float(dist_threshold), _p(votes, C.c_int32)), "pvlm_line2line_votes")
        return votes[:nei.n_segments, :ref.n_segments]


    def line2line_votes_batch(self, refs, neis, dist_threshold):
        """One launch for many (ref, nei) pairs; returns the list of n_nei_seg x n_ref_seg vote matrices."""
        n = len(refs)
        assert len(neis) == n
        hr = (C.c_void_p * max(n, 1))(*[s._h for s in refs]); hn = (C.c_void_p * max(n, 1))(*[s._h for s in neis])
        voff = np.zeros(n + 1, np.int64)
        self._check(self.lib.pvlm_line2line_votes_batch(self._h, C.c_int(n), hr, hn, C.c_float(dist_threshold), _p(voff, C.c_int64), None, C.c_int64(0)),
                    "pvlm_line2line_votes_batch")
        votes = np.zeros(max(int(voff[-1]), 1), np.int32)
        self._check(self.lib.pvlm_line2line_votes_batch(self._h, C.c_int(n), hr, hn, C.c_float(dist_threshold), _p(voff, C.c_int64), _p(votes, C.c_int32),
                                                        C.c_int64(len(votes))), "pvlm_line2line_votes_batch")
        return [votes[voff[p]:voff[p + 1]].reshape(neis[p].n_segments, refs[p].n_segments) for p in range(n)]


    def line2line_best_batch(self, refs, neis, dist_threshold):
        """pvlm_line2line_best_batch: per pair and neighbour segment (a row of the vote block) the reference segment with the most votes — the first of equals — and
        that count.  Returns (row_offsets, best_col, best_count); a pair without reference segments has no rows."""
        n = len(refs)
        assert len(neis) == n
        hr = (C.c_void_p * max(n, 1))(*[s._h for s in refs]); hn = (C.c_void_p * max(n, 1))(*[s._h for s in neis])
        roff = np.zeros(n + 1, np.int64)
        self._check(self.lib.pvlm_line2line_best_batch(self._h, C.c_int(n), hr, hn, C.c_float(dist_threshold), _p(roff, C.c_int64), None, None, C.c_int64(0)), "pvlm_line2line_best_batch")
        col = np.zeros(max(int(roff[-1]), 1), np.int32); cnt = np.zeros_like(col)
        self._check(self.lib.pvlm_line2line_best_batch(self._h, C.c_int(n), hr, hn, C.c_float(dist_threshold), _p(roff, C.c_int64), _p(col, C.c_int32), _p(cnt, C.c_int32),
                                                       C.c_int64(len(col))), "pvlm_line2line_best_batch")
        return roff, col[:int(roff[-1])], cnt[:int(roff[-1])]


class MvsViews:
    """Resident set of equally sized MVS views (pvlm_mvs_views_*): maps stay on the GPU between the scoring pass, the
    PatchMatch sweeps and the fusion filter."""

    def __init__(self, ctx, rows, cols, n_views):
        self.ctx, self.rows, self.cols, self.n = ctx, rows, cols, n_views
        self._h = C.c_void_p()
        ctx._check(ctx.lib.pvlm_mvs_views_create(ctx._h, C.c_int(rows), C.c_int(cols), C.c_int(n_views), C.byref(self._h)), "pvlm_mvs_views_create")

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_mvs_views_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, view, gray=None, depth=None, normal=None, conf=None):
        g = None if gray is None else np.ascontiguousarray(gray, np.uint8)
        d = None if depth is None else np.ascontiguousarray(depth, np.float32)
        n = None if normal is None else np.ascontiguousarray(normal, np.float32)
        c = None if conf is None else np.ascontiguousarray(conf, np.float32)
        for a, shape in ((g, (self.rows, self.cols)), (d, (self.rows, self.cols)), (n, (self.rows, self.cols, 3)), (c, (self.rows, self.cols))):
            assert a is None or a.shape == shape
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_upload(self.ctx._h, self._h, C.c_int(view), _p(g, C.c_ubyte), _p(d, C.c_float), _p(n, C.c_float),
                                                           _p(c, C.c_float)), "pvlm_mvs_views_upload")

    def download(self, view, what=("depth", "normal", "conf", "depth_filter", "conf_filter")):
        out = {k: np.zeros((self.rows, self.cols, 3) if k == "normal" else (self.rows, self.cols), np.float32) for k in what}
        ptr = lambda k: _p(out[k], C.c_float) if k in out else None
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_download(self.ctx._h, self._h, C.c_int(view), ptr("depth"), ptr("normal"), ptr("conf"), ptr("depth_filter"),
                                                             ptr("conf_filter")), "pvlm_mvs_views_download")
        return out

    def snapshot_depth(self, view):
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_snapshot_depth(self.ctx._h, self._h, C.c_int(view)), "pvlm_mvs_views_snapshot_depth")

    def _nb(self, nei, R_nr, t_nr):
        ids = _i32(np.asarray(nei, np.int32).reshape(-1))
        return ids, _f32(R_nr).reshape(-1), _f32(t_nr).reshape(-1)

    def estimate(self, ref, nei, R_nr, t_nr, half_window=3, step=1, use_geometry=False, depth_constant=None, min_depth=0.1, max_depth=20.0, seed=1, max_iter=-1,
                 conf_threshold=-1.0, sequential=False):
        """max_iter < 0: InitConfMap; otherwise EstimateDepthMapSingle with max_iter iterations (checkerboard, or the sequential sweep)."""
        ids, R, t = self._nb(nei, R_nr, t_nr)
        dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
        fn = self.ctx.lib.pvlm_mvs_views_estimate_sequential if (sequential and max_iter >= 0) else self.ctx.lib.pvlm_mvs_views_estimate
        self.ctx._check(fn(self.ctx._h, self._h, C.c_int(ref), C.c_int(len(ids)), _p(ids, C.c_int), _p(R, C.c_float), _p(t, C.c_float),
                                                             C.c_int(half_window), C.c_int(step), C.c_int(1 if use_geometry else 0), _p(dc, C.c_ubyte),
                                                             C.c_float(min_depth), C.c_float(max_depth), C.c_ulonglong(seed), C.c_int(max_iter),
                                                             C.c_float(conf_threshold)), "pvlm_mvs_views_estimate")

    def estimate_sequential_batch(self, jobs, half_window=3, step=1, use_geometry=False, min_depth=0.1, max_depth=20.0, max_iter=1, conf_threshold=-1.0):
        """jobs: list of dicts (ref, nei, R_nr, t_nr, seed[, depth_constant]) with distinct refs: the sequential sweep of all of them, one
        launch per anti-diagonal for the whole batch (pvlm_mvs_views_estimate_sequential_batch)."""
        n = len(jobs)
        refs = _i32([j["ref"] for j in jobs]); cnt = _i32([len(j["nei"]) for j in jobs])
        ids = _i32([b for j in jobs for b in j["nei"]]) if int(cnt.sum()) else np.zeros(1, np.int32)
        R = _f32(np.concatenate([np.asarray(j["R_nr"], np.float32).reshape(-1) for j in jobs])) if int(cnt.sum()) else np.zeros(9, np.float32)
        t = _f32(np.concatenate([np.asarray(j["t_nr"], np.float32).reshape(-1) for j in jobs])) if int(cnt.sum()) else np.zeros(3, np.float32)
        seeds = np.ascontiguousarray([int(j.get("seed", 1)) for j in jobs], np.uint64)
        dcs = [None if j.get("depth_constant") is None else np.ascontiguousarray(j["depth_constant"], np.uint8) for j in jobs]
        dptr = None
        if any(d is not None for d in dcs):
            dptr = (C.POINTER(C.c_ubyte) * max(n, 1))(*[None if d is None else d.ctypes.data_as(C.POINTER(C.c_ubyte)) for d in dcs])
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_estimate_sequential_batch(
            self.ctx._h, self._h, C.c_int(n), _p(refs, C.c_int), _p(cnt, C.c_int), _p(ids, C.c_int), _p(R, C.c_float), _p(t, C.c_float), C.c_int(half_window), C.c_int(step),
            C.c_int(1 if use_geometry else 0), dptr, C.c_float(min_depth), C.c_float(max_depth), seeds.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_int(max_iter),
            C.c_float(conf_threshold)), "pvlm_mvs_views_estimate_sequential_batch")

    def depth_to_cloud(self, view, bgr, T_wc, max_depth=20.0, filter_sky=True, use_filtered_depth=True, with_normal=False):
        """MVS::DepthImageToCloud / DepthNormalToCloud of a resident view (its depth_filter or depth map, its normal map)."""
        c = np.ascontiguousarray(bgr, np.uint8); assert c.shape == (self.rows, self.cols, 3)
        T = np.ascontiguousarray(np.asarray(T_wc, np.float64).reshape(-1)[:12])
        npix = self.rows * self.cols
        xyz = np.empty((npix, 3), np.float32); rgb = np.empty((npix, 3), np.uint8)
        nout = np.empty((npix, 3), np.float32) if with_normal else None
        n = C.c_longlong(0)
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_depth_to_cloud(self.ctx._h, self._h, C.c_int(view), C.c_int(1 if use_filtered_depth else 0), _p(c, C.c_ubyte),
                                                                   _p(T, C.c_double), C.c_float(max_depth), C.c_int(1 if filter_sky else 0), _p(xyz, C.c_float),
                                                                   _p(rgb, C.c_ubyte), _p(nout, C.c_float), C.byref(n)), "pvlm_mvs_views_depth_to_cloud")
        k = n.value
        return (xyz[:k].copy(), rgb[:k].copy()) if nout is None else (xyz[:k].copy(), rgb[:k].copy(), nout[:k].copy())

    def filter_refine(self, ref, nei, R_nr, t_nr, depth_constant=None, thr=0.01, min_depth=0.1, max_depth=20.0):
        ids, R, t = self._nb(nei, R_nr, t_nr)
        dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
        self.ctx._check(self.ctx.lib.pvlm_mvs_views_filter_refine(self.ctx._h, self._h, C.c_int(ref), C.c_int(len(ids)), _p(ids, C.c_int), _p(R, C.c_float),
                                                                  _p(t, C.c_float), _p(dc, C.c_ubyte), C.c_float(thr), C.c_float(min_depth),
                                                                  C.c_float(max_depth)), "pvlm_mvs_views_filter_refine")


class ResidualSet:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        n = C.c_int64(); p = C.c_int(); k = C.c_int(); f = C.c_uint()
        ctx._check(ctx.lib.pvlm_resset_info(self._h, C.byref(n), C.byref(p), C.byref(k), C.byref(f)), "pvlm_resset_info")
        self.n, self.n_pairs, self.kind, self.flags = n.value, p.value, k.value, f.value

    @classmethod
    def upload(cls, ctx, kind, rows, pair_offsets, pair_ref, pair_nei, flags=0, weight=1.0):
        stride = STRIDE.get(kind, np.shape(rows)[-1] if np.ndim(rows) == 2 else 1)   # unknown kinds are rejected by the library
        rows = _f64(rows).reshape(-1, stride) if np.size(rows) else np.zeros((0, stride))
        po = _i64(pair_offsets); pr = _i32(pair_ref); pn = _i32(pair_nei)
        h = C.c_void_p()
        ctx._check(ctx.lib.pvlm_resset_upload(ctx._h, C.c_int(kind), C.c_uint(flags), C.c_double(weight), C.c_int64(rows.shape[0]),
                                              C.c_int(len(pr)), _p(po, C.c_int64), _p(pr, C.c_int), _p(pn, C.c_int),
                                              _p(rows, C.c_double), C.c_int(stride), C.byref(h)), "pvlm_resset_upload")
        return cls(ctx, h)

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_resset_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def download(self):
        po = np.empty(self.n_pairs + 1, np.int64); pr = np.empty(max(self.n_pairs, 1), np.int32); pn = np.empty(max(self.n_pairs, 1), np.int32)
        rows = np.empty((max(self.n, 1), STRIDE[self.kind]), np.float64)
        self.ctx._check(self.ctx.lib.pvlm_resset_download(self.ctx._h, self._h, _p(po, C.c_int64), _p(pr, C.c_int), _p(pn, C.c_int),
                                                          _p(rows, C.c_double)), "pvlm_resset_download")
        return po, pr[:self.n_pairs], pn[:self.n_pairs], rows[:self.n]

    def set_pose_ids(self, pair_ref, pair_nei):
        """Renumbers the segments' poses (one pose table for every set of a problem)."""
        pr = _i32(pair_ref); pn = _i32(pair_nei)
        assert len(pr) == self.n_pairs and len(pn) == self.n_pairs
        self.ctx._check(self.ctx.lib.pvlm_resset_set_pose_ids(self.ctx._h, self._h, _p(pr, C.c_int), _p(pn, C.c_int)), "pvlm_resset_set_pose_ids")

    def eval(self, jac=True):
        r = np.empty(max(self.n, 1), np.float64)
        J = np.empty((max(self.n, 1), 12), np.float64) if jac else None
        self.ctx._check(self.ctx.lib.pvlm_eval(self.ctx._h, self._h, _p(r, C.c_double), _p(J, C.c_double)), "pvlm_eval")
        return r[:self.n], (J[:self.n] if jac else None)

    def eval_dev(self, d_r_ptr, d_J_ptr):
        self.ctx._check(self.ctx.lib.pvlm_eval_dev(self.ctx._h, self._h, C.c_void_p(d_r_ptr), C.c_void_p(d_J_ptr or 0)), "pvlm_eval_dev")

    def pair_blocks(self, loss=LOSS_NONE, loss_a=0.0):
        out = np.empty((max(self.n_pairs, 1), PAIR_BLOCK), np.float64)
        self.ctx._check(self.ctx.lib.pvlm_eval_pair_blocks(self.ctx._h, self._h, C.c_int(loss), C.c_double(loss_a), _p(out, C.c_double)),
                        "pvlm_eval_pair_blocks")
        return out[:self.n_pairs]

    def pair_blocks_dev(self, d_out_ptr, loss=LOSS_NONE, loss_a=0.0):
        self.ctx._check(self.ctx.lib.pvlm_eval_pair_blocks_dev(self.ctx._h, self._h, C.c_int(loss), C.c_double(loss_a), C.c_void_p(d_out_ptr)),
                        "pvlm_eval_pair_blocks_dev")

    def eval_host_async(self, r_host, J_host=None):
        """pvlm_eval_host_async into (preferably pinned) host arrays; complete after ctx.synchronize()."""
        self.ctx._check(self.ctx.lib.pvlm_eval_host_async(self.ctx._h, self._h, C.c_void_p(r_host.ctypes.data),
                                                          C.c_void_p(J_host.ctypes.data if J_host is not None else 0)), "pvlm_eval_host_async")

    def eval_wrench_host_async(self, w_host, tab_host):
        self.ctx._check(self.ctx.lib.pvlm_eval_wrench_host_async(self.ctx._h, self._h, C.c_void_p(w_host.ctypes.data), C.c_void_p(tab_host.ctypes.data)),
                        "pvlm_eval_wrench_host_async")

    def eval_force_host_async(self, f_host, tab_host):
        """[r | g(3)] rows (32 B per block, point functors) + pair tables into page-locked arrays (Context.host_alloc); complete after synchronize()."""
        self.ctx._check(self.ctx.lib.pvlm_eval_force_host_async(self.ctx._h, self._h, _p(f_host, C.c_double), _p(tab_host, C.c_double)), "pvlm_eval_force_host_async")

    def assoc_exact_fits(self):
        """Queries of the association whose plane came from the exact QR because the certified fast fit refused."""
        n = C.c_int64(0)
        self.ctx._check(self.ctx.lib.pvlm_assoc_point2plane_stats(self._h, C.byref(n)), "pvlm_assoc_point2plane_stats")
        return int(n.value)

    def assoc_stats(self):
        n, b, e = C.c_int64(0), C.c_int(0), C.c_int(0)
        self.ctx._check(self.ctx.lib.pvlm_assoc_point2plane_stats2(self._h, C.byref(n), C.byref(b), C.byref(e)), "pvlm_assoc_point2plane_stats2")
        return dict(exact_fits=int(n.value), batches=b.value, exact_kernel_batches=e.value)

    def assoc_debug(self):
        q = np.empty(max(self.n, 1), np.int32); nn = np.empty((max(self.n, 1), 10), np.int32)
        self.ctx._check(self.ctx.lib.pvlm_assoc_point2plane_debug(self.ctx._h, self._h, _p(q, C.c_int32), _p(nn, C.c_int32)),
                        "pvlm_assoc_point2plane_debug")
        return q[:self.n], nn[:self.n]


class NormalEq:
    def __init__(self, ctx, n_poses, upair_i, upair_j):
        self.ctx = ctx
        ui = _i32(upair_i); uj = _i32(upair_j)
        self._h = C.c_void_p()
        ctx._check(ctx.lib.pvlm_neq_create(ctx._h, C.c_int(n_poses), C.c_int(len(ui)), _p(ui, C.c_int), _p(uj, C.c_int), C.byref(self._h)),
                   "pvlm_neq_create")
        self.n_poses, self.n_upairs = n_poses, len(ui)
        self.size = int(ctx.lib.pvlm_neq_size(self._h))

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_neq_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def accumulate(self, rs, loss=LOSS_NONE, loss_a=0.0, packed=None):
        zero_first = packed is None
        buf = np.zeros(self.size, np.float64) if packed is None else _f64(packed)
        self.ctx._check(self.ctx.lib.pvlm_neq_accumulate(self.ctx._h, self._h, rs._h, C.c_int(loss), C.c_double(loss_a),
                                                         C.c_int(1 if zero_first else 0), _p(buf, C.c_double)), "pvlm_neq_accumulate")
        return buf

    def accumulate_dev(self, rs, d_packed_ptr, loss=LOSS_NONE, loss_a=0.0, zero_first=True):
        self.ctx._check(self.ctx.lib.pvlm_neq_accumulate_dev(self.ctx._h, self._h, rs._h, C.c_int(loss), C.c_double(loss_a),
                                                             C.c_int(1 if zero_first else 0), C.c_void_p(d_packed_ptr)), "pvlm_neq_accumulate_dev")

    def accumulate_async(self, rs, packed, loss=LOSS_NONE, loss_a=0.0):
        """Queues the linearisation of `rs` into this structure's own device buffer and its copy into `packed` (float64 array of
        `size` elements that must stay alive); complete after ctx.synchronize().  One call per residual set of a problem, one
        synchronisation for all of them."""
        assert packed.dtype == np.float64 and packed.flags["C_CONTIGUOUS"] and packed.size == self.size
        self.ctx._check(self.ctx.lib.pvlm_neq_accumulate_async(self.ctx._h, self._h, rs._h, C.c_int(loss), C.c_double(loss_a),
                                                               packed.ctypes.data_as(C.POINTER(C.c_double))), "pvlm_neq_accumulate_async")

    @staticmethod
    def accumulate_sets(ctx, neqs, sets, losses, loss_as, packed):
        """All sets linearised and summed on the device into neqs[0]'s buffer (identical structures), one queued copy into `packed`;
        complete after ctx.synchronize()."""
        n = len(neqs)
        assert packed.dtype == np.float64 and packed.flags["C_CONTIGUOUS"] and packed.size == neqs[0].size
        qa = (C.c_void_p * n)(*[q._h for q in neqs]); ra = (C.c_void_p * n)(*[r._h for r in sets])
        la = (C.c_int * n)(*[int(l) for l in losses]); aa = (C.c_double * n)(*[float(a) for a in loss_as])
        ctx._check(ctx.lib.pvlm_neq_accumulate_sets(ctx._h, C.c_int(n), qa, ra, la, aa, packed.ctypes.data_as(C.POINTER(C.c_double))), "pvlm_neq_accumulate_sets")

    def unpack(self, packed):
        n, u = self.n_poses, self.n_upairs
        Hd = packed[:n * 36].reshape(n, 6, 6); Ho = packed[n * 36:(n + u) * 36].reshape(u, 6, 6)
        g = packed[(n + u) * 36:(n + u) * 36 + n * 6].reshape(n, 6)
        return Hd, Ho, g, float(packed[-1])


class BundleSet:
    """Reprojection blocks (PanoramaReprojResidual_1Angle) with the 3-D points resident on the GPU and eliminated
    there (pvlm_ba_* of include/pvlm.h).  Camera poses come from Context.set_poses (angleAxis_cw, t_cw)."""

    def __init__(self, ctx, point_offsets, cam_ids, bearings, points, weight=1.0):
        self.ctx = ctx
        off = _i64(point_offsets); cam = _i32(cam_ids); b = _f64(bearings); X = _f64(points)
        self._h = C.c_void_p()
        ctx._check(ctx.lib.pvlm_ba_create(ctx._h, C.c_int(len(off) - 1), C.c_int64(len(cam)), _p(off, C.c_int64), _p(cam, C.c_int),
                                          _p(b, C.c_double), _p(X, C.c_double), C.c_double(weight), C.byref(self._h)), "pvlm_ba_create")
        npts = C.c_int(); nobs = C.c_int64(); ncam = C.c_int(); nup = C.c_int()
        ctx.lib.pvlm_ba_structure(self._h, C.byref(npts), C.byref(nobs), C.byref(ncam), C.byref(nup), None, None)
        self.n_points, self.n_obs, self.n_cams, self.n_upairs = npts.value, nobs.value, ncam.value, nup.value
        self.ui = np.zeros(self.n_upairs, np.int32); self.uj = np.zeros(self.n_upairs, np.int32)
        ctx.lib.pvlm_ba_structure(self._h, None, None, None, None, _p(self.ui, C.c_int), _p(self.uj, C.c_int))
        self.size = int(ctx.lib.pvlm_ba_packed_size(self._h))

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_ba_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def points(self, candidate=False):
        X = np.zeros((self.n_points, 3), np.float64)
        self.ctx._check(self.ctx.lib.pvlm_ba_get_points(self.ctx._h, self._h, C.c_int(1 if candidate else 0), _p(X, C.c_double)), "pvlm_ba_get_points")
        return X

    def set_points(self, X):
        X = _f64(X)
        assert X.shape == (self.n_points, 3)
        self.ctx._check(self.ctx.lib.pvlm_ba_set_points(self.ctx._h, self._h, _p(X, C.c_double)), "pvlm_ba_set_points")

    def set_constant(self, mask):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        assert m is None or m.shape == (self.n_points,)
        self.ctx._check(self.ctx.lib.pvlm_ba_set_constant(self.ctx._h, self._h, _p(m, C.c_ubyte)), "pvlm_ba_set_constant")

    def evaluate(self, jac=True):
        r = np.zeros(self.n_obs, np.float64)
        J = np.zeros((self.n_obs, 9), np.float64) if jac else None
        self.ctx._check(self.ctx.lib.pvlm_ba_eval(self.ctx._h, self._h, _p(r, C.c_double), _p(J, C.c_double)), "pvlm_ba_eval")
        return r, J

    def reduce(self, loss=LOSS_NONE, loss_a=0.0, init_scale=False, radius=1e4, min_diag=1e-6, max_diag=1e32):
        packed = np.zeros(self.size, np.float64)
        self.ctx._check(self.ctx.lib.pvlm_ba_reduce(self.ctx._h, self._h, C.c_int(loss), C.c_double(loss_a), C.c_int(1 if init_scale else 0),
                                                    C.c_double(radius), C.c_double(min_diag), C.c_double(max_diag), _p(packed, C.c_double)), "pvlm_ba_reduce")
        return packed

    def step(self, dcam, loss=LOSS_NONE, loss_a=0.0):
        d = _f64(dcam)
        assert d.shape == (self.n_cams, 6)
        out3 = np.zeros(3, np.float64)
        self.ctx._check(self.ctx.lib.pvlm_ba_step(self.ctx._h, self._h, C.c_int(loss), C.c_double(loss_a), _p(d, C.c_double), _p(out3, C.c_double)), "pvlm_ba_step")
        return out3

    def cost(self, loss=LOSS_NONE, loss_a=0.0, candidate=False):
        c = C.c_double()
        self.ctx._check(self.ctx.lib.pvlm_ba_cost(self.ctx._h, self._h, C.c_int(loss), C.c_double(loss_a), C.c_int(1 if candidate else 0), C.byref(c)), "pvlm_ba_cost")
        return c.value

    def accept(self):
        self.ctx._check(self.ctx.lib.pvlm_ba_accept(self.ctx._h, self._h), "pvlm_ba_accept")


class Comm:
    """RCCL communicator of the C ABI (for hosts without their own collective layer)."""

    def __init__(self, ctx, world_size, rank, unique_id=None):
        self.ctx = ctx
        if unique_id is None:
            buf = (C.c_ubyte * 128)()
            ctx._check(ctx.lib.pvlm_comm_unique_id(ctx._h, buf), "pvlm_comm_unique_id")
            unique_id = bytes(buf)
        self.unique_id = unique_id
        self._h = C.c_void_p()
        ctx._check(ctx.lib.pvlm_comm_create(ctx._h, C.c_int(world_size), C.c_int(rank), (C.c_ubyte * 128).from_buffer_copy(unique_id),
                                            C.byref(self._h)), "pvlm_comm_create")

    def allreduce_sum_f64(self, d_ptr, count):
        self.ctx._check(self.ctx.lib.pvlm_allreduce_sum_f64(self.ctx._h, self._h, C.c_void_p(d_ptr), C.c_int64(count)), "pvlm_allreduce_sum_f64")

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_comm_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()


class ScanDesc(C.Structure):
    _fields_ = [
        ("id", C.c_int), ("R_wl", C.POINTER(C.c_double)), ("t_wl", C.POINTER(C.c_double)),
        ("n_surf_flat", C.c_int), ("surf_flat_xyz", C.POINTER(C.c_float)), ("surf_flat_tag", C.POINTER(C.c_float)),
        ("n_surf_less_flat", C.c_int), ("surf_less_flat_xyz", C.POINTER(C.c_float)), ("surf_less_flat_tag", C.POINTER(C.c_float)),
        ("n_corner", C.c_int), ("corner_xyz", C.POINTER(C.c_float)),
        ("p2s_offsets", C.POINTER(C.c_int)), ("p2s_ids", C.POINTER(C.c_int)),
        ("n_segments", C.c_int), ("segment_size", C.POINTER(C.c_int)),
        ("segment_coeffs", C.POINTER(C.c_double)), ("end_points", C.POINTER(C.c_double)),
        ("seg_points_xyz", C.POINTER(C.c_float)), ("point_stride_floats", C.c_int),
    ]


class Scan:
    """Device-resident feature clouds of one LiDAR scan (the sensors/Velodyne.h:80-91 contract).
    `scan` is a dict: id, R_wl, t_wl, flat_xyz, flat_tag, less_xyz, less_tag, corner_xyz, p2s,
    seg_size, seg_coeffs, end_points (all optional except the pose)."""

    @staticmethod
    def _describe(scan):
        """ScanDesc of a scan dict + the arrays it points into (to be kept alive across the call)."""
        g = scan.get
        R = _f64(g("R_wl", np.eye(3))).reshape(9); t = _f64(g("t_wl", np.zeros(3)))
        flat = _f32(g("flat_xyz", np.zeros((0, 3)))); flat_tag = _f32(g("flat_tag", np.ones(len(flat))))
        less = _f32(g("less_xyz", np.zeros((0, 3)))); less_tag = _f32(g("less_tag", np.ones(len(less))))
        corner = _f32(g("corner_xyz", np.zeros((0, 3))))
        p2s = g("p2s", None)
        if p2s is None:
            p2s = [[] for _ in range(len(corner))]
        off = np.zeros(len(p2s) + 1, np.int32)
        for i, l in enumerate(p2s):
            off[i + 1] = off[i] + len(l)
        ids = _i32([v for l in p2s for v in l]) if off[-1] > 0 else np.zeros(1, np.int32)
        seg_size = _i32(g("seg_size", np.zeros(0)))
        seg_coeffs = _f64(g("seg_coeffs", np.zeros((len(seg_size), 6))))
        end_points = _f64(g("end_points", np.zeros((len(seg_size), 6))))
        d = ScanDesc()
        d.id = int(g("id", 0)); d.R_wl = _p(R, C.c_double); d.t_wl = _p(t, C.c_double)
        d.n_surf_flat = len(flat); d.surf_flat_xyz = _p(flat, C.c_float); d.surf_flat_tag = _p(flat_tag, C.c_float)
        d.n_surf_less_flat = len(less); d.surf_less_flat_xyz = _p(less, C.c_float); d.surf_less_flat_tag = _p(less_tag, C.c_float)
        d.n_corner = len(corner); d.corner_xyz = _p(corner, C.c_float)
        d.p2s_offsets = _p(off, C.c_int); d.p2s_ids = _p(ids, C.c_int)
        d.n_segments = len(seg_size); d.segment_size = _p(seg_size, C.c_int)
        d.segment_coeffs = _p(seg_coeffs, C.c_double); d.end_points = _p(end_points, C.c_double)
        seg_xyz = g("seg_points_xyz", None)          # optional: the points of every segment, concatenated (world frame)
        seg_xyz = _f32(seg_xyz).reshape(-1, 3) if seg_xyz is not None else None
        if seg_xyz is not None:
            assert len(seg_xyz) == int(seg_size.sum())
        d.seg_points_xyz = _p(seg_xyz, C.c_float) if seg_xyz is not None and len(seg_xyz) else None
        keep = (R, t, flat, flat_tag, less, less_tag, corner, off, ids, seg_size, seg_coeffs, end_points, seg_xyz)
        if g("point_records", False):
            # the clouds handed over as pcl::PointXYZI-style records {x, y, z, intensity} (point_stride_floats = 4): xyz and tag point INTO one n x 4 array per cloud
            rec = [np.ascontiguousarray(np.concatenate([xyz, tag[:, None]], axis=1), np.float32) if len(xyz) else np.zeros((0, 4), np.float32)
                   for xyz, tag in ((flat, flat_tag), (less, less_tag), (corner, np.zeros(len(corner), np.float32)))]
            at = lambda a, k: C.cast(a.ctypes.data + 4 * k, C.POINTER(C.c_float)) if len(a) else None
            d.point_stride_floats = 4
            d.surf_flat_xyz, d.surf_flat_tag = at(rec[0], 0), at(rec[0], 3)
            d.surf_less_flat_xyz, d.surf_less_flat_tag = at(rec[1], 0), at(rec[1], 3)
            d.corner_xyz = at(rec[2], 0)
            keep = keep + tuple(rec)
        return d, keep

    def _adopt(self, ctx, d, handle):
        self.ctx = ctx
        self._h = handle
        self.id = d.id
        self.n_segments = d.n_segments
        self.n_flat, self.n_less, self.n_corner = d.n_surf_flat, d.n_surf_less_flat, d.n_corner

    def __init__(self, ctx, scan):
        d, keep = self._describe(scan)
        h = C.c_void_p()
        self._h = None
        ctx._check(ctx.lib.pvlm_scan_upload(ctx._h, C.byref(d), C.byref(h)), "pvlm_scan_upload")
        self._adopt(ctx, d, h)

    @classmethod
    def upload_batch(cls, ctx, scans):
        """pvlm_scan_upload_batch: every scan of `scans` (dicts) in one staging copy / one device slab / one grid build."""
        pairs = [cls._describe(s) for s in scans]
        n = len(pairs)
        descs = (ScanDesc * max(n, 1))()
        for k, (d, _) in enumerate(pairs):
            descs[k] = d
        handles = (C.c_void_p * max(n, 1))()
        ctx._check(ctx.lib.pvlm_scan_upload_batch(ctx._h, n, descs, handles), "pvlm_scan_upload_batch")
        out = []
        for k, (d, _) in enumerate(pairs):
            o = cls.__new__(cls)
            o._adopt(ctx, d, C.c_void_p(handles[k]))
            out.append(o)
        return out

    @staticmethod
    def transform_batch(ctx, scans, T, rebuild_grids=True):
        """pvlm_scan_transform_batch: scan k's resident float clouds are replaced by T[k] (3x4 or 4x4, [R | t]) applied as
        pcl::transformPointCloud applies an Eigen::Matrix4d; rebuild_grids: the voxel grids follow."""
        n = len(scans)
        T12 = np.ascontiguousarray(np.stack([np.asarray(t, np.float64)[:3, :4] for t in T]).reshape(n, 12)) if n else np.zeros((0, 12))
        handles = (C.c_void_p * max(n, 1))(*[s._h for s in scans])
        ctx._check(ctx.lib.pvlm_scan_transform_batch(ctx._h, n, handles, _p(T12, C.c_double), 1 if rebuild_grids else 0), "pvlm_scan_transform_batch")

    def set_pose(self, R_wl, t_wl):
        R = _f64(R_wl).reshape(9); t = _f64(t_wl).reshape(3)
        self.ctx._check(self.ctx.lib.pvlm_scan_set_pose(self.ctx._h, self._h, _p(R, C.c_double), _p(t, C.c_double)), "pvlm_scan_set_pose")

    def cloud_info(self, which):
        info = GridInfo()
        self.ctx._check(self.ctx.lib.pvlm_scan_cloud_info(self._h, int(which), C.byref(info)), "pvlm_scan_cloud_info")
        return info

    def fetch_cloud(self, which, grid=False):
        """The resident floats of cloud `which` (0 surfFlat, 1 surfLessFlat, 2 cornerLessSharp, 3 segment points) and, with grid=True, a
        CANONICAL form of its voxel grid: the plan and the cell-sorted records ordered by (cell — dense: cell index, hashed: 64-bit key —, original
        index): cell_key, index, points — independent of the slot / in-cell order the atomics of a build produce."""
        info = self.cloud_info(which)
        xyz = np.zeros((info.n, 3), np.float32)
        if not grid or not info.has_grid:
            self.ctx._check(self.ctx.lib.pvlm_scan_cloud_fetch(self.ctx._h, self._h, int(which), _p(xyz, C.c_float), None, None, None, None), "pvlm_scan_cloud_fetch")
            return xyz if not grid else (xyz, None)
        T = info.table_size
        count = np.zeros(T, np.int32); start = np.zeros(T, np.int32); keys = np.zeros(T, np.uint64); srt = np.zeros((info.n, 4), np.float32)
        self.ctx._check(self.ctx.lib.pvlm_scan_cloud_fetch(self.ctx._h, self._h, int(which), _p(xyz, C.c_float), _p(count, C.c_int), _p(start, C.c_int),
                                                           _p(keys, C.c_uint64) if not info.dense else None, _p(srt, C.c_float)), "pvlm_scan_cloud_fetch")
        plan = dict(n=info.n, dense=info.dense, nx=info.nx, ny=info.ny, nz=info.nz, xf=info.xf, table_size=T, cell=np.float32(info.cell).tobytes(),
                    origin=np.asarray(list(info.origin), np.float32).tobytes(), stale=info.stale)
        # per sorted position the key of its cell: the occupied slots in order of their start cover [0, n) without gaps
        slots = np.nonzero(count)[0]
        slots = slots[np.argsort(start[slots], kind="stable")]
        assert int(count[slots].sum()) == info.n and np.array_equal(start[slots], np.concatenate([[0], np.cumsum(count[slots])[:-1]]))
        cell_key = np.repeat(slots.astype(np.uint64) if info.dense else keys[slots], count[slots])
        idx = srt[:, 3].view(np.int32)
        order = np.lexsort((idx, cell_key))
        return xyz, dict(plan=plan, cell_key=cell_key[order], index=idx[order].copy(), points=srt[order, :3].copy())

    def close(self):
        if self._h:
            self.ctx.lib.pvlm_scan_destroy(self.ctx._h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridInfo(C.Structure):
    _fields_ = [("n", C.c_int), ("has_grid", C.c_int), ("stale", C.c_int), ("dense", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("xf", C.c_int), ("table_size", C.c_int), ("cell", C.c_float), ("origin", C.c_float * 3)]


class UndistortScanDesc(C.Structure):
    _fields_ = [("xyzi", C.POINTER(C.c_float)), ("n", C.c_int), ("stride_floats", C.c_int), ("R_wl", C.POINTER(C.c_double)), ("t_wl", C.POINTER(C.c_double)),
                ("R_we", C.POINTER(C.c_double)), ("t_we", C.POINTER(C.c_double))]


def undistort_batch(ctx, clouds, start_poses, end_poses, inplace=False):
    """pvlm_undistort_batch: clouds — list of n x 4 float32 arrays; start_poses / end_poses — lists of (R 3x3, t 3), world <- sensor.  Returns the
    motion-compensated copies (Velodyne::UndistortCloud); inplace=True: the arrays themselves (C-contiguous float32 n x 4) are updated, as the C call does."""
    if inplace:
        out = list(clouds)
        for c in out:
            assert c.dtype == np.float32 and c.flags["C_CONTIGUOUS"] and c.ndim == 2 and c.shape[1] == 4
    else:
        out = [np.ascontiguousarray(c, np.float32).reshape(-1, 4).copy() for c in clouds]
    keep = []
    descs = (UndistortScanDesc * max(len(out), 1))()
    for k, c in enumerate(out):
        arrs = [_f64(start_poses[k][0]).reshape(9), _f64(start_poses[k][1]).reshape(3), _f64(end_poses[k][0]).reshape(9), _f64(end_poses[k][1]).reshape(3)]
        keep.append(arrs)
        descs[k].xyzi = _p(c, C.c_float); descs[k].n = len(c); descs[k].stride_floats = 4
        descs[k].R_wl, descs[k].t_wl, descs[k].R_we, descs[k].t_we = (_p(a, C.c_double) for a in arrs)
    ctx._check(ctx.lib.pvlm_undistort_batch(ctx._h, C.c_int(len(out)), descs), "pvlm_undistort_batch")
    return out


def device_sort(ctx, keys):
    """pvlm_ring_debug_sort: the permutation the device's std::sort restatement leaves for uint32 keys (tests)."""
    keys = np.ascontiguousarray(keys, np.uint32)
    order = np.zeros(len(keys), np.int32)
    ctx._check(ctx.lib.pvlm_ring_debug_sort(ctx._h, _p(keys, C.c_uint), C.c_int(len(keys)), _p(order, C.c_int)), "pvlm_ring_debug_sort")
    return order


class RawScanDesc(C.Structure):
    _fields_ = [("xyzi", C.POINTER(C.c_float)), ("n", C.c_int), ("stride_floats", C.c_int)]


class RingResultDesc(C.Structure):
    _fields_ = [("n_raw", C.c_int), ("n_reordered", C.c_int), ("n_kept", C.c_int), ("resolved_points", C.c_int), ("resolved_edges", C.c_int), ("replayed", C.c_int),
                ("ring_count_reordered", C.POINTER(C.c_int)), ("ring_count", C.POINTER(C.c_int)), ("source", C.POINTER(C.c_int)), ("ring_col", C.POINTER(C.c_int)),
                ("curvature", C.POINTER(C.c_float)), ("half_window", C.POINTER(C.c_int)), ("range", C.POINTER(C.c_float)), ("sorted", C.POINTER(C.c_int)),
                ("sector_host", C.POINTER(C.c_ubyte)), ("picks", C.c_int), ("max_curvature", C.c_float), ("intersect_angle_threshold", C.c_float),
                ("state", C.POINTER(C.c_ubyte)), ("corner", C.POINTER(C.c_int)), ("flat", C.POINTER(C.c_int)), ("voxel_span", C.POINTER(C.c_int)),
                ("voxels", C.POINTER(C.c_float)), ("ring_host", C.POINTER(C.c_ubyte))]


class RingBatch:
    """pvlm_ring_extract_batch: ReOrderVLP + Segmentation + adaptive curvature of a batch of raw scans (n x 4 float32 each) on the GPU."""

    def __init__(self, ctx, raw_scans, n_rings=16, horizon=1800, segment=True, picks=None, keep_arrays=True):
        """picks = (max_curvature, intersect_angle_threshold): also run K24 (feature picks + voxel grid) — see RingBatch.picks().  keep_arrays=False (with picks): the
        per-point arrays curvature / half_window / range / sorted come down only for scans with a ring left to the host, as the C++ host mirror asks for them."""
        self.ctx = ctx
        self._raw = [_f32(r).reshape(-1, 4) for r in raw_scans]
        self.n_rings, self.horizon = n_rings, horizon
        descs = (RawScanDesc * max(len(self._raw), 1))()
        for k, r in enumerate(self._raw):
            descs[k].xyzi = _p(r, C.c_float); descs[k].n = len(r); descs[k].stride_floats = 4
        self._h = C.c_void_p()
        if picks is None:
            ctx._check(ctx.lib.pvlm_ring_extract_batch(ctx._h, C.c_int(len(self._raw)), descs, C.c_int(n_rings), C.c_int(horizon), C.c_int(1 if segment else 0),
                                                       C.byref(self._h)), "pvlm_ring_extract_batch")
        else:
            ctx._check(ctx.lib.pvlm_ring_extract_batch_picks(ctx._h, C.c_int(len(self._raw)), descs, C.c_int(n_rings), C.c_int(horizon), C.c_int((1 if segment else 0) | (2 if keep_arrays else 0)),
                                                             C.c_float(picks[0]), C.c_float(picks[1]), C.byref(self._h)), "pvlm_ring_extract_batch_picks")

    def close(self):
        if self._h and self.ctx._h:
            self.ctx.lib.pvlm_ring_batch_destroy(self.ctx._h, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def timing(self):
        ms = np.zeros(8)
        self.ctx._check(self.ctx.lib.pvlm_ring_batch_timing(self._h, _p(ms, C.c_double)), "pvlm_ring_batch_timing")
        return dict(zip(("upload", "classify", "columns", "scatter", "edges", "components", "compact_curvature", "download"), ms.tolist()))

    def result(self, scan):
        """The kept cloud of one scan as numpy copies (the picks of the host read these arrays in place)."""
        r = RingResultDesc()
        self.ctx._check(self.ctx.lib.pvlm_ring_batch_scan(self._h, C.c_int(scan), C.byref(r)), "pvlm_ring_batch_scan")
        m = r.n_kept
        arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n > 0 and ptr else np.zeros(0, dt)
        out = dict(n_raw=r.n_raw, n_reordered=r.n_reordered, n_kept=m, resolved_points=r.resolved_points, resolved_edges=r.resolved_edges, replayed=r.replayed,
                   ring_count_reordered=arr(r.ring_count_reordered, 64, np.int32), ring_count=arr(r.ring_count, 64, np.int32), source=arr(r.source, m, np.int32),
                   ring_col=arr(r.ring_col, m, np.int32), curvature=arr(r.curvature, m, np.float32), half_window=arr(r.half_window, m, np.int32),
                   range=arr(r.range, m, np.float32), sorted=arr(r.sorted, m, np.int32), sector_host=arr(r.sector_host, self.n_rings * 6, np.uint8))
        return out

    def picks(self, scan):
        """K24's results for one scan: dict(ring_host (n_rings flags), state (per kept point), corner / sharp (edge picks in upstream's order and which of them are
        sharp), flat (plane picks), less_flat (m x 4 centroids, ring by ring)) — or None when the batch was made without picks."""
        r = RingResultDesc()
        self.ctx._check(self.ctx.lib.pvlm_ring_batch_scan(self._h, C.c_int(scan), C.byref(r)), "pvlm_ring_batch_scan")
        if not r.picks:
            return None
        R = self.n_rings
        ring_host = np.ctypeslib.as_array(r.ring_host, shape=(R,)).copy()
        state = np.ctypeslib.as_array(r.state, shape=(max(r.n_kept, 1),))[:r.n_kept].copy()
        corner_t = np.ctypeslib.as_array(r.corner, shape=(R, 181)); flat_t = np.ctypeslib.as_array(r.flat, shape=(R, 25)); span = np.ctypeslib.as_array(r.voxel_span, shape=(R, 2))
        corner = np.concatenate([corner_t[k, 1:1 + corner_t[k, 0]] for k in range(R)]) if R else np.zeros(0, np.int32)
        flat = np.concatenate([flat_t[k, 1:1 + flat_t[k, 0]] for k in range(R)]) if R else np.zeros(0, np.int32)
        less = [np.ctypeslib.as_array(C.cast(C.addressof(r.voxels.contents) + 16 * int(span[k, 0]), C.POINTER(C.c_float)), shape=(int(span[k, 1]), 4)).copy()
                for k in range(R) if span[k, 1] > 0]
        return dict(ring_host=ring_host, state=state, corner=(corner & 0x7FFFFFFF).astype(np.int32), sharp=(corner.view(np.uint32) >> 31).astype(bool),
                    flat=flat.astype(np.int32), less_flat=np.concatenate(less) if less else np.zeros((0, 4), np.float32))

    def fetch(self, scan, state):
        """Device-resident arrays of one scan: state 0 after ReOrderVLP, 1 after Segmentation."""
        res = self.result(scan)
        n = res["n_kept"] if state else res["n_reordered"]
        cloud = np.zeros((max(n, 1), 4), np.float32); rc = np.zeros((max(n, 1), 2), np.int32)
        image = np.zeros((self.n_rings, self.horizon), np.float32); i2p = np.zeros((self.n_rings, self.horizon), np.int32)
        self.ctx._check(self.ctx.lib.pvlm_ring_batch_fetch(self.ctx._h, self._h, C.c_int(scan), C.c_int(state), _p(cloud, C.c_float), _p(rc, C.c_int), _p(image, C.c_float),
                                                           _p(i2p, C.c_int)), "pvlm_ring_batch_fetch")
        return cloud[:n], rc[:n], image, i2p

    def arrays(self, scan):
        """Everything tests/ring_cases.assert_matches_oracle compares, for one scan."""
        res = self.result(scan)
        c0, rc0, image, i2p0 = self.fetch(scan, 0)
        c1, rc1, _, i2p1 = self.fetch(scan, 1)
        return dict(n_reordered=res["n_reordered"], n_kept=res["n_kept"], cloud_reordered=c0, rc_reordered=rc0, range_image=image, image_to_point_reordered=i2p0,
                    ring_count_reordered=res["ring_count_reordered"], cloud_kept=c1, rc_kept=rc1, image_to_point_kept=i2p1, ring_count=res["ring_count"],
                    curvature=res["curvature"], half_window=res["half_window"], range=res["range"], source=res["source"], ring_col=res["ring_col"],
                    resolved_points=res["resolved_points"], resolved_edges=res["resolved_edges"], replayed=res["replayed"], sorted=res["sorted"],
                    sector_host=res["sector_host"])
