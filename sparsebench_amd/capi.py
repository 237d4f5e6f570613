"""ctypes binding of the C-ABI HIP layer (include/sbhip.h -> lib/libsbhip.so).

This is plumbing only: every call goes straight into the shared library.  There is
no CPU fallback -- if the library is missing, or no MI355X is visible, using the
hot path raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBHIP_LIBRARY") or os.path.join(HERE, "lib", "libsbhip.so")  # override: lab builds

# every symbol include/sbhip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "sb_init", "sb_finalize", "sb_is_initialized", "sb_device_count", "sb_device_name",
    "sb_num_cus", "sb_sync", "sb_stream", "sb_malloc", "sb_free", "sb_memset", "sb_h2d",
    "sb_d2h", "sb_d2d", "sb_is_device_ptr", "sb_event_create", "sb_event_record",
    "sb_event_elapsed_ms", "sb_event_destroy", "sb_crs_upload", "sb_scs_upload",
    "sb_matrix_free", "sb_matrix_nr", "sb_matrix_nc", "sb_matrix_is_permuted",
    "sb_matrix_spmv_bytes", "sb_spmv", "sb_spmv_native", "sb_permute", "sb_unpermute",
    "sb_waxpby", "sb_ddot_async", "sb_ddot", "sb_ddot_partials", "sb_reduce_final",
    "sb_comm_unique_id", "sb_comm_init", "sb_comm_finalize", "sb_comm_rank", "sb_comm_size",
    "sb_comm_reduction", "sb_halo_create", "sb_halo_free", "sb_halo_exchange", "sb_cg_create",
    "sb_cg_free", "sb_cg_set_fused", "sb_cg_set_graph", "sb_cg_solve", "sb_cg_run_iters",
    "sb_cg_history", "sb_cg_solution", "sb_cg_check_residual", "sb_cg_region_ms", "sb_version",
    "sb_comm_allgather_bytes", "sb_comm_alltoallv_ints", "sb_comm_barrier", "sb_cg_loop_ms",
    "sb_cg_spmv_timing", "sb_cg_spmv_ms", "sb_cg_spmv_us_series", "sb_cg_counters", "sb_debug_stream_read_gbs",
    "sb_matrix_pack_level", "sb_set_external_ids", "sb_spmv_native_dot", "sb_matrix_use_packed", "sb_matrix_stream_bytes",
    "sb_matrix_packed_mode", "sb_matrix_crs_kernel", "sb_matrix_lds_window", "sb_matrix_pattern_classes", "sb_matrix_row_patterns", "sb_matrix_row_programs", "sb_comm_p2p_handle", "sb_comm_p2p_open", "sb_comm_p2p_enabled", "sb_halo_p2p_enabled", "sb_cg_start", "sb_cg_finish", "sb_cg_vector_phase", "sb_cg_launches_per_body",
    "sb_comm_init_transport", "sb_comm_p2p_reason", "sb_halo_p2p_reason",
    "sb_comm_data_plane", "sb_comm_data_plane_selected", "sb_comm_rccl_info", "sb_cg_phase_timing", "sb_cg_phase_ms",
    "sb_comm_halo_push_inside", "sb_lab_build", "sb_cg_collectives_per_body",
    "sb_cg_set_fuse_p", "sb_cg_fuse_p", "sb_cg_set_fuse_alpha", "sb_cg_set_fuse_beta",
    "sb_malloc_host_visible", "sb_host_visible_reason", "sb_malloc_pinned_host", "sb_free_pinned_host", "sb_copy_counters",
    "sb_region_begin", "sb_region_end", "sb_region_seconds", "sb_region_reset",
    "sb_matrix_place", "sb_matrix_place_at", "sb_matrix_place_home", "sb_placement_arena_bytes", "sb_placement_probe", "sb_matrix_place_fresh", "sb_matrix_place_commit", "sb_matrix_placement", "sb_matrix_placement_report", "sb_matrix_debug_ptrs", "sb_cg_debug_ptrs",
]

_lib = None
vp = C.c_void_p
u32 = C.c_uint32


def load():
    """Load libsbhip.so (no device is touched until sb_init)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "sparsebench_amd: %s is missing -- build it with `make hip` (hipcc, gfx950). "
            "There is no CPU fallback for the hot path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    sig = {
        "sb_init": (None, [C.c_int]),
        "sb_finalize": (None, []),
        "sb_is_initialized": (C.c_int, []),
        "sb_device_count": (C.c_int, []),
        "sb_device_name": (C.c_char_p, []),
        "sb_num_cus": (C.c_int, []),
        "sb_sync": (None, []),
        "sb_stream": (vp, []),
        "sb_malloc": (vp, [C.c_size_t]),
        "sb_free": (None, [vp]),
        "sb_memset": (None, [vp, C.c_int, C.c_size_t]),
        "sb_h2d": (None, [vp, vp, C.c_size_t]),
        "sb_d2h": (None, [vp, vp, C.c_size_t]),
        "sb_d2d": (None, [vp, vp, C.c_size_t]),
        "sb_is_device_ptr": (C.c_int, [vp]),
        "sb_event_create": (vp, []),
        "sb_event_record": (None, [vp]),
        "sb_event_elapsed_ms": (C.c_float, [vp, vp]),
        "sb_event_destroy": (None, [vp]),
        "sb_crs_upload": (vp, [u32, u32, vp, vp, vp]),
        "sb_scs_upload": (vp, [u32, u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp]),
        "sb_matrix_free": (None, [vp]),
        "sb_matrix_nr": (u32, [vp]),
        "sb_matrix_nc": (u32, [vp]),
        "sb_matrix_is_permuted": (C.c_int, [vp]),
        "sb_matrix_spmv_bytes": (C.c_double, [vp]),
        "sb_spmv": (None, [vp, vp, vp]),
        "sb_spmv_native": (None, [vp, vp, vp]),
        "sb_permute": (None, [vp, vp, vp]),
        "sb_unpermute": (None, [vp, vp, vp]),
        "sb_waxpby": (None, [u32, C.c_double, vp, C.c_double, vp, vp]),
        "sb_ddot_async": (None, [u32, vp, vp, vp]),
        "sb_ddot": (C.c_double, [u32, vp, vp]),
        "sb_ddot_partials": (None, [u32, vp, vp, vp]),
        "sb_reduce_final": (None, [u32, vp, vp]),
        "sb_comm_unique_id": (None, [vp]),
        "sb_comm_init": (None, [C.c_int, C.c_int, vp]),
        "sb_comm_finalize": (None, []),
        "sb_comm_rank": (C.c_int, []),
        "sb_comm_size": (C.c_int, []),
        "sb_comm_reduction": (None, [vp, C.c_int]),
        "sb_halo_create": (vp, [u32, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int,
                                C.c_int, vp]),
        "sb_halo_free": (None, [vp]),
        "sb_halo_exchange": (None, [vp, vp]),
        "sb_cg_create": (vp, [vp, vp, vp, vp]),
        "sb_cg_free": (None, [vp]),
        "sb_cg_set_fused": (None, [vp, C.c_int]),
        "sb_cg_set_graph": (None, [vp, C.c_int]),
        "sb_cg_solve": (C.c_int, [vp, C.c_int, C.c_double]),
        "sb_cg_run_iters": (None, [vp, C.c_int]),
        "sb_cg_history": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]),
        "sb_cg_solution": (None, [vp, vp]),
        "sb_cg_check_residual": (C.c_double, [vp]),
        "sb_cg_region_ms": (None, [vp, vp]),
        "sb_version": (C.c_char_p, []),
        "sb_comm_allgather_bytes": (None, [vp, C.c_int, vp]),
        "sb_comm_alltoallv_ints": (None, [vp, vp, vp, vp, vp, vp]),
        "sb_comm_barrier": (None, []),
        "sb_cg_loop_ms": (C.c_double, [vp]),
        "sb_cg_spmv_timing": (None, [vp, C.c_int]),
        "sb_cg_spmv_ms": (C.c_double, [vp, C.POINTER(C.c_int)]),
        "sb_cg_spmv_us_series": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int]),
        "sb_cg_counters": (None, [vp, vp]),
        "sb_debug_stream_read_gbs": (C.c_double, [C.c_size_t, C.c_int]),
        "sb_matrix_pack_level": (C.c_int, [vp]),
        "sb_matrix_use_packed": (None, [vp, C.c_int]),
        "sb_set_external_ids": (None, [vp, C.c_uint32]),
        "sb_spmv_native_dot": (C.c_int, [vp, vp, vp, vp]),
        "sb_matrix_stream_bytes": (C.c_double, [vp]),
        "sb_matrix_packed_mode": (C.c_int, [vp]),
        "sb_matrix_crs_kernel": (C.c_int, [vp]),
        "sb_matrix_lds_window": (C.c_uint32, [vp]),
        "sb_matrix_pattern_classes": (C.c_uint32, [vp]),
        "sb_matrix_row_patterns": (C.c_uint32, [vp, C.POINTER(C.c_uint32)]),
        "sb_matrix_row_programs": (C.c_uint32, [vp, C.POINTER(C.c_uint32)]),
        "sb_cg_vector_phase": (C.c_int, [vp]),
        "sb_cg_launches_per_body": (C.c_int, [vp]),
        "sb_cg_start": (None, [vp, C.c_int, C.c_double]),
        "sb_cg_finish": (C.c_int, [vp]),
        "sb_comm_init_transport": (None, [C.c_int, C.c_int, vp]),
        "sb_comm_p2p_handle": (C.c_int, [vp]),
        "sb_comm_p2p_open": (C.c_int, [vp]),
        "sb_comm_p2p_enabled": (C.c_int, []),
        "sb_halo_p2p_enabled": (C.c_int, [vp]),
        "sb_comm_p2p_reason": (C.c_char_p, []),
        "sb_halo_p2p_reason": (C.c_char_p, [vp]),
        "sb_comm_data_plane": (None, [C.c_int]),
        "sb_comm_data_plane_selected": (C.c_int, []),
        "sb_comm_halo_push_inside": (None, [C.c_int]),
        "sb_lab_build": (C.c_int, []),
        "sb_cg_collectives_per_body": (C.c_int, [vp]),
        "sb_cg_set_fuse_p": (None, [vp, C.c_int]),
        "sb_cg_fuse_p": (C.c_int, [vp]),
        "sb_cg_set_fuse_alpha": (None, [vp, C.c_int]),
        "sb_cg_set_fuse_beta": (None, [vp, C.c_int]),
        "sb_comm_rccl_info": (C.c_int, [C.POINTER(C.c_int)]),
        "sb_cg_phase_timing": (None, [vp, C.c_int]),
        "sb_cg_phase_ms": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "sb_malloc_host_visible": (vp, [C.c_size_t]),
        "sb_host_visible_reason": (C.c_char_p, []),
        "sb_malloc_pinned_host": (vp, [C.c_size_t]),
        "sb_free_pinned_host": (None, [vp]),
        "sb_copy_counters": (None, [C.POINTER(C.c_uint64)]),
        "sb_region_begin": (None, [C.c_int]),
        "sb_region_end": (None, [C.c_int]),
        "sb_region_seconds": (C.c_double, [C.c_int, C.POINTER(C.c_uint64)]),
        "sb_region_reset": (None, []),
        "sb_matrix_place": (None, [vp, C.c_int, C.c_int]),
        "sb_matrix_place_commit": (None, [vp]),
        "sb_matrix_place_fresh": (None, [vp]),
        "sb_matrix_place_at": (None, [vp, vp, vp]),
        "sb_matrix_place_home": (None, [vp]),
        "sb_placement_arena_bytes": (C.c_size_t, [vp]),
        "sb_placement_probe": (C.c_float, [vp, vp]),
        "sb_matrix_placement": (None, [vp, C.POINTER(C.c_int)]),
        "sb_matrix_debug_ptrs": (None, [vp, C.POINTER(C.c_uint64)]),
        "sb_matrix_placement_report": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "sb_cg_debug_ptrs": (None, [vp, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# include/sbhip.h: sb_transport
ALLREDUCE_FN = C.CFUNCTYPE(None, vp, vp, C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(None, vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                          vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))


ALLGATHER_BYTES_FN = C.CFUNCTYPE(None, vp, vp, C.c_int, vp)


class TransportS(C.Structure):
    _fields_ = [("ctx", vp), ("allreduce", ALLREDUCE_FN), ("neighbour_exchange", EXCHANGE_FN),
                ("allgather_bytes", ALLGATHER_BYTES_FN)]


def init(device=None):
    """Select the GPU (LOCAL_RANK by default) and create the layer's stream."""
    L = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if L.sb_device_count() == 0:
        raise RuntimeError("sparsebench_amd: no HIP device visible; the HIP path is the only "
                           "path (no CPU fallback)")
    L.sb_init(device)
    return L


def _hp(a):
    return a.ctypes.data_as(vp)


class DeviceVector:
    """A vector of doubles in HBM owned through the C-ABI (sb_malloc / sb_free)."""

    def __init__(self, n, host=None):
        self.L = load()
        self.n = int(n)
        self.ptr = self.L.sb_malloc(max(self.n, 1) * 8)
        if host is not None:
            self.set(host)

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return cls(len(a), a)

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert len(a) == self.n
        if self.n:
            self.L.sb_h2d(self.ptr, _hp(a), self.n * 8)

    def get(self):
        out = np.empty(self.n, dtype=np.float64)
        if self.n:
            self.L.sb_d2h(_hp(out), self.ptr, self.n * 8)
        return out

    def zero(self):
        self.L.sb_memset(self.ptr, 0, self.n * 8)

    def free(self):
        if self.ptr:
            self.L.sb_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
