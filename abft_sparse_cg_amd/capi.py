"""ctypes view of the C ABI in include/abft_hip.h (libabft_hip.so).

The library is the product: there is no Python or CPU implementation behind
these calls.  If the shared object is missing or a call fails, this module
raises -- it never falls back to anything.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABFT_HIP_LIB") or os.path.join(_HERE, "libabft_hip.so")  # override: A/B builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "abft_hip.h")

OK = 0
MODES = ("none", "constraints", "sed", "sec7", "sec8", "secded")
MODE_ID = {m: i for i, m in enumerate(MODES)}
MODE_ID["sec"] = MODE_ID["sec7"]  # BASELINE.json config 5 spells sec7 "sec"
FMT_CSR, FMT_COO = 0, 1
FLIP_ANY, FLIP_VALUE, FLIP_INDEX = 0, 1, 2
PART_ALL, PART_INTERIOR, PART_BOUNDARY = 0, 1, 2
K_SPMV, K_DOT, K_CALC_XR, K_CALC_P = 0, 1, 2, 3

u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)


class AbftError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("abft_hip error %d: %s" % (code, msg))
        self.code = code


class Event(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("bit", C.c_uint32), ("fmt", C.c_uint32)]

    def tup(self):
        return (self.kind, self.index, self.bit)


# name -> (restype, argtypes); every symbol include/abft_hip.h declares
SIGNATURES = {
    "abft_hip_last_error": (C.c_char_p, []),
    "abft_hip_device_count": (C.c_int, [i32p]),
    "abft_hip_init": (C.c_int, [C.c_int, vpp]),
    "abft_hip_shutdown": (C.c_int, [vp]),
    "abft_hip_set_stream": (C.c_int, [vp, vp]),
    "abft_hip_get_stream": (vp, [vp]),
    "abft_hip_synchronize": (C.c_int, [vp]),
    "abft_hip_synchronize_timeout": (C.c_int, [vp, C.c_double]),
    "abft_hip_matrix_create_csr": (C.c_int, [vp, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int, vpp]),
    "abft_hip_matrix_create_coo": (C.c_int, [vp, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int, vpp]),
    "abft_hip_matrix_create_shard": (C.c_int, [vp, C.c_int, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int,
                                               C.c_int, C.c_uint32, vpp]),
    "abft_hip_matrix_create_shard_indexed": (C.c_int, [vp, C.c_int, C.c_int, u32p, u32p, f64p, C.c_int, C.c_int,
                                                       C.c_int, u32p, vpp]),
    "abft_hip_matrix_destroy": (C.c_int, [vp]),
    "abft_hip_matrix_info": (C.c_int, [vp, i32p, i32p]),
    "abft_hip_matrix_read_element": (C.c_int, [vp, C.c_uint32, u32p]),
    "abft_hip_matrix_read_csr": (C.c_int, [vp, vp, vp, vp]),
    "abft_hip_matrix_read_coo": (C.c_int, [vp, vp]),
    "abft_hip_inject": (C.c_int, [vp, C.c_uint32, i32p, C.c_int]),
    "abft_hip_inject_rowptr": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "abft_hip_vector_create": (C.c_int, [vp, C.c_int, vpp]),
    "abft_hip_vector_view": (C.c_int, [vp, C.c_int, C.c_int, vpp]),
    "abft_hip_vector_destroy": (C.c_int, [vp]),
    "abft_hip_vector_map": (C.c_int, [vp, C.POINTER(f64p)]),
    "abft_hip_vector_unmap": (C.c_int, [vp, f64p]),
    "abft_hip_vector_copy": (C.c_int, [vp, vp]),
    "abft_hip_vector_device_ptr": (vp, [vp]),
    "abft_hip_vector_length": (C.c_int, [vp]),
    "abft_hip_dot": (C.c_int, [vp, vp, vp, f64p]),
    "abft_hip_calc_xr": (C.c_int, [vp, vp, vp, vp, vp, C.c_double, f64p]),
    "abft_hip_calc_p": (C.c_int, [vp, vp, vp, C.c_double]),
    "abft_hip_spmv": (C.c_int, [vp, vp, vp, vp]),
    "abft_hip_dot_dev": (C.c_int, [vp, vp, vp, vp]),
    "abft_hip_calc_xr_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_double, vp]),
    "abft_hip_read_pair": (C.c_int, [vp, vp, f64p, f64p]),
    "abft_hip_spmv_dot_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp]),
    "abft_hip_matrix_set_interior": (C.c_int, [vp, C.c_int, C.c_int]),
    "abft_hip_spmv_part": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "abft_hip_spmv_dot_part_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_int]),
    "abft_hip_calc_xr_ratio_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "abft_hip_calc_p_ratio_dev": (C.c_int, [vp, vp, vp, vp, vp]),
    "abft_hip_set_sharers": (C.c_int, [vp, C.c_int]),
    "abft_hip_speculation_stats": (C.c_int, [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "abft_hip_cg_iteration_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "abft_hip_write_pair": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "abft_hip_peer_board_bytes": (C.c_size_t, []),
    "abft_hip_peer_board_attach": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_double]),
    "abft_hip_peer_board_detach": (C.c_int, [vp]),
    "abft_hip_allreduce_pair_peers": (C.c_int, [vp, vp]),
    "abft_hip_peer_board_failed": (C.c_int, [vp]),
    "abft_hip_peer_board_ipc_handle_bytes": (C.c_size_t, []),
    "abft_hip_peer_board_ipc_export": (C.c_int, [vp, vp]),
    "abft_hip_peer_board_ipc_attach": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double]),
    "abft_hip_peer_board_device_alloc": (C.c_int, [vp, C.POINTER(vp)]),
    "abft_hip_peer_board_device_free": (C.c_int, [vp, vp]),
    "abft_hip_peer_board_attach_device": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_double]),
    "abft_hip_peer_board_fuse": (C.c_int, [vp, C.c_int]),
    "abft_hip_peer_exchange_bytes": (C.c_size_t, [C.c_int, C.c_size_t]),
    "abft_hip_peer_exchange_attach": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, vp,
                                                C.c_int, C.c_double]),
    "abft_hip_peer_exchange_detach": (C.c_int, [vp]),
    "abft_hip_peer_exchange": (C.c_int, [vp, vp]),
    "abft_hip_peer_exchange_begin": (C.c_int, [vp, vp, C.c_int]),
    "abft_hip_peer_exchange_finish": (C.c_int, [vp]),
    "abft_hip_peer_exchange_failed": (C.c_int, [vp]),
    "abft_hip_peer_exchange_ipc_export": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
    "abft_hip_peer_exchange_ipc_attach": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, vp, C.c_int,
                                                    C.c_double]),
    "abft_hip_peer_exchange_device_alloc": (C.c_int, [vp, C.c_int, C.c_size_t, C.POINTER(vp)]),
    "abft_hip_peer_exchange_device_free": (C.c_int, [vp, vp]),
    "abft_hip_peer_exchange_attach_device": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_size_t, vp, C.c_int, vp,
                                                       C.c_int, C.c_double]),
    "abft_hip_matrix_panels": (C.c_int, [vp, i32p, i32p]),
    "abft_hip_spmv_dot_range_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int]),
    "abft_hip_graph_begin": (C.c_int, [vp]),
    "abft_hip_graph_end": (C.c_int, [vp, vpp]),
    "abft_hip_graph_launch": (C.c_int, [vp]),
    "abft_hip_graph_destroy": (C.c_int, [vp]),
    "abft_hip_drain_events": (C.c_int, [vp, C.POINTER(Event), C.c_int, i32p, i32p]),
    "abft_hip_event_capacity": (C.c_int, []),
    "abft_hip_pending_events": (C.c_int, [vp]),
    "abft_format_event": (C.c_int, [C.POINTER(Event), C.c_char_p, C.c_size_t]),
    "abft_event_is_fatal": (C.c_int, [C.c_uint32]),
    "abft_hip_profile_enable": (C.c_int, [vp, C.c_int]),
    "abft_hip_profile_stride": (C.c_int, [vp, C.c_int]),
    "abft_hip_profile_reset": (C.c_int, [vp]),
    "abft_hip_profile_read": (C.c_int, [vp, C.c_int, f64p, C.POINTER(C.c_long)]),
    "abft_hip_stream_probe": (C.c_int, [vp, C.c_size_t, C.c_int, f64p, f64p]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same soname as /opt/rocm's); if libabft_hip.so pulled in
    ROCm's copy and torch later loaded its own, the second runtime to start finds
    no device.  When torch is installed, load ITS runtime first (without importing
    torch): libabft_hip.so then binds to it by soname, and so does torch whenever
    it is imported.  Plain C++ users of the library (cg-csr / cg-coo) never load
    torch and run on /opt/rocm's runtime.  ABFT_HIP_SYSTEM_RUNTIME=1 skips this."""
    if os.environ.get("ABFT_HIP_SYSTEM_RUNTIME") == "1":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def load():
    """Load libabft_hip.so (built by `make -C abft_sparse_cg_amd/csrc`).  Raises
    if it is missing: the HIP library is the only implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C %s` (hipcc --offload-arch=gfx950); "
                              "there is no CPU fallback" % (LIB_PATH, os.path.join(_HERE, "csrc")))
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != OK:
        raise AbftError(rc, (load().abft_hip_last_error() or b"").decode(errors="replace"))


def is_fatal(kind):
    """same rule as abft_event_is_fatal"""
    return kind in (1, 4) or kind >= 5


def format_event(kind, index, bit, fmt):
    buf = C.create_string_buffer(160)
    ev = Event(kind, index, bit, fmt)
    load().abft_format_event(C.byref(ev), buf, 160)
    return buf.value.decode()
