"""ctypes loader of csrc/libvlq_ivfpq.so (the C ABI of include/vlq_ivfpq.h)."""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("VLQ_LIB_PATH") or os.path.join(_CSRC, "libvlq_ivfpq.so")   # override: A/B builds of the library

# every symbol include/vlq_ivfpq.h declares
SYMBOLS = [
    "vlq_version", "vlq_last_error", "vlq_device_count", "vlq_ivfpq_create", "vlq_ivfpq_destroy",
    "vlq_ivfpq_set_stream", "vlq_ivfpq_set_coarse_centroids", "vlq_ivfpq_set_imi_centroids",
    "vlq_ivfpq_set_pq_centroids",
    "vlq_ivfpq_set_search_options", "vlq_ivfpq_set_float16_tables", "vlq_ivfpq_set_scan_schedule", "vlq_ivfpq_set_coarse_screen", "vlq_ivfpq_coarse_screen_state", "vlq_ivfpq_set_lists", "vlq_ivfpq_add", "vlq_ivfpq_reserve_memory", "vlq_ivfpq_reclaim_memory", "vlq_ivfpq_encode", "vlq_ivfpq_encode_preassigned",
    "vlq_ivfpq_ntotal", "vlq_ivfpq_list_length", "vlq_ivfpq_get_list", "vlq_ivfpq_search",
    "vlq_ivfpq_search_preassigned", "vlq_ivfpq_coarse_search", "vlq_ivfpq_query_tables",
    "vlq_ivfpq_get_precomputed_table", "vlq_ivfpq_stats", "vlq_ivfpq_profile",
    "vlq_ivfpq_profile_read", "vlq_merge_topk", "vlq_ivfpq_last_scan_info", "vlq_ivfpq_reset_walk_state",
    # include/vlq_line.h
    "vlq_line_set_float16_tables", "vlq_line_set_row_mode", "vlq_line_set_scan_parts", "vlq_line_create", "vlq_line_destroy", "vlq_line_set_stream", "vlq_line_set_coarse_centroids",
    "vlq_line_set_pq_centroids", "vlq_line_set_lambda_codebook", "vlq_line_set_graph",
    "vlq_line_build_graph", "vlq_line_assign", "vlq_line_residuals", "vlq_line_encode", "vlq_line_add",
    "vlq_line_set_lists", "vlq_line_ntotal", "vlq_line_list_length", "vlq_line_get_list",
    "vlq_line_search", "vlq_line_stats", "vlq_line_profile", "vlq_line_profile_read",
]


class VlqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vlq error %d: %s" % (code, msg))
        self.code = code


def library_path():
    return _SO


def build_library(force=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", _CSRC])
    return _SO


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (SONAME libamdhip64.so.7); if this library were loaded first it would
    pull the system copy and a later `import torch` would bring a second runtime
    that cannot open the device.  So when torch is installed but not yet imported,
    map torch's copy first: our NEEDED libamdhip64.so.7 then binds to it by SONAME."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise VlqError(-1, "HIP library %s is not built (run `make -C %s`); there is no "
                               "fallback path" % (_SO, _CSRC))
        _share_hip_runtime_with_torch()
        L = C.CDLL(_SO)
        L.vlq_last_error.restype = C.c_char_p
        L.vlq_ivfpq_ntotal.restype = C.c_int64
        L.vlq_ivfpq_destroy.restype = None
        L.vlq_line_ntotal.restype = C.c_int64
        L.vlq_line_destroy.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise VlqError(rc, lib().vlq_last_error().decode(errors="replace"))


def device_count():
    return lib().vlq_device_count()
