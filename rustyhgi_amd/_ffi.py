"""ctypes binding of libhgi_hip.so -- the C ABI declared in include/hgi.h.

There is no fallback: if the HIP library is missing or no device is usable, every
operation raises.  Nothing here imports the CPU oracle.
"""
import ctypes
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HGI_LIB_PATH") or os.path.join(_HERE, "libhgi_hip.so")

OK, EINVAL, ENOMEM, EDEVICE, EUNSUPPORTED = 0, 1, 2, 3, 4
PATH_AUTO, PATH_LEVELWISE, PATH_FUSED = 0, 1, 2
SYNTH_XY, SYNTH_NOISE, SYNTH_RAMP = 0, 1, 2

# every symbol include/hgi.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _int, _sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_size_t
SYMBOLS = [
    ("hgi_ctx_create", _int, [_int, ctypes.POINTER(_vp)]),
    ("hgi_ctx_destroy", None, [_vp]),
    ("hgi_ctx_set_stream", _int, [_vp, _vp]),
    ("hgi_ctx_use_own_stream", _int, [_vp]),
    ("hgi_ctx_set_path", _int, [_vp, _int]),
    ("hgi_ctx_reserve", _int, [_vp, _u32, _u32, _u32, _sz]),
    ("hgi_ctx_scratch_bytes", _int, [_vp, ctypes.POINTER(_sz)]),
    ("hgi_histogram_u8_dev", _int, [_vp, _vp, _u32, _u32, _sz, _sz, _vp]),
    ("hgi_encode_u8_batch", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp, _vp, _sz, _sz]),
    ("hgi_decode_u8_batch", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp, _sz, _sz]),
    ("hgi_sync", _int, [_vp]),
    ("hgi_last_error", ctypes.c_char_p, []),
    ("hgi_version", ctypes.c_char_p, []),
    ("hgi_linear_lut", _int, [_int, _vp, _vp]),
    ("hgi_noop_lut", None, [_vp]),
    ("hgi_encode_u8", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp, _vp]),
    ("hgi_decode_u8", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp]),
    ("hgi_encode_u8_dev", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp, _vp, _sz, _sz]),
    ("hgi_decode_u8_dev", _int, [_vp, _vp, _u32, _u32, _u32, _int, _vp, _sz, _sz]),
    ("hgi_synth_u8_dev", _int, [_vp, _int, _u64, _u64, _u32, _u32, _vp, _sz, _sz]),
    ("hgi_copy_u8_dev", _int, [_vp, _vp, _vp, _sz]),
    ("hgi_diff_stats_dev", _int, [_vp, _vp, _vp, _u32, _u32, _sz, _sz, _vp]),
    ("hgi_deflate_grid_dev", _int, [_vp, _vp, _u32, _u32, _vp, _sz, ctypes.POINTER(_sz)]),
    ("hgi_deflate_grids_dev", _int, [_vp, _vp, _u32, _u32, _sz, _sz, _vp, _sz, _vp]),
    ("hgi_deflate_grids_packed_dev", _int, [_vp, _vp, _u32, _u32, _sz, _sz, _vp, _sz, _vp, _vp]),
    ("hgi_deflate_grid", _int, [_vp, _vp, _u32, _u32, _vp, _sz, ctypes.POINTER(_sz)]),
    ("hgi_huffman_plan", _int, [_vp, _vp, _vp, _vp, _sz, ctypes.POINTER(_sz)]),
    ("hgi_planes_alloc", _int, [_vp, _sz, _u32, ctypes.POINTER(_vp), ctypes.POINTER(_int)]),
    ("hgi_planes_free", _int, [_vp, _u32, ctypes.POINTER(_vp)]),
    ("hgi_planes_report", ctypes.c_char_p, [_vp]),
    ("hgi_probe_pair_u8_dev", _int, [_vp, _vp, _vp, _sz, ctypes.POINTER(ctypes.c_float)]),
    ("hgi_timer_start", _int, [_vp]),
    ("hgi_timer_stop", _int, [_vp, ctypes.POINTER(ctypes.c_float)]),
]


class HgiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("hgi status %d: %s" % (status, message))
        self.status = status


_lib = None


def _share_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  torch ships its own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's); if two copies get mapped, the second one finds no
    device.  So when torch is installed but not imported yet, map torch's copy first: libhgi_hip.so
    then binds to it by SONAME, and so does torch when it is imported later.  Without torch the
    library uses the system ROCm runtime."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load libhgi_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C rustyhgi_amd/csrc` "
                              "(there is no CPU fallback)" % LIB_PATH)
        _share_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise HgiError(status, lib().hgi_last_error().decode("utf-8", "replace"))


class Context:
    """hgi_ctx: device id + stream + scratch.  Not thread-safe (include/hgi.h)."""

    def __init__(self, device=0):
        h = _vp()
        check(lib().hgi_ctx_create(int(device), ctypes.byref(h)))
        self.handle, self.device = h, int(device)

    def close(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.hgi_ctx_destroy(self.handle)
        self.handle = None

    __del__ = close

    def set_stream(self, stream_ptr):
        """Borrow a hipStream_t verbatim (0/None = HIP's default stream, torch's default)."""
        if stream_ptr != getattr(self, "_stream", "own"):
            check(lib().hgi_ctx_set_stream(self.handle, _vp(stream_ptr or 0)))
            self._stream = stream_ptr

    def use_own_stream(self):
        check(lib().hgi_ctx_use_own_stream(self.handle))
        self._stream = "own"

    def set_path(self, path):
        check(lib().hgi_ctx_set_path(self.handle, int(path)))

    def reserve(self, w, h, levels, batch=1):
        check(lib().hgi_ctx_reserve(self.handle, w, h, levels, batch))

    def scratch_bytes(self):
        """Bytes of device scratch the context owns right now (hgi_ctx_scratch_bytes)."""
        n = ctypes.c_size_t(0)
        check(lib().hgi_ctx_scratch_bytes(self.handle, ctypes.byref(n)))
        return n.value

    def sync(self):
        check(lib().hgi_sync(self.handle))

    def timer_start(self):
        check(lib().hgi_timer_start(self.handle))

    def timer_stop(self):
        ms = ctypes.c_float(0)
        check(lib().hgi_timer_stop(self.handle, ctypes.byref(ms)))
        return ms.value


_default = {}


def default_context(device=0):
    ctx = _default.get(device)
    if ctx is None or ctx.handle is None:
        ctx = _default[device] = Context(device)
    return ctx
