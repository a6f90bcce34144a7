"""fusion-power-video_amd -- MI355X-native predictor/residual engine of the
fusion-power-video lossless 16-bit video codec.

The product is native: `lib/libfpvhip.so` (hand-written gfx950 kernels behind
the C ABI of include/fpv_hip.h) and `lib/libfpv_amd.so` (the host codec with the
reference's `fusion_power_video.h` API, include/fusion_power_video.h).  This
Python package is only the ctypes binding used by the tests, `bench.py` and
`__graft_entry__.py`; torch is used for device memory, streams and
`torch.distributed`, nothing else.

The directory name is not an importable identifier; load it with
`importlib` (see `__graft_entry__.load_package`) under the name `fpv_amd`.

There is no CPU fallback: every entry point raises when the HIP library is
missing or no GPU is usable.
"""
import ctypes
import os

import torch  # must be imported BEFORE libfpvhip.so so that both share one HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(_HERE, "lib")

USE_DELTA, USE_CG, NO_LOW_BYTES = 1, 2, 4


class FpvError(RuntimeError):
    pass


_hip = None
_host = None


def _declare_hip(lib):
    c = ctypes
    vp, u32, i32 = c.c_void_p, c.c_uint32, c.c_int
    lib.fpvhip_abi_version.restype = i32
    lib.fpvhip_device_count.restype = i32
    lib.fpvhip_last_error.restype = c.c_char_p
    lib.fpvhip_last_error.argtypes = [vp]
    lib.fpvhip_create.restype = vp
    lib.fpvhip_create.argtypes = [i32, u32, u32, i32, i32, u32, u32]
    lib.fpvhip_destroy.argtypes = [vp]
    lib.fpvhip_destroy.restype = None
    lib.fpvhip_npix.restype = c.c_size_t
    lib.fpvhip_npix.argtypes = [vp]
    lib.fpvhip_npreview.restype = c.c_size_t
    lib.fpvhip_npreview.argtypes = [vp]
    lib.fpvhip_set_delta_raw.argtypes = [vp, vp]
    lib.fpvhip_set_delta_decoded.argtypes = [vp, vp]
    lib.fpvhip_stage_in.restype = vp
    lib.fpvhip_stage_in.argtypes = [vp, u32, u32]
    lib.fpvhip_predict_submit.argtypes = [vp, u32, u32, i32]
    lib.fpvhip_predict_wait.argtypes = [vp, u32] + [c.POINTER(vp)] * 4
    lib.fpvhip_stage_plane.restype = vp
    lib.fpvhip_stage_plane.argtypes = [vp, u32, u32, i32]
    lib.fpvhip_inverse_submit.argtypes = [vp, u32, u32, vp]
    lib.fpvhip_inverse_wait.argtypes = [vp, u32, c.POINTER(vp)]
    lib.fpvhip_predict_device.argtypes = [vp, vp, u32, i32, vp, vp, vp, vp, vp]
    lib.fpvhip_inverse_device.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.fpvhip_time_predict_device.restype = c.c_double
    lib.fpvhip_time_predict_device.argtypes = [vp, vp, u32, i32, vp, vp, vp, vp, vp, i32,
                                               c.POINTER(c.c_double)]
    lib.fpvhip_time_copy_device.restype = c.c_double
    lib.fpvhip_time_copy_device.argtypes = [vp, vp, c.c_size_t, vp, i32]
    lib.fpvhip_time_copy_shape.restype = c.c_double
    lib.fpvhip_time_copy_shape.argtypes = [vp, vp, c.c_size_t, vp, i32, i32]
    lib.fpvhip_time_inverse_device.restype = c.c_double
    lib.fpvhip_time_inverse_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, i32]
    if hasattr(lib, "fpvhip_inverse_kernel_name"):   # absent from older diagnostic builds (tools/ab_*.py)
        lib.fpvhip_inverse_kernel_name.restype = c.c_char_p
        lib.fpvhip_inverse_kernel_name.argtypes = [vp, u32]
    if hasattr(lib, "fpvhip_forward_kernel_names"):
        lib.fpvhip_forward_kernel_names.restype = c.c_char_p
        lib.fpvhip_forward_kernel_names.argtypes = [vp]
    lib.fpvhip_debug_counters.restype = i32
    lib.fpvhip_debug_counters.argtypes = [vp, c.POINTER(c.c_uint32), u32]
    lib.fpvhip_inverse_pass_stats.restype = i32
    lib.fpvhip_inverse_pass_stats.argtypes = [vp, c.POINTER(c.c_uint32), i32]
    if hasattr(lib, "fpvhip_forward_tile_shape"):
        lib.fpvhip_forward_tile_shape.restype = i32
        lib.fpvhip_forward_tile_shape.argtypes = [vp, c.POINTER(u32)]
    if hasattr(lib, "fpvhip_debug_set_chain_mask"):
        lib.fpvhip_debug_set_chain_mask.restype = i32
        lib.fpvhip_debug_set_chain_mask.argtypes = [vp, u32]
    for name in ("fpvhip_set_delta_raw", "fpvhip_set_delta_decoded", "fpvhip_predict_submit",
                 "fpvhip_predict_wait", "fpvhip_inverse_submit", "fpvhip_inverse_wait",
                 "fpvhip_predict_device", "fpvhip_inverse_device"):
        getattr(lib, name).restype = i32


def hip_lib():
    """The C-ABI library (include/fpv_hip.h).  Raises when it is not built."""
    global _hip
    if _hip is None:
        path = os.path.join(LIBDIR, "libfpvhip.so")  # diagnostic builds are loaded explicitly (tools/ab_forward.py), never through the environment
        if not os.path.exists(path):
            raise FpvError(f"{path} is missing: run `make product` (no CPU fallback exists)")
        _hip = ctypes.CDLL(path)
        _declare_hip(_hip)
    return _hip


def host_lib():
    """The host codec library (fusion_power_video.h API + its C shim)."""
    global _host
    if _host is None:
        hip_lib()
        path = os.path.join(LIBDIR, "libfpv_amd.so")
        if not os.path.exists(path):
            raise FpvError(f"{path} is missing: run `make product`")
        _host = ctypes.CDLL(path)
        from . import codec  # noqa: F401  (declares signatures)
        codec.declare(_host)
    return _host


def _err(ctx=None):
    return (hip_lib().fpvhip_last_error(ctx) or b"").decode()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def copy_bandwidth_GBps(nbytes=1 << 30, iters=10, device=0, shape=1):
    """Measured device-copy rate (read + write bytes per second) on this GPU.  shape 1: one
    16-byte element per thread, dispatch-ordered (MI355X_MICROARCH.md's float4 copy); 0: round
    2's chunked persistent copy."""
    lib = hip_lib()
    src = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device}").random_(0, 255)
    dst = torch.empty_like(src)
    ms = lib.fpvhip_time_copy_shape(_ptr(dst), _ptr(src), nbytes,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), iters, shape)
    if ms <= 0:
        raise FpvError("copy timing failed")
    return 2.0 * nbytes * iters / (ms / 1e3) / 1e9


class Predictor:
    """Device-resident view of one fpvhip context (tests and bench).

    Tensors live on `cuda:<device>`; kernels are launched on torch's current
    stream.  Mirrors Frame::Predict / DecompressImage's predictor half:
      predict(frames[u16 n,ys,xs]) -> high, low, preview, flags
      inverse(high, low, flags)    -> frames
    """

    def __init__(self, xs, ys, shift=0, big_endian=False, max_batch=64, slots=0, device=0):
        lib = hip_lib()
        if not torch.cuda.is_available():
            raise FpvError("no GPU visible to torch: the HIP path has no CPU fallback")
        torch.cuda.set_device(device)
        torch.cuda.current_stream()  # make sure torch's HIP context exists
        self.lib, self.xs, self.ys, self.shift = lib, xs, ys, shift
        self.big_endian, self.max_batch, self.device = big_endian, max_batch, device
        self.ctx = lib.fpvhip_create(device, xs, ys, shift, int(big_endian), max_batch, slots)
        if not self.ctx:
            raise FpvError("fpvhip_create: " + _err())
        self.npix = lib.fpvhip_npix(self.ctx)
        self.npv = lib.fpvhip_npreview(self.ctx)
        self.has_delta = False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.fpvhip_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise FpvError(f"{what}: {_err(self.ctx)}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def set_delta_raw(self, delta_np):
        import numpy as np
        d = np.ascontiguousarray(delta_np, dtype=np.uint16)
        assert d.size == self.npix
        self._check(self.lib.fpvhip_set_delta_raw(self.ctx, d.ctypes.data), "set_delta_raw")
        self.has_delta = True

    def set_delta_decoded(self, delta_np):
        import numpy as np
        d = np.ascontiguousarray(delta_np, dtype=np.uint16)
        assert d.size == self.npix
        self._check(self.lib.fpvhip_set_delta_decoded(self.ctx, d.ctypes.data), "set_delta_decoded")
        self.has_delta = True

    def alloc_outputs(self, n):
        dev = f"cuda:{self.device}"
        return (torch.empty((n, self.npix), dtype=torch.uint8, device=dev),
                torch.empty((n, self.npix), dtype=torch.uint8, device=dev),
                torch.empty((n, self.npv), dtype=torch.uint8, device=dev),
                torch.empty((n,), dtype=torch.uint8, device=dev))

    def predict(self, frames, with_delta=None, out=None):
        """frames: uint16 tensor on the GPU, n*npix elements."""
        n = frames.numel() // self.npix
        assert frames.dtype == torch.uint16 and frames.is_contiguous() and frames.is_cuda
        if with_delta is None:
            with_delta = self.has_delta
        high, low, pv, flags = out if out is not None else self.alloc_outputs(n)
        self._check(self.lib.fpvhip_predict_device(
            self.ctx, _ptr(frames), n, int(with_delta), _ptr(high), _ptr(low), _ptr(pv),
            _ptr(flags), self._stream()), "predict_device")
        return high, low, pv, flags

    def inverse(self, high, low, flags, out=None):
        n = flags.numel()
        if out is None:
            out = torch.empty((n, self.ys, self.xs), dtype=torch.uint16, device=high.device)
        self._check(self.lib.fpvhip_inverse_device(
            self.ctx, _ptr(high), _ptr(low), _ptr(flags), n, _ptr(out), self._stream()),
            "inverse_device")
        return out

    def time_predict(self, frames, out, iters, with_delta=None, per_kernel=False):
        n = frames.numel() // self.npix
        if with_delta is None:
            with_delta = self.has_delta
        high, low, pv, flags = out
        km = (ctypes.c_double * 4)() if per_kernel else None
        ms = self.lib.fpvhip_time_predict_device(
            self.ctx, _ptr(frames), n, int(with_delta), _ptr(high), _ptr(low), _ptr(pv),
            _ptr(flags), self._stream(), iters, km)
        if ms < 0:
            raise FpvError("time_predict: " + _err(self.ctx))
        return (ms, list(km)) if per_kernel else ms

    def debug_counters(self, n=16):
        """Counter block of the last forward batch: [0] frames whose delta decision was
        mis-speculated, [1] frames redone under their true flags, [4..7] decisions by value."""
        buf = (ctypes.c_uint32 * n)()
        self._check(self.lib.fpvhip_debug_counters(self.ctx, buf, n), "debug_counters")
        return [int(v) for v in buf]

    def set_chain_mask(self, value):
        """Overwrites the three-wave inverse kernel's per-CU "SIMD taken" words (0: all free)."""
        self._check(self.lib.fpvhip_debug_set_chain_mask(self.ctx, value), "debug_set_chain_mask")

    def forward_tile_shape(self):
        """(groups per tile, threads per workgroup, tiles per frame, log2 frames per group) of the tile pass."""
        out = (ctypes.c_uint32 * 4)()
        self._check(self.lib.fpvhip_forward_tile_shape(self.ctx, out), "forward_tile_shape")
        return tuple(out)

    def forward_kernel_names(self):
        """The three launches of one forward batch, in the order of time_predict(per_kernel=True)."""
        if not hasattr(self.lib, "fpvhip_forward_kernel_names"):
            return ["k_stream", "k_redo_decide", "k_redo_stream"]
        return self.lib.fpvhip_forward_kernel_names(self.ctx).decode().split(",")

    def inverse_kernel_name(self, nframes):
        """Which inverse kernel launch_inverse picks for this batch size (FPV_INVERSE unset)."""
        return self.lib.fpvhip_inverse_kernel_name(self.ctx, nframes).decode()

    def inverse_pass_stats(self, reset=True):
        """(supersteps, passes) of the speculative inverse kernel since the last reset;
        `self.last_pass_extremes` = (most passes on one frame, most on one superstep)."""
        out = (ctypes.c_uint32 * 4)()
        self._check(self.lib.fpvhip_inverse_pass_stats(self.ctx, out, int(reset)), "inverse_pass_stats")
        self.last_pass_extremes = (int(out[2]), int(out[3]))
        return int(out[0]), int(out[1])

    def time_inverse(self, high, low, flags, out, iters):
        ms = self.lib.fpvhip_time_inverse_device(
            self.ctx, _ptr(high), _ptr(low), _ptr(flags), flags.numel(), _ptr(out),
            self._stream(), iters)
        if ms < 0:
            raise FpvError("time_inverse: " + _err(self.ctx))
        return ms
