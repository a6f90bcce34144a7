"""ctypes bindings of the TEST-ONLY checkers (see fpv_oracle.h).

  Oracle   oracle/liboracle.so       plain-C restatement (travels to the GPU box)
  Ref      oracle/_ref/libfpvref.so  the real reference behind ref_shim.cc
                                     (built only where /root/reference exists)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
c = ctypes
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")


def _opt(a):
    return a.ctypes.data_as(c.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C oracle`")
        L = self.lib = c.CDLL(path)
        L.fpvo_clamped_gradient.restype = c.c_uint8
        L.fpvo_clamped_gradient.argtypes = [c.c_uint8] * 3
        L.fpvo_estimate_entropy.restype = c.c_uint64
        L.fpvo_estimate_entropy.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
        L.fpvo_predict.restype = c.c_uint8
        L.fpvo_predict.argtypes = [_u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int,
                                   c.c_void_p, c.c_void_p, _u8p, _u8p, _u8p]
        L.fpvo_split.restype = c.c_int
        L.fpvo_split.argtypes = [_u16p, c.c_size_t, c.c_int, c.c_int, _u8p, _u8p]
        L.fpvo_inverse.restype = None
        L.fpvo_inverse.argtypes = [_u8p, c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t,
                                   c.c_uint8, _u16p]
        L.fpvo_unextract.restype = None
        L.fpvo_unextract.argtypes = [_u16p, c.c_size_t, c.c_int, c.c_int, _u8p]
        L.fpvo_max_compressed_size.restype = c.c_size_t
        L.fpvo_max_compressed_size.argtypes = [c.c_size_t]
        L.fpvo_encode_stream.restype = c.c_size_t
        L.fpvo_encode_stream.argtypes = [_u16p, c.c_size_t, _u16p, c.c_size_t, c.c_size_t,
                                         c.c_int, c.c_int, _u8p, c.c_size_t]
        L.fpvo_decode_stream.restype = c.c_long
        L.fpvo_decode_stream.argtypes = [_u8p, c.c_size_t, _u16p, c.c_size_t,
                                         c.POINTER(c.c_size_t), c.POINTER(c.c_size_t)]

    def cg(self, n, w, nw):
        return self.lib.fpvo_clamped_gradient(n, w, nw)

    def entropy(self, hist):
        return int(self.lib.fpvo_estimate_entropy(np.ascontiguousarray(hist, dtype=np.uint64)))

    def split(self, img, shift=0, big_endian=False):
        img = np.ascontiguousarray(img, dtype=np.uint16)
        n = img.size
        high, low = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        no_low = self.lib.fpvo_split(img.reshape(-1), n, shift, int(big_endian), high, low)
        return high, low, bool(no_low)

    def predict(self, img, xs, ys, shift=0, big_endian=False, delta_raw=None):
        """Returns high, low, preview, flags for one frame."""
        img = np.ascontiguousarray(img, dtype=np.uint16).reshape(-1)
        n = xs * ys
        dh = dl = None
        if delta_raw is not None:
            dh, dl, _ = self.split(delta_raw, shift, big_endian)
        high, low = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        pv = np.zeros(max(n // 16, (xs // 4) * (ys // 4)) + 1, np.uint8)
        flags = self.lib.fpvo_predict(img, xs, ys, shift, int(big_endian), _opt(dh), _opt(dl),
                                      high, low, pv)
        return high, low, pv[:(xs // 4) * (ys // 4)].copy(), int(flags)

    def inverse(self, high, low, delta, xs, ys, flags):
        high = np.array(high, dtype=np.uint8, copy=True).reshape(-1)
        low = None if low is None else np.ascontiguousarray(low, dtype=np.uint8).reshape(-1)
        delta = None if delta is None else np.ascontiguousarray(delta, dtype=np.uint16).reshape(-1)
        img = np.zeros(xs * ys, np.uint16)
        self.lib.fpvo_inverse(high, _opt(low), _opt(delta), xs, ys, flags, img)
        return img.reshape(ys, xs)

    def unextract(self, img, shift=0, big_endian=False):
        img = np.ascontiguousarray(img, dtype=np.uint16).reshape(-1)
        out = np.zeros(img.size * 2, np.uint8)
        self.lib.fpvo_unextract(img, img.size, shift, int(big_endian), out)
        return out

    def encode_stream(self, frames, delta_raw, xs, ys, shift=0, big_endian=False):
        frames = np.ascontiguousarray(frames, dtype=np.uint16).reshape(-1)
        delta_raw = np.ascontiguousarray(delta_raw, dtype=np.uint16).reshape(-1)
        n = frames.size // (xs * ys)
        cap = 64 + (n + 1) * (3 * self.lib.fpvo_max_compressed_size(xs * ys) + 64)
        out = np.zeros(cap, np.uint8)
        sz = self.lib.fpvo_encode_stream(frames, n, delta_raw, xs, ys, shift, int(big_endian),
                                         out, cap)
        if sz == 0:
            raise RuntimeError("oracle encode_stream failed")
        return out[:sz].copy()

    def decode_stream(self, data, max_frames=1 << 16):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        xs = int.from_bytes(data[0:4].tobytes(), "little")
        ys = int.from_bytes(data[4:8].tobytes(), "little")
        n = int.from_bytes(data[-8:].tobytes(), "little")
        if n > max_frames or xs == 0 or ys == 0 or xs > 65536 or ys > 65536:
            raise RuntimeError("bad stream")
        out = np.zeros(max(n, 1) * xs * ys, np.uint16)
        xo, yo = c.c_size_t(), c.c_size_t()
        got = self.lib.fpvo_decode_stream(data, data.size, out, n, c.byref(xo), c.byref(yo))
        if got < 0:
            raise RuntimeError("oracle decode_stream failed")
        return out[:got * xs * ys].reshape(got, ys, xs)


class Ref:
    """The real reference library (only where oracle/_ref was built)."""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libfpvref.so"))

    def __init__(self):
        L = self.lib = c.CDLL(os.path.join(_HERE, "_ref", "libfpvref.so"))
        st = c.POINTER(c.c_size_t)
        L.ref_predict.restype = c.c_int
        L.ref_predict.argtypes = [_u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, c.c_void_p,
                                  _u8p, _u8p, _u8p, st, st]
        L.ref_split.restype = c.c_int
        L.ref_split.argtypes = [_u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, _u8p, _u8p, st]
        L.ref_encode_stream.restype = c.c_size_t
        L.ref_encode_stream.argtypes = [_u16p, c.c_size_t, _u16p, c.c_size_t, c.c_size_t,
                                        c.c_int, c.c_int, c.c_size_t, _u8p, c.c_size_t]
        L.ref_decode_stream.restype = c.c_long
        L.ref_decode_stream.argtypes = [_u8p, c.c_size_t, _u16p, c.c_size_t, c.c_void_p, st, st]
        L.ref_stream_decode.restype = c.c_long
        L.ref_stream_decode.argtypes = [_u8p, c.c_size_t, c.c_size_t, _u16p, c.c_size_t]
        L.ref_unextract.restype = None
        L.ref_unextract.argtypes = [_u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, _u8p]

    def predict(self, img, xs, ys, shift=0, big_endian=False, delta_raw=None):
        img = np.ascontiguousarray(img, dtype=np.uint16).reshape(-1)
        d = None if delta_raw is None else np.ascontiguousarray(delta_raw, dtype=np.uint16).reshape(-1)
        n = xs * ys
        high, low, pv = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n // 16 + 16, np.uint8)
        ls, ps = c.c_size_t(), c.c_size_t()
        flags = self.lib.ref_predict(img, xs, ys, shift, int(big_endian), _opt(d), high, low, pv,
                                     c.byref(ls), c.byref(ps))
        return high, low[:ls.value].copy(), pv[:ps.value].copy(), int(flags)

    def encode_stream(self, frames, delta_raw, xs, ys, shift=0, big_endian=False, threads=0):
        frames = np.ascontiguousarray(frames, dtype=np.uint16).reshape(-1)
        delta_raw = np.ascontiguousarray(delta_raw, dtype=np.uint16).reshape(-1)
        n = frames.size // (xs * ys)
        cap = 64 + (n + 1) * (3 * (xs * ys + 1024) + 64)
        out = np.zeros(cap, np.uint8)
        sz = self.lib.ref_encode_stream(frames, n, delta_raw, xs, ys, shift, int(big_endian),
                                        threads, out, cap)
        if sz == 0:
            raise RuntimeError("reference encode failed")
        return out[:sz].copy()

    def decode_stream(self, data, with_previews=False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        xs = int.from_bytes(data[0:4].tobytes(), "little")
        ys = int.from_bytes(data[4:8].tobytes(), "little")
        n = int.from_bytes(data[-8:].tobytes(), "little")
        out = np.zeros(max(n, 1) * xs * ys, np.uint16)
        pv = np.zeros(max(n, 1) * (xs // 4) * (ys // 4), np.uint8) if with_previews else None
        xo, yo = c.c_size_t(), c.c_size_t()
        got = self.lib.ref_decode_stream(data, data.size, out, n, _opt(pv), c.byref(xo), c.byref(yo))
        if got < 0:
            raise RuntimeError("reference decode failed")
        frames = out[:got * xs * ys].reshape(got, ys, xs)
        if with_previews:
            return frames, pv.reshape(got, ys // 4, xs // 4)
        return frames

    def stream_decode(self, data, xs, ys, nframes, block=65536):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(max(nframes, 1) * xs * ys, np.uint16)
        got = self.lib.ref_stream_decode(data, data.size, block, out, nframes)
        if got < 0:
            raise RuntimeError("reference streaming decode failed")
        return out[:got * xs * ys].reshape(got, ys, xs)
