"""ctypes view of include/fpv_c_api.h (the host codec: fpvc::Encoder,
RandomAccessDecoder, StreamingDecoder behind plain-C entry points).

`Codec()` binds the product library (GPU predictor, no CPU fallback).  Tests
without a GPU pass the path of tests/shim/libfpv_amd_cpushim.so instead, which
links the same host sources against an oracle-backed stand-in for the C ABI.
"""
import ctypes as c

import numpy as np

_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
WRITE_FN = c.CFUNCTYPE(None, c.POINTER(c.c_uint8), c.c_size_t, c.c_void_p)


def declare(lib):
    st = c.POINTER(c.c_size_t)
    lib.fpvc_encoder_new.restype = c.c_void_p
    lib.fpvc_encoder_new.argtypes = [c.c_size_t, c.c_int, c.c_int]
    lib.fpvc_encoder_max_queued.restype = c.c_size_t
    lib.fpvc_encoder_max_queued.argtypes = [c.c_void_p]
    lib.fpvc_encoder_init.restype = None
    lib.fpvc_encoder_init.argtypes = [c.c_void_p, _u16p, c.c_size_t, c.c_size_t, WRITE_FN, c.c_void_p]
    lib.fpvc_encoder_compress_frame.restype = None
    lib.fpvc_encoder_compress_frame.argtypes = [c.c_void_p, _u16p, WRITE_FN, c.c_void_p]
    lib.fpvc_encoder_finish.restype = None
    lib.fpvc_encoder_finish.argtypes = [c.c_void_p, WRITE_FN, c.c_void_p]
    lib.fpvc_encoder_free.restype = None
    lib.fpvc_encoder_free.argtypes = [c.c_void_p]
    lib.fpvc_encoder_last_error.restype = c.c_char_p
    lib.fpvc_encoder_last_error.argtypes = [c.c_void_p]
    lib.fpvc_encode_stream.restype = c.c_size_t
    lib.fpvc_encode_stream.argtypes = [_u16p, c.c_size_t, _u16p, c.c_size_t, c.c_size_t, c.c_int,
                                       c.c_int, c.c_size_t, _u8p, c.c_size_t]
    lib.fpvc_decode_stream.restype = c.c_long
    lib.fpvc_decode_stream.argtypes = [_u8p, c.c_size_t, _u16p, c.c_size_t, c.c_void_p, c.c_int,
                                       c.c_size_t]
    lib.fpvc_decode_stream_raw.restype = c.c_long
    lib.fpvc_decode_stream_raw.argtypes = [_u8p, c.c_size_t, _u8p, c.c_size_t, c.c_int, c.c_int, c.c_size_t]
    lib.fpvc_stream_decode.restype = c.c_long
    lib.fpvc_stream_decode.argtypes = [_u8p, c.c_size_t, c.c_size_t, _u16p, c.c_size_t]
    lib.fpvc_stream_info.restype = c.c_int
    lib.fpvc_stream_info.argtypes = [_u8p, c.c_size_t, st, st, st]
    lib.fpvc_unextract.restype = None
    lib.fpvc_unextract.argtypes = [_u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, _u8p]


class Codec:
    def __init__(self, lib_path=None):
        if lib_path is None:
            from . import host_lib
            self.lib = host_lib()
        else:
            self.lib = c.CDLL(lib_path)
            declare(self.lib)

    def encode_stream(self, frames, delta_raw, xs, ys, shift=0, big_endian=False, threads=4):
        frames = np.ascontiguousarray(frames, dtype=np.uint16).reshape(-1)
        delta_raw = np.ascontiguousarray(delta_raw, dtype=np.uint16).reshape(-1)
        n = frames.size // (xs * ys)
        cap = 64 + 16 * n + (n + 1) * (3 * (xs * ys + xs * ys // 8 + 1024))
        out = np.zeros(cap, np.uint8)
        if n == 0:
            frames = np.zeros(1, np.uint16)
        sz = self.lib.fpvc_encode_stream(frames, n, delta_raw, xs, ys, shift, int(big_endian),
                                         threads, out, cap)
        if sz == 0:
            raise RuntimeError("encode_stream: output did not fit")
        return out[:sz].copy()

    def encode_with_callbacks(self, frames, delta_raw, xs, ys, shift=0, big_endian=False, threads=4):
        """Drives Encoder::Init/CompressFrame/Finish directly; returns the list of
        (payload_id, bytes) in callback order."""
        frames = np.ascontiguousarray(frames, dtype=np.uint16).reshape(-1, xs * ys)
        delta_raw = np.ascontiguousarray(delta_raw, dtype=np.uint16).reshape(-1)
        got = []

        def on_write(data, size, payload):
            got.append((payload or 0, bytes(c.cast(data, c.POINTER(c.c_uint8 * size)).contents) if size else b""))

        cb = WRITE_FN(on_write)
        enc = self.lib.fpvc_encoder_new(threads, shift, int(big_endian))
        mq = self.lib.fpvc_encoder_max_queued(enc)
        self.lib.fpvc_encoder_init(enc, delta_raw, xs, ys, cb, c.c_void_p(1000000))
        for i in range(frames.shape[0]):
            self.lib.fpvc_encoder_compress_frame(enc, np.ascontiguousarray(frames[i]), cb, c.c_void_p(i + 1))
        self.lib.fpvc_encoder_finish(enc, cb, c.c_void_p(2000000))
        self.lib.fpvc_encoder_finish(enc, cb, c.c_void_p(3000000))   # second call: no-op
        self.last_error = self.lib.fpvc_encoder_last_error(enc).decode()
        self.lib.fpvc_encoder_free(enc)
        return mq, got

    def stream_info(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        xs, ys, n = c.c_size_t(), c.c_size_t(), c.c_size_t()
        if self.lib.fpvc_stream_info(data, data.size, c.byref(xs), c.byref(ys), c.byref(n)):
            raise RuntimeError("stream too short")
        return xs.value, ys.value, n.value

    def decode_stream(self, data, with_previews=False, batched=False, threads=4):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        xs, ys, n = self.stream_info(data)
        if xs == 0 or ys == 0 or xs > 65536 or ys > 65536 or n > data.size:
            xs = ys = 1
            n = 0
        out = np.zeros(max(n, 1) * xs * ys, np.uint16)
        pv = np.zeros(max(n, 1) * (xs // 4) * (ys // 4) + 1, np.uint8) if with_previews else None
        got = self.lib.fpvc_decode_stream(data, data.size, out, n,
                                          pv.ctypes.data_as(c.c_void_p) if with_previews else None,
                                          int(batched), threads)
        if got < 0:
            raise RuntimeError("decode_stream failed")
        frames = out[:got * xs * ys].reshape(got, ys, xs)
        if with_previews:
            return frames, pv[:got * (xs // 4) * (ys // 4)].reshape(got, ys // 4, xs // 4)
        return frames

    def decode_stream_raw(self, data, shift=0, big_endian=False, threads=4):
        """Whole stream -> raw file bytes per frame (UnextractFrame fused on the GPU)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        xs, ys, n = self.stream_info(data)
        out = np.zeros(max(n, 1) * xs * ys * 2, np.uint8)
        got = self.lib.fpvc_decode_stream_raw(data, data.size, out, n, shift, int(big_endian), threads)
        if got < 0:
            raise RuntimeError("decode_stream_raw failed")
        return out[:got * xs * ys * 2].reshape(got, xs * ys * 2)

    def stream_decode(self, data, xs, ys, nframes, block=65536):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(max(nframes, 1) * xs * ys, np.uint16)
        got = self.lib.fpvc_stream_decode(data, data.size, block, out, nframes)
        if got < 0:
            err = RuntimeError("stream_decode failed")
            err.delivered = out[:(-got - 1) * xs * ys].reshape(-got - 1, ys, xs)  # frames in front of the failure
            raise err
        return out[:got * xs * ys].reshape(got, ys, xs)

    def unextract(self, img, xs, ys, shift=0, big_endian=False):
        img = np.ascontiguousarray(img, dtype=np.uint16).reshape(-1)
        out = np.zeros(xs * ys * 2, np.uint8)
        self.lib.fpvc_unextract(img, xs, ys, shift, int(big_endian), out)
        return out
