"""ctypes driver for the Frame-API C entry points (fpv_frame_capi.inc), usable
with the product library (prefix fpvc_), the CPU test shim, or the reference
shim oracle/_ref (prefix ref_)."""
import ctypes as c

import numpy as np

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C")


class FrameApi:
    def __init__(self, lib, prefix):
        f = getattr(lib, prefix + "frame_compress_full")
        f.restype = c.c_size_t
        f.argtypes = [u16p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, u8p, c.c_size_t]
        g = getattr(lib, prefix + "frame_cycle")
        g.restype = c.c_uint64
        g.argtypes = [u16p, u16p, c.c_size_t, c.c_size_t, c.c_int, c.c_int, c.c_int] + [u8p] * 7
        h = getattr(lib, prefix + "frame8_compress_core")
        h.restype = c.c_size_t
        h.argtypes = [u8p, u8p, c.c_size_t, c.c_size_t, u8p, c.c_size_t]
        self.f, self.g, self.h = f, g, h

    def compress_full(self, img, delta, xs, ys, shift, be):
        img = np.ascontiguousarray(img, np.uint16).reshape(-1)
        d = None if delta is None else np.ascontiguousarray(delta, np.uint16).reshape(-1)
        out = np.zeros(4 * xs * ys + 4096, np.uint8)
        sz = self.f(img, None if d is None else d.ctypes.data, xs, ys, shift, int(be), out, out.size)
        return out[:sz].copy()

    def cycle(self, img, delta, xs, ys, shift, be, parallel=True):
        n = xs * ys
        img = np.ascontiguousarray(img, np.uint16).reshape(-1)
        d = np.ascontiguousarray(delta, np.uint16).reshape(-1)
        bufs = [np.zeros(n + 16, np.uint8) for _ in range(3)] + [np.zeros(4, np.uint8)] + \
               [np.zeros(n + 16, np.uint8) for _ in range(3)]
        packed = self.g(img, d, xs, ys, shift, int(be), int(parallel), *bufs)
        return dict(sizes=np.array([packed & 0x1fffff, (packed >> 21) & 0x1fffff, packed >> 42], np.int64),
                    pred_high=bufs[0][:n], pred_low=bufs[1][:n], pred_preview=bufs[2][:n // 16],
                    info=bufs[3], rest_high=bufs[4][:n], rest_low=bufs[5][:n], rest_preview=bufs[6][:n // 16])

    def compress8_core(self, img8, delta8, xs, ys):
        out = np.zeros(2 * xs * ys + 4096, np.uint8)
        sz = self.h(np.ascontiguousarray(img8, np.uint8).reshape(-1),
                    np.ascontiguousarray(delta8, np.uint8).reshape(-1), xs, ys, out, out.size)
        return out[:sz].copy()


def cases(synth):
    """(name, img, delta-or-None, shift, big_endian) at 64x48."""
    xs, ys = 64, 48
    fr = synth.plasma_frames(3, xs, ys, bits=12, seed=3)
    nz = synth.noise_frames(3, xs, ys, seed=4)
    return xs, ys, [
        ("plasma_s4", fr[1], fr[0], 4, False),
        ("plasma_s4_nodelta", fr[1], None, 4, False),
        ("noise", nz[1], nz[0], 0, False),
        ("noise_be", nz[2].byteswap(), nz[0].byteswap(), 0, True),
        ("plasma8_s8", (fr[2] >> 4).astype(np.uint16), (fr[0] >> 4).astype(np.uint16), 8, False),
        ("zero_low_vs_delta", (nz[1] & 0xFF00).astype(np.uint16), nz[0], 0, False),
        ("same_as_delta", fr[0], fr[0], 4, False),
    ]
