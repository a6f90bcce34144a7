"""The C-ABI libraries load without a GPU and export every symbol their headers
declare; without a GPU they fail loudly instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fusion-power-video_amd", "lib")


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_hip_library_exports_every_declared_symbol():
    path = os.path.join(LIB, "libfpvhip.so")
    assert os.path.exists(path), "run `make product`"
    lib = ctypes.CDLL(path)
    names = declared("fpv_hip.h", "fpvhip_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fpv_hip.h but not exported"
    lib.fpvhip_abi_version.restype = ctypes.c_int
    assert lib.fpvhip_abi_version() == 1


def test_host_library_exports_c_api_and_cxx_api():
    lib = ctypes.CDLL(os.path.join(LIB, "libfpv_amd.so"))
    for n in declared("fpv_c_api.h", "fpvc_"):
        assert hasattr(lib, n), f"{n} declared in include/fpv_c_api.h but not exported"
    # the C++ API the upstream CLIs link against (Itanium-mangled)
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB, "libfpv_amd.so")],
                          capture_output=True, text=True).stdout
    for m in ("_ZN4fpvc7EncoderC1Emib", "_ZN4fpvc7Encoder4InitEPKtmmSt8functionIFvPKhmPvEES6_",
              "_ZN4fpvc7Encoder13CompressFrameEPKtSt8functionIFvPKhmPvEES6_", "_ZNK4fpvc7Encoder9MaxQueuedEv",
              "_ZN4fpvc16StreamingDecoder6DecodeEPKhmSt8functionIFvbPtmmPvEES5_",
              "_ZN4fpvc19RandomAccessDecoder4InitEPKhm", "_ZNK4fpvc19RandomAccessDecoder11DecodeFrameEmPt",
              "_ZNK4fpvc19RandomAccessDecoder13DecodePreviewEmPh", "_ZN4fpvc14UnextractFrameEPKtmmibPh"):
        assert m in syms, m


def test_every_hip_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "fpv_hip.h")).read()
    assert text.count("cc:") >= 10 and "h:68-73" in text


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = ctypes.CDLL(os.path.join(LIB, "libfpvhip.so"))
    lib.fpvhip_create.restype = ctypes.c_void_p
    lib.fpvhip_last_error.restype = ctypes.c_char_p
    lib.fpvhip_last_error.argtypes = [ctypes.c_void_p]
    assert lib.fpvhip_device_count() <= 0
    assert not lib.fpvhip_create(0, 64, 48, 0, 0, 4, 1)
    assert lib.fpvhip_last_error(None)


def test_product_never_links_the_oracle():
    import subprocess
    for so in ("libfpvhip.so", "libfpv_amd.so"):
        out = subprocess.run(["nm", "-D", os.path.join(LIB, so)], capture_output=True, text=True).stdout
        assert "fpvo_" not in out, f"{so} references the test oracle"


def test_unsupported_geometry_is_a_clean_error():
    """Geometries the forward kernels cannot take fail in fpvhip_create with a message (no
    abort, no GPU needed to find out): odd sizes, frames wider than the LDS line buffers."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "fusion-power-video_amd", "lib", "libfpvhip.so"))
    lib.fpvhip_create.restype = ctypes.c_void_p
    lib.fpvhip_create.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_uint32, ctypes.c_uint32]
    lib.fpvhip_last_error.restype = ctypes.c_char_p
    lib.fpvhip_last_error.argtypes = [ctypes.c_void_p]
    for xs, ys, shift, what in ((65536, 4, 0, b"too large"), (1282, 800, 0, b"multiples of 4"),
                                (1280, 798, 0, b"multiples of 4"), (1280, 800, 16, b"shift")):
        assert not lib.fpvhip_create(0, xs, ys, shift, 0, 1, 1)
        assert what in lib.fpvhip_last_error(None), (xs, ys, lib.fpvhip_last_error(None))
