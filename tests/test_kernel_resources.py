"""What the compiler made of the kernels, read from the code object inside the built libfpvhip.so
(no GPU needed): no kernel of the library may use scratch memory (a spill in the tile pass costs a
wait for the load it spills -- round 4 found one that way), and the register counts that decide the
workgroups per CU must stay inside their budgets:
  k_tile     <= 128 VGPRs (four waves per SIMD: two 512-thread workgroups per CU)
  k_tile8<4> <=  64 VGPRs (eight waves per SIMD: two 1024-thread workgroups per CU)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "fusion-power-video_amd", "lib", "libfpvhip.so")


def _kernels(tmp_path):
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not in this image")
    if not os.path.exists(LIB):
        pytest.skip("libfpvhip.so not built")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "co.elf")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                     for k in ("private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size")}
        out[name]["agpr_count"] = int(re.match(r"\s*(\d+)", blk).group(1))
    return out


def test_no_kernel_spills_and_register_budgets_hold(tmp_path):
    ks = _kernels(tmp_path)
    assert len(ks) > 40, len(ks)
    spilling = {n: k["private_segment_fixed_size"] for n, k in ks.items() if k["private_segment_fixed_size"]}
    assert not spilling, spilling
    tile = [k for n, k in ks.items() if re.search(r"\d6k_tileILb", n)]
    tile8 = [k for n, k in ks.items() if re.search(r"\d7k_tile8ILb[01]ELb[01]ELi4E", n)]
    assert len(tile) == 4 and len(tile8) == 4
    assert all(k["vgpr_count"] + k["agpr_count"] <= 128 for k in tile), tile
    assert all(k["vgpr_count"] + k["agpr_count"] <= 64 for k in tile8), tile8
    # the three-wave inverse of the 1000-frame batch at 1280x800: four 192-thread workgroups per CU (1024 frames resident at
    # once) are three waves per SIMD, i.e. <= 168 registers -- round 5's k_inverse_trio<5, 2, true> (in place, blocks of two
    # supersteps at the geometry's own segment length); round 4's choice k_inverse_trio<4, 4> (both forms) stays within 128
    # (<5, 2, true, RES>: both the form that reads its north bytes from the ring and the one that keeps them in registers; the
    # latter in blocks of four, <5, 4, true, true>, is what partial batches take: three workgroups per CU are enough there)
    trio = {}
    for n, k in ks.items():   # k_inverse_trio<K4, M, IP, RES, M2>
        m = re.search(r"k_inverse_trioILi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELb([01])E", n)
        if m:
            trio[tuple(int(v) for v in m.groups())] = k["vgpr_count"] + k["agpr_count"]
    assert trio[(5, 2, 1, 0, 0)] <= 168 and trio[(5, 2, 1, 1, 0)] <= 168, trio
    assert trio[(5, 4, 1, 1, 0)] <= 168, trio
    assert all(trio[(4, 4, ip, res, 0)] <= 128 for ip, res in ((0, 0), (1, 0), (1, 1))), trio
    # rows of 2 ... 4 supersteps (m + 1 register sets), the long segments (32 / 40 / 48 pixels per lane) and the forms with the memo
    # behind the second pass: partial batches only, at most two workgroups per CU -- no spills (checked above), registers within the file
    multi = [v for (k4, mb, ip, res, m2), v in trio.items() if res >= 2 and not m2]
    memo2 = [v for (k4, mb, ip, res, m2), v in trio.items() if m2]
    assert len(multi) == 8 and max(multi) <= 256, trio
    assert len(memo2) == 17 and max(memo2) <= 256, trio
    assert trio[(5, 2, 1, 0, 1)] <= 168, trio   # full batches of 1280 x 800: four workgroups per CU
