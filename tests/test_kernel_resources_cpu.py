"""CPU: register and LDS budgets of the kernels that must share a CU with the LK launch, read from the built library.

DESIGN.md section 3: the LK kernel ALLOCATES 136 VGPRs so that three of its wavefronts per SIMD leave 512 - 3 * 136 = 104
registers per lane and 37 KB of LDS per CU to the helper kernels (frame preparation, compaction) that run beside it.  A
helper whose allocation exceeds that no longer fits beside LK and only runs in the gaps between launches: round 3 lost 25 %
of the 4K pipeline that way for an afternoon (one extra branch took min-eig from 64 to 106 registers) and only the final
profile run noticed.  This test reads the AMDGPU metadata notes of the gfx950 code object inside libpolychase_hip.so."""
import os
import struct

import pytest

from polychase_amd import build

HELPER_VGPR_BUDGET = 104          # 512 - 3 * 136, allocation granularity 8
HELPER_LDS_BUDGET = 18 * 1024     # two workgroups of a helper kernel per CU beside 12 LK wavefronts (120 of 160 KB)
# kernels enqueued by the analyzer while LK launches run (substring of the mangled name)
HELPERS = ["level_kernel", "min_eig_kernelILi0", "min_eig_kernelILi1", "min_eig_kernelILi2", "min_eig_kernelILi3", "min_eig_fused_kernelILi0", "min_eig_fused_kernelILi1",
           "min_eig_fused_kernelILi2", "min_eig_fused_kernelILi3", "nms_kernel", "bucket_scatter_kernel", "bucket_sort_kernel", "suppress_sorted_kernel",
           "accept_all_kernel", "accepted_scatter_kernel", "bin_scatter_kernel", "compact_count_kernel", "compact_scatter_kernel",
           "copy_keypoints_kernel", "lk_gate_kernel"]


def _code_objects(path):
    """the gfx950 ELFs inside the library's .hip_fatbin section (one clang offload bundle per translation unit)"""
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], blob.find(magic)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(magic))
        o = at + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tlen].decode()
            o += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + len(magic))
    assert out, "no gfx950 code object in the library"
    return out


def _kernel_metadata(elf):
    import msgpack

    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "expected a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:      # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:          # NT_AMDGPU_METADATA
                return msgpack.unpackb(elf[d:d + descsz], raw=False)["amdhsa.kernels"]
            p = d + (descsz + 3) // 4 * 4
    raise AssertionError("no AMDGPU metadata note")


@pytest.fixture(scope="module")
def kernels():
    path = build.hip_library_path()
    if not os.path.exists(path):
        build.build_hip()
    return {k[".name"]: k for elf in _code_objects(path) for k in _kernel_metadata(elf)}


def test_helper_kernels_fit_beside_three_lk_wavefronts(kernels):
    seen = set()
    for name, k in kernels.items():
        for h in HELPERS:
            if h in name:
                seen.add(h)
                alloc = (k[".vgpr_count"] + 7) // 8 * 8
                assert alloc <= HELPER_VGPR_BUDGET, f"{name}: {k['.vgpr_count']} VGPRs do not fit beside three LK wavefronts per SIMD"
                assert k[".group_segment_fixed_size"] <= HELPER_LDS_BUDGET, f"{name}: {k['.group_segment_fixed_size']} B of LDS"
                assert k.get(".private_segment_fixed_size", 0) == 0, f"{name}: spills to scratch"
    assert seen == set(HELPERS), f"kernels not found in the library: {set(HELPERS) - seen}"


def test_lk_kernel_allocation_is_what_the_budget_assumes(kernels):
    # ELb0: the canonical arithmetic; ELb1: the x86 summation order (PC_ARITH_LK_X86_ORDER) -- the same budget for both
    for variant in ("lk3_kernelILi10ELb0", "lk3_kernelILi10ELb1"):
        lk = [k for n, k in kernels.items() if variant in n]
        assert len(lk) == 1, variant
        k = lk[0]
        assert 129 <= k[".vgpr_count"] <= 136, (variant, k[".vgpr_count"])   # three wavefronts per SIMD, not two, not four
        assert k.get(".agpr_count", 0) == 0 and k.get(".private_segment_fixed_size", 0) == 0, variant
        assert k[".group_segment_fixed_size"] <= 10 * 1024 + 512, variant       # 12 wavefronts per CU hold <= 126 KB of the 160 KB
