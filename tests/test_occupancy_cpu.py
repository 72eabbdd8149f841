"""The register / LDS budget of the persistent kernels, read from the code objects inside libegohmr_hip.so (no GPU needed).

Every tile engine here launches 2 x CUs blocks of 256 threads and counts on BOTH being resident on a CU: the chained hidden convs poll
counters of blocks that must be running, the stream-K convs wait for partial sums of their neighbours, and the host-side plans (ticket
queues, tile rounds, stream-K runs) are sized for 2 blocks per CU.  That holds while a block needs at most 80 KiB of LDS and 256 registers
(VGPRs + AGPRs) per lane.  hipcc does not enforce `__launch_bounds__(256, 2)`: when something else already costs the second block (4 bytes of LDS
too many are enough) it quietly spends more registers, and the kernel runs at half occupancy with every test still green - round 5 shipped the
stream-K conv that way for a while (`__syncthreads_or` brings its own LDS word: 183 VGPRs + 96 AGPRs, 187 -> 269 us per launch)."""
import struct

import pytest

msgpack = pytest.importorskip("msgpack")


def _device_kernels(path):
    """{kernel name: metadata dict} of every AMDGPU code object embedded in the shared library"""
    blob = open(path, "rb").read()
    out, pos = {}, 0
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            break
        elf = blob[pos:]
        pos += 4
        if len(elf) < 64 or elf[4] != 2 or struct.unpack_from("<H", elf, 18)[0] != 224:      # ELF64, EM_AMDGPU
            continue
        shoff, = struct.unpack_from("<Q", elf, 40)
        shentsize, shnum = struct.unpack_from("<HH", elf, 58)
        for i in range(shnum):
            sh = shoff + i * shentsize
            sh_type, = struct.unpack_from("<I", elf, sh + 4)
            off, size = struct.unpack_from("<QQ", elf, sh + 24)
            if sh_type != 7:                                                                  # SHT_NOTE
                continue
            p = off
            while p + 12 <= off + size:
                namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
                d0 = p + 12 + (namesz + 3) // 4 * 4
                if name == b"AMDGPU" and ntype == 32:                                         # NT_AMDGPU_METADATA
                    md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                    for k in md.get("amdhsa.kernels", []):
                        out[k[".name"]] = k
                p = d0 + (descsz + 3) // 4 * 4
    return out


@pytest.fixture(scope="module")
def kernels():
    from egohmr_amd import _lib
    _lib.build()
    ks = _device_kernels(_lib.LIB_PATH)
    assert len(ks) > 40, f"only {len(ks)} kernels found in {_lib.LIB_PATH}: the code-object parser no longer matches the library"
    return ks


def _regs(k):
    return k[".vgpr_count"] + k.get(".agpr_count", 0)


# (substring of the mangled name, threads per block, blocks per CU the host side counts on)
TWO_PER_CU = ["conv_x2_tile_kernel", "linear_tile_kernel", "gcn_hidden_chain_kernelILi3ELi4E", "gcn_hidden_chain_kernelILi1ELi4E",
              "gcn_hidden_tile_kernel"]


def test_tile_engines_keep_two_blocks_per_cu(kernels):
    seen = 0
    for name, k in kernels.items():
        if not any(t in name for t in TWO_PER_CU):
            continue
        seen += 1
        assert k[".max_flat_workgroup_size"] == 256, name
        assert k[".group_segment_fixed_size"] <= 80 * 1024, f"{name}: {k['.group_segment_fixed_size']} bytes of LDS - the second block of a CU does not fit"
        assert _regs(k) <= 256, f"{name}: {k['.vgpr_count']} VGPRs + {k.get('.agpr_count', 0)} AGPRs - one wave per SIMD only"
        assert k[".private_segment_fixed_size"] == 0, f"{name}: {k['.private_segment_fixed_size']} bytes of scratch per lane"
    assert seen >= 20, f"{seen} tile-engine kernels matched: the name patterns are stale"


def test_eight_wave_chain_kernel_is_one_block_of_the_whole_cu(kernels):
    ks = [k for n, k in kernels.items() if "gcn_hidden_chain_kernelILi1ELi8E" in n]
    assert len(ks) == 1
    k = ks[0]
    assert k[".max_flat_workgroup_size"] == 512 and k[".group_segment_fixed_size"] <= 160 * 1024 and _regs(k) <= 256
    assert k[".private_segment_fixed_size"] == 0


def test_per_body_step_kernels_do_not_spill(kernels):
    seen = 0
    for name, k in kernels.items():
        if "step_fused_kernel" in name:
            seen += 1
            nt = k[".max_flat_workgroup_size"]
            assert nt in (256, 512, 1024), name
            assert _regs(k) <= 128, f"{name}: {_regs(k)} registers - a CU no longer holds 2048 of these threads"
            # (the 512- / 256-thread forms - more bodies than CUs - keep a dozen spilled dwords at their 128-register cap; the 1024-thread form none)
            assert k[".private_segment_fixed_size"] <= (64 if nt <= 512 else 0), f"{name}: {k['.private_segment_fixed_size']} bytes of scratch per lane"
            if nt == 256:                                   # four of these per CU: LDS must leave room for them
                assert k[".group_segment_fixed_size"] <= 40 * 1024, f"{name}: {k['.group_segment_fixed_size']} bytes of LDS"
    assert seen >= 4


def test_skinning_keeps_four_blocks_per_cu(kernels):
    ks = [k for n, k in kernels.items() if "skin_mfma_kernel" in n or "skin_input_kernel" in n]
    assert len(ks) == 4
    for k in ks:
        assert _regs(k) <= 128 and k[".group_segment_fixed_size"] <= 40 * 1024 and k[".private_segment_fixed_size"] == 0, k[".name"]
