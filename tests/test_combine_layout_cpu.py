"""Arithmetic of the fused final combine's arrival counting (csrc/fm_kernel_parts.hpp: combine_groups, block_combine),
restated: every workgroup of a row belongs to exactly one group, the group sizes the kernel computes add up to the row, the
group partials fit behind the workgroup partials, and the constants the host allocates with agree with the device header."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")


def combine_groups(blocks):          # fm_kernel_parts.hpp
    return (7 if blocks >= 1792 else blocks >> 8) if blocks >= 512 else 1


def test_groups_partition_the_row():
    for blocks in list(range(1, 2600)) + [4096, 8191, 8192, 8193, 65535, 65536]:
        G = combine_groups(blocks)
        assert 1 <= G <= 7
        members = [(blocks - g + G - 1) // G for g in range(G)]
        assert sum(members) == blocks and min(members) >= 1
        for g in range(G):                                            # member k of group g is workgroup g + k·G
            assert g + (members[g] - 1) * G < blocks <= g + members[g] * G + (G - 1)
        assert (G == 1) == (blocks < 512)


def test_constants_agree_between_host_and_device():
    parts = open(os.path.join(CSRC, "fm_kernel_parts.hpp")).read()
    prog = open(os.path.join(CSRC, "fm_program.h")).read()
    runtime = open(os.path.join(CSRC, "runtime.cpp")).read()
    slots = int(re.search(r"FM_COMBINE_GROUP_SLOTS\s*=\s*(\d+)", parts).group(1))
    planes = int(re.search(r"FM_COUNTER_PLANES\s*=\s*(\d+)", prog).group(1))
    assert slots >= 7 and planes >= 8                                  # 7 group counters + the second-level counter
    assert re.search(r"\(size_t\)bpr \+ %d\)" % slots, runtime), "the host allocates bpr + FM_COMBINE_GROUP_SLOTS partial slots per row"
    assert "(size_t)7 * FM_COUNTER_PLANE" in parts                     # second-level counter = plane 7
