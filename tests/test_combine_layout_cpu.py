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
    """Q = workgroups per span of the reduction tree: 1 (a workgroup takes a span) or 4 (one unit each); the groups count SPANS."""
    for Q in (1, 4):
        for blocks in list(range(1, 2600)) + [4096, 8191, 8192, 8193, 65535, 65536]:
            spans = (blocks + Q - 1) // Q
            G = combine_groups(spans)
            assert 1 <= G <= 7
            group_spans = [(spans - g + G - 1) // G for g in range(G)]
            assert sum(group_spans) == spans and min(group_spans) >= 1
            # what block_combine computes: workgroups of a group = Q per span, the row's last span may have fewer
            members = [group_spans[g] * Q - ((spans * Q - blocks) if (spans - 1) % G == g else 0) for g in range(G)]
            counted = [0] * G
            for b in range(blocks):
                counted[(b // Q) % G] += 1
            assert counted == members and sum(members) == blocks
            for g in range(G):                                            # span k of group g is span g + k·G
                assert g + (group_spans[g] - 1) * G < spans <= g + group_spans[g] * G + (G - 1)
            assert (G == 1) == (spans < 512)


def test_constants_agree_between_host_and_device():
    parts = open(os.path.join(CSRC, "fm_kernel_parts.hpp")).read()
    prog = open(os.path.join(CSRC, "fm_program.h")).read()
    runtime = open(os.path.join(CSRC, "runtime.cpp")).read()
    slots = int(re.search(r"FM_COMBINE_GROUP_SLOTS\s*=\s*(\d+)", parts).group(1))
    planes = int(re.search(r"FM_COUNTER_PLANES\s*=\s*(\d+)", prog).group(1))
    assert slots >= 7 and planes >= 8                                  # 7 group counters + the second-level counter
    assert re.search(r"\(blocks_per_row \+ %d\)" % slots, runtime), "the host allocates blocks + FM_COMBINE_GROUP_SLOTS partial slots per row"
    assert "(size_t)7 * FM_COUNTER_PLANE" in parts                     # second-level counter = plane 7
