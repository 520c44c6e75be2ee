#!/bin/bash
# The read pattern of the valuation chains and the generated valuation kernel in isolation (benchmarks/read_pattern.hip,
# benchmarks/peeled_harness.hip; the ph_* binaries are built from the generated source and hand-edited variants: DESIGN.md §4.5).
echo "# benchmarks/read_pattern.hip, one MI355X; this box:"
finmath-lib-cuda-extensions_amd/bin/box_speed
timeout -k 5 200 benchmarks/build/read_pattern 88 61 4
echo
echo "# benchmarks/peeled_harness.hip on the generated valuation kernel (88 rows x 58 periods x 1 M paths) and hand-edited variants of its source:"
echo "#   A as generated (round-3 pipeline, reduces its root, stores it); B no reduction; C ping-pong pipeline; D the loop's division replaced by a product;"
echo "#   E scalar operands as literals; F = B + E; G no input behind the loop; H that input loaded before the loop; I no arithmetic in the loop; K = I without the store of the root"
for v in A B C D E F G H I K; do echo -n "$v: "; timeout -k 5 60 benchmarks/build/ph_$v 88 58 1000000 10; done
