"""CPU-side checks of the specialised-kernel tier: the source generator (needs no device) and, when hiprtc is usable in
this container, that the generated source compiles for gfx950 with the library's own embedded headers."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(fm):
    p = fm.Program(3)
    t = p.op("SUB", p.op("MULT", p.op("DIV_S", p.op("ADD_S", 0, s=4.0), s=2.0), 1), 2)
    u = p.op("SQRT", p.op("ABS", p.op("LOG", p.op("EXP", t))))
    w = p.op("CHOOSE", t, u, 0)
    p.output(w)
    p.reduce(w)
    return p


def test_generated_source_shape():
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    src = build(fm).source()
    assert src.count('extern "C" __global__') == 2                      # row block inline (batch 1) and row table flavours
    assert "fm_jit_inline" in src and "fm_jit_table" in src
    assert '#include "fm_kernel_parts.hpp"' in src
    assert "red_accumulate<E>" in src and "block_combine<1>" in src     # the interpreter's own reduction code
    assert src.count("ueval<") == 2 * 8                                 # LDA + 7 micro-ops (DIV_S by 2 became MULT_S; CHOOSE reuses t) …
    assert src.count("sqrt_all<E>(a)") == 2                             # … sqrt, whose E elements share one special-case branch …
    assert src.count("log_all<4>(a + g)") == 2                          # … and log, likewise, in groups of four (bounded live fp64 temporaries)
    # deterministic: the text is the cache key of the compiled kernel
    assert src == build(fm).source()


def test_source_of_a_bad_description_is_an_error():
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    import pytest
    p = fm.Program(1)
    p.op("ADD", 0, 5)                   # operand 5 does not exist
    p.output(1)
    with pytest.raises(fm.FmhipError):
        p.source()


def build_dividing(fm):
    p = fm.Program(3)
    a = p.op("DIV", 0, 1)
    b = p.op("DISCOUNT", a, 2, s=0.5)
    c = p.op("VID_S", p.op("INVERT", b), s=3.0)
    d = p.op("SUBRATIO", p.op("ADDRATIO", c, 0, 1), 2, 1)
    e = p.op("DIV_S", d, s=3.0)                                          # not a power of two: stays a division
    p.output(e)
    return p


def test_divisions_are_evaluated_in_pairs_and_the_source_compiles(tmp_path):
    """Micro-ops with a division go through ueval_div_all (IEEE division with packed multiply-adds, fm_device_math.hpp); the
    generated translation unit must compile for gfx950 against the device headers (hipcc cross-compiles without a GPU)."""
    import shutil
    import subprocess
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    src = build_dividing(fm).source()
    assert src.count("ueval_div_all<") == 2 * 7
    assert " / " not in src.split("fm_jit_inline", 1)[1].replace("E / FM_VEC", "").replace("- 1u) / gridDim.x", "").replace("- 1u) / tile_step", "")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    f = tmp_path / "dividing.hip"
    f.write_text(src)
    csrc = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
                          "-I", csrc, "--cuda-device-only", "-S", "-o", str(tmp_path / "dividing.s"), str(f)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    asm = (tmp_path / "dividing.s").read_text()
    assert "v_pk_fma_f32" in asm and "v_div_fixup_f32" in asm


def test_kernel_pack_is_built_from_its_descriptions():
    """csrc/kernel_pack.txt (program and rolled-loop descriptions of the flagship workloads) is compiled at build time, without a
    device, into lib/jit_pack — one code object per distinct description (the build fails on an entry that does not parse or compile)."""
    pkg = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd")
    with open(os.path.join(pkg, "csrc", "kernel_pack.txt")) as fh:
        lines = {ln.strip() for ln in fh if ln.strip() and not ln.startswith("#")}
    assert len(lines) >= 20 and all(ln.startswith(("program variant ", "rolled elems ")) for ln in lines)
    assert sum(ln.startswith("rolled ") for ln in lines) >= 4
    pack = os.path.join(pkg, "lib", "jit_pack")
    if not os.path.isdir(pack):
        import __graft_entry__
        __graft_entry__.build()
    objects = [f for f in os.listdir(pack) if f.endswith(".co")]
    assert len(objects) == len(lines)
    with open(os.path.join(pack, objects[0]), "rb") as fh:
        assert fh.read(8) == b"FMJITCO1"                                # the cache's own container (jit.cpp: cache_store)
    # every description survives parse → describe unchanged (what is compiled is what was recorded)
    import subprocess
    tool = os.path.join(pkg, "build", "jit_pack_tool")
    out = subprocess.run([tool, "--check", os.path.join(pkg, "csrc", "kernel_pack.txt")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith(f"{len(lines)} descriptions, 0 round-trip differences")


VALUATION = ("rolled elems 8 log 0 globals 0 inputs 1 carried 3 final 3 out body 14:l0:-:-:s 16:v0:-:-:s 22:c0:v1:-:- 30:v2:l0:-:s peel 2 1 init p6 preout postout 1 "
             "finalstore 0 pre 14:x0:-:-:s 16:p0:-:-:s 30:p1:x0:-:s 14:x1:-:-:s 16:p3:-:-:s 22:p2:p4:-:- 30:p5:x1:-:s post 12:F0:-:-:s 26:q0:x2:-:- reduce q1")


def test_merged_chain_kernel_source(tmp_path):
    """The merged form of a loop shape (jit.hpp: RolledBody::chains; the swaption's backward induction of SwaptionSimple): K chains over one
    sequence of vectors.  Every step loads its vector once and prepares the discount denominator once; each chain runs the head's two
    stages, then the loop body, on that vector; the tails share the numeraire; K reductions, one hand-off.  A shape that does not meet the
    preconditions has no merged form (empty source: the engine then leaves its components alone).  The source compiles for gfx950."""
    import shutil
    import subprocess
    import pytest
    pkg = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd")
    tool = os.path.join(pkg, "build", "jit_pack_tool")
    if not os.path.exists(tool):
        import __graft_entry__
        __graft_entry__.build()
    K = 5
    desc = tmp_path / "merged.txt"
    desc.write_text(VALUATION + f" chains {K} sden 1\n")
    chk = subprocess.run([tool, "--check", str(desc)], capture_output=True, text=True, timeout=60)
    assert chk.returncode == 0 and chk.stdout.startswith("1 descriptions, 0 round-trip differences"), chk.stdout + chk.stderr
    src = subprocess.run([tool, "--source", str(desc), "rolled"], capture_output=True, text=True, timeout=60).stdout
    assert f"merged loop: {K} chains" in src and "shared denominators" in src
    assert src.count("load_stream(p, i4c[t])") == 3                       # the first step's vector, the next step's inside the loop, the numeraire: nothing per chain
    assert src.count("div_prepare_discount<E>(D, l, sden)") == 1            # the denominator of every discount of a step, once
    assert src.count("ueval_div_prepared<E>(") == 3 * K                   # head stage 0, head stage 1, body: per chain
    assert src.count("ueval_div_all<26u, E>(") == K                       # the tail's division by the numeraire
    assert src.count("red_chain_unit<K, E>(") == K and src.count("block_combine_values<K>(") == 1
    assert src.count("if (t >= t0_") == K and src.count("__builtin_nontemporal_store(") == K      # a chain joins at its own first step; its root is stored once
    # no merged form: a shape whose loop stores a value in every iteration
    bad = tmp_path / "bad.txt"
    bad.write_text(VALUATION.replace(" out body", " out 3 body") + f" chains {K} sden 1\n")
    assert subprocess.run([tool, "--source", str(bad), "rolled"], capture_output=True, text=True, timeout=60).stdout == ""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    f = tmp_path / "merged.hip"
    f.write_text(src)
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-structurizecfg-skip-uniform-regions",
                          "-I", os.path.join(pkg, "csrc"), "--cuda-device-only", "-S", "-o", str(tmp_path / "merged.s"), str(f)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    asm = (tmp_path / "merged.s").read_text()
    assert ".private_segment_fixed_size: 0" in asm.split("fm_jit_table", 1)[1]        # nothing spills


def test_cross_workgroup_hand_off_in_the_machine_code(tmp_path):
    """The hand-off of the reduction partials between workgroups (fm_kernel_parts.hpp: block_combine), checked where it counts — in the
    gfx950 instructions hipcc emits for the same header both tiers are built from: every arrival-counter add is preceded by a full drain
    of the vector-memory counter with no store in between (the sc1 partial stores are in memory before the workgroup is counted), the
    partial stores and EVERY load behind the first add carry sc1 (no stale line from the L1 or a foreign XCD's L2), and an agent-scope
    acquire (buffer_inv sc1) stands between each add and the loads of the workgroup that arrived last."""
    import re
    import shutil
    import subprocess
    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this environment")
    fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
    src = tmp_path / "reduce.hip"
    src.write_text(build(fm).source())
    out = tmp_path / "reduce.s"
    csrc = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-structurizecfg-skip-uniform-regions",
                           "-I", csrc, "-S", "--cuda-device-only", "-o", str(out), str(src)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    for kernel in ("fm_jit_inline", "fm_jit_table"):
        body = text[text.index(kernel + ":"):]
        body = body[:body.index(".Lfunc_end")]                         # (the kernel has several exits: waves that are done leave early)
        ins = [ln.strip() for ln in body.splitlines() if re.match(r"\s+[a-z]", ln)]
        adds = [i for i, x in enumerate(ins) if x.startswith("global_atomic_add")]
        assert len(adds) == 3, (kernel, len(adds))                      # the group counter, the count of counted groups of a one-group row (round 4), the second-level counter
        for a in adds:
            back = ins[:a][::-1]
            drain = next(i for i, x in enumerate(back) if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
            assert not any(x.startswith(("global_store", "global_atomic", "flat_store")) for x in back[:drain]), (kernel, "a store between the drain and the counter")
            assert sum(1 for x in back[drain:] if x.startswith("global_store_dwordx2") and "sc1" in x) >= 4, (kernel, "no sc1 partial stores in front of the drain")
            ahead = ins[a + 1:]
            inv = next(i for i, x in enumerate(ahead) if x.startswith("buffer_inv"))
            first_load = next(i for i, x in enumerate(ahead) if x.startswith("global_load"))
            assert "sc1" in ahead[inv] and inv < first_load, (kernel, "no acquire between the counter and the partial loads")
        behind = [x for x in ins[adds[0] + 1:] if x.startswith(("global_load", "flat_load", "buffer_load"))]
        assert behind and all("sc1" in x for x in behind), (kernel, [x for x in behind if "sc1" not in x][:3])


def test_pack_code_objects_hand_off(tmp_path):
    """The same check on the code objects that actually ship (lib/jit_pack, compiled with hiprtc at build time): disassembled with
    llvm-objdump, every kernel that counts arrivals drains before the add and invalidates behind it."""
    import glob
    import re
    import subprocess
    import pytest
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    pack = sorted(glob.glob(os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "lib", "jit_pack", "*.co")))
    if not os.path.exists(objdump) or not pack:
        pytest.skip("no kernel pack / llvm-objdump in this environment")
    checked = 0
    for path in pack:
        raw = open(path, "rb").read()
        assert raw[:8] == b"FMJITCO1"
        co = tmp_path / "k.co"
        co.write_bytes(raw[24:])                                         # magic, check word, size; then the ELF code object
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True).stdout
        ins = [ln.split("//")[0].strip() for ln in dis.splitlines() if re.match(r"\s+[a-z_0-9]+ ", ln)]
        adds = [i for i, x in enumerate(ins) if x.startswith("global_atomic_add")]
        for a in adds:
            back = ins[:a][::-1]
            drain = next(i for i, x in enumerate(back) if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
            assert not any(x.startswith(("global_store", "flat_store")) for x in back[:drain]), path
            ahead = ins[a + 1:a + 200]
            assert any(x.startswith("buffer_inv") and "sc1" in x for x in ahead), path
            checked += 1
    assert checked >= 4                                                  # stream S, the stand-alone reduction …: two adds per kernel flavour
