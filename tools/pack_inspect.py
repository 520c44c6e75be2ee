#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a code-object directory (lib/jit_pack or ~/.cache/fmhip-jit): the 24-byte header of a
.co file is stripped and the AMDGPU metadata notes are read with llvm-readelf.  --dump NAME also writes the disassembly of that kernel's
code object to stdout (llvm-objdump -d).   usage: tools/pack_inspect.py [directory] [--dump fm_jit_<hash>]"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(path):
    blob = open(path, "rb").read()
    if blob[:8] != b"FMJITCO1":
        return [], None
    with tempfile.NamedTemporaryFile(suffix=".o", delete=False) as t:
        t.write(blob[24:])
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", t.name], capture_output=True, text=True).stdout
    found = []
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
        found.append({"name": m.group(2), "vgprs": int(m.group(5)), "sgprs": int(m.group(4)), "lds": int(m.group(1)), "scratch": int(m.group(3))})
    return found, t.name


def main():
    argv = sys.argv[1:]
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        del argv[i:i + 2]
    args = argv
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    directory = args[0] if args else os.path.join(here, "finmath-lib-cuda-extensions_amd", "lib", "jit_pack")
    rows = []
    for f in sorted(glob.glob(os.path.join(directory, "*.co"))):
        ks, obj = kernels(f)
        for k in ks:
            k["file"] = os.path.basename(f)
            rows.append(k)
            if dump and k["name"].startswith(dump) and k["name"].endswith("_t"):
                sys.stdout.write(subprocess.run([LLVM + "/llvm-objdump", "-d", "--symbolize-operands", obj], capture_output=True, text=True).stdout)
        if obj:
            os.unlink(obj)
    if dump:
        return
    print(f"{'kernel':34s} {'vgprs':>5s} {'sgprs':>5s} {'lds':>6s} {'scratch':>7s}  waves/SIMD  file")
    for k in sorted(rows, key=lambda k: (-k["vgprs"], k["name"])):
        waves = min(8, 512 // max(1, (k["vgprs"] + 7) // 8 * 8))
        print(f"{k['name']:34s} {k['vgprs']:5d} {k['sgprs']:5d} {k['lds']:6d} {k['scratch']:7d}  {waves:10d}  {k['file']}")


if __name__ == "__main__":
    main()
