"""the Java / JNI shim exists as SOURCE (java/net/finmath/hip/*.java, src/jni/fmhip_jni.cpp) and cannot
be compiled here (no JDK, no jni.h, no finmath-lib jar).  What CAN be checked without a JVM, and is on every run:

  * every function include/fmhip.h exports has exactly one native method in Native.java and one
    Java_net_finmath_hip_Native_* function in fmhip_jni.cpp — and nothing is bound that the header does not export;
  * the JNI functions call the C function they are named after;
  * Opcode.java carries the header's opcode numbers;
  * RandomVariableHip overrides the 64 methods RandomVariableCuda overrides (the method set of the interface the reference
    implements, RandomVariableCuda.java — counted from the list in SURVEY.md §8a), the factory its two, BrownianMotionHip the
    BrownianMotion methods of BrownianMotionCudaWithRandomVariableCuda.java:111-259."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "net", "finmath", "hip")


def read(*parts):
    with open(os.path.join(*parts)) as fh:
        return fh.read()


def camel(c_name):
    parts = c_name[len("fmhip_"):].split("_")
    return parts[0] + "".join(p.capitalize() for p in parts[1:])


def header_exports():
    text = re.sub(r"/\*.*?\*/", "", read(ROOT, "include", "fmhip.h"), flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|double|const char\*)\s+(fmhip_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_every_export_has_one_native_method_and_one_jni_function():
    exports = header_exports()
    assert len(exports) >= 51
    native = re.findall(r"static native [\w\[\]]+\s+(\w+)\s*\(", read(JAVA, "Native.java"))
    jni_text = read(ROOT, "src", "jni", "fmhip_jni.cpp")
    jni = re.findall(r"^FMJ\(\w+,\s*(\w+)\)", jni_text, flags=re.M)
    assert len(native) == len(set(native)) and len(jni) == len(set(jni)), "duplicate binding"
    wanted = {camel(e): e for e in exports}
    assert set(native) == set(wanted), (sorted(set(wanted) - set(native)), sorted(set(native) - set(wanted)))
    assert set(jni) == set(wanted), (sorted(set(wanted) - set(jni)), sorted(set(jni) - set(wanted)))
    # each JNI function calls the C function it is named after
    bodies = re.split(r"(?m)^FMJ\(\w+,\s*", jni_text)[1:]
    for body in bodies:
        name = body[:body.index(")")]
        assert wanted[name] + "(" in body, f"{name} does not call {wanted[name]}"


def test_opcodes_match_the_header():
    header = dict((k, int(v)) for k, v in re.findall(r"FMHIP_OP_([A-Z_]+?)\s*=\s*(\d+)", read(ROOT, "include", "fmhip.h")) if k != "_COUNT")
    java = dict((k, int(v)) for k, v in re.findall(r"\b([A-Z][A-Z_]+)\s*=\s*(\d+)", read(JAVA, "Opcode.java")))
    assert java == header and len(java) == 31


def test_the_classes_override_the_interface_methods_the_reference_overrides():
    rv = read(JAVA, "RandomVariableHip.java")
    assert "implements RandomVariable" in rv
    overrides = re.findall(r"@Override\s+public\s+[\w\[\]<>.]+\s+(\w+)\s*\(([^)]*)\)", rv)
    assert len(overrides) == 64                                   # RandomVariableCuda.java: 64 @Override
    names = {n for n, _ in overrides}
    for method in ("cap floor add sub bus mult div vid pow squared sqrt exp log sin cos invert abs isNaN accrue discount choose addProduct addRatio "
                   "subRatio getAverage getVariance getSampleVariance getStandardDeviation getStandardError getMin getMax getQuantile "
                   "getQuantileExpectation getHistogram getRealizations doubleValue isDeterministic size get getFiltrationTime getTypePriority "
                   "cache apply average equals getOperator getRealizationsStream").split():
        assert method in names, method
    factory = read(JAVA, "RandomVariableHipFactory.java")
    assert "extends AbstractRandomVariableFactory" in factory and factory.count("@Override") == 2      # RandomVariableCudaFactory.java:27-34
    bm = read(JAVA, "BrownianMotionHip.java")
    assert "implements BrownianMotion" in bm
    for method in ("getCloneWithModifiedSeed", "getCloneWithModifiedTimeDiscretization", "getBrownianIncrement", "getTimeDiscretization",
                   "getNumberOfFactors", "getNumberOfPaths", "getRandomVariableForConstant", "getIncrement", "getSeed", "equals", "hashCode", "toString"):
        assert re.search(r"public\s+[\w.]+\s+" + method + r"\s*\(", bm), method
    assert "Native.bmGenerate(" in bm                              # ONE native call for all steps x factors


def test_status_is_stated():
    """The sources say of themselves that they have never been compiled here, and so do README and INTEGRATION."""
    for path in (os.path.join(JAVA, "Native.java"), os.path.join(JAVA, "RandomVariableHip.java"), os.path.join(ROOT, "src", "jni", "fmhip_jni.cpp"),
                 os.path.join(ROOT, "CMakeLists.txt")):
        assert re.search(r"UNCOMPILED|never configured or compiled", read(path)), path
    for doc in ("README.md", "INTEGRATION.md"):
        assert re.search(r"uncompiled|not compiled|never compiled", read(ROOT, doc), flags=re.I), doc


def test_jni_layer_compiles_against_a_declaration_stub_and_links_against_the_library(tmp_path):
    """No JDK here — but a hand-written declaration stub of <jni.h> (tests/jni_stub/jni.h: the specification's types and the JNIEnv
    members the layer uses, nothing with a body) is enough to put src/jni/fmhip_jni.cpp through a C++ compiler with all warnings on
    and to link it against libfmhip.so: every C++ error, every call that does not match include/fmhip.h, every fmhip_* function
    that the library does not export shows up in this test instead of at a user's first build.  The resulting library exports
    exactly one Java_net_finmath_hip_Native_* symbol per native method of Native.java."""
    import shutil
    import subprocess
    import pytest
    gxx = shutil.which("g++")
    lib = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "lib", "libfmhip.so")
    if not gxx or not os.path.exists(lib):
        pytest.skip("needs g++ and the built libfmhip.so")
    out = tmp_path / "libfmhip_jni.so"
    res = subprocess.run([gxx, "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-Wl,--no-undefined-version",
                          "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "src", "jni", "fmhip_jni.cpp"), "-o", str(out), "-L", os.path.dirname(lib), "-lfmhip"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    defined = subprocess.run(["nm", "-D", "--defined-only", str(out)], capture_output=True, text=True).stdout
    exported = sorted(re.findall(r"Java_net_finmath_hip_Native_(\w+)", defined))
    native = sorted(re.findall(r"static native [\w\[\]]+\s+(\w+)\s*\(", read(JAVA, "Native.java")))
    assert exported == native
    # every fmhip_* symbol the layer needs is one the library really exports (the header alone could promise more than the library has)
    needed = set(re.findall(r"\bU\s+(fmhip_\w+)", subprocess.run(["nm", "-D", "--undefined-only", str(out)], capture_output=True, text=True).stdout))
    have = set(re.findall(r"\bT\s+(fmhip_\w+)", subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout))
    assert needed and needed <= have, sorted(needed - have)
    assert needed == set(header_exports())
