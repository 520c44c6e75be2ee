// fmhip_jni.cpp — the JNI layer of net.finmath.hip.Native: ONE function per C-ABI function of include/fmhip.h, no logic
// beyond argument marshalling ("JNI layer = one Java_net_finmath_hip_Native_* function per C-ABI function").
//
// UNCOMPILED against a real JDK / UNTESTED in a JVM in this repository: the build image has no JDK (no jni.h).  CMakeLists.txt at
// the repository root builds it only when find_package(JNI) succeeds.  What IS done on every run of the CPU test suite
// (tests/test_jni_binding_cpu.py): this file is compiled against a hand-written declaration stub of <jni.h> (tests/jni_stub: the
// types and the JNIEnv members used here) and linked against libfmhip.so — every C++ error and every mismatch with
// include/fmhip.h shows — and the set of functions here, the native methods of java/net/finmath/hip/Native.java and the exports
// of the header are checked to be the same set.
//
// Marshalling rules: every array is accessed with Get/Release<Type>ArrayElements (may copy; other JNI calls stay legal) — no
// critical region is held around an engine call (an upload or a read-back synchronises with the device and may compile a
// kernel: a pinned array would stall the collector for that long); EVERY array length is checked against what the C function
// will read or write before it is called (FMHIP_ERR_INVALID_ARGUMENT otherwise: a short array from Java must never become a
// native overrun); out-parameters are 1-element (or documented-length) arrays; a null array where the C function accepts NULL
// is passed as NULL.
#include <jni.h>
#include <cstdint>
#include <string>
#include <vector>
#include "fmhip.h"

namespace {

// RAII access to the elements of a primitive Java array (may be null).  Pin<T> = Get/Release<Type>ArrayElements: may copy, and
// other JNI calls (further arrays, GetArrayLength) stay legal while it is held — used wherever several arrays travel together.
template <typename T> struct ArrayOps;
template <> struct ArrayOps<jint>    { static jint*    get(JNIEnv* e, jarray a) { return e->GetIntArrayElements((jintArray)a, nullptr); }       static void put(JNIEnv* e, jarray a, jint* p, jint m)    { e->ReleaseIntArrayElements((jintArray)a, p, m); } };
template <> struct ArrayOps<jlong>   { static jlong*   get(JNIEnv* e, jarray a) { return e->GetLongArrayElements((jlongArray)a, nullptr); }     static void put(JNIEnv* e, jarray a, jlong* p, jint m)   { e->ReleaseLongArrayElements((jlongArray)a, p, m); } };
template <> struct ArrayOps<jdouble> { static jdouble* get(JNIEnv* e, jarray a) { return e->GetDoubleArrayElements((jdoubleArray)a, nullptr); } static void put(JNIEnv* e, jarray a, jdouble* p, jint m) { e->ReleaseDoubleArrayElements((jdoubleArray)a, p, m); } };
template <> struct ArrayOps<jfloat>  { static jfloat*  get(JNIEnv* e, jarray a) { return e->GetFloatArrayElements((jfloatArray)a, nullptr); }   static void put(JNIEnv* e, jarray a, jfloat* p, jint m)  { e->ReleaseFloatArrayElements((jfloatArray)a, p, m); } };
template <typename T>
struct Pin {
    JNIEnv* env; jarray arr; T* p; jint mode; jsize len;
    Pin(JNIEnv* e, jarray a, jint release_mode = 0) : env(e), arr(a), p(nullptr), mode(release_mode), len(a ? e->GetArrayLength(a) : 0) { if (a) p = ArrayOps<T>::get(e, a); }
    ~Pin() { if (arr && p) ArrayOps<T>::put(env, arr, p, mode); }
    jsize length() const { return len; }
};
// fmhip_prog_op[] from the parallel arrays of Native.programCreate / programSource
// (false: an operand array is missing or shorter than the opcode array)
bool program_ops(JNIEnv* env, jintArray opcode, jintArray a, jintArray b, jintArray c, jdoubleArray scalar, std::vector<fmhip_prog_op>& ops) {
    const jsize n = opcode ? env->GetArrayLength(opcode) : 0;
    ops.assign((size_t)n, fmhip_prog_op{});
    if (n == 0) return true;
    Pin<jint> po(env, opcode, JNI_ABORT), pa(env, a, JNI_ABORT), pb(env, b, JNI_ABORT), pc(env, c, JNI_ABORT);
    Pin<jdouble> ps(env, scalar, JNI_ABORT);
    if (!po.p || !pa.p || !pb.p || !pc.p || pa.length() < n || pb.length() < n || pc.length() < n || (ps.p && ps.length() < n)) return false;
    for (jsize i = 0; i < n; ++i) ops[(size_t)i] = { po.p[i], pa.p[i], pb.p[i], pc.p[i], ps.p ? ps.p[i] : 0.0 };
    return true;
}

inline void set1(JNIEnv* env, jintArray arr, jint v) { if (arr && env->GetArrayLength(arr) > 0) env->SetIntArrayRegion(arr, 0, 1, &v); }
inline void set1(JNIEnv* env, jlongArray arr, jlong v) { if (arr && env->GetArrayLength(arr) > 0) env->SetLongArrayRegion(arr, 0, 1, &v); }

} // namespace

#define FMJ(ret, name) extern "C" JNIEXPORT ret JNICALL Java_net_finmath_hip_Native_##name

// ---------------------------------------------------------------- lifecycle
FMJ(jint, init)(JNIEnv*, jclass, jint deviceIndex) { return fmhip_init(deviceIndex); }
FMJ(jint, initDevices)(JNIEnv* env, jclass, jintArray devices) {
    Pin<jint> pd(env, devices, JNI_ABORT);
    if (!pd.p) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_init_devices((const int*)pd.p, pd.length());
}
FMJ(jint, deviceCount)(JNIEnv* env, jclass, jintArray count) {
    if (!count || env->GetArrayLength(count) < 1) return FMHIP_ERR_INVALID_ARGUMENT;
    int c = 0;
    const int st = fmhip_device_count(&c);
    if (st == FMHIP_OK) { const jint v = c; env->SetIntArrayRegion(count, 0, 1, &v); }
    return st;
}
FMJ(jint, setThreadEngines)(JNIEnv* env, jclass, jint enabled, jintArray previous) {
    int prev = 0;
    const int st = fmhip_set_thread_engines(enabled, &prev);
    if (st == FMHIP_OK && previous && env->GetArrayLength(previous) >= 1) { const jint v = prev; env->SetIntArrayRegion(previous, 0, 1, &v); }
    return st;
}
FMJ(jint, shutdown)(JNIEnv*, jclass) { return fmhip_shutdown(); }
FMJ(jint, isInitialized)(JNIEnv*, jclass) { return fmhip_is_initialized(); }
FMJ(jint, abiVersion)(JNIEnv*, jclass) { return fmhip_abi_version(); }
FMJ(jstring, lastError)(JNIEnv* env, jclass) { return env->NewStringUTF(fmhip_last_error()); }
FMJ(jint, deviceInfo)(JNIEnv* env, jclass, jobjectArray name, jintArray computeUnits, jlongArray hbmBytes) {
    char buf[256] = { 0 }; int cus = 0; int64_t hbm = 0;
    const int st = fmhip_device_info(buf, (int)sizeof buf, &cus, &hbm);
    if (st == FMHIP_OK) {
        if (name && env->GetArrayLength(name) > 0) env->SetObjectArrayElement(name, 0, env->NewStringUTF(buf));
        set1(env, computeUnits, (jint)cus); set1(env, hbmBytes, (jlong)hbm);
    }
    return st;
}
FMJ(jint, synchronize)(JNIEnv*, jclass) { return fmhip_synchronize(); }
FMJ(jint, getStream)(JNIEnv* env, jclass, jlongArray stream) {
    void* s = nullptr;
    const int st = fmhip_get_stream(&s);
    if (st == FMHIP_OK) set1(env, stream, (jlong)(intptr_t)s);
    return st;
}

// ---------------------------------------------------------------- vectors
FMJ(jlong, vecCreateFromDouble)(JNIEnv* env, jclass, jdoubleArray values) {
    Pin<jdouble> p(env, values, JNI_ABORT);
    fmhip_vec out = 0;
    return fmhip_vec_create_from_double(p.p, p.length(), &out) == FMHIP_OK ? (jlong)out : 0;
}
FMJ(jlong, vecCreateFromFloat)(JNIEnv* env, jclass, jfloatArray values) {
    Pin<jfloat> p(env, values, JNI_ABORT);
    fmhip_vec out = 0;
    return fmhip_vec_create_from_float(p.p, p.length(), &out) == FMHIP_OK ? (jlong)out : 0;
}
FMJ(jlong, vecCreateFilled)(JNIEnv*, jclass, jlong n, jdouble value) { fmhip_vec out = 0; return fmhip_vec_create_filled(n, value, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jlong, vecCreateUninitialized)(JNIEnv*, jclass, jlong n) { fmhip_vec out = 0; return fmhip_vec_create_uninitialized(n, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jint, vecRetain)(JNIEnv*, jclass, jlong v) { return fmhip_vec_retain(v); }
FMJ(jint, vecRelease)(JNIEnv*, jclass, jlong v) { return fmhip_vec_release(v); }
FMJ(jint, vecSize)(JNIEnv* env, jclass, jlong v, jlongArray size) {
    int64_t n = 0;
    const int st = fmhip_vec_size(v, &n);
    if (st == FMHIP_OK) set1(env, size, (jlong)n);
    return st;
}
FMJ(jint, vecReadDouble)(JNIEnv* env, jclass, jlong v, jdoubleArray out) { Pin<jdouble> p(env, out); return fmhip_vec_read_double(v, p.p, p.length()); }     // (the engine checks the length against the vector's)
FMJ(jint, vecReadFloat)(JNIEnv* env, jclass, jlong v, jfloatArray out) { Pin<jfloat> p(env, out); return fmhip_vec_read_float(v, p.p, p.length()); }
FMJ(jint, vecDevicePtr)(JNIEnv* env, jclass, jlong v, jlongArray devicePointer) {
    void* ptr = nullptr;
    const int st = fmhip_vec_device_ptr(v, &ptr);
    if (st == FMHIP_OK) set1(env, devicePointer, (jlong)(intptr_t)ptr);
    return st;
}

// ---------------------------------------------------------------- element-wise operations (the hot path: no allocation, no array)
FMJ(jlong, callV1s0)(JNIEnv*, jclass, jint opcode, jlong a) { fmhip_vec out = 0; return fmhip_call_v1s0(opcode, a, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jlong, callV1s1)(JNIEnv*, jclass, jint opcode, jlong a, jdouble s) { fmhip_vec out = 0; return fmhip_call_v1s1(opcode, a, s, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jlong, callV2s0)(JNIEnv*, jclass, jint opcode, jlong a, jlong b) { fmhip_vec out = 0; return fmhip_call_v2s0(opcode, a, b, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jlong, callV2s1)(JNIEnv*, jclass, jint opcode, jlong a, jlong b, jdouble s) { fmhip_vec out = 0; return fmhip_call_v2s1(opcode, a, b, s, &out) == FMHIP_OK ? (jlong)out : 0; }
FMJ(jlong, callV3s0)(JNIEnv*, jclass, jint opcode, jlong a, jlong b, jlong c) { fmhip_vec out = 0; return fmhip_call_v3s0(opcode, a, b, c, &out) == FMHIP_OK ? (jlong)out : 0; }

// ---------------------------------------------------------------- lazy fusion front-end
FMJ(jint, setFusion)(JNIEnv* env, jclass, jint enabled, jintArray previous) { int prev = 0; const int st = fmhip_set_fusion(enabled, &prev); if (st == FMHIP_OK) set1(env, previous, prev); return st; }
FMJ(jint, flush)(JNIEnv*, jclass) { return fmhip_flush(); }
FMJ(jint, setStepGrouping)(JNIEnv* env, jclass, jint steps, jintArray previous) { int prev = 0; const int st = fmhip_set_step_grouping(steps, &prev); if (st == FMHIP_OK) set1(env, previous, prev); return st; }
FMJ(jint, fusionHold)(JNIEnv* env, jclass, jint hold, jintArray previous) { int prev = 0; const int st = fmhip_fusion_hold(hold, &prev); if (st == FMHIP_OK) set1(env, previous, prev); return st; }
FMJ(jint, graphClone)(JNIEnv* env, jclass, jlongArray roots, jint nCopies, jlongArray leafFrom, jlongArray leafTo, jdoubleArray scalars, jint nScalars, jlongArray out) {
    Pin<jlong> pr(env, roots, JNI_ABORT), pf(env, leafFrom, JNI_ABORT), pt(env, leafTo, JNI_ABORT), po(env, out); Pin<jdouble> ps(env, scalars, JNI_ABORT);
    if (nCopies < 0 || nScalars < 0 || !pr.p || !po.p || (int64_t)pt.length() < (int64_t)pf.length() * nCopies || (int64_t)po.length() < (int64_t)pr.length() * nCopies ||
        (ps.p && (int64_t)ps.length() < (int64_t)nScalars * nCopies)) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_graph_clone((const fmhip_vec*)pr.p, pr.length(), nCopies, (const fmhip_vec*)pf.p, (const fmhip_vec*)pt.p, pf.length(), ps.p, nScalars, (fmhip_vec*)po.p);
}
FMJ(jint, graphScalars)(JNIEnv* env, jclass, jlongArray roots, jdoubleArray scalarsOut, jintArray count) {
    int n = 0, st;
    { Pin<jlong> pr(env, roots, JNI_ABORT); Pin<jdouble> ps(env, scalarsOut); st = fmhip_graph_scalars((const fmhip_vec*)pr.p, pr.length(), ps.p, ps.length(), &n); }     // capacity = the array's own length
    if (st == FMHIP_OK) set1(env, count, n);
    return st;
}
FMJ(jint, setMathMode)(JNIEnv* env, jclass, jint mode, jintArray previous) { int prev = 0; const int st = fmhip_set_math_mode(mode, &prev); if (st == FMHIP_OK) set1(env, previous, prev); return st; }

// ---------------------------------------------------------------- reductions
FMJ(jint, reduceMoments)(JNIEnv* env, jclass, jlong v, jdouble shift, jdoubleArray moments4) {
    fmhip_moments m;
    const int st = fmhip_reduce_moments(v, shift, &m);
    if (st == FMHIP_OK && moments4 && env->GetArrayLength(moments4) >= 4) { const jdouble d[4] = { m.sum, m.sumsq, m.min, m.max }; env->SetDoubleArrayRegion(moments4, 0, 4, d); }
    return st;
}
FMJ(jint, reduceMomentsDevice)(JNIEnv*, jclass, jlong v, jdouble shift, jlong deviceOut) { return fmhip_reduce_moments_device(v, shift, (void*)(intptr_t)deviceOut); }
FMJ(jint, reduceMomentsBatch)(JNIEnv* env, jclass, jlongArray vectors, jdoubleArray shifts, jdoubleArray moments4PerVector) {
    Pin<jlong> pv(env, vectors, JNI_ABORT); Pin<jdouble> ps(env, shifts, JNI_ABORT); Pin<jdouble> pm(env, moments4PerVector);
    static_assert(sizeof(fmhip_moments) == 4 * sizeof(double), "moments travel as 4 doubles");
    if (!pv.p || !pm.p || (int64_t)pm.length() < 4 * (int64_t)pv.length() || (ps.p && ps.length() < pv.length())) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_reduce_moments_batch((const fmhip_vec*)pv.p, pv.length(), ps.p, (fmhip_moments*)pm.p);
}
FMJ(jint, reduceMomentsBatchDevice)(JNIEnv* env, jclass, jlongArray vectors, jdoubleArray shifts, jlong deviceOut) {
    Pin<jlong> pv(env, vectors, JNI_ABORT); Pin<jdouble> ps(env, shifts, JNI_ABORT);
    if (!pv.p || (ps.p && ps.length() < pv.length())) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_reduce_moments_batch_device((const fmhip_vec*)pv.p, pv.length(), ps.p, (void*)(intptr_t)deviceOut);
}
FMJ(jint, reduceMomentsBatchDevices)(JNIEnv* env, jclass, jlongArray vectors, jdoubleArray shifts, jlongArray deviceOutPerDevice) {
    Pin<jlong> pv(env, vectors, JNI_ABORT); Pin<jdouble> ps(env, shifts, JNI_ABORT); Pin<jlong> po(env, deviceOutPerDevice, JNI_ABORT);
    if (!pv.p || !po.p || (ps.p && ps.length() < pv.length())) return FMHIP_ERR_INVALID_ARGUMENT;
    std::vector<void*> out((size_t)po.length());
    for (int d = 0; d < po.length(); ++d) out[(size_t)d] = (void*)(intptr_t)po.p[d];
    return fmhip_reduce_moments_batch_devices((const fmhip_vec*)pv.p, pv.length(), ps.p, out.data(), po.length());
}
FMJ(jint, getStreamOf)(JNIEnv* env, jclass, jint shard, jlongArray stream) {
    void* s = nullptr;
    const int st = fmhip_get_stream_of(shard, &s);
    if (st == FMHIP_OK) set1(env, stream, (jlong)(intptr_t)s);
    return st;
}
FMJ(jint, expectationCollective)(JNIEnv* env, jclass, jintArray kind) {
    int k = 0;
    const int st = fmhip_expectation_collective(&k, nullptr, 0);
    if (st == FMHIP_OK && kind && env->GetArrayLength(kind) >= 1) { const jint v = k; env->SetIntArrayRegion(kind, 0, 1, &v); }
    return st;
}
FMJ(jint, reduceMomentsBatchBegin)(JNIEnv* env, jclass, jlongArray vectors, jdoubleArray shifts, jlongArray ticket) {
    Pin<jlong> pv(env, vectors, JNI_ABORT); Pin<jdouble> ps(env, shifts, JNI_ABORT);
    if (!pv.p || !ticket || env->GetArrayLength(ticket) < 1 || (ps.p && ps.length() < pv.length())) return FMHIP_ERR_INVALID_ARGUMENT;
    fmhip_ticket t = 0;
    const int st = fmhip_reduce_moments_batch_begin((const fmhip_vec*)pv.p, pv.length(), ps.p, &t);
    if (st == FMHIP_OK) { const jlong v = (jlong)t; env->SetLongArrayRegion(ticket, 0, 1, &v); }
    return st;
}
FMJ(jint, vecGiveUpValues)(JNIEnv* env, jclass, jlongArray vectors) {
    Pin<jlong> pv(env, vectors, JNI_ABORT);
    if (!pv.p) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_vec_give_up_values((const fmhip_vec*)pv.p, pv.length());
}
FMJ(jint, reduceMomentsBatchEnd)(JNIEnv* env, jclass, jlong ticket, jdoubleArray moments4PerVector, jint count) {
    Pin<jdouble> pm(env, moments4PerVector);
    if (!pm.p || count <= 0 || (int64_t)pm.length() < 4 * (int64_t)count) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_reduce_moments_batch_end((fmhip_ticket)ticket, (fmhip_moments*)pm.p, count);
}
FMJ(jint, setExpectationComm)(JNIEnv*, jclass, jint world, jint rank, jlong gatherFunction, jlong context) {
    return fmhip_set_expectation_comm(world, rank, (fmhip_gather_fn)(intptr_t)gatherFunction, (void*)(intptr_t)context);
}
FMJ(jint, expectationWorld)(JNIEnv* env, jclass, jintArray world, jintArray rank) {
    int w = 1, r = 0;
    const int st = fmhip_expectation_world(&w, &r);
    if (st == FMHIP_OK) { set1(env, world, w); set1(env, rank, r); }
    return st;
}
FMJ(jint, expectationCombine)(JNIEnv* env, jclass, jdoubleArray gathered, jint world, jint count, jdoubleArray moments4PerVector) {
    Pin<jdouble> pg(env, gathered, JNI_ABORT), pm(env, moments4PerVector);
    if (world < 1 || count < 0 || !pg.p || !pm.p || (int64_t)pg.length() < 4 * (int64_t)world * count || (int64_t)pm.length() < 4 * (int64_t)count) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_expectation_combine((const fmhip_moments*)pg.p, world, count, (fmhip_moments*)pm.p);
}

// ---------------------------------------------------------------- explicit fused programs
FMJ(jlong, programCreate)(JNIEnv* env, jclass, jintArray opcode, jintArray a, jintArray b, jintArray c, jdoubleArray scalar, jint nInputs, jintArray outValues, jintArray reduceValues) {
    std::vector<fmhip_prog_op> ops;
    if (!program_ops(env, opcode, a, b, c, scalar, ops)) return 0;
    Pin<jint> po(env, outValues, JNI_ABORT), pr(env, reduceValues, JNI_ABORT);
    fmhip_program out = 0;
    return fmhip_program_create(ops.data(), (int)ops.size(), nInputs, (const int32_t*)po.p, po.length(), (const int32_t*)pr.p, pr.length(), &out) == FMHIP_OK ? (jlong)out : 0;
}
FMJ(jint, programRelease)(JNIEnv*, jclass, jlong p) { return fmhip_program_release(p); }
FMJ(jint, programLaunchCount)(JNIEnv* env, jclass, jlong p, jintArray launches) { int n = 0; const int st = fmhip_program_launch_count(p, &n); if (st == FMHIP_OK) set1(env, launches, n); return st; }
FMJ(jint, programShape)(JNIEnv* env, jclass, jlong p, jintArray inputsOutputsReductions3) {
    int shape[3] = { 0, 0, 0 };
    const int st = fmhip_program_shape(p, &shape[0], &shape[1], &shape[2]);
    if (st == FMHIP_OK && inputsOutputsReductions3 && env->GetArrayLength(inputsOutputsReductions3) >= 3) { const jint v[3] = { shape[0], shape[1], shape[2] }; env->SetIntArrayRegion(inputsOutputsReductions3, 0, 3, v); }
    return st;
}
namespace {
// the arrays of programRun / programRunInto against the program's shape: batch x inputs handles in, batch x outputs handles,
// one shift per reduction, 4 doubles per row and reduction
bool run_arrays_fit(jlong p, jint batch, const Pin<jlong>& in, const Pin<jlong>& out, const Pin<jdouble>& shift, const Pin<jdouble>& moments) {
    int n_in = 0, n_out = 0, n_red = 0;
    if (batch <= 0 || fmhip_program_shape(p, &n_in, &n_out, &n_red) != FMHIP_OK) return false;
    const int64_t b = batch;
    return in.p && (int64_t)in.length() >= b * n_in && (n_out == 0 || (out.p && (int64_t)out.length() >= b * n_out)) &&
           (!shift.p || shift.length() >= n_red) && (!moments.p || (int64_t)moments.length() >= 4 * b * n_red);
}
}
FMJ(jint, programRun)(JNIEnv* env, jclass, jlong p, jint batch, jlongArray inputs, jlongArray outputs, jdoubleArray reduceShift, jdoubleArray moments4, jlong deviceMoments) {
    Pin<jlong> pi(env, inputs, JNI_ABORT), po(env, outputs); Pin<jdouble> ps(env, reduceShift, JNI_ABORT), pm(env, moments4);
    if (!run_arrays_fit(p, batch, pi, po, ps, pm)) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_program_run(p, batch, (const fmhip_vec*)pi.p, (fmhip_vec*)po.p, ps.p, (fmhip_moments*)pm.p, (void*)(intptr_t)deviceMoments);
}
FMJ(jint, programRunInto)(JNIEnv* env, jclass, jlong p, jint batch, jlongArray inputs, jlongArray outputs, jdoubleArray reduceShift, jdoubleArray moments4, jlong deviceMoments) {
    Pin<jlong> pi(env, inputs, JNI_ABORT), po(env, outputs, JNI_ABORT); Pin<jdouble> ps(env, reduceShift, JNI_ABORT), pm(env, moments4);
    if (!run_arrays_fit(p, batch, pi, po, ps, pm)) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_program_run_into(p, batch, (const fmhip_vec*)pi.p, (const fmhip_vec*)po.p, ps.p, (fmhip_moments*)pm.p, (void*)(intptr_t)deviceMoments);
}

// ---------------------------------------------------------------- execution tiers
FMJ(jint, setJit)(JNIEnv* env, jclass, jint mode, jintArray previous) { int prev = 0; const int st = fmhip_set_jit(mode, &prev); if (st == FMHIP_OK) set1(env, previous, prev); return st; }
FMJ(jint, jitWait)(JNIEnv*, jclass) { return fmhip_jit_wait(); }
FMJ(jint, jitStats)(JNIEnv* env, jclass, jlongArray compiledFailedPendingDiskHits, jdoubleArray compileSeconds) {
    int64_t compiled = 0, failed = 0, pending = 0, disk = 0; double seconds = 0.0;
    const int st = fmhip_jit_stats(&compiled, &failed, &pending, &seconds, &disk);
    if (st == FMHIP_OK) {
        if (compiledFailedPendingDiskHits && env->GetArrayLength(compiledFailedPendingDiskHits) >= 4) { const jlong v[4] = { (jlong)compiled, (jlong)failed, (jlong)pending, (jlong)disk }; env->SetLongArrayRegion(compiledFailedPendingDiskHits, 0, 4, v); }
        if (compileSeconds && env->GetArrayLength(compileSeconds) >= 1) env->SetDoubleArrayRegion(compileSeconds, 0, 1, &seconds);
    }
    return st;
}
FMJ(jint, programTier)(JNIEnv* env, jclass, jlong p, jintArray tierAndVgprs) {
    int tier = 0, vgprs = 0;
    const int st = fmhip_program_tier(p, &tier, &vgprs);
    if (st == FMHIP_OK && tierAndVgprs && env->GetArrayLength(tierAndVgprs) >= 2) { const jint v[2] = { tier, vgprs }; env->SetIntArrayRegion(tierAndVgprs, 0, 2, v); }
    return st;
}
FMJ(jstring, programSource)(JNIEnv* env, jclass, jintArray opcode, jintArray a, jintArray b, jintArray c, jdoubleArray scalar, jint nInputs, jintArray outValues, jintArray reduceValues) {
    std::vector<fmhip_prog_op> ops;
    if (!program_ops(env, opcode, a, b, c, scalar, ops)) return nullptr;
    std::vector<int32_t> outs, reds;
    { Pin<jint> po(env, outValues, JNI_ABORT), pr(env, reduceValues, JNI_ABORT); outs.assign(po.p, po.p + po.length()); reds.assign(pr.p, pr.p + pr.length()); }
    int64_t needed = 0;
    if (fmhip_program_source(ops.data(), (int)ops.size(), nInputs, outs.data(), (int)outs.size(), reds.data(), (int)reds.size(), nullptr, 0, &needed) != FMHIP_OK) return nullptr;
    std::string text((size_t)needed + 1, '\0');
    if (fmhip_program_source(ops.data(), (int)ops.size(), nInputs, outs.data(), (int)outs.size(), reds.data(), (int)reds.size(), &text[0], needed + 1, &needed) != FMHIP_OK) return nullptr;
    return env->NewStringUTF(text.c_str());
}

// ---------------------------------------------------------------- Brownian increments
FMJ(jint, bmGenerate)(JNIEnv* env, jclass, jlong seed, jint nSteps, jint nFactors, jlong nPaths, jlong pathOffset, jdoubleArray dt, jlongArray outHandles) {
    Pin<jdouble> pd(env, dt, JNI_ABORT); Pin<jlong> po(env, outHandles);
    if (nSteps <= 0 || nFactors <= 0 || !po.p || !pd.p || (int64_t)po.length() < (int64_t)nSteps * nFactors || pd.length() < nSteps) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_bm_generate(seed, nSteps, nFactors, nPaths, pathOffset, pd.p, (fmhip_vec*)po.p);
}
FMJ(jint, mersenneIncrements)(JNIEnv* env, jclass, jint seed, jint nSteps, jint nFactors, jlong nPaths, jdoubleArray dt, jdoubleArray hostOut) {
    Pin<jdouble> pd(env, dt, JNI_ABORT), po(env, hostOut);
    if (nSteps <= 0 || nFactors <= 0 || nPaths < 0 || !po.p || !pd.p || (int64_t)po.length() < (int64_t)nSteps * nFactors * nPaths || pd.length() < nSteps) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_mersenne_increments(seed, nSteps, nFactors, nPaths, pd.p, po.p);
}
FMJ(jint, bmGenerateMersenne)(JNIEnv* env, jclass, jint seed, jint nSteps, jint nFactors, jlong nPaths, jdoubleArray dt, jlongArray outHandles) {
    Pin<jdouble> pd(env, dt, JNI_ABORT); Pin<jlong> po(env, outHandles);
    if (nSteps <= 0 || nFactors <= 0 || !po.p || !pd.p || (int64_t)po.length() < (int64_t)nSteps * nFactors || pd.length() < nSteps) return FMHIP_ERR_INVALID_ARGUMENT;
    return fmhip_bm_generate_mersenne(seed, nSteps, nFactors, nPaths, pd.p, (fmhip_vec*)po.p);
}
FMJ(jdouble, inverseNormalCdf)(JNIEnv*, jclass, jdouble p) { return fmhip_inverse_normal_cdf(p); }

// ---------------------------------------------------------------- pool
FMJ(jint, poolClean)(JNIEnv*, jclass) { return fmhip_pool_clean(); }
FMJ(jint, poolPurge)(JNIEnv*, jclass) { return fmhip_pool_purge(); }
FMJ(jint, poolStats)(JNIEnv* env, jclass, jlongArray stats10) {
    fmhip_pool_stats_t s;
    const int st = fmhip_pool_stats(&s);
    static_assert(sizeof(fmhip_pool_stats_t) == 10 * sizeof(int64_t), "pool statistics travel as 10 longs");
    if (st == FMHIP_OK && stats10 && env->GetArrayLength(stats10) >= 10) env->SetLongArrayRegion(stats10, 0, 10, (const jlong*)&s);
    return st;
}

// ---------------------------------------------------------------- measurement
FMJ(jint, profileEnable)(JNIEnv*, jclass, jint enabled) { return fmhip_profile_enable(enabled); }
FMJ(jint, trafficStats)(JNIEnv* env, jclass, jlongArray algorithmicBytesAndSpecialisedLaunches) {
    int64_t bytes = 0, launches = 0;
    const int st = fmhip_traffic_stats(&bytes, &launches);
    if (st == FMHIP_OK && algorithmicBytesAndSpecialisedLaunches && env->GetArrayLength(algorithmicBytesAndSpecialisedLaunches) >= 2) { const jlong v[2] = { (jlong)bytes, (jlong)launches }; env->SetLongArrayRegion(algorithmicBytesAndSpecialisedLaunches, 0, 2, v); }
    return st;
}
FMJ(jint, engineStats)(JNIEnv* env, jclass, jlongArray stats17) {
    fmhip_engine_stats_t s;
    const int st = fmhip_engine_stats(&s);
    static_assert(sizeof(fmhip_engine_stats_t) == 17 * sizeof(int64_t), "engine statistics travel as 17 longs");
    if (st == FMHIP_OK && stats17) { const jsize len = env->GetArrayLength(stats17); env->SetLongArrayRegion(stats17, 0, len < 17 ? len : 17, (const jlong*)&s); }
    return st;
}
FMJ(jint, profileRead)(JNIEnv* env, jclass, jdoubleArray kernelMsTotal, jlongArray launches) {
    double ms = 0.0; int64_t n = 0;
    const int st = fmhip_profile_read(&ms, &n);
    if (st == FMHIP_OK) { if (kernelMsTotal && env->GetArrayLength(kernelMsTotal) >= 1) env->SetDoubleArrayRegion(kernelMsTotal, 0, 1, &ms); set1(env, launches, (jlong)n); }
    return st;
}
