/*
 * Native.java — 1:1 image of include/fmhip.h (the C-ABI of libfmhip.so): one static native method per exported function,
 * no logic.  The JNI side is src/jni/fmhip_jni.cpp (one Java_net_finmath_hip_Native_* function per method).
 *
 * UNCOMPILED / UNTESTED in this repository: the build image has no JDK, no jni.h and no finmath-lib jar (DESIGN.md §1).
 * What IS checked here, on every test run: that this class, the JNI file and the header declare exactly the same set of
 * entry points (tests/test_jni_binding_cpu.py).
 *
 * Naming: fmhip_vec_create_from_double → vecCreateFromDouble.  Conventions:
 *   - functions that CREATE a vector or a program return the new handle (0 = failure, see lastError()): they are the hot path
 *     of every RandomVariable method and must not allocate an out-array per call;
 *   - every other function returns the C status (0 = FMHIP_OK) and writes results into caller-provided arrays;
 *   - device pointers and streams travel as long.
 * Replaces the JCuda / JCurand imports of RandomVariableCuda.java:8-16 and BrownianMotionCudaWithRandomVariableCuda.java:153-181.
 */
package net.finmath.hip;

final class Native {

	static {
		System.loadLibrary("fmhip_jni");		// libfmhip_jni.so, linked against libfmhip.so
	}

	private Native() { }

	// ---- lifecycle (fmhip.h: fmhip_init … fmhip_get_stream)
	static native int init(int deviceIndex);
	static native int initDevices(int[] devices);
	static native int deviceCount(int[] count);
	static native int    setThreadEngines(int enabled, int[] previous);   // fmhip_set_thread_engines: an engine per caller thread
	static native int shutdown();
	static native int isInitialized();
	static native int abiVersion();
	static native String lastError();
	static native int deviceInfo(String[] name, int[] computeUnits, long[] hbmBytes);
	static native int synchronize();
	static native int getStream(long[] stream);

	// ---- vectors
	static native long vecCreateFromDouble(double[] values);
	static native long vecCreateFromFloat(float[] values);
	static native long vecCreateFilled(long n, double value);
	static native long vecCreateUninitialized(long n);
	static native int vecRetain(long vector);
	static native int vecRelease(long vector);
	static native int vecSize(long vector, long[] size);
	static native int vecReadDouble(long vector, double[] out);
	static native int vecReadFloat(long vector, float[] out);
	static native int vecDevicePtr(long vector, long[] devicePointer);

	// ---- element-wise operations (replace callFunctionv1s0 … v2s1, RandomVariableCuda.java:483-557)
	static native long callV1s0(int opcode, long a);
	static native long callV1s1(int opcode, long a, double s);
	static native long callV2s0(int opcode, long a, long b);
	static native long callV2s1(int opcode, long a, long b, double s);
	static native long callV3s0(int opcode, long a, long b, long c);

	// ---- lazy fusion front-end
	static native int setFusion(int enabled, int[] previous);
	static native int flush();
	static native int fusionHold(int hold, int[] previous);
	static native int setStepGrouping(int steps, int[] previous);
	static native int graphClone(long[] roots, int nCopies, long[] leafFrom, long[] leafTo, double[] scalarsOrNull, int nScalars, long[] out);
	static native int graphScalars(long[] roots, double[] scalarsOutOrNull, int[] count);
	static native int setMathMode(int mode, int[] previous);

	// ---- reductions: {sum, sumsq, min, max} per vector (replace getAverage … getMax, RandomVariableCuda.java:830-901)
	static native int reduceMoments(long vector, double shift, double[] moments4);
	static native int reduceMomentsDevice(long vector, double shift, long deviceOut4Doubles);
	static native int reduceMomentsBatch(long[] vectors, double[] shiftsOrNull, double[] moments4PerVector);
	static native int reduceMomentsBatchDevice(long[] vectors, double[] shiftsOrNull, long deviceOut);
	/** With a device list: one device buffer per listed device (0 = not wanted there), each receives the moments of the whole vectors. */
	static native int reduceMomentsBatchDevices(long[] vectors, double[] shiftsOrNull, long[] deviceOutPerDevice);
	static native int getStreamOf(int shard, long[] stream);
	/** kind[0]: 0 = one device, 1 = grouped RCCL all-gather over the listed devices, 2 = combined on the host. */
	static native int expectationCollective(int[] kind);
	// the same reduction in two halves: begin enqueues and returns a ticket (ticket[0]); end waits for that reduction only and retires the ticket
	static native int reduceMomentsBatchBegin(long[] vectors, double[] shiftsOrNull, long[] ticket);
	static native int vecGiveUpValues(long[] vectors);
	static native int reduceMomentsBatchEnd(long ticket, double[] moments4PerVector, int count);
	// expectation communicator (paths sharded over processes): gatherFunction = address of a C function of type fmhip_gather_fn,
	// e.g. from an MPI / RCCL helper library; context is handed back to it.  0 removes the communicator.
	static native int setExpectationComm(int world, int rank, long gatherFunction, long context);
	static native int expectationWorld(int[] world, int[] rankOrNull);
	static native int expectationCombine(double[] gatheredMoments4PerRankAndVector, int world, int count, double[] moments4PerVector);

	// ---- explicit fused programs: ops as parallel arrays {opcode, a, b, c, scalar}
	static native long programCreate(int[] opcode, int[] a, int[] b, int[] c, double[] scalar, int nInputs, int[] outValues, int[] reduceValues);
	static native int programRelease(long program);
	static native int programLaunchCount(long program, int[] launches);
	static native int programShape(long program, int[] inputsOutputsReductions3);
	static native int programRun(long program, int batch, long[] inputs, long[] outputs, double[] reduceShiftOrNull, double[] moments4OrNull, long deviceMomentsOrZero);
	static native int programRunInto(long program, int batch, long[] inputs, long[] outputs, double[] reduceShiftOrNull, double[] moments4OrNull, long deviceMomentsOrZero);

	// ---- execution tiers (replace JCudaUtils.preparePtxFile, JCudaUtils.java:37-122)
	static native int setJit(int mode, int[] previous);
	static native int jitWait();
	static native int jitStats(long[] compiledFailedPendingDiskHits, double[] compileSeconds);
	static native int programTier(long program, int[] tierAndVgprs);
	static native String programSource(int[] opcode, int[] a, int[] b, int[] c, double[] scalar, int nInputs, int[] outValues, int[] reduceValues);

	// ---- Brownian increments (replace curandGenerateNormal, BrownianMotionCudaWithRandomVariableCuda.java:168-178)
	static native int bmGenerate(long seed, int nSteps, int nFactors, long nPaths, long pathOffset, double[] dt, long[] outHandles);
	static native int mersenneIncrements(int seed, int nSteps, int nFactors, long nPaths, double[] dt, double[] hostOut);
	static native int bmGenerateMersenne(int seed, int nSteps, int nFactors, long nPaths, double[] dt, long[] outHandles);
	static native double inverseNormalCdf(double p);

	// ---- pool (replace DeviceMemoryPool.clean / purge / getDeviceFreeMemPercentage, RandomVariableCuda.java:393-449)
	static native int poolClean();
	static native int poolPurge();
	static native int poolStats(long[] stats10);

	// ---- measurement
	static native int profileEnable(int enabled);
	static native int trafficStats(long[] algorithmicBytesAndSpecialisedLaunches);
	/** {size, kernelLaunches, specialisedLaunches, interpreterLaunches, algorithmicBytes, algorithmicBytesWritten, valuesDeferred, valuesDeferredNow, valuesDemanded, pendingOperations, peakBytesReserved,
	 *  lateReleasesWhileWaiting, lateReleasesAtOnce, lateReleaseNanoseconds, mergedLaunches, mergedChains, commonRows} */
	static native int engineStats(long[] stats17);
	static native int profileRead(double[] kernelMsTotal, long[] launches);

	// ---- helpers (plain Java)
	static long checkHandle(final long handle) {
		if(handle == 0) {
			throw new RuntimeException("fmhip: " + lastError());
		}
		return handle;
	}

	static void check(final int status) {
		if(status == 0) {
			return;
		}
		final String message = "fmhip error " + status + ": " + lastError();
		if(status == -3) {
			throw new OutOfMemoryError(message);					// as RandomVariableCuda.java:375
		}
		if(status == -7) {
			throw new UnsupportedOperationException(message);
		}
		throw new RuntimeException(message);
	}
}
