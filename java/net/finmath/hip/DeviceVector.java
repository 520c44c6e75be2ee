/*
 * DeviceVector.java — owner of one engine handle (an fp32 vector in HBM, materialised or still a pending expression of
 * the lazy front-end).  Lifetime: the handle is released by a java.lang.ref.Cleaner action when the owner becomes
 * unreachable — explicit and O(1) on the native side; the engine never polls the garbage collector and never calls
 * hipMemGetInfo on the hot path (the reference does both on every pool miss: ReferenceQueue polling, System.gc(),
 * cudaMemGetInfo — RandomVariableCuda.java:293-335).
 * UNCOMPILED / UNTESTED here (no JDK in the build image).
 */
package net.finmath.hip;

import java.lang.ref.Cleaner;
import java.lang.ref.Reference;

final class DeviceVector {

	private static final Cleaner cleaner = Cleaner.create();

	private static final class Release implements Runnable {
		private final long handle;
		Release(final long handle) { this.handle = handle; }
		@Override public void run() { Native.vecRelease(handle); }
	}

	final long handle;
	final long size;

	DeviceVector(final long handle, final long size) {
		this.handle = Native.checkHandle(handle);
		this.size = size;
		cleaner.register(this, new Release(handle));
	}

	static DeviceVector fromHost(final double[] values) {			// (float)values[i], RandomVariableCuda.java:768-774 — narrowed in the engine
		return new DeviceVector(afterCollection(Native.vecCreateFromDouble(values), () -> Native.vecCreateFromDouble(values)), values.length);
	}

	static DeviceVector filled(final long size, final double value) {
		return new DeviceVector(afterCollection(Native.vecCreateFilled(size, value), () -> Native.vecCreateFilled(size, value)), size);
	}

	double[] toDoubleArray() {
		final double[] out = new double[(int)size];
		try { Native.check(Native.vecReadDouble(handle, out)); }
		finally { Reference.reachabilityFence(this); }
		return out;
	}

	/** {sum (x - shift), sum (x - shift)^2, min, max} in fp64, computed on the device; 32 bytes come back. */
	double[] moments(final double shift) {
		final double[] m = new double[4];
		try { Native.check(Native.reduceMoments(handle, shift, m)); }
		finally { Reference.reachabilityFence(this); }
		return m;
	}

	// The operands stay reachable until the native call has returned (the engine then holds its own references to them):
	// without the fences the JIT may treat `this` / `b` as dead once their handle field has been read, and the Cleaner could
	// release a handle that the call is about to use.
	/*
	 * A device allocation that fails while dead wrappers wait for the collector: the collector is asked to run, the cleaner given a moment,
	 * the call tried once more — what the reference's pool does when device memory runs short (System.gc() and a wait on the reference
	 * queue, RandomVariableCuda.java:311-335).  Only on the failure path: the hot path allocates nothing for it.  (The C++ mirror models
	 * this caller: host/random_variable.hpp, ReleaseLag.)
	 */
	private static long afterCollection(final long handle, final java.util.function.LongSupplier again) {
		if(handle != 0 || !Native.lastError().startsWith("device allocation")) {
			return handle;
		}
		System.gc();
		try { Thread.sleep(50); } catch(final InterruptedException e) { Thread.currentThread().interrupt(); }
		return again.getAsLong();
	}

	DeviceVector v1s0(final int opcode) {
		try { return new DeviceVector(afterCollection(Native.callV1s0(opcode, handle), () -> Native.callV1s0(opcode, handle)), size); }
		finally { Reference.reachabilityFence(this); }
	}
	DeviceVector v1s1(final int opcode, final double s) {
		try { return new DeviceVector(afterCollection(Native.callV1s1(opcode, handle, s), () -> Native.callV1s1(opcode, handle, s)), size); }
		finally { Reference.reachabilityFence(this); }
	}
	DeviceVector v2s0(final int opcode, final DeviceVector b) {
		try { return new DeviceVector(afterCollection(Native.callV2s0(opcode, handle, b.handle), () -> Native.callV2s0(opcode, handle, b.handle)), size); }
		finally { Reference.reachabilityFence(this); Reference.reachabilityFence(b); }
	}
	DeviceVector v2s1(final int opcode, final DeviceVector b, final double s) {
		try { return new DeviceVector(afterCollection(Native.callV2s1(opcode, handle, b.handle, s), () -> Native.callV2s1(opcode, handle, b.handle, s)), size); }
		finally { Reference.reachabilityFence(this); Reference.reachabilityFence(b); }
	}
	DeviceVector v3s0(final int opcode, final DeviceVector b, final DeviceVector c) {
		try { return new DeviceVector(afterCollection(Native.callV3s0(opcode, handle, b.handle, c.handle), () -> Native.callV3s0(opcode, handle, b.handle, c.handle)), size); }
		finally { Reference.reachabilityFence(this); Reference.reachabilityFence(b); Reference.reachabilityFence(c); }
	}
}
