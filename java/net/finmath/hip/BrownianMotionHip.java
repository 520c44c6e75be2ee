/*
 * BrownianMotionHip.java — net.finmath.montecarlo.BrownianMotion whose increments are generated ON the device.
 *
 * Takes the place of BrownianMotionCudaWithRandomVariableCuda (…/alternative/BrownianMotionCudaWithRandomVariableCuda.java,
 * cited as :line): same constructor arguments, same lazy generation of ALL increments on first access under a lock (:123-139),
 * same time stamp t_{i+1} and scaling sqrt(dt_i) per increment (:170-175).  What changes: the reference creates a cuRAND
 * XORWOW generator and calls curandGenerateNormal once per (time step, factor) into a freshly allocated vector (:153-181);
 * here ONE native call fills one slab with all steps x factors vectors — counter-based Philox4x32-10 and an LDS-table-driven
 * inverse normal CDF (DESIGN.md §4.3).  An increment is a pure function of (seed, time index, factor, GLOBAL path index):
 * `pathOffset` shards the paths over several GPUs / processes without changing a single number.
 * cuRAND's bit stream is not reproduced — the reference's tests pin the increments statistically only
 * (BrownianMotionTest.java:120-121).  UNCOMPILED / UNTESTED here (no JDK, no finmath-lib jar).
 */
package net.finmath.hip;

import java.io.Serializable;

import net.finmath.montecarlo.BrownianMotion;
import net.finmath.stochastic.RandomVariable;
import net.finmath.time.TimeDiscretization;

public class BrownianMotionHip implements BrownianMotion, Serializable {

	private static final long serialVersionUID = 1L;

	private final TimeDiscretization timeDiscretization;
	private final int numberOfFactors;
	private final int numberOfPaths;
	private final int seed;
	private final long pathOffset;

	private transient RandomVariable[][] brownianIncrements;
	private final Object brownianIncrementsLazyInitLock = new Object();

	/*
	 * Time-step grouping on the caller's behalf is the ENGINE's business (fmhip_set_step_grouping, include/fmhip.h): finmath-lib's
	 * EulerSchemeFromProcessModel reads the increments of time index i exactly while it computes step i, the engine watches for the
	 * first use of an increment with a new time index and executes the methods recorded between S such boundaries together (whole
	 * time steps, their periodic part as one rolled-loop launch) instead of cutting the stream every ~40 methods.  The system
	 * property net.finmath.hip.groupTimeSteps overrides the engine's default (4; 0 = off).
	 */
	static {
		final Integer steps = Integer.getInteger("net.finmath.hip.groupTimeSteps");
		if(steps != null) {
			Native.check(Native.setStepGrouping(steps, null));
		}
	}

	/**
	 * @param timeDiscretization the time grid; increment i covers [t_i, t_{i+1}]
	 * @param numberOfFactors independent components per time step
	 * @param numberOfPaths paths held by THIS process
	 * @param seed the seed (the Philox key)
	 * @param pathOffset global index of this process's first path: rank·numberOfPaths when the paths are sharded over GPUs, else 0
	 */
	public BrownianMotionHip(final TimeDiscretization timeDiscretization, final int numberOfFactors, final int numberOfPaths, final int seed, final long pathOffset) {
		this.timeDiscretization = timeDiscretization;
		this.numberOfFactors = numberOfFactors;
		this.numberOfPaths = numberOfPaths;
		this.seed = seed;
		this.pathOffset = pathOffset;
	}

	public BrownianMotionHip(final TimeDiscretization timeDiscretization, final int numberOfFactors, final int numberOfPaths, final int seed) {
		this(timeDiscretization, numberOfFactors, numberOfPaths, seed, 0L);
	}

	@Override
	public BrownianMotion getCloneWithModifiedSeed(final int seed) {
		return new BrownianMotionHip(getTimeDiscretization(), getNumberOfFactors(), getNumberOfPaths(), seed, pathOffset);
	}

	@Override
	public BrownianMotion getCloneWithModifiedTimeDiscretization(final TimeDiscretization newTimeDiscretization) {
		return new BrownianMotionHip(newTimeDiscretization, getNumberOfFactors(), getNumberOfPaths(), getSeed(), pathOffset);
	}

	@Override
	public RandomVariable getBrownianIncrement(final int timeIndex, final int factor) {
		synchronized(brownianIncrementsLazyInitLock) {
			if(brownianIncrements == null) {
				doGenerateBrownianMotion();
			}
		}
		return brownianIncrements[timeIndex][factor];
	}

	/** Number of time steps the engine executes together on the caller's behalf (0 = off); process-wide. */
	public static void setGroupSteps(final int steps) {
		Native.check(Native.setStepGrouping(steps, null));
	}

	private void doGenerateBrownianMotion() {
		final int numberOfTimeSteps = timeDiscretization.getNumberOfTimeSteps();
		final double[] timeSteps = new double[numberOfTimeSteps];
		for(int timeIndex = 0; timeIndex < numberOfTimeSteps; timeIndex++) {
			timeSteps[timeIndex] = timeDiscretization.getTimeStep(timeIndex);
		}
		final long[] handles = new long[numberOfTimeSteps * numberOfFactors];
		Native.check(Native.bmGenerate(seed, numberOfTimeSteps, numberOfFactors, numberOfPaths, pathOffset, timeSteps, handles));
		final RandomVariable[][] increments = new RandomVariable[numberOfTimeSteps][numberOfFactors];
		for(int timeIndex = 0; timeIndex < numberOfTimeSteps; timeIndex++) {
			final double time = timeDiscretization.getTime(timeIndex + 1);					// :175
			for(int factor = 0; factor < numberOfFactors; factor++) {
				increments[timeIndex][factor] = new RandomVariableHip(time, new DeviceVector(handles[timeIndex * numberOfFactors + factor], numberOfPaths));
			}
		}
		brownianIncrements = increments;
	}

	@Override
	public TimeDiscretization getTimeDiscretization() {
		return timeDiscretization;
	}

	@Override
	public int getNumberOfFactors() {
		return numberOfFactors;
	}

	@Override
	public int getNumberOfPaths() {
		return numberOfPaths;
	}

	@Override
	public RandomVariable getRandomVariableForConstant(final double value) {
		return new RandomVariableHip(value);
	}

	@Override
	public RandomVariable getIncrement(final int timeIndex, final int factor) {
		return getBrownianIncrement(timeIndex, factor);
	}

	public int getSeed() {
		return seed;
	}

	public long getPathOffset() {
		return pathOffset;
	}

	@Override
	public String toString() {
		return super.toString() + "\n" + "timeDiscretization: " + timeDiscretization.toString() + "\n" + "numberOfPaths: " + numberOfPaths + "\n"
				+ "numberOfFactors: " + numberOfFactors + "\n" + "seed: " + seed + "\n" + "pathOffset: " + pathOffset;
	}

	@Override
	public boolean equals(final Object o) {
		if(this == o) {
			return true;
		}
		if(o == null || getClass() != o.getClass()) {
			return false;
		}
		final BrownianMotionHip that = (BrownianMotionHip)o;
		return numberOfFactors == that.numberOfFactors && numberOfPaths == that.numberOfPaths && seed == that.seed && pathOffset == that.pathOffset
				&& timeDiscretization.equals(that.timeDiscretization);
	}

	@Override
	public int hashCode() {
		int result = timeDiscretization.hashCode();
		result = 31 * result + numberOfFactors;
		result = 31 * result + numberOfPaths;
		result = 31 * result + seed;
		result = 31 * result + Long.hashCode(pathOffset);
		return result;
	}
}
