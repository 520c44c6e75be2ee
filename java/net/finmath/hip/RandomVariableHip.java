/*
 * RandomVariableHip.java — net.finmath.stochastic.RandomVariable on the MI355X engine (libfmhip.so through Native).
 *
 * Takes the place of RandomVariableCuda (src/main/java/net/finmath/cuda/montecarlo/RandomVariableCuda.java, cited as :line):
 * the same immutable data model — filtration time, type priority 20 (:568), EITHER a constant held as a double OR a vector
 * of fp32 realizations on the device (:566-577) — and the same dispatch in every method (higher type priority takes over,
 * maximum of the filtration times, host arithmetic when everything is constant, the scalar kernel when one side is).
 * What is gone: the inner DeviceMemoryPool (:119-558) with its single-thread executor, GC polling and cudaMemGetInfo; what
 * is new: a method call only RECORDS an operation when the engine's fusion front-end is on (Native.setFusion) — chains of
 * calls run as one fused launch when a value is read or reduced — and reductions return 32 bytes instead of the whole vector.
 *
 * Deliberate deviations from RandomVariableCuda, each following the reference's own CPU twin RandomVariableFromFloatArray
 * (listed in DESIGN.md §2): the new filtration time is kept in add/sub/bus, accrue/discount of a constant receiver,
 * addRatio/subRatio and vid(constant); cap(RandomVariable) with a constant argument works; the priority branch of vid is
 * div; choose, isNaN, sin, cos are implemented; variance is the twin's two-pass formula (second pass on the device);
 * min/max have java.lang.Math semantics.
 *
 * This is the Java rendering of the executable specifications host/random_variable.hpp (C++) and random_variable.py
 * (Python), which are what this repository's GPU tests and differential fuzzers run.
 * UNCOMPILED / UNTESTED here: no JDK, no finmath-lib-5.1.3.jar in the build image.
 */
package net.finmath.hip;

import java.util.Arrays;
import java.util.function.DoubleBinaryOperator;
import java.util.function.DoubleUnaryOperator;
import java.util.function.IntToDoubleFunction;
import java.util.stream.DoubleStream;

import net.finmath.functions.DoubleTernaryOperator;
import net.finmath.stochastic.RandomVariable;

public class RandomVariableHip implements RandomVariable {

	private static final long serialVersionUID = 1L;
	private static final int typePriorityDefault = 20;					// :568

	private final double time;											// filtration time, -infinity = no dependence (:571)
	private final int typePriority;
	private final double valueIfNonStochastic;							// used iff realizations == null
	private final transient DeviceVector realizations;					// fp32 values in HBM (or a pending expression)

	// ---- construction (:600-734)

	public RandomVariableHip(final double time, final double value) {
		this(time, value, null, typePriorityDefault);
	}

	public RandomVariableHip(final double value) {
		this(Double.NEGATIVE_INFINITY, value);
	}

	public RandomVariableHip(final double time, final double[] realisations) {
		this(time, Double.NaN, DeviceVector.fromHost(realisations), typePriorityDefault);
	}

	RandomVariableHip(final double time, final DeviceVector realizations) {
		this(time, Double.NaN, realizations, typePriorityDefault);
	}

	private RandomVariableHip(final double time, final double value, final DeviceVector realizations, final int typePriority) {
		this.time = time;
		this.valueIfNonStochastic = value;
		this.realizations = realizations;
		this.typePriority = typePriority;
	}

	static {
		Native.check(Native.init(Integer.getInteger("net.finmath.hip.device", -1)));		// -1: FMHIP_DEVICE_INDEX / LOCAL_RANK / 0
		Native.check(Native.setFusion(Boolean.parseBoolean(System.getProperty("net.finmath.hip.fusion", "true")) ? 1 : 0, null));
		// -Dnet.finmath.hip.threadEngines=true: an engine per Java thread (fmhip_set_thread_engines) — for a multi-threaded optimiser
		// (LevenbergMarquardt with numberOfThreads > 1) at path counts where the host, not the device, is the bound
		if (Boolean.getBoolean("net.finmath.hip.threadEngines")) Native.check(Native.setThreadEngines(1, null));
	}

	private static RandomVariableHip constant(final double time, final double value) {
		return new RandomVariableHip(time, value);
	}

	private static RandomVariableHip stochastic(final double time, final DeviceVector vector) {
		return new RandomVariableHip(time, vector);
	}

	/** Returns cached device buffers to the driver; the reference's tests call RandomVariableCuda.clean() / purge() in @After. */
	public static void clean() {
		Native.check(Native.poolClean());
	}

	public static void purge() {
		Native.check(Native.poolPurge());
	}

	/** Executes everything the fusion front-end has recorded (normally implicit: reading or reducing a value does it). */
	public static void flush() {
		Native.check(Native.flush());
	}

	// The device vector of any RandomVariable: ours, or an upload of getRealizations() for a foreign type (:759-766).
	private static DeviceVector vectorOf(final RandomVariable randomVariable) {
		if(randomVariable instanceof RandomVariableHip) {
			return ((RandomVariableHip)randomVariable).realizations;
		}
		return DeviceVector.fromHost(randomVariable.getRealizations());
	}

	private RandomVariable scalarOperation(final int opcode, final double scalar, final double resultIfConstant) {
		if(isDeterministic()) {
			return constant(time, resultIfConstant);
		}
		return stochastic(time, realizations.v1s1(opcode, scalar));
	}

	private RandomVariable unaryOperation(final int opcode, final double resultIfConstant) {
		if(isDeterministic()) {
			return constant(time, resultIfConstant);
		}
		return stochastic(time, realizations.v1s0(opcode));
	}

	// ---- accessors

	@Override
	public boolean equals(final RandomVariable randomVariable) {
		throw new UnsupportedOperationException();						// as :785-800
	}

	@Override
	public double getFiltrationTime() {
		return time;
	}

	@Override
	public int getTypePriority() {
		return typePriority;
	}

	@Override
	public double get(final int pathOrState) {
		if(isDeterministic()) {
			return valueIfNonStochastic;
		}
		throw new UnsupportedOperationException();						// as :812-818: no element access to device memory
	}

	@Override
	public int size() {
		return isDeterministic() ? 1 : (int)realizations.size;
	}

	@Override
	public boolean isDeterministic() {
		return realizations == null;
	}

	@Override
	public RandomVariable cache() {
		return this;													// :1099; a pending expression is executed when it is read
	}

	@Override
	public double[] getRealizations() {
		if(isDeterministic()) {
			return new double[] { valueIfNonStochastic };
		}
		return realizations.toDoubleArray();							// D2H of 4·N bytes + widening (:1115-1123)
	}

	@Override
	public Double doubleValue() {
		if(isDeterministic()) {
			return valueIfNonStochastic;
		}
		throw new UnsupportedOperationException("The random variable is non-deterministic");
	}

	@Override
	public IntToDoubleFunction getOperator() {
		if(isDeterministic()) {
			return i -> valueIfNonStochastic;
		}
		final double[] values = getRealizations();
		return i -> values[i];
	}

	@Override
	public DoubleStream getRealizationsStream() {
		return Arrays.stream(getRealizations());
	}

	// ---- reductions: on the device, 32 bytes per call come back (replaces :830-967)

	@Override
	public double getMin() {
		return isDeterministic() ? valueIfNonStochastic : realizations.moments(0.0)[2];
	}

	@Override
	public double getMax() {
		return isDeterministic() ? valueIfNonStochastic : realizations.moments(0.0)[3];
	}

	/**
	 * Paths behind an expectation: size() of this process's shard times the ranks of the expectation communicator
	 * (Native.setExpectationComm; 1 without one) - the moments the engine returns are then those of the global vector.
	 */
	private long sampleSize() {
		final int[] world = new int[1];
		Native.check(Native.expectationWorld(world, null));
		return (long)size() * world[0];
	}

	@Override
	public double getAverage() {
		if(isDeterministic()) {
			return valueIfNonStochastic;
		}
		if(size() == 0) {
			return Double.NaN;
		}
		return realizations.moments(0.0)[0] / sampleSize();
	}

	@Override
	public double getAverage(final RandomVariable probabilities) {
		return this.mult(probabilities).getAverage();					// :886-888
	}

	@Override
	public double getVariance() {
		if(isDeterministic() || size() == 1) {
			return 0.0;
		}
		if(size() == 0) {
			return Double.NaN;
		}
		final double average = getAverage();							// two passes like the twin (twin:360-382): Σ(x - mean)²/n, second pass on the device
		return realizations.moments(average)[1] / sampleSize();
	}

	@Override
	public double getVariance(final RandomVariable probabilities) {
		final double average = getAverage(probabilities);
		return this.squared().sub(average * average).getAverage(probabilities);		// :904-907
	}

	@Override
	public double getSampleVariance() {
		if(isDeterministic() || size() == 1) {
			return 0.0;
		}
		final long n = sampleSize();
		return getVariance() * n / (n - 1);
	}

	@Override
	public double getStandardDeviation() {
		return isDeterministic() ? 0.0 : Math.sqrt(getVariance());
	}

	@Override
	public double getStandardDeviation(final RandomVariable probabilities) {
		return isDeterministic() ? 0.0 : Math.sqrt(getVariance(probabilities));
	}

	@Override
	public double getStandardError() {
		return isDeterministic() ? 0.0 : getStandardDeviation() / Math.sqrt(sampleSize());
	}

	@Override
	public double getStandardError(final RandomVariable probabilities) {
		return isDeterministic() ? 0.0 : getStandardDeviation(probabilities) / Math.sqrt(sampleSize());
	}

	// ---- host-side cold paths: the reference sorts on the host as well (:970-1091)

	@Override
	public double getQuantile(final double quantile) {
		if(isDeterministic()) {
			return valueIfNonStochastic;
		}
		if(size() == 0) {
			return Double.NaN;
		}
		final double[] sorted = getRealizations();
		Arrays.sort(sorted);
		final int index = (int)Math.round((size() + 1) * (1 - quantile) - 1);			// index convention of the GPU class (:983)
		return sorted[Math.min(Math.max(index, 0), sorted.length - 1)];
	}

	@Override
	public double getQuantile(final double quantile, final RandomVariable probabilities) {
		throw new RuntimeException("Method not implemented.");			// as :989-998
	}

	// The three methods below share one sorted copy of the sample and two helpers (as random_variable.py and host/random_variable.hpp
	// do): the position a quantile falls on, and the number of sample points not above a bound (binary search).  Semantics as the
	// reference's (:1001-1091): mean of the sorted sample between two quantile positions; shares of the sample per interval
	// (points[k-1], points[k]] plus the share above the last point; a symmetric grid of points around the mean with its bin edges.

	private double[] sortedSample() {
		final double[] sample = getRealizations();
		Arrays.sort(sample);
		return sample;
	}

	/** Position of a quantile in a sorted sample of this size: round((n + 1) q - 1), kept inside the sample. */
	private int quantilePosition(final double quantile, final int sampleLength) {
		final long position = Math.round((size() + 1) * quantile - 1);
		return (int)Math.min(Math.max(position, 0L), sampleLength - 1L);
	}

	/** Number of elements of the sorted array that are <= bound. */
	private static int countNotAbove(final double[] sorted, final double bound) {
		int low = 0, high = sorted.length;
		while(low < high) {
			final int middle = (low + high) >>> 1;
			if(sorted[middle] <= bound) {
				low = middle + 1;
			}
			else {
				high = middle;
			}
		}
		return low;
	}

	@Override
	public double getQuantileExpectation(final double quantileStart, final double quantileEnd) {
		if(isDeterministic()) {
			return valueIfNonStochastic;
		}
		if(size() == 0) {
			return Double.NaN;
		}
		final double[] sample = sortedSample();
		final int from = quantilePosition(Math.min(quantileStart, quantileEnd), sample.length);
		final int to = quantilePosition(Math.max(quantileStart, quantileEnd), sample.length);
		double sum = 0.0;			// plain left-to-right sum, as the reference and the C++ / Python mirrors (Arrays.stream(...).sum() is compensated: other bits)
		for(int i = from; i <= to; i++) {
			sum += sample[i];
		}
		return sum / (to - from + 1);
	}

	@Override
	public double[] getHistogram(final double[] intervalPoints) {
		final int bins = intervalPoints.length + 1;
		final double[] shares = new double[bins];
		if(isDeterministic()) {
			// the reference's convention for a constant (:1030-1041): the first point below the value, and the last bin
			for(int k = 0; k < intervalPoints.length; k++) {
				if(valueIfNonStochastic > intervalPoints[k]) {
					shares[k] = 1.0;
					break;
				}
			}
			shares[bins - 1] = 1.0;
			return shares;
		}
		final double[] sample = sortedSample();
		int below = 0;							// sample points accounted for so far
		for(int k = 0; k < intervalPoints.length; k++) {
			final int upTo = Math.max(countNotAbove(sample, intervalPoints[k]), below);		// (points that are not ascending get empty intervals)
			shares[k] = upTo - below;
			below = upTo;
		}
		shares[bins - 1] = sample.length - below;
		if(sample.length > 0) {
			for(int k = 0; k < bins; k++) {
				shares[k] /= sample.length;
			}
		}
		return shares;
	}

	@Override
	public double[][] getHistogram(final int numberOfPoints, final double standardDeviations) {
		final double center = getAverage();
		final double radius = standardDeviations * getStandardDeviation();
		final double halfSpan = (numberOfPoints - 1) / 2.0;			// grid points per side
		final double halfBin = radius / (2 * halfSpan);
		final double[] points = new double[numberOfPoints];
		final double[] edges = new double[numberOfPoints + 1];
		for(int i = 0; i < numberOfPoints; i++) {
			points[i] = center + (i - halfSpan) / halfSpan * radius;
			edges[i] = points[i] - halfBin;
		}
		edges[numberOfPoints] = center + radius + halfBin;
		return new double[][] { edges, getHistogram(points) };
	}

	@Override
	public RandomVariable apply(final DoubleUnaryOperator function) {
		if(isDeterministic()) {
			return constant(time, function.applyAsDouble(valueIfNonStochastic));
		}
		final double[] values = getRealizations();						// the twin maps on the host (twin:667-676); RandomVariableCuda throws (:1146-1159)
		for(int i = 0; i < values.length; i++) {
			values[i] = function.applyAsDouble(values[i]);
		}
		return new RandomVariableHip(time, values);
	}

	@Override
	public RandomVariable apply(final DoubleBinaryOperator operator, final RandomVariable argument) {
		final double newTime = Math.max(time, argument.getFiltrationTime());
		final double[] x = getRealizations(), y = argument.getRealizations();
		final double[] values = new double[Math.max(x.length, y.length)];
		for(int i = 0; i < values.length; i++) {
			values[i] = operator.applyAsDouble(x[x.length == 1 ? 0 : i], y[y.length == 1 ? 0 : i]);
		}
		return values.length == 1 ? constant(newTime, values[0]) : new RandomVariableHip(newTime, values);
	}

	@Override
	public RandomVariable apply(final DoubleTernaryOperator operator, final RandomVariable argument1, final RandomVariable argument2) {
		final double newTime = Math.max(Math.max(time, argument1.getFiltrationTime()), argument2.getFiltrationTime());
		final double[] x = getRealizations(), y = argument1.getRealizations(), z = argument2.getRealizations();
		final double[] values = new double[Math.max(x.length, Math.max(y.length, z.length))];
		for(int i = 0; i < values.length; i++) {
			values[i] = operator.applyAsDouble(x[x.length == 1 ? 0 : i], y[y.length == 1 ? 0 : i], z[z.length == 1 ? 0 : i]);
		}
		return values.length == 1 ? constant(newTime, values[0]) : new RandomVariableHip(newTime, values);
	}

	// ---- scalar operand / unary (:1172-1384): fmhip_call_v1s1 / fmhip_call_v1s0

	@Override
	public RandomVariable cap(final double cap) {
		return scalarOperation(Opcode.CAP_S, cap, Math.min(valueIfNonStochastic, cap));
	}

	@Override
	public RandomVariable floor(final double floor) {
		return scalarOperation(Opcode.FLOOR_S, floor, Math.max(valueIfNonStochastic, floor));
	}

	@Override
	public RandomVariable add(final double value) {
		return scalarOperation(Opcode.ADD_S, value, valueIfNonStochastic + value);
	}

	@Override
	public RandomVariable sub(final double value) {
		return scalarOperation(Opcode.SUB_S, value, valueIfNonStochastic - value);
	}

	@Override
	public RandomVariable bus(final double value) {
		return scalarOperation(Opcode.BUS_S, value, -valueIfNonStochastic + value);
	}

	@Override
	public RandomVariable mult(final double value) {
		return scalarOperation(Opcode.MULT_S, value, valueIfNonStochastic * value);
	}

	@Override
	public RandomVariable div(final double value) {
		return scalarOperation(Opcode.DIV_S, value, valueIfNonStochastic / value);
	}

	@Override
	public RandomVariable vid(final double value) {
		return scalarOperation(Opcode.VID_S, value, value / valueIfNonStochastic);
	}

	@Override
	public RandomVariable pow(final double exponent) {
		return scalarOperation(Opcode.POW_S, exponent, Math.pow(valueIfNonStochastic, exponent));
	}

	@Override
	public RandomVariable average() {
		return constant(Double.NEGATIVE_INFINITY, getAverage());		// :1280
	}

	@Override
	public RandomVariable squared() {
		return unaryOperation(Opcode.SQUARED, valueIfNonStochastic * valueIfNonStochastic);
	}

	@Override
	public RandomVariable sqrt() {
		return unaryOperation(Opcode.SQRT, Math.sqrt(valueIfNonStochastic));
	}

	@Override
	public RandomVariable invert() {
		return unaryOperation(Opcode.INVERT, 1.0 / valueIfNonStochastic);
	}

	@Override
	public RandomVariable abs() {
		return unaryOperation(Opcode.ABS, Math.abs(valueIfNonStochastic));
	}

	@Override
	public RandomVariable exp() {
		return unaryOperation(Opcode.EXP, Math.exp(valueIfNonStochastic));
	}

	@Override
	public RandomVariable log() {
		return unaryOperation(Opcode.LOG, Math.log(valueIfNonStochastic));
	}

	@Override
	public RandomVariable sin() {
		return unaryOperation(Opcode.SIN, Math.sin(valueIfNonStochastic));			// RandomVariableCuda throws (:1355-1368); semantics of the twin (twin:927)
	}

	@Override
	public RandomVariable cos() {
		return unaryOperation(Opcode.COS, Math.cos(valueIfNonStochastic));
	}

	@Override
	public RandomVariable isNaN() {
		return unaryOperation(Opcode.ISNAN, Double.isNaN(valueIfNonStochastic) ? 1.0 : 0.0);	// RandomVariableCuda returns null (:1701-1704); twin:1441-1451
	}

	// ---- vector operand (:1391-1580): type priority → new time → constant fast paths → fmhip_call_v2s0

	private RandomVariable binary(final RandomVariable argument, final int opcodeVectorVector, final int opcodeVectorScalar, final int opcodeConstantReceiver,
			final double resultIfBothConstant) {
		final double newTime = Math.max(time, argument.getFiltrationTime());
		if(isDeterministic() && argument.isDeterministic()) {
			return constant(newTime, resultIfBothConstant);
		}
		if(isDeterministic()) {											// constant receiver: the scalar kernel on the argument
			return stochastic(newTime, vectorOf(argument).v1s1(opcodeConstantReceiver, valueIfNonStochastic));
		}
		if(argument.isDeterministic()) {
			return stochastic(newTime, realizations.v1s1(opcodeVectorScalar, argument.doubleValue()));
		}
		return stochastic(newTime, realizations.v2s0(opcodeVectorVector, vectorOf(argument)));
	}

	@Override
	public RandomVariable add(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.add(this);							// :1392-1395
		}
		return binary(randomVariable, Opcode.ADD, Opcode.ADD_S, Opcode.ADD_S, valueIfNonStochastic + (randomVariable.isDeterministic() ? randomVariable.doubleValue() : 0.0));
	}

	@Override
	public RandomVariable sub(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.bus(this);
		}
		return binary(randomVariable, Opcode.SUB, Opcode.SUB_S, Opcode.BUS_S, valueIfNonStochastic - (randomVariable.isDeterministic() ? randomVariable.doubleValue() : 0.0));
	}

	@Override
	public RandomVariable bus(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.sub(this);
		}
		final double newTime = Math.max(time, randomVariable.getFiltrationTime());
		if(isDeterministic() && randomVariable.isDeterministic()) {
			return constant(newTime, -valueIfNonStochastic + randomVariable.doubleValue());
		}
		if(isDeterministic()) {
			return stochastic(newTime, vectorOf(randomVariable).v1s1(Opcode.SUB_S, valueIfNonStochastic));
		}
		if(randomVariable.isDeterministic()) {
			return stochastic(newTime, realizations.v1s1(Opcode.BUS_S, randomVariable.doubleValue()));
		}
		return stochastic(newTime, vectorOf(randomVariable).v2s0(Opcode.SUB, realizations));		// flipped arguments (:1458)
	}

	@Override
	public RandomVariable mult(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.mult(this);
		}
		return binary(randomVariable, Opcode.MULT, Opcode.MULT_S, Opcode.MULT_S, valueIfNonStochastic * (randomVariable.isDeterministic() ? randomVariable.doubleValue() : 0.0));
	}

	@Override
	public RandomVariable div(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.vid(this);
		}
		return binary(randomVariable, Opcode.DIV, Opcode.DIV_S, Opcode.VID_S, valueIfNonStochastic / (randomVariable.isDeterministic() ? randomVariable.doubleValue() : 1.0));
	}

	@Override
	public RandomVariable vid(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.div(this);							// the twin's branch (twin:1116-1119); RandomVariableCuda calls vid here (:1513-1516)
		}
		final double newTime = Math.max(time, randomVariable.getFiltrationTime());
		if(isDeterministic() && randomVariable.isDeterministic()) {
			return constant(newTime, randomVariable.doubleValue() / valueIfNonStochastic);
		}
		if(isDeterministic()) {
			return stochastic(newTime, vectorOf(randomVariable).v1s1(Opcode.DIV_S, valueIfNonStochastic));
		}
		if(randomVariable.isDeterministic()) {
			return stochastic(newTime, realizations.v1s1(Opcode.VID_S, randomVariable.doubleValue()));	// constant narrowed to fp32 first, as :1528
		}
		return stochastic(newTime, vectorOf(randomVariable).v2s0(Opcode.DIV, realizations));		// flipped arguments (:1531)
	}

	@Override
	public RandomVariable cap(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.cap(this);
		}
		return binary(randomVariable, Opcode.CAP, Opcode.CAP_S, Opcode.CAP_S, Math.min(valueIfNonStochastic, randomVariable.isDeterministic() ? randomVariable.doubleValue() : 0.0));
	}

	@Override
	public RandomVariable floor(final RandomVariable randomVariable) {
		if(randomVariable.getTypePriority() > this.getTypePriority()) {
			return randomVariable.floor(this);
		}
		return binary(randomVariable, Opcode.FLOOR, Opcode.FLOOR_S, Opcode.FLOOR_S, Math.max(valueIfNonStochastic, randomVariable.isDeterministic() ? randomVariable.doubleValue() : 0.0));
	}

	@Override
	public RandomVariable accrue(final RandomVariable rate, final double periodLength) {
		if(rate.getTypePriority() > this.getTypePriority()) {
			return rate.mult(periodLength).add(1.0).mult(this);		// :1584-1587
		}
		final double newTime = Math.max(time, rate.getFiltrationTime());
		if(rate.isDeterministic()) {
			return this.mult(1.0 + rate.doubleValue() * periodLength);
		}
		if(isDeterministic()) {
			return stochastic(newTime, vectorOf(rate).v1s1(Opcode.MULT_S, periodLength).v1s1(Opcode.ADD_S, 1.0).v1s1(Opcode.MULT_S, valueIfNonStochastic));
		}
		return stochastic(newTime, realizations.v2s1(Opcode.ACCRUE, vectorOf(rate), periodLength));	// a·(1 + b·Δ), each product / sum rounded separately (.cu:224-231)
	}

	@Override
	public RandomVariable discount(final RandomVariable rate, final double periodLength) {
		if(rate.getTypePriority() > this.getTypePriority()) {
			return rate.mult(periodLength).add(1.0).invert().mult(this);
		}
		final double newTime = Math.max(time, rate.getFiltrationTime());
		if(rate.isDeterministic()) {
			return this.div(1.0 + rate.doubleValue() * periodLength);
		}
		if(isDeterministic()) {
			return stochastic(newTime, vectorOf(rate).v1s1(Opcode.MULT_S, periodLength).v1s1(Opcode.ADD_S, 1.0).v1s1(Opcode.VID_S, valueIfNonStochastic));
		}
		return stochastic(newTime, realizations.v2s1(Opcode.DISCOUNT, vectorOf(rate), periodLength));
	}

	@Override
	public RandomVariable choose(final RandomVariable valueIfTriggerNonNegative, final RandomVariable valueIfTriggerNegative) {
		// RandomVariableCuda returns null (:1632-1635); semantics of the twin: trigger >= 0 ? first : second (twin:1264-1285)
		final double newTime = Math.max(Math.max(time, valueIfTriggerNonNegative.getFiltrationTime()), valueIfTriggerNegative.getFiltrationTime());
		if(isDeterministic()) {
			return valueIfNonStochastic >= 0 ? valueIfTriggerNonNegative : valueIfTriggerNegative;
		}
		final DeviceVector first = valueIfTriggerNonNegative.isDeterministic() ? DeviceVector.filled(size(), valueIfTriggerNonNegative.doubleValue()) : vectorOf(valueIfTriggerNonNegative);
		final DeviceVector second = valueIfTriggerNegative.isDeterministic() ? DeviceVector.filled(size(), valueIfTriggerNegative.doubleValue()) : vectorOf(valueIfTriggerNegative);
		return stochastic(newTime, realizations.v3s0(Opcode.CHOOSE, first, second));
	}

	@Override
	public RandomVariable addProduct(final RandomVariable factor1, final double factor2) {
		if(factor1.getTypePriority() > this.getTypePriority()) {
			return factor1.mult(factor2).add(this);					// :1639-1642
		}
		final double newTime = Math.max(time, factor1.getFiltrationTime());
		if(factor1.isDeterministic()) {
			return this.add(factor1.doubleValue() * factor2);
		}
		if(!isDeterministic()) {
			return stochastic(newTime, realizations.v2s1(Opcode.ADDPRODUCT_VS, vectorOf(factor1), factor2));	// a + b·s, two roundings (.cu:257-264)
		}
		return this.add(factor1.mult(factor2));
	}

	@Override
	public RandomVariable addProduct(final RandomVariable factor1, final RandomVariable factor2) {
		if(factor1.getTypePriority() > this.getTypePriority() || factor2.getTypePriority() > this.getTypePriority()) {
			return factor1.mult(factor2).add(this);
		}
		final double newTime = Math.max(Math.max(time, factor1.getFiltrationTime()), factor2.getFiltrationTime());
		if(isDeterministic() && factor1.isDeterministic() && factor2.isDeterministic()) {
			return constant(newTime, valueIfNonStochastic + factor1.doubleValue() * factor2.doubleValue());
		}
		if(factor1.isDeterministic() && factor2.isDeterministic()) {
			return this.add(factor1.doubleValue() * factor2.doubleValue());
		}
		if(factor2.isDeterministic()) {
			return this.addProduct(factor1, factor2.doubleValue());
		}
		if(factor1.isDeterministic()) {
			return this.addProduct(factor2, factor1.doubleValue());
		}
		if(!isDeterministic()) {
			return stochastic(newTime, realizations.v3s0(Opcode.ADDPRODUCT, vectorOf(factor1), vectorOf(factor2)));	// .cu:247-254
		}
		return this.add(factor1.mult(factor2));
	}

	@Override
	public RandomVariable addRatio(final RandomVariable numerator, final RandomVariable denominator) {
		return this.add(numerator.div(denominator));					// :1686-1689; fused into one launch by the front-end
	}

	@Override
	public RandomVariable subRatio(final RandomVariable numerator, final RandomVariable denominator) {
		return this.sub(numerator.div(denominator));					// :1692-1695
	}
}
