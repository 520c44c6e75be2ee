/*
 * RandomVariableHipFactory.java — the injection point: every finmath-lib model that takes a RandomVariableFactory
 * (LIBORMarketModelFromCovarianceModel.of(…, randomVariableFactory, …), BrownianMotionFromMersenneRandomNumbers(…, factory),
 * LIBORMarketModelCalibrationATMTest.java:283,351-358) runs on the MI355X engine by passing this factory instead of
 * RandomVariableCudaFactory (RandomVariableCudaFactory.java:27-34).  UNCOMPILED / UNTESTED here (no JDK, no finmath-lib jar).
 */
package net.finmath.hip;

import net.finmath.montecarlo.AbstractRandomVariableFactory;
import net.finmath.montecarlo.RandomVariableFactory;
import net.finmath.stochastic.RandomVariable;

public class RandomVariableHipFactory extends AbstractRandomVariableFactory implements RandomVariableFactory {

	private static final long serialVersionUID = 1L;

	public RandomVariableHipFactory() {
		super();
	}

	/** A constant: lives on the host only, no device memory, no launch (as RandomVariableCuda.java:683-689). */
	@Override
	public RandomVariable createRandomVariable(final double time, final double value) {
		return new RandomVariableHip(time, value);
	}

	/** A stochastic value: the doubles are narrowed to fp32 and uploaded once. */
	@Override
	public RandomVariable createRandomVariable(final double time, final double[] values) {
		return new RandomVariableHip(time, values);
	}
}
