/*
 * Opcode.java — the fmhip_opcode values of include/fmhip.h (each names the reference kernel it replaces,
 * RandomVariableCudaKernel.cu).  tests/test_jni_binding_cpu.py checks the numbers against the header.
 */
package net.finmath.hip;

final class Opcode {
	private Opcode() { }

	static final int CAP_S = 1, FLOOR_S = 2, ADD_S = 3, SUB_S = 4, BUS_S = 5, MULT_S = 6, DIV_S = 7, VID_S = 8, POW_S = 9;
	static final int SQUARED = 10, SQRT = 11, EXP = 12, LOG = 13, INVERT = 14, ABS = 15, SIN = 16, COS = 17, ISNAN = 18;
	static final int CAP = 19, FLOOR = 20, ADD = 21, SUB = 22, MULT = 23, DIV = 24;
	static final int ACCRUE = 25, DISCOUNT = 26, ADDPRODUCT_VS = 27;
	static final int ADDPRODUCT = 28, ADDRATIO = 29, SUBRATIO = 30, CHOOSE = 31;
}
