"""The quick-start example of README.md (run on the GPU box: python benchmarks/readme_example.py)."""
import importlib, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fm = importlib.import_module("finmath-lib-cuda-extensions_amd")
fm.init(0)
fm.set_fusion(True)
factory = fm.RandomVariableHipFactory()
bm = fm.BrownianMotionHip(fm.TimeDiscretization(0.0, 10, 0.2), 1, 1_000_000, 31415)
x = factory.createRandomVariable(0.0)
for i in range(10):
    x = x.add((0.05 - 0.5 * 0.3 ** 2) * 0.2).addProduct(bm.getBrownianIncrement(i, 0), 0.3)
call = x.exp().sub(1.05).floor(0.0).div(2.718281828 ** (0.05 * 2.0))
print(call.getAverage(), call.getStandardError())
aad = fm.RandomVariableDifferentiableAADFactory(factory)
print(aad.createRandomVariable(2.0).squared().getGradient())
