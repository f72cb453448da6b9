import torch, sys
sys.path.insert(0, '.')
from differentiable_ransac_amd import ops
from tests.conftest import load_golden
g = load_golden("f8")
F, valid = ops.solve_f8(g["samples"].to('cuda'))
print(F[0], g["samples"][0].mean(0))
