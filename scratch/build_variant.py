"""Builds scratch/libdransac_<name>.so = the tree's library with ONE source recompiled under extra -D flags (in-step A/B builds).
    python scratch/build_variant.py <name> <source.hip> [-DKNOB=V ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differentiable_ransac_amd import build as B   # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_library()
obj = os.path.join(ROOT, "scratch", f"_{name}_{src[:-4]}.o")
subprocess.check_call([B.HIPCC, *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.OBJ, s[:-4] + ".o") for s in B._sources()]
out = os.path.join(ROOT, "scratch", f"libdransac_{name}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print("built", out)
