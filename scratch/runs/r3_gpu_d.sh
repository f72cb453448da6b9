#!/bin/bash
# round 3, GPU pass D: K4 parity tests on the tree library, then in-step A/B of the model-fetch variants of the scoring kernel
# (cur = real SGPR pairs + inline-asm prefetch awaited at the end of the iteration; spair = pairs only; asmonly = asm prefetch
# with op_sel broadcasts; base = round-2 form), two rounds each
mkdir -p gpurun_out/r3d
(timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log)
tail -4 gpurun_out/r3d/pytest.log
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur base spair asmonly 2>&1 | tee gpurun_out/r3d/ab_model_fetch.log
