#!/bin/bash
# round 3, GPU pass Q: K4 with 128-thread workgroups (one 64-slot tile each, "h1") against the two-halves workgroups ("cur")
mkdir -p gpurun_out/r3q
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur h1 2>&1 | tee gpurun_out/r3q/ab_halves.log
