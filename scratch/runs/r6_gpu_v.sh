#!/bin/bash
# round 6, run v: the full GPU suite once more at HEAD on a fresh box + the solver fuzz of round 4 (asm chain evaluation in place)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 600 python scratch/fuzz_r4.py 2>&1 | tail -12
