#!/bin/bash
# round 5: variants of the 8-points-per-lane rigid residual kernel in the config-4 step (group of models per scalar fetch, occupancy hints)
cd $GRAFT_REPO_ROOT
AB_ARGS="--workload c4" bash scratch/ab_step.sh cur k4r8g2 k4r8g1 k4r8g3
