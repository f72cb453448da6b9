#!/bin/bash
# the whole GPU suite + smoke at the tree's state, then the headline and train lines
mkdir -p gpurun_out/suite
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/suite/pytest.log 2>&1; tail -3 gpurun_out/suite/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scratch/ab_step.sh cur 2>&1 | grep -v amdgpu.ids
AB_ARGS="--mode train" bash scratch/ab_step.sh cur 2>&1 | grep -v amdgpu.ids
