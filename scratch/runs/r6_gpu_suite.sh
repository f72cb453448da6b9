#!/bin/bash
# the whole GPU suite + smoke at the tree's state
mkdir -p gpurun_out/suite
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/suite/pytest.log 2>&1; tail -3 gpurun_out/suite/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
