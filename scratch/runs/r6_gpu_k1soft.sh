#!/bin/bash
# round 6: the mask selection in every mode of the register kernel: identity with the list selection, the whole GPU suite
mkdir -p gpurun_out/k1sel
cd $GRAFT_REPO_ROOT
timeout 900 python scratch/k1_select_check.py > gpurun_out/k1sel/check.txt 2>&1; echo "check rc $?"
grep -v amdgpu.ids gpurun_out/k1sel/check.txt | grep "us$\|ALL\|DIFFER\|Error\|error" | head
grep -c identical gpurun_out/k1sel/check.txt
rm -f gpurun_out/k1sel/*.pt
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/k1sel/pytest.log 2>&1; tail -3 gpurun_out/k1sel/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
