#!/bin/bash
# round 6, run q: full GPU suite + smoke + build() from scratch on the box (what the driver does at round end)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -q -m "not gpu" 2>&1 | tail -2
