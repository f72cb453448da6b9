#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m "not gpu" 2>&1 | grep -E "FAILED|Error|assert " | head -30
