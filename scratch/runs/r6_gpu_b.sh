cd $GRAFT_REPO_ROOT
DROPIN_RBS=64 timeout 300 python scratch/dropin_loop.py 2>&1 | tail -12
