#!/bin/bash
# round 6, run t: the flag-compatible harness end to end (test / train / F / f64 / 3-D) after the round's changes
cd $GRAFT_REPO_ROOT
python tools/run_path.py -nf 2000 -bs 32 -rbs 1024 -sam 2 -tr 0 -t 0.75 2>&1 | tail -3
python tools/run_path.py -nf 2000 -bs 32 -rbs 64 -sam 2 -tr 0 -t 0.75 2>&1 | tail -3
python tools/run_path.py -nf 2000 -bs 32 -rbs 1024 -sam 2 -tr 1 -t 0.75 2>&1 | tail -3
python tools/run_path.py -nf 2000 -bs 8 -rbs 64 -sam 3 -fmat 1 -tr 0 -t 0.75 2>&1 | tail -3
python tools/run_path.py -nf 500 -bs 4 -rbs 256 -sam 3 -tr 1 -pr 2 -t 0.75 2>&1 | tail -3
python tools/run_path.py --three-d -nf 5000 -bs 2 -rbs 512 -sam 2 -tr 1 2>&1 | tail -3
