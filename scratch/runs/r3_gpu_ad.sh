#!/bin/bash
# round 3, GPU pass AD: the converged-iterate fix of the safeguarded Newton step (tree) against the old rule ("oldnewton"), and
# shorter refine schedules on top of the fix (sch66 / sch47 / sch38 = BIS_LAST NEWT_LAST 6 6 | 4 7 | 3 8; tree 10 6)
mkdir -p gpurun_out/r3ad
sed -i 's/^for method in (1,):/for method in (1, 0):/' scratch/roots_stats.py
for n in oldnewton cur sch66 sch47 sch38; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  echo "== $n" | tee -a gpurun_out/r3ad/roots.log
  DRANSAC_LIB=$lib timeout 120 python scratch/roots_stats.py 2>&1 | tail -4 | tee -a gpurun_out/r3ad/roots.log
done
DRANSAC_LIB=$PWD/scratch/libdransac_oldnewton.so timeout 120 python scratch/k3_ab.py gpurun_out/r3ad/old.npz 2>&1 | grep K3 | tee gpurun_out/r3ad/k3.log
for n in cur sch66 sch47 sch38; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  echo "== $n" | tee -a gpurun_out/r3ad/k3.log
  DRANSAC_LIB=$lib timeout 120 python scratch/k3_ab.py gpurun_out/r3ad/$n.npz 2>&1 | grep K3 | tee -a gpurun_out/r3ad/k3.log
  python scratch/k3_ab.py cmp gpurun_out/r3ad/old.npz gpurun_out/r3ad/$n.npz 2>&1 | grep -v "^a ms" | tee -a gpurun_out/r3ad/k3.log
done
rm -f gpurun_out/r3ad/*.npz
