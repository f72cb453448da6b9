#!/bin/bash
# round 3, GPU pass S: K3 -- residual cached for the verification ("k3pre2"), and shorter bisection / Newton schedules of the
# root search (k3a..k3e = BIS_LOW NEWT_LOW BIS_LAST NEWT_LAST: 3 4 10 6 | 4 3 10 6 | 6 4 6 6 | 3 4 5 6 | 2 5 4 7)
mkdir -p gpurun_out/r3s
python scratch/k3_ab.py gpurun_out/r3s/cur.npz 2>&1 | grep K3 | tee gpurun_out/r3s/k3.log
for n in k3pre k3pre2 k3a k3b k3c k3d k3e; do
  echo "== $n" | tee -a gpurun_out/r3s/k3.log
  DRANSAC_LIB=$PWD/scratch/libdransac_$n.so python scratch/k3_ab.py gpurun_out/r3s/$n.npz 2>&1 | grep K3 | tee -a gpurun_out/r3s/k3.log
  python scratch/k3_ab.py cmp gpurun_out/r3s/cur.npz gpurun_out/r3s/$n.npz 2>&1 | tee -a gpurun_out/r3s/k3.log
done
rm -f gpurun_out/r3s/*.npz
