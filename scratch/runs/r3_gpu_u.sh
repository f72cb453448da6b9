#!/bin/bash
# round 3, GPU pass U: K3 with the three-product Jacobian in the Gauss-Newton step ("k3jac") against the tree ("cur")
mkdir -p gpurun_out/r3u
python scratch/k3_ab.py gpurun_out/r3u/cur.npz 2>&1 | grep K3 | tee gpurun_out/r3u/k3.log
for n in k3jac k3jac2 k3jac3; do
  echo "== $n" | tee -a gpurun_out/r3u/k3.log
  DRANSAC_LIB=$PWD/scratch/libdransac_$n.so python scratch/k3_ab.py gpurun_out/r3u/$n.npz 2>&1 | grep K3 | tee -a gpurun_out/r3u/k3.log
  python scratch/k3_ab.py cmp gpurun_out/r3u/cur.npz gpurun_out/r3u/$n.npz 2>&1 | tee -a gpurun_out/r3u/k3.log
done
rm -f gpurun_out/r3u/*.npz
AB_ARGS="--segments 3 --prewarm-s 0.3" bash scratch/ab_step.sh cur k3jac3 2>&1 | tee gpurun_out/r3u/ab.log
