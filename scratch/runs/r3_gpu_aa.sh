#!/bin/bash
# round 3, GPU pass AA: stopping tolerance of the f32 Gauss-Newton polish (1e-17 tree, 1e-16, 1e-15)
mkdir -p gpurun_out/r3aa
timeout 120 python scratch/k3_ab.py gpurun_out/r3aa/cur.npz 2>&1 | grep K3 | tee gpurun_out/r3aa/k3.log
for n in tol1e-16 tol1e-15; do
  echo "== $n" | tee -a gpurun_out/r3aa/k3.log
  DRANSAC_LIB=$PWD/scratch/libdransac_$n.so timeout 120 python scratch/k3_ab.py gpurun_out/r3aa/$n.npz 2>&1 | grep K3 | tee -a gpurun_out/r3aa/k3.log
  python scratch/k3_ab.py cmp gpurun_out/r3aa/cur.npz gpurun_out/r3aa/$n.npz 2>&1 | grep -v "^a ms" | tee -a gpurun_out/r3aa/k3.log
done
rm -f gpurun_out/r3aa/*.npz
