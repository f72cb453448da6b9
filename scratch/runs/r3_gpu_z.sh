#!/bin/bash
# round 3, GPU pass Z: K3 time with and without the final stage, derivative chain vs Sturm isolation
for n in k3nosturm skipf_s0 cur skipf_s1; do
  lib=""; [ "$n" != "cur" ] && lib=$PWD/scratch/libdransac_$n.so
  echo "== $n"; DRANSAC_LIB=$lib timeout 120 python scratch/k3_ab.py /tmp/x.npz 2>&1 | grep "K3 ms"
done
