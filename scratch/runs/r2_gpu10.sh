#!/bin/bash
# memory-path counters of the general scoring kernel at 128 pairs (and of a same-size fill kernel for comparison)
mkdir -p gpurun_out/r2n
O=$GRAFT_REPO_ROOT/gpurun_out/r2n
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -oE "(TCC|TCP|TA|TD|SQ|GRBM|SPI)_[A-Z0-9_]+" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
grep -E "WRREQ|WRITE|STALL|BUSY|VMEM|PENDING|TA_|ATOMIC" $O/counter_names.txt | tr '\n' ' ' | head -c 6000; echo
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES" \
           "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY TCC_REQ" \
           "TCC_EA0_WRREQ_64B TCC_WRITE TCC_TAG_STALL TCC_HIT" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_GATE_EN1 TCP_TA_TCP_STATE_READ" \
           "TA_BUSY TA_TA_BUSY TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p$i -- python $R/scratch/k4_general_run.py > $O/p$i.log 2>&1 || echo "set $i failed: $(tail -2 $O/p$i.log | head -c 300)"
done
cd $R
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for name, counter, n, avg in rows:
        if 'msac_score' in name or 'FillFunctor' in name or 'fill' in name.lower():
            print(f"{name.split('(')[0][-44:]:44s} {counter:34s} n={n} avg={avg:.5g}")
PY
rm -rf $O/p*/
