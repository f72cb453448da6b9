#!/bin/bash
# round-2 final measurements after the K3 work: default bench line, train line, rocprofv3 kernel stats of c2 (128 and 32 pairs), c3, train
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; head -c 900 $O/bench_default.json; echo
timeout 300 python bench.py --mode train --steps 300 > $O/bench_train.json 2> $O/bench_train.err; head -c 400 $O/bench_train.json; echo
timeout 300 python bench.py --pairs 32 --steps 500 --no-configs --no-cpu-baseline > $O/bench_p32.json 2> $O/bench_p32.err; head -c 300 $O/bench_p32.json; echo
cd /tmp
prof() {  # name, title, bench args...
  local name=$1; local title=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o bench -- python $R/bench.py "$@" > $O/prof_$name.json 2> $O/prof_$name.err
  python $R/tools/rocprof_summary.py $(find $O/prof_$name -name "*results.db" | head -1) $O/r2_kernel_stats_$name.md "$title" first 105
  rm -rf $O/prof_$name
}
prof c2 "python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-extras   (default workload: c2, 128 pairs per step)" --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-extras
prof c2_p32 "python bench.py --workload c2 --pairs 32 --steps 100 --warmup 5 --no-configs --no-cpu-baseline" --workload c2 --pairs 32 --steps 100 --warmup 5 --no-configs --no-cpu-baseline
prof c3 "python bench.py --workload c3 --steps 100 --warmup 5 --no-configs --no-cpu-baseline" --workload c3 --steps 100 --warmup 5 --no-configs --no-cpu-baseline
prof c4 "python bench.py --workload c4 --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-extras" --workload c4 --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-extras
prof train "python bench.py --mode train --graph off --steps 100 --warmup 5   (32 pairs per step, eager so that the launches are visible one by one)" --mode train --graph off --steps 100 --warmup 5
ls $O
