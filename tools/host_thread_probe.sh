#!/bin/bash
# the ~0.7 busy core a rank burns beside its main thread (a spinning runtime thread).  Bench line under HIP / HSA runtime knobs -> frames/s, process CPU s per step, per-thread CPU.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/r04_host_thread_probe.txt; : > $out
run() {
  tag="$1"; shift
  line=$(env "$@" python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-parity-check 2>/dev/null | tail -1)
  echo "$tag $(echo $line | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("frames/s", d["value"], "host_cpu_s_per_step", d["host_cpu_s_per_step"], "threads", [(t["comm"], t["cpu_s"], t["thread"]) for t in d["host_threads_cpu_s_per_step"]])')" >> $out
}
if [ "$1" != flags ]; then
run "baseline                     " X=1
run "HSA_ENABLE_INTERRUPT=0       " HSA_ENABLE_INTERRUPT=0
run "HIP_FORCE_DEV_KERNARG=1      " HIP_FORCE_DEV_KERNARG=1
run "GPU_MAX_HW_QUEUES=2          " GPU_MAX_HW_QUEUES=2
run "AMD_DIRECT_DISPATCH=0        " AMD_DIRECT_DISPATCH=0
run "HIP_LAUNCH_BLOCKING=0 yield  " HIP_LAUNCH_BLOCKING=0 X=1
fi
[ "$1" != flags ] && cat $out
# which part of the run keeps the second thread busy?  (appended in round 4: bench flags instead of runtime knobs)
run2() {
  tag="$1"; shift
  line=$(python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-parity-check "$@" 2>/dev/null | tail -1)
  echo "$tag $(echo $line | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("frames/s", d["value"], "host_cpu_s_per_step", d["host_cpu_s_per_step"], "threads", [(t["comm"], t["cpu_s"], t["thread"]) for t in d["host_threads_cpu_s_per_step"]])')" >> $out
}
if [ "$1" = flags ]; then
  : > $out
  run2 "--no-prof                    " --no-prof
  run2 "--no-prof --no-overlap       " --no-prof --no-overlap
  run2 "--no-prof --responses 0      " --no-prof --responses 0
  run2 "--no-prof --host-sync blocking" --no-prof --host-sync blocking
  cat $out
fi
