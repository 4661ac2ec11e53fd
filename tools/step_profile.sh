#!/bin/bash
# rocprofv3 kernel trace of LLM steps of S rows at 15 k context (tools/step_classes_probe.py) -> gpurun_out/<tag>_step<S>_kernels.txt
tag=${1:-r04}; S=${2:-49}; W=${3:-bf16}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_step
rocprofv3 --kernel-trace -d $O/prof_step -o trace -- python3 $R/tools/step_classes_probe.py 15000 $W $S > $O/${tag}_step${S}_probe.log 2>&1
db=$(ls $O/prof_step/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 40 > $O/${tag}_step${S}_${W}_kernels.txt
rm -rf $O/prof_step
tail -3 $O/${tag}_step${S}_probe.log
