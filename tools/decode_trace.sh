#!/bin/bash
# rocprofv3 kernel trace of tools/decode_probe.py (64 tokens at a context): per-kernel table into gpurun_out/<tag>_decode_kernel_stats.txt
#   tools/decode_trace.sh <tag> [context=15000] [weights=bf16]      (environment switches such as MMDUET_ATTN_DECODE_RING=0 pass through)
tag=${1:-decode}; ctx=${2:-15000}; w=${3:-bf16}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$tag
rocprofv3 --kernel-trace -d $O/prof_$tag -o trace -- python3 $R/tools/decode_probe.py 64 $ctx $w > $O/${tag}_decode.log 2>&1
db=$(ls $O/prof_$tag/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 16 > $O/${tag}_decode_kernel_stats.txt
rm -rf $O/prof_$tag
tail -3 $O/${tag}_decode.log; head -14 $O/${tag}_decode_kernel_stats.txt | cut -c1-150
