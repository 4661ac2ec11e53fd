#!/bin/bash
# tower (MFMA-bound) under a decode burst (HBM-bound) -- 64 decoded tokens at 7.6 k context + 8 tower batches of 35 frames, wall time per form of the tower GEMMs.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/r04_overlap_sweep.txt; : > $out
run() { tag="$1"; shift; echo "== $tag" >> $out; env "$@" python3 $R/tools/probes/overlap_probe.py 8 2>/dev/null | tail -1 >> $out; }
if [ "$1" != more ]; then
run "8-wave persistent ring, all 256 CUs (one block per CU)" MMDUET_TOWER_RING=16
run "8-wave persistent ring capped at 128 blocks (the shipped burst form: chip split in space)" MMDUET_TOWER_RING=16 MMDUET_TOWER_RING_BLOCKS=128
run "8-wave ring, NON-persistent (one tile per block, lowest stream priority: decode kernels enter at tile ends)" MMDUET_TOWER_RING=16 MMDUET_TOWER_RING_BLOCKS=-1
fi
# (the fp16 tower always runs the shipped 8-wave instantiation: the 4-wave forms need the bf16 tower)
run "bf16 tower, 8-wave persistent ring, all CUs (reference for the 4-wave rows)" OVERLAP_TOWER_DTYPE=bf16 MMDUET_TOWER_RING=16
run "bf16 tower, 4-wave 256x128 ring, one block per CU (256 blocks: 1 wave per SIMD, ~256 registers per SIMD left to the GEMV waves)" OVERLAP_TOWER_DTYPE=bf16 MMDUET_TOWER_RING=17 MMDUET_TOWER_RING_BLOCKS=256
run "bf16 tower, 4-wave 256x128 ring, two blocks per CU (512 blocks)" OVERLAP_TOWER_DTYPE=bf16 MMDUET_TOWER_RING=17
run "decode on a HIGH-priority stream; 8-wave persistent ring, all CUs" MMDUET_TOWER_RING=16 OVERLAP_DECODE_PRIO=-1
run "decode on a HIGH-priority stream; 8-wave ring capped at 128 blocks" MMDUET_TOWER_RING=16 MMDUET_TOWER_RING_BLOCKS=128 OVERLAP_DECODE_PRIO=-1
run "decode on a HIGH-priority stream; 8-wave ring NON-persistent" MMDUET_TOWER_RING=16 MMDUET_TOWER_RING_BLOCKS=-1 OVERLAP_DECODE_PRIO=-1
cat $out
