#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from two tools/pmc_summary.py outputs (FETCH_SIZE pass, WRITE_SIZE pass) of the same workload.

    python tools/pmc_traffic.py profiles/r01_final_pmc_FETCH_SIZE.txt profiles/r01_final_pmc_WRITE_SIZE.txt > profiles/r01_pmc_traffic.json

HBM bytes per launch of each kernel class bench.py reports (`mmd_prof_*` classes): counters are in KB; FETCH_SIZE is doubled (gfx950
tallies the 128-B requests of 16-B/lane streaming reads at 64 B -- MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated.
A class launch = one main kernel; helper kernels of the class (split-K reduce, attention combine) add bytes, not launches."""
import json, re, sys

CLASSES = {     # class -> (main kernel prefixes, helper kernel prefixes)
    'gemm_tile': (('void gemm_ringx_kernel', 'void gemm_ring256_kernel', 'void gemm_big_kernel', 'void gemm_tile_kernel'), ('void splitk_reduce_kernel',)),
    'gemm_skinny': (('void gemm_gemv16_kernel', 'void gemm_skinny_kernel'), ()),
    'attn_llm': (('void attn_gqa128_kernel', 'void attn_mfma_kernel'), ('attn_combine128_kernel', 'attn_combine_kernel')),
    'attn_vit': (('void attn_rowmajor_kernel',), ()),
}


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(' '):
            name = line.strip()
        else:
            m = re.match(r'\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)', line)
            if m and m.group(1) == counter:
                out[name] = (int(m.group(2)), float(m.group(3)))
    return out


def main(fetch_path, write_path):
    f, w = parse(fetch_path, 'FETCH_SIZE'), parse(write_path, 'WRITE_SIZE')
    res = {}
    for cls, (mains, helpers) in CLASSES.items():
        n = sum(c for k, (c, _) in f.items() if k.startswith(mains))
        if not n:
            continue
        fb = sum(c * a for k, (c, a) in f.items() if k.startswith(mains + helpers)) * 1024 * 2
        wb = sum(c * a for k, (c, a) in w.items() if k.startswith(mains + helpers)) * 1024
        res[cls] = dict(launches=n, fetch_bytes_per_launch=fb / n, write_bytes_per_launch=wb / n, hbm_bytes_per_launch=(fb + wb) / n)
    res['_note'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 0 --no-prof --no-overlap '
                    '--multi-stream 0`; FETCH_SIZE doubled per the MI355X guide (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams); '
                    'WRITE_SIZE uncalibrated; KB -> bytes; tools/pmc_traffic.py')
    json.dump(res, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
