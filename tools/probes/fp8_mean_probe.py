#!/usr/bin/env python
"""Where does the fp8 build's extra MEAN head-logit error come from (youcook2 0.0301 against the bf16 oracle's 0.0241)?
A: bf16 weights on the youcook2 schedule (remove_assistant_turns + KV stash, 12 responses) -- isolates the stash path.
B: fp8 weights on the ground600 schedule (no responses) -- isolates the fp8 GEMMs.      python tools/probes/fp8_mean_probe.py"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_gpu_fullsize as T

def run(tag, weights, cfgname):
    model, tok, w = T._build_product(weights, 600)
    if weights == 'fp8':
        w32 = T._dequantised(w); w16 = {k: v.to(torch.bfloat16) for k, v in w32.items()}
    else:
        w32 = {k: v.float() for k, v in w.items()}; w16 = w
    try:
        r = T._case(cfgname, model, tok, w32, w16, e2e=False)
    except AssertionError as e:
        r = json.load(open(os.path.join(R, 'gpurun_out', 'parity_full_size.json')))[cfgname]
    ls = r['llm_side']
    print(f"PROBE {tag}: ours max {ls['ours_vs_fp32']:.4f} mean {ls['ours_vs_fp32_mean']:.5f} | bf16 oracle max {ls['bf16_oracle_vs_fp32']:.4f} mean {ls['bf16_oracle_vs_fp32_mean']:.5f}", flush=True)
    del model, w, w32, w16
    torch.cuda.empty_cache()

run('A bf16 weights, youcook2 schedule', 'bf16', 'youcook2')
run('B fp8 weights, ground600 schedule', 'fp8', 'ground600')
