#!/usr/bin/env python
"""Hidden-state error of a 1274-row chunk through L decoder layers at the true width: fp8 build vs bf16 build, each against the fp32 oracle on ITS weights (the fp8 build:
dequantised), next to the bf16 oracle's own distance -- where does the fp8 build's extra mean error come from?     python tools/probes/fp8_layer_probe.py [scale]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from oracle import duet_oracle as O
from oracle.stream_check import dequantised_fp8
from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
from mmduet_amd.weights import synthetic_weights
scale = sys.argv[1] if len(sys.argv) > 1 else 'init02'
dev = torch.device('cuda', 0)
def rms(a, b): return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()
for L in (1, 4):
    for S in (1274, 49):
        for wd in (None, 'fp8_e4m3'):
            pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=2048, num_hidden_layers=L, vit_num_hidden_layers=2, vit_layers_removed=1, frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>')
            pcfg.weight_dtype = wd
            ocfg = O.OracleConfig(vocab_size=2048, num_hidden_layers=L, vit_layers=1)
            m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=torch.bfloat16, max_vit_batch=2, max_step_tokens=1536, kv_initial_tokens=4096)
            w = {}
            for name, t in synthetic_weights(pcfg, seed=3, device=dev, dtype=torch.bfloat16, scale=scale):
                m.load_tensor(name, t); w[name] = t
            m.finalize()
            w32 = dequantised_fp8(w) if wd else {k: v.float() for k, v in w.items()}
            w16 = {k: v.to(torch.bfloat16) for k, v in w32.items()}
            g = torch.Generator(device=dev).manual_seed(2)
            prompt = (torch.randn(1, 29, 3584, generator=g, device=dev) * 0.5).to(torch.bfloat16)
            x = (torch.randn(1, S, 3584, generator=g, device=dev) * 0.5).to(torch.bfloat16)
            base = m(inputs_embeds=prompt).past_key_values
            h = m(inputs_embeds=x, past_key_values=base).hidden_states[0]
            _, c32 = O.llm_forward(w32, ocfg, prompt[0].float(), None)
            h32, _ = O.llm_forward(w32, ocfg, x[0].float(), c32)
            _, c16 = O.llm_forward(w16, ocfg, prompt[0], None)
            h16, _ = O.llm_forward(w16, ocfg, x[0], c16)
            print(f'LAYERS {L} rows {S} weights {wd or "bf16":8s} ({scale}): ours vs fp32 {rms(h, h32):.4e}   bf16 oracle vs fp32 {rms(h16, h32):.4e}   ratio {rms(h, h32) / rms(h16, h32):.3f}', flush=True)
            del m, w, w32, w16
            torch.cuda.empty_cache()
