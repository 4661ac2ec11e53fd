"""Shared builders for the test-suite (oracle-side and HIP-side models from the same golden weights)."""
import json, os
import numpy as np
import torch
from conftest import GOLDEN, load_golden_weights, load_npz


def product_config(cfgd, **over):
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    kw = dict(vocab_size=cfgd['vocab_size'], hidden_size=cfgd['hidden_size'], intermediate_size=cfgd['intermediate_size'],
              num_hidden_layers=cfgd['num_hidden_layers'], num_attention_heads=cfgd['num_attention_heads'],
              num_key_value_heads=cfgd['num_key_value_heads'], rope_theta=cfgd['rope_theta'], rms_norm_eps=cfgd['rms_norm_eps'],
              vit_hidden_size=cfgd['vit_hidden_size'], vit_intermediate_size=cfgd['vit_intermediate_size'],
              vit_num_hidden_layers=cfgd['vit_layers'] + 1, vit_layers_removed=1, vit_num_attention_heads=cfgd['vit_heads'],
              vit_image_size=cfgd['vit_image_size'], vit_patch_size=cfgd['vit_patch_size'],
              video_pooling_stride=cfgd['video_pooling_stride'], mm_spatial_pool_mode=cfgd['mm_spatial_pool_mode'],
              frame_num_tokens=cfgd['frame_num_tokens'], frame_resolution=cfgd['frame_resolution'], v_placeholder='<image>')
    kw.update(over)
    return VideoHeadLiveLlavaQwenConfig(**kw)


def make_args(**over):
    from mmduet_amd.arguments_live import LiveTestArguments
    a = LiveTestArguments(llm_pretrained='synthetic:0', frame_fps=1.0, system_prompt='A tiny assistant.')
    for k, v in over.items():
        setattr(a, k, v)
    return a


def oracle_model(tag, dtype=torch.float32, **cfg_over):
    from oracle import duet_oracle as O
    cfgd, w = load_golden_weights(tag)
    cfg = O.OracleConfig(**{**cfgd, **cfg_over})
    return O.OracleModel(cfg, {k: v.to(dtype) for k, v in w.items()}), cfgd, w


def tokenizer_for(config):
    from mmduet_amd.tokenization_live import build_live_tokenizer_and_update_config
    return build_live_tokenizer_and_update_config('synthetic:tiny', config)


def hip_model(tag, dtype=torch.float32, weights=None, **cfg_over):
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    cfgd, w = load_golden_weights(tag)
    if weights is not None:
        w = weights
    config = product_config(cfgd, **cfg_over)
    m = VideoHeadLiveLlavaQwenForCausalLM(config, torch_dtype=dtype, max_vit_batch=8, max_step_tokens=512, kv_initial_tokens=512)
    m.load_state_dict(w)
    return m, cfgd, w


def stream_cases():
    return json.load(open(os.path.join(GOLDEN, 'cfgA_streams.json')))


def stream_frames(name):
    return torch.from_numpy(np.load(os.path.join(GOLDEN, f'stream_{name}_frames.npy')))


def run_stream_case(driver_cls, model, name, case, meta, frames_per_forward=1, dtype=torch.float32):
    """Run one golden stream case through the product driver around `model`; returns the driver."""
    opts = case['opts']
    args = make_args(frame_fps=case['fps'], system_prompt=meta['system_prompt'], max_new_tokens=12,
                     stream_end_prob_threshold=opts.get('stream_end_prob_threshold'),
                     stream_end_score_sum_threshold=opts.get('stream_end_score_sum_threshold'),
                     score_heads=opts.get('score_heads', 'informative_score'),
                     remove_assistant_turns=opts.get('remove_assistant_turns', False),
                     repetition_penalty=opts.get('repetition_penalty'),
                     running_list_length=opts.get('running_list_length', 20),
                     frames_per_forward=frames_per_forward, bf16=(dtype == torch.bfloat16))
    tok = tokenizer_for(model.config)
    model.config.eos_token_id = meta['eos_token_id']      # the fixture's synthetic "eos" (see make_golden.py)
    d = driver_cls(args, model=model, tokenizer=tok)
    d.input_video_stream(stream_frames(name))
    d.input_query_stream(case['conversation'])
    d.responses = d.inference()
    return d
