"""Checkpoint -> native weight loading (host-side; replaces `from_pretrained` + `PeftModel.from_pretrained`,
models/modeling_live.py:96-99,123).

  * `load_pretrained_into`  reads HF safetensors shards of a LLaVA-OneVision-Qwen2 checkpoint and hands every tensor
    on the path to `mmd_load_tensor` under its checkpoint name (the native side fuses q/k/v, gate/up, pads, ...).
  * `load_lora_into`        merges a peft LoRA adapter (r=16, alpha=32 by default, targets q/k/v/o/gate/up/down,
    models/arguments_live.py:13-15) into the base matrices in fp32: W += (alpha/r) * B @ A, and overwrites the
    `modules_to_save` tensors (mm_projector, heads).
  * `synthetic_weights`     seeded random weights at the configured shape (no checkpoints exist offline): N(0, 0.02)
    matrices, unit norm gains, zero biases or the O(1)-activation 'unit' scaling used by parity tests.
"""
import json
import math
import os
import re
import torch

VT = 'model.vision_tower.vision_tower.vision_model.'


def expected_tensors(config):
    """name -> shape of every tensor the native model consumes."""
    H, I, V = config.hidden_size, config.intermediate_size, config.vocab_size
    d, nh, nkv = config.head_dim, config.num_attention_heads, config.num_key_value_heads
    s = {'model.embed_tokens.weight': (V, H), 'model.norm.weight': (H,), 'lm_head.weight': (V, H),
         'informative_head.weight': (2, H), 'relevance_head.weight': (2, H)}
    for i in range(config.num_hidden_layers):
        p = f'model.layers.{i}.'
        s[p + 'input_layernorm.weight'] = (H,)
        s[p + 'post_attention_layernorm.weight'] = (H,)
        for nm, rows in (('q_proj', nh * d), ('k_proj', nkv * d), ('v_proj', nkv * d)):
            s[p + f'self_attn.{nm}.weight'] = (rows, H)
            s[p + f'self_attn.{nm}.bias'] = (rows,)
        s[p + 'self_attn.o_proj.weight'] = (H, nh * d)
        s[p + 'mlp.gate_proj.weight'] = (I, H)
        s[p + 'mlp.up_proj.weight'] = (I, H)
        s[p + 'mlp.down_proj.weight'] = (H, I)
    C, CI, P = config.vit_hidden_size, config.vit_intermediate_size, config.vit_patch_size
    s[VT + 'embeddings.patch_embedding.weight'] = (C, 3, P, P)
    s[VT + 'embeddings.patch_embedding.bias'] = (C,)
    s[VT + 'embeddings.position_embedding.weight'] = (config.vit_grid ** 2, C)
    for i in range(config.vit_layers_run):
        p = VT + f'encoder.layers.{i}.'
        for ln in ('layer_norm1', 'layer_norm2'):
            s[p + ln + '.weight'] = (C,)
            s[p + ln + '.bias'] = (C,)
        for lin in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            s[p + f'self_attn.{lin}.weight'] = (C, C)
            s[p + f'self_attn.{lin}.bias'] = (C,)
        s[p + 'mlp.fc1.weight'] = (CI, C); s[p + 'mlp.fc1.bias'] = (CI,)
        s[p + 'mlp.fc2.weight'] = (C, CI); s[p + 'mlp.fc2.bias'] = (C,)
    if config.vit_post_layernorm:
        s[VT + 'post_layernorm.weight'] = (C,); s[VT + 'post_layernorm.bias'] = (C,)
    s['model.mm_projector.0.weight'] = (H, C); s['model.mm_projector.0.bias'] = (H,)
    s['model.mm_projector.2.weight'] = (H, H); s['model.mm_projector.2.bias'] = (H,)
    return s


def synthetic_weights(config, seed=0, device='cuda', dtype=torch.bfloat16, scale='init02'):
    """Yield (name, tensor) with a per-tensor seeded generator so the stream is order-independent."""
    names = expected_tensors(config)
    for idx, (name, shape) in enumerate(names.items()):
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + idx)
        if scale == 'init02':
            if len(shape) >= 2:
                t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
            elif name.endswith('weight'):
                t = torch.ones(shape, device=device)
            else:
                t = torch.zeros(shape, device=device)
        else:
            if len(shape) >= 2:
                fan_in = math.prod(shape[1:])
                t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * (0.5 if 'embed' in name else 0.7 / math.sqrt(fan_in))
            elif 'norm' in name and name.endswith('weight'):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
            else:
                t = 0.1 * torch.randn(shape, generator=g, device=device)
        yield name, t.to(dtype)


def _keep(name, config):
    """Is this checkpoint tensor on the path?"""
    if name in ('model.image_newline',):
        return False
    m = re.match(r'model\.vision_tower\.vision_tower\.(vision_model\.)?encoder\.layers\.(\d+)\.', name)
    if m and int(m.group(2)) >= config.vit_layers_run:
        return False          # LLaVA deletes the last encoder layer(s)
    if '.vision_model.head.' in name or '.vision_tower.head.' in name:
        return False          # SigLIP pooling head is replaced by Identity
    if 'post_layernorm' in name and not config.vit_post_layernorm:
        return False
    return True


def _resolve_dir(path_or_id):
    if os.path.isdir(path_or_id):
        return path_or_id
    from huggingface_hub import snapshot_download     # offline boxes: must already be in the local cache
    return snapshot_download(path_or_id, allow_patterns=['*.safetensors', '*.json'])


def load_pretrained_into(model, path_or_id):
    from safetensors import safe_open
    d = _resolve_dir(path_or_id)
    files = sorted(f for f in os.listdir(d) if f.endswith('.safetensors') and not f.startswith('adapter'))
    if not files:
        raise FileNotFoundError(f'no *.safetensors under {d}')
    need = set(expected_tensors(model.config))
    seen = set()
    for f in files:
        with safe_open(os.path.join(d, f), framework='pt', device='cpu') as sf:
            for name in sf.keys():
                if not _keep(name, model.config):
                    continue
                canon = name.replace('model.vision_tower.vision_tower.encoder', VT + 'encoder').replace(
                    'model.vision_tower.vision_tower.embeddings', VT + 'embeddings').replace(
                    'model.vision_tower.vision_tower.post_layernorm', VT + 'post_layernorm')
                if canon not in need:
                    continue
                model.load_tensor(canon, sf.get_tensor(name))
                seen.add(canon)
    # heads absent from a plain LLaVA-OV checkpoint arrive with the LoRA adapter's modules_to_save (or stay missing
    # and finalize() reports them)
    return sorted(need - seen)


def load_lora_into(model, lora_dir):
    """peft adapter: `base_model.model.<module>.lora_A[.default].weight` [r,in], `.lora_B[.default].weight` [out,r];
    scale = lora_alpha / r from adapter_config.json; modules_to_save tensors replace the base ones."""
    from safetensors import safe_open
    d = _resolve_dir(lora_dir)
    cfg = json.load(open(os.path.join(d, 'adapter_config.json')))
    scale = cfg['lora_alpha'] / cfg['r']
    path = os.path.join(d, 'adapter_model.safetensors')
    A, B = {}, {}
    need = set(expected_tensors(model.config))
    with safe_open(path, framework='pt', device='cpu') as sf:
        for key in sf.keys():
            name = key[len('base_model.model.'):] if key.startswith('base_model.model.') else key
            name = name.replace('.default.', '.')
            if '.lora_A.' in name:
                A[name.replace('.lora_A.weight', '.weight')] = sf.get_tensor(key)
            elif '.lora_B.' in name:
                B[name.replace('.lora_B.weight', '.weight')] = sf.get_tensor(key)
            elif '.original_module.' in name:
                continue
            else:
                name = name.replace('.modules_to_save', '')
                if name in need:
                    model.load_tensor(name, sf.get_tensor(key))
    for w in sorted(A):
        if w not in B:
            raise KeyError(f'LoRA adapter has lora_A but no lora_B for {w}')
        model.merge_lora(w, A[w], B[w], scale)
    return len(A)


def save_checkpoint(named_tensors, out_dir, config=None):
    """Write a HF-layout checkpoint (used by tests to round-trip the loader)."""
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    save_file({k: v.detach().cpu().contiguous() for k, v in named_tensors.items()}, os.path.join(out_dir, 'model.safetensors'))
    if config is not None:
        config.save_pretrained(out_dir)
