"""CPU restatement of the secondary frame encoders.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows models/vision_live.py:11-64 (`_siglip_vision_encode`, `_clip_vision_encode`, `build_live_vision`) with the tower arithmetic of the
HF vision models they call written out in plain torch ops ([3P] = installed transformers 5.15.0):
  SiglipVisionModel   siglip/modeling_siglip.py: embeddings (conv patch embed + learned positions), pre-norm encoder layers
                      (gelu_pytorch_tanh), post_layernorm on the sequence, SiglipMultiheadAttentionPoolingHead -> pooler_output
  CLIPVisionModel     clip/modeling_clip.py: class embedding + bias-free conv patch embed + positions, pre_layrnorm, pre-norm
                      encoder layers (quick_gelu), last_hidden_state WITHOUT post_layernorm
Pinned against outputs of the reference functions themselves: tests/golden/vision_live.npz (tests/golden/make_vision_golden.py).
Weights: flat dict with the HF state-dict names of the vision model (`embeddings.patch_embedding.weight`, `encoder.layers.N...`, `head...`).
"""
import math
import torch
import torch.nn.functional as F

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]      # transformers/utils/constants.py [3P]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def normalize(frames, mean, std, rescale_factor=0.00392156862745098):
    """torchvision.transforms.functional.normalize(frames * rescale_factor, mean, std) -- models/vision_live.py:13,36"""
    x = frames * rescale_factor
    mean = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(-1, 1, 1); std = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
    return (x - mean) / std


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _encoder(w, x, n_layers, n_heads, eps, act):
    """pre-norm encoder layers shared by SiglipEncoderLayer / CLIPEncoderLayer: LN1 -> MHSA(scale d^-0.5, biased q/k/v/out) -> +res -> LN2 -> fc1, act, fc2 -> +res"""
    B, N, C = x.shape
    hd = C // n_heads
    for i in range(n_layers):
        p = f'encoder.layers.{i}.'
        h = _ln(x, w[p + 'layer_norm1.weight'], w[p + 'layer_norm1.bias'], eps)
        q = F.linear(h, w[p + 'self_attn.q_proj.weight'], w[p + 'self_attn.q_proj.bias']).view(B, N, n_heads, hd).transpose(1, 2)
        k = F.linear(h, w[p + 'self_attn.k_proj.weight'], w[p + 'self_attn.k_proj.bias']).view(B, N, n_heads, hd).transpose(1, 2)
        v = F.linear(h, w[p + 'self_attn.v_proj.weight'], w[p + 'self_attn.v_proj.bias']).view(B, N, n_heads, hd).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, N, C)
        x = x + F.linear(o, w[p + 'self_attn.out_proj.weight'], w[p + 'self_attn.out_proj.bias'])
        h = _ln(x, w[p + 'layer_norm2.weight'], w[p + 'layer_norm2.bias'], eps)
        h = act(F.linear(h, w[p + 'mlp.fc1.weight'], w[p + 'mlp.fc1.bias']))
        x = x + F.linear(h, w[p + 'mlp.fc2.weight'], w[p + 'mlp.fc2.bias'])
    return x


def _patches(pixel_values, P):
    B, C, H, W = pixel_values.shape
    g = H // P
    return pixel_values[:, :, :g * P, :g * P].reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)


def gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def siglip_vision_model(w, cfg, pixel_values):
    """-> (last_hidden_state [B, g*g, C] after post_layernorm, pooler_output [B, C])   (SiglipVisionModel.forward [3P])"""
    P, eps, nh = cfg['patch_size'], cfg['layer_norm_eps'], cfg['num_attention_heads']
    x = F.linear(_patches(pixel_values, P), w['embeddings.patch_embedding.weight'].flatten(1), w['embeddings.patch_embedding.bias'])
    x = x + w['embeddings.position_embedding.weight'][None]
    x = _encoder(w, x, cfg['num_hidden_layers'], nh, eps, gelu_tanh)
    x = _ln(x, w['post_layernorm.weight'], w['post_layernorm.bias'], eps)
    # SiglipMultiheadAttentionPoolingHead: nn.MultiheadAttention(probe, x, x) -> residual + mlp(layernorm(.)) -> [:, 0]
    B, N, C = x.shape
    hd = C // nh
    wi, bi = w['head.attention.in_proj_weight'], w['head.attention.in_proj_bias']
    q = F.linear(w['head.probe'].reshape(1, 1, C).expand(B, 1, C), wi[:C], bi[:C]).view(B, 1, nh, hd).transpose(1, 2)
    k = F.linear(x, wi[C:2 * C], bi[C:2 * C]).view(B, N, nh, hd).transpose(1, 2)
    v = F.linear(x, wi[2 * C:], bi[2 * C:]).view(B, N, nh, hd).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5, dim=-1)
    h = F.linear(torch.matmul(a, v).transpose(1, 2).reshape(B, 1, C), w['head.attention.out_proj.weight'], w['head.attention.out_proj.bias'])
    r = h
    h = _ln(h, w['head.layernorm.weight'], w['head.layernorm.bias'], eps)
    h = r + F.linear(gelu_tanh(F.linear(h, w['head.mlp.fc1.weight'], w['head.mlp.fc1.bias'])), w['head.mlp.fc2.weight'], w['head.mlp.fc2.bias'])
    return x, h[:, 0]


def clip_vision_model(w, cfg, pixel_values):
    """-> last_hidden_state [B, 1 + g*g, C] (class token first, no post_layernorm)   (CLIPVisionModel.forward [3P])"""
    P, eps = cfg['patch_size'], cfg['layer_norm_eps']
    x = F.linear(_patches(pixel_values, P), w['embeddings.patch_embedding.weight'].flatten(1))
    x = torch.cat([w['embeddings.class_embedding'].reshape(1, 1, -1).expand(x.shape[0], 1, -1), x], 1) + w['embeddings.position_embedding.weight'][None]
    x = _ln(x, w['pre_layrnorm.weight'], w['pre_layrnorm.bias'], eps)
    return _encoder(w, x, cfg['num_hidden_layers'], cfg['num_attention_heads'], eps, quick_gelu)


def _adaptive_tokens(x, out_hw):
    """adaptive_avg_pool2d of the s x s token grid (models/vision_live.py:17-24): bin i covers [floor(i*s/o), ceil((i+1)*s/o))"""
    from .duet_oracle import adaptive_avg_pool_tokens
    return adaptive_avg_pool_tokens(x, out_hw)


def siglip_vision_encode(w, cfg, frames, frame_token_cls, frame_token_pooled, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """models/vision_live.py:11-31"""
    last, pooled = siglip_vision_model(w, cfg, normalize(frames, mean, std))
    if frame_token_pooled:
        spatial = _adaptive_tokens(last, tuple(frame_token_pooled))
        if not frame_token_cls:
            return spatial
    if frame_token_cls:
        cls = pooled[:, None]
        if not frame_token_pooled:
            return cls
    return torch.cat([cls, spatial], dim=1)


def clip_vision_encode(w, cfg, frames, frame_token_cls, frame_token_pooled, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD):
    """models/vision_live.py:34-54.  With cls AND pooled the reference concatenates a [B,C] with a [B,hw,C] tensor and raises; the only
    consistent reading (and what the SigLIP twin does) is cls[:, None] -- used here and marked as a deviation in the product."""
    last = clip_vision_model(w, cfg, normalize(frames, mean, std))
    if frame_token_pooled:
        spatial = _adaptive_tokens(last[:, 1:], tuple(frame_token_pooled))
        if not frame_token_cls:
            return spatial
    if frame_token_cls:
        cls = last[:, 0]
        if not frame_token_pooled:
            return cls
    return torch.cat([cls[:, None], spatial], dim=1)
