"""CPU restatement (plain torch ops) of MMDuet's streaming forward path.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Every function cites the reference file:line (relative to /root/reference) or, where the arithmetic lives in a
third-party dependency of the reference, the installed transformers 5.15.0 source ([3P]) or the recalled LLaVA-NeXT
source ([3P-recalled]).

Weights are a flat dict name -> tensor using the checkpoint names of the reference model
(`model.layers.N.self_attn.q_proj.weight`, `model.vision_tower.vision_tower.vision_model.encoder.layers.N...`,
`model.mm_projector.0.weight`, `lm_head.weight`, `informative_head.weight`, `relevance_head.weight`).

Precision: all functions run in the dtype of the tensors they are given.  With float32 weights this is the fp32
reference; with bfloat16 weights every torch op rounds its result to bf16, which reproduces the rounding points of the
reference's eager bf16 execution (RMSNorm in fp32 then cast *before* the gain multiply, RoPE tables computed in fp32
then cast, softmax in fp32, heads `.float()` after a bf16 GEMM).
"""
from __future__ import annotations
import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional
import torch
import torch.nn.functional as F

VT = 'model.vision_tower.vision_tower.vision_model.'     # LLaVA-NeXT checkpoint prefix of the SigLIP tower [3P-recalled]


# ----------------------------------------------------------------------------------------------------------------
# configuration (shape only)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # Qwen2 decoder
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    rope_theta: float = 1e6
    rms_norm_eps: float = 1e-6
    # SigLIP tower (layers = number that RUN, i.e. after LLaVA deleted the last one)
    vit_hidden_size: int = 1152
    vit_intermediate_size: int = 4304
    vit_layers: int = 26
    vit_heads: int = 16
    vit_image_size: int = 384
    vit_patch_size: int = 14
    vit_layer_norm_eps: float = 1e-6
    vit_post_layernorm: bool = False
    # connector / pooling  (models/arguments_live.py:20-22, models/live_llava/video_head_live_llava_qwen.py:100-119)
    video_pooling_stride: int = 4
    mm_spatial_pool_mode: str = 'bilinear'
    frame_num_tokens: int = 49
    frame_resolution: int = 384
    v_placeholder: str = '<image>'
    v_placeholder_id: Optional[int] = None
    eos_token_id: Optional[int] = None

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def vit_grid(self):
        return self.vit_image_size // self.vit_patch_size

    @property
    def vit_tokens(self):
        return self.vit_grid ** 2


def tiny_config(**over) -> OracleConfig:
    """The tiny seeded configuration used by the golden fixtures (BASELINE.json configs[0] plumbing case)."""
    base = dict(vocab_size=512, hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6,
                vit_hidden_size=32, vit_intermediate_size=64, vit_layers=2, vit_heads=4, vit_image_size=56,
                vit_patch_size=14, video_pooling_stride=2, mm_spatial_pool_mode='bilinear', frame_num_tokens=4,
                frame_resolution=56)
    base.update(over)
    return OracleConfig(**base)


# ----------------------------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------------------------
def weight_shapes(cfg: OracleConfig) -> dict:
    """name -> shape for every tensor on the path (checkpoint layout: nn.Linear weights are [out, in])."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    hd = cfg.head_dim
    s = {'model.embed_tokens.weight': (V, H), 'model.norm.weight': (H,), 'lm_head.weight': (V, H),
         'informative_head.weight': (2, H), 'relevance_head.weight': (2, H)}
    for i in range(cfg.num_hidden_layers):
        p = f'model.layers.{i}.'
        s[p + 'input_layernorm.weight'] = (H,)
        s[p + 'post_attention_layernorm.weight'] = (H,)
        s[p + 'self_attn.q_proj.weight'] = (cfg.num_attention_heads * hd, H); s[p + 'self_attn.q_proj.bias'] = (cfg.num_attention_heads * hd,)
        s[p + 'self_attn.k_proj.weight'] = (cfg.num_key_value_heads * hd, H); s[p + 'self_attn.k_proj.bias'] = (cfg.num_key_value_heads * hd,)
        s[p + 'self_attn.v_proj.weight'] = (cfg.num_key_value_heads * hd, H); s[p + 'self_attn.v_proj.bias'] = (cfg.num_key_value_heads * hd,)
        s[p + 'self_attn.o_proj.weight'] = (H, cfg.num_attention_heads * hd)
        s[p + 'mlp.gate_proj.weight'] = (I, H); s[p + 'mlp.up_proj.weight'] = (I, H); s[p + 'mlp.down_proj.weight'] = (H, I)
    C, CI, P = cfg.vit_hidden_size, cfg.vit_intermediate_size, cfg.vit_patch_size
    s[VT + 'embeddings.patch_embedding.weight'] = (C, 3, P, P); s[VT + 'embeddings.patch_embedding.bias'] = (C,)
    s[VT + 'embeddings.position_embedding.weight'] = (cfg.vit_tokens, C)
    for i in range(cfg.vit_layers):
        p = VT + f'encoder.layers.{i}.'
        for ln in ('layer_norm1', 'layer_norm2'):
            s[p + ln + '.weight'] = (C,); s[p + ln + '.bias'] = (C,)
        for lin in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            s[p + f'self_attn.{lin}.weight'] = (C, C); s[p + f'self_attn.{lin}.bias'] = (C,)
        s[p + 'mlp.fc1.weight'] = (CI, C); s[p + 'mlp.fc1.bias'] = (CI,)
        s[p + 'mlp.fc2.weight'] = (C, CI); s[p + 'mlp.fc2.bias'] = (C,)
    s[VT + 'post_layernorm.weight'] = (C,); s[VT + 'post_layernorm.bias'] = (C,)
    s['model.mm_projector.0.weight'] = (H, C); s['model.mm_projector.0.bias'] = (H,)
    s['model.mm_projector.2.weight'] = (H, H); s['model.mm_projector.2.bias'] = (H,)
    return s


def random_weights(cfg: OracleConfig, seed=0, dtype=torch.float32, scale='unit') -> dict:
    """Seeded random weights.  scale='unit' keeps activations O(1) through the stack (variance-preserving matrices,
    gains near 1, small biases) so that parity tests exercise every term; scale='init02' is the N(0,0.02) /
    ones / zeros initialisation SURVEY.md §8(d) names for the true-shape benchmark."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shape in weight_shapes(cfg).items():
        if scale == 'init02':
            if len(shape) >= 2:
                t = torch.randn(shape, generator=g) * 0.02
            elif name.endswith('weight'):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
        else:
            if len(shape) >= 2:
                fan_in = math.prod(shape[1:])
                t = torch.randn(shape, generator=g) * (0.5 if 'embed' in name else 0.7 / math.sqrt(fan_in))
            elif 'norm' in name and name.endswith('weight'):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        w[name] = t.to(dtype)
    return w


def weights_from_reference_state_dict(sd: dict) -> dict:
    """Rename a state_dict of the reference model built by tests/golden/ref_harness.py (transformers 5.x SigLIP has no
    inner `vision_model.` level) to the checkpoint names used here."""
    out = {}
    for k, v in sd.items():
        if k.startswith('model.vision_tower.vision_tower.') and not k.startswith(VT):
            k = VT + k[len('model.vision_tower.vision_tower.'):]
        if '.head.' in k:
            continue
        out[k] = v.detach().clone()
    return out


# ----------------------------------------------------------------------------------------------------------------
# vision side
# ----------------------------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    # nn.LayerNorm (transformers siglip/modeling_siglip.py:329-331 [3P]); statistics in fp32 like torch's kernel
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    return (((xf - mu) * torch.rsqrt(var + eps)) * w.float() + b.float()).to(x.dtype)


def gelu_tanh(x):
    # ACT2FN['gelu_pytorch_tanh'] (SigLIP hidden_act) [3P]
    xf = x.float()
    return (0.5 * xf * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (xf + 0.044715 * xf ** 3)))).to(x.dtype)


def gelu_erf(x):
    # nn.GELU() in LLaVA's mlp2x_gelu projector [3P-recalled]
    xf = x.float()
    return (0.5 * xf * (1.0 + torch.erf(xf / math.sqrt(2.0)))).to(x.dtype)


def linear(x, w, b=None):
    return F.linear(x, w, b)


def vit_patch_embed(w, cfg: OracleConfig, pixel_values):
    """SiglipVisionEmbeddings.forward (siglip/modeling_siglip.py:175-186 [3P]): conv(k=stride=patch, valid) as an
    im2col GEMM with K ordered (channel, ky, kx) exactly as conv weight.flatten(1), + learned position embedding."""
    B, C, Himg, Wimg = pixel_values.shape
    P, g = cfg.vit_patch_size, cfg.vit_grid
    x = pixel_values.to(w[VT + 'embeddings.patch_embedding.weight'].dtype)
    x = x[:, :, :g * P, :g * P].reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    wt = w[VT + 'embeddings.patch_embedding.weight'].flatten(1)
    h = linear(x, wt, w[VT + 'embeddings.patch_embedding.bias'])
    return h + w[VT + 'embeddings.position_embedding.weight'][None]


def vit_attention(w, cfg: OracleConfig, p, x):
    """SiglipAttention.forward (siglip/modeling_siglip.py:273-307 [3P]): non-causal MHSA, scale head_dim^-0.5,
    softmax in fp32 (eager_attention_forward)."""
    B, N, C = x.shape
    nh = cfg.vit_heads; hd = C // nh
    q = linear(x, w[p + 'self_attn.q_proj.weight'], w[p + 'self_attn.q_proj.bias']).view(B, N, nh, hd).transpose(1, 2)
    k = linear(x, w[p + 'self_attn.k_proj.weight'], w[p + 'self_attn.k_proj.bias']).view(B, N, nh, hd).transpose(1, 2)
    v = linear(x, w[p + 'self_attn.v_proj.weight'], w[p + 'self_attn.v_proj.bias']).view(B, N, nh, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    a = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, N, C)
    return linear(o, w[p + 'self_attn.out_proj.weight'], w[p + 'self_attn.out_proj.bias'])


def vit_forward(w, cfg: OracleConfig, pixel_values):
    """models/live_llava/video_head_live_llava_qwen.py:96-98 -> SigLipVisionTower.forward [3P-recalled]: embeddings,
    then the `vit_layers` encoder layers that survive LLaVA's `del layers[-1:]`; returns hidden_states[-1], i.e. NO
    post_layernorm and no pooling head (unless cfg.vit_post_layernorm).  Layer math: SiglipEncoderLayer.forward
    (siglip/modeling_siglip.py:335-357 [3P])."""
    h = vit_patch_embed(w, cfg, pixel_values)
    for i in range(cfg.vit_layers):
        p = VT + f'encoder.layers.{i}.'
        r = h
        h = layer_norm(h, w[p + 'layer_norm1.weight'], w[p + 'layer_norm1.bias'], cfg.vit_layer_norm_eps)
        h = r + vit_attention(w, cfg, p, h)
        r = h
        h = layer_norm(h, w[p + 'layer_norm2.weight'], w[p + 'layer_norm2.bias'], cfg.vit_layer_norm_eps)
        h = gelu_tanh(linear(h, w[p + 'mlp.fc1.weight'], w[p + 'mlp.fc1.bias']))
        h = r + linear(h, w[p + 'mlp.fc2.weight'], w[p + 'mlp.fc2.bias'])
    if cfg.vit_post_layernorm:
        h = layer_norm(h, w[VT + 'post_layernorm.weight'], w[VT + 'post_layernorm.bias'], cfg.vit_layer_norm_eps)
    return h.to(pixel_values.dtype)


def vit_forward_autocast_fp16(w, cfg: OracleConfig, pixel_values):
    """The tower as the reference's GPU path runs it: `with torch.cuda.amp.autocast(): frames = self.vision_encode(...)` (models/modeling_live.py:28).
    Restated from the CUDA autocast op policy [3P, torch/csrc/autocast] applied to SigLipVisionTower / SiglipEncoderLayer [3P-recalled]:
      * conv / linear / matmul run with fp16 inputs and weights (fp32 accumulate), result fp16; the bias add is inside the op;
      * layer_norm and softmax are on the fp32 list: computed AND returned in fp32 (the next linear casts its input to fp16);
      * `patch_embeds (fp16) + position_embedding (model dtype)` and every residual `hidden + sublayer_out (fp16)` promote to fp32: the residual stream is fp32;
      * gelu_pytorch_tanh on an fp16 tensor: fp16 in, fp16 out;
      * the tower returns `hidden_states[-1].to(images.dtype)`.
    PARITY UNPINNED: `torch.cuda.amp.autocast` is a no-op without CUDA and CPU autocast follows a different op list, so the reference classes cannot be run
    this way in the build container; the plain (non-autocast) tower restatement above IS pinned against them (tests/test_oracle_vs_golden.py)."""
    h16 = lambda t: t.to(torch.float16)
    B, C, Himg, Wimg = pixel_values.shape
    P, g = cfg.vit_patch_size, cfg.vit_grid
    x = pixel_values[:, :, :g * P, :g * P].reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    h = linear(h16(x), h16(w[VT + 'embeddings.patch_embedding.weight'].flatten(1)), h16(w[VT + 'embeddings.patch_embedding.bias']))
    h = h.float() + w[VT + 'embeddings.position_embedding.weight'][None].float()
    nh = cfg.vit_heads
    for i in range(cfg.vit_layers):
        p = VT + f'encoder.layers.{i}.'
        r = h
        x = layer_norm(h, w[p + 'layer_norm1.weight'], w[p + 'layer_norm1.bias'], cfg.vit_layer_norm_eps)          # fp32 in, fp32 out
        N, Cw = x.shape[1], x.shape[2]
        hd = Cw // nh
        q, k, v = (linear(h16(x), h16(w[p + f'self_attn.{n}.weight']), h16(w[p + f'self_attn.{n}.bias'])).view(B, N, nh, hd).transpose(1, 2) for n in ('q_proj', 'k_proj', 'v_proj'))
        s = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)                                                         # fp16
        a = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, N, Cw)
        h = r + linear(o, h16(w[p + 'self_attn.out_proj.weight']), h16(w[p + 'self_attn.out_proj.bias'])).float()
        r = h
        x = layer_norm(h, w[p + 'layer_norm2.weight'], w[p + 'layer_norm2.bias'], cfg.vit_layer_norm_eps)
        x = gelu_tanh(linear(h16(x), h16(w[p + 'mlp.fc1.weight']), h16(w[p + 'mlp.fc1.bias'])))
        h = r + linear(x, h16(w[p + 'mlp.fc2.weight']), h16(w[p + 'mlp.fc2.bias'])).float()
    if cfg.vit_post_layernorm:
        h = layer_norm(h, w[VT + 'post_layernorm.weight'], w[VT + 'post_layernorm.bias'], cfg.vit_layer_norm_eps)
    return h.to(pixel_values.dtype)


def connector(w, x):
    """models/live_llava/video_head_live_llava_qwen.py:90-91 -> mm_projector = Linear, GELU(erf), Linear [3P-recalled]."""
    h = gelu_erf(linear(x, w['model.mm_projector.0.weight'], w['model.mm_projector.0.bias']))
    return linear(h, w['model.mm_projector.2.weight'], w['model.mm_projector.2.bias'])


def bilinear_taps(n_in: int, n_out: int):
    """Source taps of F.interpolate(mode='bilinear', align_corners=False, antialias=False) along one axis:
    src = (o + 0.5) * n_in/n_out - 0.5 clamped at 0; i0 = floor(src); i1 = min(i0 + 1, n_in - 1); lam = src - i0."""
    scale = n_in / n_out
    taps = []
    for o in range(n_out):
        src = max((o + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        taps.append((i0, i1, src - i0))
    return taps


def post_projector_pooling(cfg: OracleConfig, x):
    """models/live_llava/video_head_live_llava_qwen.py:100-119.  x [B, g*g, H] -> [B, out*out, H].
    bilinear: interpolate to ceil(g/stride) per side (:111-114); average/max: pool2d(kernel=stride=stride) (:107-110)."""
    B, N, H = x.shape
    g = int(round(math.sqrt(N)))
    s = cfg.video_pooling_stride
    img = x.view(B, g, g, H)
    mode = cfg.mm_spatial_pool_mode
    if mode == 'bilinear':
        out = math.ceil(g / s)
        taps = bilinear_taps(g, out)
        xf = img.float()
        rows = []
        for (y0, y1, ly) in taps:
            cols = []
            for (x0, x1, lx) in taps:
                top = xf[:, y0, x0] * (1 - lx) + xf[:, y0, x1] * lx
                bot = xf[:, y1, x0] * (1 - lx) + xf[:, y1, x1] * lx
                cols.append(top * (1 - ly) + bot * ly)
            rows.append(torch.stack(cols, 1))
        return torch.stack(rows, 1).reshape(B, out * out, H).to(x.dtype)
    if mode in ('average', 'max'):
        out = g // s
        blk = img[:, :out * s, :out * s].reshape(B, out, s, out, s, H)
        if mode == 'average':
            r = blk.float().mean(dim=(2, 4)).to(x.dtype)
        else:
            r = blk.amax(dim=(2, 4))
        return r.reshape(B, out * out, H)
    raise ValueError(f'Unexpected mm_spatial_pool_mode: {mode}')


def adaptive_avg_pool_tokens(x, out_hw):
    """models/vision_live.py:17-24 (secondary encoder path): adaptive_avg_pool2d of the s x s token grid to out_hw.
    Bin i covers [floor(i*s/o), ceil((i+1)*s/o))."""
    B, N, C = x.shape
    s = int(math.sqrt(N))
    img = x.view(B, s, s, C).float()
    oh, ow = out_hw
    rows = []
    for i in range(oh):
        y0, y1 = (i * s) // oh, -((-(i + 1) * s) // oh)
        cols = []
        for j in range(ow):
            x0, x1 = (j * s) // ow, -((-(j + 1) * s) // ow)
            cols.append(img[:, y0:y1, x0:x1].mean(dim=(1, 2)))
        rows.append(torch.stack(cols, 1))
    return torch.stack(rows, 1).reshape(B, oh * ow, C).to(x.dtype)


def visual_embed(w, cfg: OracleConfig, pixel_values, tower_autocast_fp16=False):
    """models/modeling_live.py:26-33: tower (under autocast on the reference's GPU path, :28) -> connector -> pooling -> flatten to [B*frame_num_tokens, hidden]."""
    h = vit_forward_autocast_fp16(w, cfg, pixel_values) if tower_autocast_fp16 else vit_forward(w, cfg, pixel_values)
    h = connector(w, h)
    h = post_projector_pooling(cfg, h)
    return h.reshape(-1, h.shape[-1])


# ----------------------------------------------------------------------------------------------------------------
# language side
# ----------------------------------------------------------------------------------------------------------------
class KVHandle:
    """Functional KV cache handle: (per-layer K, per-layer V) each [n_kv_heads, n, head_dim].  A handle is never
    mutated, so one held before `fast_greedy_generate` still denotes the pre-generation context afterwards -- the
    semantic `remove_assistant_turns` needs (test/inference.py:265-269; pinned transformers 4.44.2 behaviour, see
    SURVEY.md §8c TRAP).  Falsy when empty (test/inference.py:229)."""
    __slots__ = ('k', 'v')

    def __init__(self, k=None, v=None):
        self.k = k or []
        self.v = v or []

    def __len__(self):
        return 0 if not self.k else self.k[0].shape[1]

    def __bool__(self):
        return len(self) > 0

    def get_seq_length(self):
        return len(self)


def rms_norm(x, w, eps):
    """Qwen2RMSNorm.forward (qwen2/modeling_qwen2.py:248-253 [3P]): fp32 statistics, cast back, THEN times gain."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_tables(cfg: OracleConfig, positions, dtype):
    """Qwen2RotaryEmbedding (qwen2/modeling_qwen2.py:84-103 [3P]): inv_freq = theta^(-2i/d); freqs = pos x inv_freq
    in fp32; emb = cat(freqs, freqs); cos/sin cast to the activation dtype."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = positions.float()[:, None] * inv.to(positions.device)[None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def apply_rope(x, cos, sin):
    # apply_rotary_pos_emb (qwen2/modeling_qwen2.py:113-135 [3P]); x [heads, S, d], cos/sin [S, d]
    return x * cos[None] + rotate_half(x) * sin[None]


def llm_layer(w, cfg: OracleConfig, i, h, cos, sin, past_k, past_v):
    """Qwen2DecoderLayer.forward / Qwen2Attention.forward / Qwen2MLP.forward (qwen2/modeling_qwen2.py:36-48,
    200-240, 270-300 [3P]).  h [S, H]; past_k/v [n_kv, n, d] or None.  Causal mask with query offset n."""
    p = f'model.layers.{i}.'
    S, H = h.shape
    nh, nkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    r = h
    x = rms_norm(h, w[p + 'input_layernorm.weight'], cfg.rms_norm_eps)
    q = linear(x, w[p + 'self_attn.q_proj.weight'], w[p + 'self_attn.q_proj.bias']).view(S, nh, d).transpose(0, 1)
    k = linear(x, w[p + 'self_attn.k_proj.weight'], w[p + 'self_attn.k_proj.bias']).view(S, nkv, d).transpose(0, 1)
    v = linear(x, w[p + 'self_attn.v_proj.weight'], w[p + 'self_attn.v_proj.bias']).view(S, nkv, d).transpose(0, 1)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    if past_k is not None:
        k = torch.cat([past_k, k], 1); v = torch.cat([past_v, v], 1)
    n_tot = k.shape[1]; n = n_tot - S
    rep = nh // nkv
    kk = k[:, None].expand(nkv, rep, n_tot, d).reshape(nh, n_tot, d)
    vv = v[:, None].expand(nkv, rep, n_tot, d).reshape(nh, n_tot, d)
    s = torch.matmul(q, kk.transpose(1, 2)) * (d ** -0.5)
    qpos = torch.arange(S, device=h.device)[:, None] + n
    mask = torch.arange(n_tot, device=h.device)[None, :] > qpos
    s = s.masked_fill(mask[None], float('-inf'))
    a = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, vv).transpose(0, 1).reshape(S, nh * d)
    h = r + linear(o, w[p + 'self_attn.o_proj.weight'])
    r = h
    x = rms_norm(h, w[p + 'post_attention_layernorm.weight'], cfg.rms_norm_eps)
    g = linear(x, w[p + 'mlp.gate_proj.weight'])
    u = linear(x, w[p + 'mlp.up_proj.weight'])
    h = r + linear(F.silu(g) * u, w[p + 'mlp.down_proj.weight'])
    return h, k, v


def llm_forward(w, cfg: OracleConfig, inputs_embeds, past: Optional[KVHandle]):
    """Qwen2Model.forward (qwen2/modeling_qwen2.py:339-402 [3P]) for batch 1: position_ids = n + arange(S),
    28 x decoder layer, final RMSNorm.  Returns (hidden [S,H], new KVHandle)."""
    h = inputs_embeds
    S = h.shape[0]
    n = len(past) if past else 0
    cos, sin = rope_tables(cfg, torch.arange(n, n + S, device=h.device), h.dtype)
    ks, vs = [], []
    for i in range(cfg.num_hidden_layers):
        pk = past.k[i] if n else None
        pv = past.v[i] if n else None
        h, k, v = llm_layer(w, cfg, i, h, cos, sin, pk, pv)
        ks.append(k); vs.append(v)
    h = rms_norm(h, w['model.norm.weight'], cfg.rms_norm_eps)
    return h, KVHandle(ks, vs)


@dataclass
class OracleOutput:
    """Mirror of VideoHeadCausalLMOutputWithPast (models/live_llava/video_head_live_llava_qwen.py:48-58)."""
    logits: torch.Tensor = None
    informative_logits: torch.Tensor = None
    relevance_logits: torch.Tensor = None
    past_key_values: KVHandle = None
    loss: float = 0.0
    hidden_states: Optional[torch.Tensor] = None


class _Embedding:
    def __init__(self, table):
        self.weight = table

    def __call__(self, ids):
        # nn.Embedding; must accept k = 0 (`torch.tensor([[]])`, test/inference.py:234)
        ids = ids.long()
        return self.weight[ids.reshape(-1)].reshape(*ids.shape, self.weight.shape[1])


class _ImageProcessor:
    def __init__(self, size):
        self.size = size

    def preprocess(self, images, return_tensors='pt'):
        from .preprocess import siglip_preprocess
        return {'pixel_values': siglip_preprocess(images, self.size)}


class OracleModel:
    """Duck-type of VideoHeadLiveLlavaQwenForCausalLM as the stream driver uses it."""

    def __init__(self, cfg: OracleConfig, weights: dict):
        self.cfg = cfg
        self.w = weights
        self.dtype = weights['model.embed_tokens.weight'].dtype
        self.config = SimpleNamespace(hidden_size=cfg.hidden_size, frame_resolution=cfg.frame_resolution,
                                      frame_num_tokens=cfg.frame_num_tokens, v_placeholder=cfg.v_placeholder,
                                      v_placeholder_id=cfg.v_placeholder_id, eos_token_id=cfg.eos_token_id,
                                      vocab_size=cfg.vocab_size, video_pooling_stride=cfg.video_pooling_stride,
                                      mm_spatial_pool_mode=cfg.mm_spatial_pool_mode, vit_grid=cfg.vit_grid, vit_hidden_size=cfg.vit_hidden_size)
        self._embed = _Embedding(weights['model.embed_tokens.weight'])
        self._tower = SimpleNamespace(image_processor=_ImageProcessor(cfg.vit_image_size),
                                      num_patches_per_side=cfg.vit_grid)
        self.vocab_size = cfg.vocab_size

    def eval(self):
        return self

    def get_vision_tower(self):
        return self._tower

    def get_input_embeddings(self):
        return self._embed

    tower_autocast_fp16 = False          # True: the tower as under torch.cuda.amp.autocast() (vit_forward_autocast_fp16)

    def visual_embed(self, frames):
        return visual_embed(self.w, self.cfg, frames.to(self.dtype), tower_autocast_fp16=self.tower_autocast_fp16)

    def connector_pool(self, tower_features, out=None):
        """models/modeling_live.py:30-33 for pre-extracted tower features (no `vision_encode`): connector -> pooling -> flatten."""
        h = post_projector_pooling(self.cfg, connector(self.w, tower_features.to(self.dtype)))
        h = h.reshape(-1, h.shape[-1])
        if out is not None:
            out.copy_(h)
            return out
        return h

    def tower_features(self, frames):
        return vit_forward(self.w, self.cfg, frames.to(self.dtype))

    def cache_prefix(self, handle, length):
        return KVHandle([k[:, :length] for k in handle.k], [v[:, :length] for v in handle.v])

    def joint_embed(self, input_ids=None, frames=None):
        """models/modeling_live.py:35-48."""
        if frames is None:
            return self._embed(input_ids)
        if input_ids is None:
            return self.visual_embed(frames)
        e = self._embed(input_ids.clamp(max=self.vocab_size - 1)).clone()
        m = input_ids == self.cfg.v_placeholder_id
        if m.any():
            e[m] = self.visual_embed(frames).to(e.dtype)
        return e

    def __call__(self, input_ids=None, inputs_embeds=None, past_key_values=None, use_cache=True, return_dict=True,
                 frames=None, **kw):
        """models/live_llava/video_head_live_llava_qwen.py:121-205: body -> lm_head / informative_head / relevance_head on the
        post-final-norm hidden state, all `.float()` (:155,:160-161)."""
        if inputs_embeds is None:
            inputs_embeds = self.joint_embed(input_ids, frames)
        assert inputs_embeds.shape[0] == 1
        h, cache = llm_forward(self.w, self.cfg, inputs_embeds[0].to(self.dtype), past_key_values)
        out = OracleOutput(
            logits=linear(h, self.w['lm_head.weight']).float()[None],
            informative_logits=linear(h, self.w['informative_head.weight']).float()[None],
            relevance_logits=linear(h, self.w['relevance_head.weight']).float()[None],
            past_key_values=cache, hidden_states=h[None])
        return out


def repetition_penalty_(scores, prev_ids, penalty):
    """RepetitionPenaltyLogitsProcessor.__call__ [3P]: score = score/p if score > 0 else score*p on seen ids."""
    ids = torch.as_tensor(prev_ids, dtype=torch.long, device=scores.device)
    s = scores.clone()
    g = s[ids]
    s[ids] = torch.where(g < 0, g * penalty, g / penalty)
    return s


def fast_greedy_generate(*, model, inputs_embeds, past_key_values, eos_token_id, inplace_output_ids,
                         repetition_penalty=None, generated_token_ids=None):
    """models/modeling_live.py:51-77.  EOS is written to the output but neither fed back nor added to the penalty
    list (:68-75); the penalty list persists across turns of one video."""
    if generated_token_ids is None:
        generated_token_ids = []
    i = 0
    for i in range(inplace_output_ids.size(1)):
        out = model(inputs_embeds=inputs_embeds, past_key_values=past_key_values, use_cache=True, return_dict=True)
        past_key_values = out.past_key_values
        last = out.logits[0, -1]
        if repetition_penalty is not None and len(generated_token_ids) > 0:
            last = repetition_penalty_(last, generated_token_ids, repetition_penalty)
        tok = int(last.argmax(-1))
        if repetition_penalty is not None and tok != eos_token_id:
            generated_token_ids.append(tok)
        inplace_output_ids[:, i] = tok
        if tok == eos_token_id:
            break
        inputs_embeds = model.get_input_embeddings()(torch.tensor([[tok]], device=inputs_embeds.device))
    return inplace_output_ids[:, :i + 1], past_key_values, generated_token_ids
