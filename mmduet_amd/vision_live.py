"""Secondary frame encoders: stand-alone HF SigLIP / CLIP towers with adaptive pooling (+ CLS), on the HIP library.

Mirrors models/vision_live.py of the reference:
  `_siglip_vision_encode` (:11-31)  normalize(frames / 255, .5, .5) -> SiglipVisionModel -> adaptive_avg_pool2d of the s x s tokens to
                                    `frame_token_pooled`, optional `pooler_output` (attention-pooling head) in front when `frame_token_cls`
  `_clip_vision_encode`   (:34-54)  normalize with the OpenAI CLIP mean / std -> CLIPVisionModel (class token, pre_layrnorm, quick_gelu) ->
                                    pooled spatial tokens (class token skipped), optional raw token 0 as CLS
  `build_live_vision`     (:57-64)  picks the encoder from `config.vision_pretrained`
In the shipped LLaVA model this path is dead (the model owns a tower, SURVEY.md section 2 note A); it is the encoder of the offline feature
extraction (`distributed_encode(vision_encode=...)`, data/utils.py:99-117).  Everything runs in libmmduet_hip.so through a *vision-only*
context (mmd_config.vision_only): mmd_normalize_frames, mmd_vision_tower, mmd_vision_pool_tokens, mmd_vision_pool_head.

One deviation, stated: `_clip_vision_encode` with BOTH frame_token_cls and frame_token_pooled concatenates a [B,C] with a [B,hw,C] tensor and
raises in the reference (:51-54); here the CLS row is unsqueezed like in the SigLIP twin.  `build_live_vision`'s CLIP branch also binds `config`
to the wrong positional (`partial(_clip_vision_encode, config)`, :62); the flags are passed by keyword here.
"""
import ctypes as C
import threading
from functools import partial
import torch

from ._lib import lib, check, MmdConfig, MMD_BF16, MMD_F32

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_T2M = {torch.float32: MMD_F32, torch.bfloat16: MMD_BF16}

# tower shapes of the checkpoints `build_live_vision` accepts ([3P-recalled] model cards; read from config.json when loading from disk)
KNOWN = {
    'google/siglip-large-patch16-384': ('siglip', dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=384,
                                                       patch_size=16, layer_norm_eps=1e-6)),
    'openai/clip-vit-large-patch14-336': ('clip', dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                                                       patch_size=14, layer_norm_eps=1e-5)),
    'laion/CLIP-ViT-L-14-DataComp.XL-s13B-b90k': ('clip', dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                                                               image_size=224, patch_size=14, layer_norm_eps=1e-5)),
}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class LiveVisionEncoder:
    """`vision_model` + its `vision_encode` in one object: `enc(frames)` == `vision_encode(vision_model, frames)` of the reference."""

    def __init__(self, kind, vcfg, torch_dtype=torch.bfloat16, device=None, frame_token_cls=False, frame_token_pooled=(7, 7), max_batch=32):
        if kind not in ('siglip', 'clip'):
            raise ValueError(f'Unverified vision encoder kind: {kind}')
        if not torch.cuda.is_available():
            from ._lib import MmduetError
            raise MmduetError('no HIP device visible: this implementation runs on MI355X only (no CPU fallback)')
        self.kind, self.vcfg, self.dtype = kind, dict(vcfg), torch_dtype
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.frame_token_cls = bool(frame_token_cls)
        self.frame_token_pooled = tuple(frame_token_pooled) if frame_token_pooled else None
        if self.frame_token_pooled and self.frame_token_pooled[0] != self.frame_token_pooled[1]:
            raise ValueError('frame_token_pooled must be square')
        if not self.frame_token_cls and not self.frame_token_pooled:
            raise ValueError('neither frame_token_cls nor frame_token_pooled: the reference function has nothing to return')
        self.mean, self.std = ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)) if kind == 'siglip' else (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)
        self.rescale_factor = 0.00392156862745098
        self.max_batch = max_batch
        c = MmdConfig()
        c.struct_size = C.sizeof(MmdConfig)
        c.dtype = _T2M[torch_dtype]
        c.vision_only = 1
        H = vcfg['hidden_size']
        c.vocab_size, c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim = 0, H, 0, 0, 1, 1, 2
        c.rope_theta, c.rms_norm_eps = 1e4, 1e-6
        c.vit_hidden, c.vit_intermediate, c.vit_layers, c.vit_heads = H, vcfg['intermediate_size'], vcfg['num_hidden_layers'], vcfg['num_attention_heads']
        c.vit_image, c.vit_patch, c.vit_ln_eps = vcfg['image_size'], vcfg['patch_size'], float(vcfg['layer_norm_eps'])
        c.vit_post_layernorm = 1 if kind == 'siglip' else 0          # SigLIP: last_hidden_state is post-LN; CLIP: post_layernorm only touches the pooled token
        c.vit_class_token = c.vit_pre_layernorm = 1 if kind == 'clip' else 0
        c.vit_act = 1 if kind == 'clip' else 0
        c.vit_pool_head = 1 if (kind == 'siglip' and self.frame_token_cls) else 0
        c.pool_mode, c.pool_stride, c.frame_num_tokens = 3, (self.frame_token_pooled or (1, 1))[0], 1
        c.max_vit_batch, c.max_step_tokens = max_batch, 1
        self._cfg = c
        self.grid = vcfg['image_size'] // vcfg['patch_size']
        self.seq = self.grid ** 2 + (1 if kind == 'clip' else 0)
        self._ctx = None
        h = C.c_void_p()
        check(lib().mmd_create(C.byref(c), self.device.index or 0, C.byref(h)), None, 'mmd_create')
        self._ctx = h
        self._lock = threading.RLock()

    def __del__(self):
        try:
            if self._ctx:
                lib().mmd_destroy(self._ctx)
        except Exception:
            pass
        self._ctx = None

    # ---- weights ------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """HF vision-model state dict (`embeddings.patch_embedding.weight`, `encoder.layers.N...`, `post_layernorm...`, `head...`; a leading
        `vision_model.` is accepted)."""
        need_head = bool(self._cfg.vit_pool_head)
        H = self.vcfg['hidden_size']
        with self._lock:
            lib().mmd_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            seen = set()
            for name, t in sd.items():
                name = name[len('vision_model.'):] if name.startswith('vision_model.') else name
                if 'position_ids' in name or (name.startswith('head.') and not need_head) or (self.kind == 'clip' and name.startswith('post_layernorm')):
                    continue
                t = t.detach()
                if t.dtype not in _T2M:
                    t = t.float()
                t = t.contiguous()
                shape = (C.c_int64 * t.ndim)(*t.shape)
                check(lib().mmd_load_tensor(self._ctx, ('vit.' + name).encode(), _p(t), _T2M[t.dtype], shape, t.ndim, 1 if t.is_cuda else 0), self._ctx, f'load {name}')
                seen.add(name)
            if 'embeddings.patch_embedding.bias' not in seen:            # CLIP's patch conv has no bias
                z = torch.zeros(H, dtype=torch.float32)
                check(lib().mmd_load_tensor(self._ctx, b'vit.embeddings.patch_embedding.bias', _p(z), MMD_F32, (C.c_int64 * 1)(H), 1, 0), self._ctx, 'patch bias')
            torch.cuda.synchronize(self.device)
            check(lib().mmd_finalize_weights(self._ctx), self._ctx, 'mmd_finalize_weights')
        return self

    @classmethod
    def from_state_dict(cls, kind, vcfg, sd, **kw):
        return cls(kind, vcfg, **kw).load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, name_or_path, **kw):
        """A local HF checkpoint directory (config.json + *.safetensors) of a SigLIP / CLIP model, or one of the ids `build_live_vision` accepts
        (must then be in the local HF cache: there is no network on the target boxes)."""
        import json, os
        from safetensors import safe_open
        from .weights import _resolve_dir
        d = _resolve_dir(name_or_path)
        cfg = json.load(open(os.path.join(d, 'config.json')))
        vc = cfg.get('vision_config', cfg)
        mt = cfg.get('model_type', vc.get('model_type', ''))
        kind = 'siglip' if 'siglip' in mt else ('clip' if 'clip' in mt else None)
        known = KNOWN.get(name_or_path)
        if kind is None and known:
            kind = known[0]
        if kind is None:
            raise ValueError(f'Unverified vision_pretrained: {name_or_path}')
        base = dict(known[1]) if known else {}
        base.update({k: vc[k] for k in ('hidden_size', 'intermediate_size', 'num_hidden_layers', 'num_attention_heads', 'image_size', 'patch_size', 'layer_norm_eps') if k in vc})
        sd = {}
        for f in sorted(x for x in os.listdir(d) if x.endswith('.safetensors')):
            with safe_open(os.path.join(d, f), 'pt') as sf:
                for k in sf.keys():
                    if k.startswith('vision_model.'):
                        sd[k] = sf.get_tensor(k)
        if not sd:
            raise FileNotFoundError(f'no vision_model.* tensors under {d}')
        return cls.from_state_dict(kind, base, sd, **kw)

    # ---- encode ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, frames):
        """frames uint8 or float [B,3,R,R] (R = the tower's image size) -> [B, (1+)h*w, C] (or the reference's other return shapes)."""
        frames = torch.as_tensor(frames)
        R = self.vcfg['image_size']
        if frames.ndim != 4 or frames.shape[1] != 3 or frames.shape[2] != R or frames.shape[3] != R:
            raise ValueError(f"Input image size ({frames.shape[-2]}*{frames.shape[-1]}) doesn't match model ({R}*{R}).")       # HF embeddings' own check
        frames = frames.to(self.device)
        kind = 0 if frames.dtype == torch.uint8 else 1
        if kind == 1:
            frames = frames.float()
        frames = frames.contiguous()
        B, Cv = frames.shape[0], self.vcfg['hidden_size']
        n_sp = (self.frame_token_pooled[0] * self.frame_token_pooled[1]) if self.frame_token_pooled else 0
        spatial = torch.empty(B, n_sp, Cv, dtype=self.dtype, device=self.device) if n_sp else None
        cls = torch.empty(B, Cv, dtype=self.dtype, device=self.device) if self.frame_token_cls else None
        mean, std = (C.c_float * 3)(*self.mean), (C.c_float * 3)(*self.std)
        with self._lock:
            L = lib()
            L.mmd_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch); n = b1 - b0
                px = torch.empty(n, 3, R, R, dtype=self.dtype, device=self.device)
                check(L.mmd_normalize_frames(self._ctx, _p(frames[b0:b1]), kind, n, R, mean, std, self.rescale_factor, _p(px)), self._ctx, 'mmd_normalize_frames')
                feats = torch.empty(n, self.seq, Cv, dtype=self.dtype, device=self.device)
                check(L.mmd_vision_tower(self._ctx, _p(px), n, _p(feats)), self._ctx, 'mmd_vision_tower')
                if spatial is not None:
                    check(L.mmd_vision_pool_tokens(self._ctx, _p(feats), n, self.frame_token_pooled[0], self.frame_token_pooled[1], _p(spatial[b0:b1])), self._ctx, 'mmd_vision_pool_tokens')
                if cls is not None:
                    if self.kind == 'siglip':
                        check(L.mmd_vision_pool_head(self._ctx, _p(feats), n, _p(cls[b0:b1])), self._ctx, 'mmd_vision_pool_head')
                    else:
                        cls[b0:b1] = feats[:, 0]
        if spatial is not None and cls is None:
            return spatial
        if spatial is None:
            return cls[:, None] if self.kind == 'siglip' else cls           # models/vision_live.py:28-30 vs :51-53
        return torch.cat([cls[:, None], spatial], dim=1)


def _vision_encode(vision_model, frames, **_ignored):
    """`vision_encode(vision_encoder, frames)` as LiveMixin.visual_embed calls it (models/modeling_live.py:29)."""
    return vision_model(frames)


def build_live_vision(config, torch_dtype=torch.bfloat16):
    """models/vision_live.py:57-64: -> (vision_model, vision_encode)."""
    name = config.vision_pretrained
    if name not in KNOWN:
        raise ValueError(f'Unverified vision_pretrained: {name}')
    enc = LiveVisionEncoder.from_pretrained(name, torch_dtype=torch_dtype, frame_token_cls=config.frame_token_cls, frame_token_pooled=config.frame_token_pooled)
    return enc, partial(_vision_encode)
