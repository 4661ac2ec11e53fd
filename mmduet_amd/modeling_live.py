"""HIP-backed LiveLlava model with the reference's Python surface.

Mirrors (names, argument meaning, error behaviour) the pieces of the reference the stream driver touches:
  VideoHeadLiveLlavaQwenForCausalLM   models/live_llava/video_head_live_llava_qwen.py:67-242
  VideoHeadCausalLMOutputWithPast     models/live_llava/video_head_live_llava_qwen.py:48-58
  LiveMixin.visual_embed / joint_embed   models/modeling_live.py:26-48
  fast_greedy_generate                models/modeling_live.py:51-77
  build_live / build_model_and_tokenizer   models/modeling_live.py:80-129, models/__init__.py:8-13
All arithmetic runs in libmmduet_hip.so (include/mmduet.h); torch only provides device buffers and the stream.
"""
from __future__ import annotations
import ctypes as C
import threading
import weakref
from typing import Optional
import torch

from . import _lib
from ._lib import lib, check, MmdConfig, MMD_BF16, MMD_F32, POOL_MODES
from .configuration_live import VideoHeadLiveLlavaQwenConfig
from .tokenization_live import build_live_tokenizer_and_update_config

_TORCH2MMD = {torch.float32: MMD_F32, torch.bfloat16: MMD_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------------------------------
# KV cache handles
# ----------------------------------------------------------------------------------------------------------------
class _KVArena:
    """Owns one native KV arena (mmd_stream): contiguous per-layer K/V with O(1) append / truncate.

    Arenas are recycled through a small per-model pool: creating one is two hipMalloc + memset of ~1 GB (30 ms, 3 % of a 300-frame
    stream), reusing one is an O(1) truncate to length 0 (every slot holds finite data, which is all the masked tail of a key
    tile needs).  Only arenas up to POOL_MAX_TOKENS are kept, at most POOL_SIZE of them; `model.release_pooled_arenas()` frees them."""
    POOL_SIZE = 8
    POOL_MAX_TOKENS = 1 << 16

    def __init__(self, model, initial_tokens):
        self.model = model
        self.handles = weakref.WeakSet()
        pool = model.__dict__.setdefault('_arena_pool', [])
        for i, h in enumerate(pool):
            if lib().mmd_kv_capacity(h) >= int(initial_tokens):
                pool.pop(i)
                check(lib().mmd_kv_truncate(h, 0), model._ctx, 'mmd_kv_truncate')
                self.h = h
                return
        h = C.c_void_p()
        check(lib().mmd_stream_create(model._ctx, int(initial_tokens), C.byref(h)), model._ctx, 'mmd_stream_create')
        self.h = h

    def length(self):
        return int(lib().mmd_kv_len(self.h))

    def truncate(self, n):
        if n < self.length():
            for hd in list(self.handles):
                if hd.length > n:
                    hd.stale = True
            check(lib().mmd_kv_truncate(self.h, int(n)), self.model._ctx, 'mmd_kv_truncate')

    def __del__(self):
        try:
            if self.h and self.model._ctx:
                pool = self.model.__dict__.get('_arena_pool')
                if pool is not None and len(pool) < self.POOL_SIZE and lib().mmd_kv_capacity(self.h) <= self.POOL_MAX_TOKENS:
                    pool.append(self.h)
                else:
                    lib().mmd_stream_destroy(self.h)
        except Exception:
            pass
        self.h = None


class KVCacheHandle:
    """`past_key_values` as callers see it: an opaque (arena, length) pair.

    Semantics the reference driver relies on:
      * falsy when empty (`if not self.past_key_values`, test/inference.py:229);
      * functional: a handle held before `fast_greedy_generate` still denotes the pre-generation context
        afterwards (remove_assistant_turns, test/inference.py:265-269).  Continuing from an older handle truncates
        the arena back to its length in O(1); handles that pointed beyond that become stale and raise if used.
    """
    __slots__ = ('arena', 'length', 'stale', '__weakref__')

    def __init__(self, arena: _KVArena, length: int):
        self.arena, self.length, self.stale = arena, int(length), False
        arena.handles.add(self)

    def __len__(self):
        return self.length

    def __bool__(self):
        return self.length > 0

    def get_seq_length(self, layer_idx=0):
        return self.length


class SamplerHandle:
    """mmd_sampler: greedy sampling state of one stream on the device (models/modeling_live.py:51-77 run by mmd_round_multi)."""

    def __init__(self, model):
        self.model = model
        h = C.c_void_p()
        with model._lock:
            check(lib().mmd_sampler_create(model._ctx, C.byref(h)), model._ctx, 'mmd_sampler_create')
        self.h = h.value

    def begin(self, eos_token_id, repetition_penalty, generated_token_ids, max_new_tokens):
        """Start of a response: the penalty list is every id generated so far in this video (it persists across turns, models/modeling_live.py:60-66)."""
        pen = float(repetition_penalty) if repetition_penalty is not None else 0.0
        prev = list(generated_token_ids) if (generated_token_ids is not None and pen > 0) else []
        arr = (C.c_int64 * max(1, len(prev)))(*prev)
        with self.model._lock:
            self.model._bind_stream()
            check(lib().mmd_sampler_begin(self.h, int(eos_token_id if eos_token_id is not None else -1), pen, arr, len(prev), int(max_new_tokens)), self.model._ctx, 'mmd_sampler_begin')

    def __del__(self):
        try:
            if getattr(self, 'h', None) and getattr(self.model, '_ctx', None):
                lib().mmd_sampler_destroy(self.h)
        except Exception:
            pass
        self.h = None


class VideoHeadCausalLMOutputWithPast:
    """Output of the model call.  `.logits` (lm_head) is computed lazily and, unless
    `config.all_position_logits` is set, only for the LAST position ([1,1,V]): the streaming loop reads nothing else
    (`outputs.logits[:, -1:]`), and the reference's all-position lm_head is 53 GFLOP + 30 MB per frame of waste."""

    def __init__(self, model, hidden, cache):
        self._model, self._hidden = model, hidden          # hidden: [S, H] post-final-norm
        self.past_key_values = cache
        self.loss = 0.0
        self.lm_loss = 0.0
        self.video_loss = 0.0
        self.attentions = None
        self._logits = None
        self._heads = None

    @property
    def hidden_states(self):
        return self._hidden[None]

    @property
    def logits(self):
        if self._logits is None:
            rows = self._hidden if self._model.config.all_position_logits else self._hidden[-1:]
            self._logits = self._model.lm_head(rows)[None]
        return self._logits

    def _video_heads(self):
        if self._heads is None:
            self._heads = self._model.video_heads(self._hidden)
        return self._heads

    @property
    def informative_logits(self):
        return self._video_heads()[None, :, 0:2]

    @property
    def relevance_logits(self):
        return self._video_heads()[None, :, 2:4]


# ----------------------------------------------------------------------------------------------------------------
class _Embedding:
    """`model.get_input_embeddings()`: callable on LongTensor [..., k] (k may be 0, test/inference.py:234)."""

    def __init__(self, model):
        self._m = model

    def __call__(self, ids, out=None):
        """`out` (optional): a contiguous [k, hidden] slice of a caller's buffer to gather into (the stream driver's step buffer: no concatenation kernel)."""
        m = self._m
        ids = ids.to(device=m.device, dtype=torch.long).contiguous()
        k = ids.numel()
        if out is None:
            out = torch.empty(*ids.shape, m.config.hidden_size, dtype=m.dtype, device=m.device)
        elif out.shape != (k, m.config.hidden_size) or out.dtype != m.dtype or not out.is_contiguous():
            raise ValueError('embedding `out` must be a contiguous [k, hidden] tensor of the model dtype')
        if k:
            with m._lock:
                m._bind_stream()
                check(lib().mmd_embed_tokens(m._ctx, _ptr(ids), k, _ptr(out)), m._ctx, 'mmd_embed_tokens')
        return out


class SigLipImageProcessor:
    """`model.get_vision_tower().image_processor` (LLaVA SigLipImageProcessor, used at test/inference.py:203).
    preprocess = bicubic resize to the tower resolution (bit-exact with Pillow), 1/255, (x-.5)/.5 -- executed by the
    HIP kernel behind mmd_preprocess_frames; the result stays on the device in the model dtype, so the driver's
    `.to('cuda').to(dtype)` are no-ops."""

    def __init__(self, model):
        self._m = model
        s = model.config.vit_image_size
        self.size = (s, s)
        self.image_mean = self.image_std = (0.5, 0.5, 0.5)
        self.rescale_factor = 1 / 255

    def preprocess(self, images, return_tensors='pt'):
        m = self._m
        if isinstance(images, (list, tuple)):
            images = torch.stack([torch.as_tensor(i) for i in images])
        images = torch.as_tensor(images)
        if images.dtype != torch.uint8 or images.ndim != 4 or images.shape[1] != 3 or images.shape[2] != images.shape[3]:
            raise ValueError(f'expected uint8 frames [T,3,R,R], got {tuple(images.shape)} {images.dtype}')
        fr = images.to(m.device).contiguous()
        T, _, R, _ = fr.shape
        out = torch.empty(T, 3, self.size[0], self.size[1], dtype=m.dtype, device=m.device)
        if T:
            with m._lock:
                m._bind_stream()
                check(lib().mmd_preprocess_frames(m._ctx, _ptr(fr), T, R, _ptr(out)), m._ctx, 'mmd_preprocess_frames')
        return {'pixel_values': out}


class _VisionTower:
    def __init__(self, model):
        self.image_processor = SigLipImageProcessor(model)
        self.num_patches_per_side = model.config.vit_grid
        self.hidden_size = model.config.vit_hidden_size

    def parameters(self):
        return iter(())


class VideoHeadLiveLlavaQwenForCausalLM:
    config_class = VideoHeadLiveLlavaQwenConfig

    def __init__(self, config: VideoHeadLiveLlavaQwenConfig, torch_dtype=torch.bfloat16, device=None,
                 max_vit_batch=35, max_step_tokens=1024, kv_initial_tokens=32768):
        if torch_dtype not in _TORCH2MMD:
            raise ValueError(f'torch_dtype must be bfloat16 or float32 on this implementation, got {torch_dtype}')
        if not torch.cuda.is_available():
            raise _lib.MmduetError('no HIP device visible: this implementation runs on MI355X only (no CPU fallback)')
        L = lib()
        self.config = config
        if not hasattr(config, 'all_position_logits'):
            config.all_position_logits = False
        self.dtype = torch_dtype
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.vocab_size = config.vocab_size
        self.kv_initial_tokens = kv_initial_tokens
        self.max_vit_batch, self.max_step_tokens = max_vit_batch, max_step_tokens
        c = MmdConfig()
        c.struct_size = C.sizeof(MmdConfig)
        c.dtype = _TORCH2MMD[torch_dtype]
        c.vocab_size, c.hidden_size, c.intermediate_size = config.vocab_size, config.hidden_size, config.intermediate_size
        c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim = config.num_hidden_layers, config.num_attention_heads, config.num_key_value_heads, config.head_dim
        c.rope_theta, c.rms_norm_eps = float(config.rope_theta), float(config.rms_norm_eps)
        c.vit_hidden, c.vit_intermediate, c.vit_layers, c.vit_heads = config.vit_hidden_size, config.vit_intermediate_size, config.vit_layers_run, config.vit_num_attention_heads
        c.vit_image, c.vit_patch, c.vit_ln_eps = config.vit_image_size, config.vit_patch_size, float(config.vit_layer_norm_eps)
        c.vit_post_layernorm = int(bool(config.vit_post_layernorm))
        if config.mm_spatial_pool_mode not in POOL_MODES:
            raise ValueError(f'Unexpected mm_spatial_pool_mode: {config.mm_spatial_pool_mode}')
        c.pool_mode, c.pool_stride = POOL_MODES[config.mm_spatial_pool_mode], config.video_pooling_stride
        g = config.vit_grid
        if config.mm_spatial_pool_mode == 'adaptive_avg':       # secondary path (models/vision_live.py): pool to frame_token_pooled[0] per side
            out_side = config.video_pooling_stride
        else:
            out_side = -(-g // config.video_pooling_stride) if config.mm_spatial_pool_mode == 'bilinear' else g // config.video_pooling_stride
        self.tokens_per_frame = out_side * out_side
        c.frame_num_tokens = self.tokens_per_frame
        c.max_vit_batch, c.max_step_tokens = max_vit_batch, max_step_tokens
        wd = getattr(config, 'weight_dtype', None)
        if wd not in (None, 'bf16', 'model', 'fp8', 'fp8_e4m3'):
            raise ValueError(f'unknown weight_dtype {wd!r} (None | "fp8_e4m3")')
        c.weight_dtype = 1 if wd in ('fp8', 'fp8_e4m3') else 0
        # The reference runs the tower under torch.cuda.amp.autocast() (models/modeling_live.py:28): IEEE-half matmuls, fp32 LayerNorm / softmax, whatever the model
        # dtype.  tower_dtype: None / 'auto' = do the same wherever the half forms of the kernels exist for the tower's shape (bf16 model, the LLaVA SigLIP form,
        # hidden % 64 == 0, head_dim % 8 == 0, >= 65 tokens per frame -- every real tower; measured free: 340 frames/s either way), else the model dtype;
        # 'fp16' = require it; 'bf16' / 'model' = tower in the model dtype (the round-2 behaviour).
        # 'fp16' keeps the hidden state between the fp16 matmuls in fp32, as autocast's type promotion does; 'fp16_resid16' rounds it to fp16 after every sublayer
        # (the round-3 form, a little faster, 0.022 instead of 0.020 rms from the fp32 tower at true width).
        td = getattr(config, 'tower_dtype', None)
        if td not in (None, 'auto', 'model', 'bf16', 'fp16', 'float16', 'fp16_resid16'):
            raise ValueError(f'unknown tower_dtype {td!r} (None | "auto" | "fp16" | "fp16_resid16" | "bf16")')
        half_ok = (torch_dtype == torch.bfloat16 and config.vit_hidden_size % 64 == 0 and (config.vit_hidden_size // config.vit_num_attention_heads) % 8 == 0 and
                   config.vit_grid ** 2 >= 65)
        if td in ('fp16', 'float16', 'fp16_resid16') and not half_ok:
            raise ValueError('tower_dtype=fp16 needs torch_dtype=bfloat16 and a tower with hidden % 64 == 0, head_dim % 8 == 0 and at least 65 tokens per frame')
        resid32_ok = not getattr(config, 'vit_post_layernorm', False) and config.vit_hidden_size <= 2048
        if td in ('fp16', 'float16') and not resid32_ok:
            raise ValueError('tower_dtype=fp16 (fp32 residual stream) needs a tower without post_layernorm and hidden <= 2048; use fp16_resid16')
        c.tower_f16 = 2 if (td == 'fp16_resid16' or (td in (None, 'auto') and half_ok and not resid32_ok)) else 1 if (td in ('fp16', 'float16') or (td in (None, 'auto') and half_ok)) else 0
        self.tower_dtype = {0: 'bf16' if torch_dtype == torch.bfloat16 else 'fp32', 1: 'fp16', 2: 'fp16_resid16'}[c.tower_f16]
        if c.weight_dtype and torch_dtype != torch.bfloat16:
            raise ValueError('weight_dtype=fp8_e4m3 needs torch_dtype=bfloat16 (fp8 weights x bf16 activations, fp32 accumulate)')
        self._cfg_struct = c
        self._ctx = None
        h = C.c_void_p()
        check(L.mmd_create(C.byref(c), self.device.index or 0, C.byref(h)), None, 'mmd_create')
        self._ctx = h
        self._lock = threading.RLock()      # the Gradio demo calls the model from two threads (demo/app.py:84-85)
        self._embed = _Embedding(self)
        self._tower = _VisionTower(self)
        self._finalized = False
        self.lm_loss_weight = self.video_loss_weight = 1
        # RoPE table with the reference's own expression (transformers qwen2/modeling_qwen2.py:84-85)
        d = config.head_dim
        inv = (1.0 / (float(config.rope_theta) ** (torch.arange(0, d, 2, dtype=torch.float) / d))).contiguous()
        check(L.mmd_set_rope_inv_freq(self._ctx, C.c_void_p(inv.data_ptr()), d // 2), self._ctx, 'mmd_set_rope_inv_freq')

    # ---- lifecycle ----------------------------------------------------------------------------------------------
    def release_pooled_arenas(self):
        """Free the KV arenas kept for reuse (see _KVArena)."""
        pool = self.__dict__.get('_arena_pool') or []
        while pool:
            lib().mmd_stream_destroy(pool.pop())

    def __del__(self):
        try:
            if self._ctx:
                self.release_pooled_arenas()
                lib().mmd_destroy(self._ctx)
        except Exception:
            pass
        self._ctx = None

    def _bind_stream(self):
        lib().mmd_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def set_tower_share(self, max_blocks: int):
        """Scheduling knob (results unchanged): cap the tower's ring-GEMM grids at `max_blocks` workgroups, 0 = all CUs (mmd_set_tower_share)."""
        with self._lock:
            check(lib().mmd_set_tower_share(self._ctx, int(max_blocks)), self._ctx, 'mmd_set_tower_share')

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def to(self, *a, **k):
        return self

    def parameters(self):
        return iter(())

    # ---- weights ------------------------------------------------------------------------------------------------
    def load_tensor(self, name: str, t: torch.Tensor):
        if t.dtype not in _TORCH2MMD:
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * t.ndim)(*t.shape)
        on_dev = 1 if t.is_cuda else 0
        with self._lock:
            self._bind_stream()
            check(lib().mmd_load_tensor(self._ctx, name.encode(), _ptr(t), _TORCH2MMD[t.dtype], shape, t.ndim, on_dev), self._ctx, f'load {name}')
            if on_dev:
                torch.cuda.current_stream(self.device).synchronize()

    def merge_lora(self, weight_name: str, A: torch.Tensor, B: torch.Tensor, scale: float):
        A = A.float().cpu().contiguous(); B = B.float().cpu().contiguous()
        with self._lock:
            self._bind_stream()
            check(lib().mmd_merge_lora(self._ctx, weight_name.encode(), _ptr(A), _ptr(B), A.shape[0], float(scale)), self._ctx, f'lora {weight_name}')

    def load_state_dict(self, named_tensors, strict=True):
        for name, t in (named_tensors.items() if hasattr(named_tensors, 'items') else named_tensors):
            self.load_tensor(name, t)
        self.finalize()
        return self

    def finalize(self):
        with self._lock:
            self._bind_stream()
            check(lib().mmd_finalize_weights(self._ctx), self._ctx, 'mmd_finalize_weights')
        self._finalized = True

    def weight_bytes(self):
        return int(lib().mmd_weight_bytes(self._ctx))

    # ---- reference surface ------------------------------------------------------------------------------------------
    def get_model(self):
        return self

    def get_vision_tower(self):
        return self._tower

    def get_input_embeddings(self):
        return self._embed

    def set_vision_inside(self):
        """models/modeling_live.py:14-20: the tower is always inside this model."""
        return None

    def visual_embed(self, frames: torch.Tensor, out: torch.Tensor = None):
        """models/modeling_live.py:26-33 -> [B*frame_num_tokens, hidden] in the model dtype (written into `out` when given)."""
        frames = frames.to(device=self.device, dtype=self.dtype).contiguous()
        B = frames.shape[0]
        if out is None:
            out = torch.empty(B * self.tokens_per_frame, self.config.hidden_size, dtype=self.dtype, device=self.device)
        elif out.shape != (B * self.tokens_per_frame, self.config.hidden_size) or out.dtype != self.dtype or not out.is_contiguous():
            raise ValueError('visual_embed: `out` must be a contiguous [B*frame_num_tokens, hidden] tensor of the model dtype')
        with self._lock:
            self._bind_stream()
            for b0 in range(0, B, self.max_vit_batch):
                b1 = min(B, b0 + self.max_vit_batch)
                check(lib().mmd_vit_encode(self._ctx, _ptr(frames[b0:b1]), b1 - b0, _ptr(out[b0 * self.tokens_per_frame:])), self._ctx, 'mmd_vit_encode')
        return out

    def visual_embed_frames(self, frames_u8: torch.Tensor, out: torch.Tensor = None):
        """image_processor.preprocess + visual_embed (test/inference.py:203,211) in one native pass per tower batch: uint8 [B,3,R,R] -> [B*frame_num_tokens, hidden].
        The resampled, normalised pixels are written straight into the patch-embed GEMM's operand (no pixel_values tensor); bit-identical to the two-call form."""
        fr = torch.as_tensor(frames_u8)
        if fr.dtype != torch.uint8 or fr.ndim != 4 or fr.shape[1] != 3 or fr.shape[2] != fr.shape[3]:
            raise ValueError(f'expected uint8 frames [T,3,R,R], got {tuple(fr.shape)} {fr.dtype}')
        fr = fr.to(self.device).contiguous()
        B, R = fr.shape[0], fr.shape[2]
        if out is None:
            out = torch.empty(B * self.tokens_per_frame, self.config.hidden_size, dtype=self.dtype, device=self.device)
        with self._lock:
            self._bind_stream()
            for b0 in range(0, B, self.max_vit_batch):
                b1 = min(B, b0 + self.max_vit_batch)
                check(lib().mmd_vit_encode_frames(self._ctx, _ptr(fr[b0:b1]), b1 - b0, R, _ptr(out[b0 * self.tokens_per_frame:])), self._ctx, 'mmd_vit_encode_frames')
        return out

    def connector_pool(self, tower_features: torch.Tensor, out: torch.Tensor = None):
        """visual_embed for pre-extracted tower features [B, vit_tokens, vit_hidden] (models/modeling_live.py:26-33 without `vision_encode`):
        mm_projector -> post_projector_pooling -> [B*frame_num_tokens, hidden]."""
        f = tower_features.to(device=self.device, dtype=self.dtype).contiguous()
        if f.ndim != 3 or f.shape[1] != self.config.vit_grid ** 2 or f.shape[2] != self.config.vit_hidden_size:
            raise ValueError(f'expected tower features [B, {self.config.vit_grid ** 2}, {self.config.vit_hidden_size}], got {tuple(f.shape)}')
        B = f.shape[0]
        if out is None:
            out = torch.empty(B * self.tokens_per_frame, self.config.hidden_size, dtype=self.dtype, device=self.device)
        with self._lock:
            self._bind_stream()
            for b0 in range(0, B, self.max_vit_batch):
                b1 = min(B, b0 + self.max_vit_batch)
                check(lib().mmd_connector_pool(self._ctx, _ptr(f[b0:b1]), b1 - b0, _ptr(out[b0 * self.tokens_per_frame:])), self._ctx, 'mmd_connector_pool')
        return out

    def tower_features(self, frames: torch.Tensor):
        """What the reference's `vision_encode(vision_encoder, frames)` returns for the LLaVA tower (models/live_llava/video_head_live_llava_qwen.py:96-98):
        [B, vit_tokens, vit_hidden] -- the tensor its offline feature extraction stores (data/utils.py:114)."""
        frames = frames.to(device=self.device, dtype=self.dtype).contiguous()
        B, T, C = frames.shape[0], self.config.vit_grid ** 2, self.config.vit_hidden_size
        out = torch.empty(B, T, C, dtype=self.dtype, device=self.device)
        scratch = torch.empty(min(B, self.max_vit_batch) * self.tokens_per_frame, self.config.hidden_size, dtype=self.dtype, device=self.device)
        with self._lock:
            self._bind_stream()
            was = int(lib().mmd_vit_get_full_tower(self._ctx))          # a caller's own set_full_tower(True) survives this call
            check(lib().mmd_vit_set_full_tower(self._ctx, 1), self._ctx, 'mmd_vit_set_full_tower')          # vision_encode's output is the tower over ALL tokens
            try:
                for b0 in range(0, B, self.max_vit_batch):
                    b1 = min(B, b0 + self.max_vit_batch)
                    check(lib().mmd_vit_encode(self._ctx, _ptr(frames[b0:b1]), b1 - b0, _ptr(scratch)), self._ctx, 'mmd_vit_encode')
                    check(lib().mmd_vit_debug_tap(self._ctx, 0, _ptr(out[b0:b1]), (b1 - b0) * T * C), self._ctx, 'mmd_vit_debug_tap')
            finally:
                check(lib().mmd_vit_set_full_tower(self._ctx, 1 if was > 0 else 0), self._ctx, 'mmd_vit_set_full_tower')
        return out

    def set_full_tower(self, on: bool):
        """Debug / feature extraction: have the tower compute its last layer (and the projector) for every token instead of the (2 out)^2 the bilinear pool reads;
        `vit_debug_tap` needs it switched on BEFORE the visual_embed call it inspects."""
        with self._lock:
            check(lib().mmd_vit_set_full_tower(self._ctx, 1 if on else 0), self._ctx, 'mmd_vit_set_full_tower')

    def vit_debug_tap(self, stage: int, B: int):
        n = B * self.config.vit_grid ** 2
        width = self.config.vit_hidden_size if stage == 0 else self.config.hidden_size
        out = torch.empty(n, width, dtype=self.dtype, device=self.device)
        with self._lock:
            self._bind_stream()
            check(lib().mmd_vit_debug_tap(self._ctx, stage, _ptr(out), out.numel()), self._ctx, 'mmd_vit_debug_tap')
        return out

    def joint_embed(self, input_ids: torch.Tensor = None, frames: torch.Tensor = None):
        """models/modeling_live.py:35-48."""
        if frames is None:
            return self._embed(input_ids)
        if input_ids is None:
            return self.visual_embed(frames)
        input_ids = input_ids.to(self.device)
        e = self._embed(input_ids.clamp(max=self.vocab_size - 1))
        mask = input_ids == self.config.v_placeholder_id
        if mask.any():
            e[mask] = self.visual_embed(frames).to(e.dtype)
        return e

    # ---- cache plumbing ---------------------------------------------------------------------------------------------
    def _resolve_cache(self, past_key_values):
        if past_key_values is None or (isinstance(past_key_values, KVCacheHandle) and past_key_values.arena is None):
            return _KVArena(self, self.kv_initial_tokens), 0
        if not isinstance(past_key_values, KVCacheHandle) or past_key_values.arena.model is not self:
            raise TypeError('past_key_values must be a KVCacheHandle produced by this model (or None)')
        if past_key_values.stale:
            raise RuntimeError('stale KV handle: the context it denoted was overwritten by a later forward from an older handle')
        arena = past_key_values.arena
        arena.truncate(past_key_values.length)
        return arena, past_key_values.length

    def cache_prefix(self, handle: KVCacheHandle, length: int):
        """Handle denoting the first `length` tokens of `handle`'s context (speculative multi-frame chunks)."""
        if length > handle.length:
            raise ValueError('prefix longer than the context')
        return KVCacheHandle(handle.arena, length)

    def kv_stash(self, handle: KVCacheHandle, start: int):
        """Set the KV of tokens [start, len(handle)) aside (device copy, stream-ordered); `kv_unstash` brings them back after something else used those slots."""
        if handle.stale:
            raise RuntimeError('stale KV handle')
        with self._lock:
            self._bind_stream()
            check(lib().mmd_kv_stash(handle.arena.h, int(start), int(handle.length)), self._ctx, 'mmd_kv_stash')
        return (handle.arena, int(start), int(handle.length))

    def kv_unstash(self, stash):
        """-> a handle denoting the context as it was when `kv_stash` was called (everything below `start` must be unchanged since)."""
        arena, start, end = stash
        with self._lock:
            arena.truncate(min(arena.length(), start))
            self._bind_stream()
            check(lib().mmd_kv_unstash(arena.h), self._ctx, 'mmd_kv_unstash')
        return KVCacheHandle(arena, end)

    def new_cache(self, initial_tokens=None):
        return KVCacheHandle(_KVArena(self, initial_tokens or self.kv_initial_tokens), 0)

    # ---- forward ----------------------------------------------------------------------------------------------------
    def __call__(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, frames=None,
                 return_dict=None, **kwargs):
        """models/live_llava/video_head_live_llava_qwen.py:121-205 (inference branch: labels must be None)."""
        if labels is not None:
            raise NotImplementedError('training losses are out of scope of the inference implementation')
        if inputs_embeds is None:
            inputs_embeds = self.joint_embed(input_ids, frames)
        if inputs_embeds.ndim != 3 or inputs_embeds.shape[0] != 1:
            raise ValueError(f'inputs_embeds must be [1, S, hidden] (streaming is batch 1), got {tuple(inputs_embeds.shape)}')
        x = inputs_embeds[0].to(device=self.device, dtype=self.dtype).contiguous()
        S = x.shape[0]
        hidden = torch.empty(S, self.config.hidden_size, dtype=self.dtype, device=self.device)
        with self._lock:
            arena, n = self._resolve_cache(past_key_values)
            self._bind_stream()
            for s0 in range(0, S, self.max_step_tokens):
                s1 = min(S, s0 + self.max_step_tokens)
                check(lib().mmd_llm_step(self._ctx, arena.h, _ptr(x[s0:s1]), s1 - s0, _ptr(hidden[s0:s1])), self._ctx, 'mmd_llm_step')
            cache = KVCacheHandle(arena, n + S)
        out = VideoHeadCausalLMOutputWithPast(self, hidden, cache)
        if return_dict is False:
            return (out.loss, out.logits, cache)
        return out

    forward = __call__

    def lm_head(self, hidden_rows: torch.Tensor):
        M = hidden_rows.shape[0]
        hidden_rows = hidden_rows.contiguous()
        out = torch.empty(M, self.config.vocab_size, dtype=torch.float32, device=self.device)
        if M:
            with self._lock:
                self._bind_stream()
                check(lib().mmd_lm_head(self._ctx, _ptr(hidden_rows), M, _ptr(out)), self._ctx, 'mmd_lm_head')
        return out

    def video_heads(self, hidden_rows: torch.Tensor):
        """-> [M,4] fp32 = informative(2) | relevance(2)."""
        M = hidden_rows.shape[0]
        hidden_rows = hidden_rows.contiguous()
        out = torch.empty(M, 4, dtype=torch.float32, device=self.device)
        if M:
            with self._lock:
                self._bind_stream()
                check(lib().mmd_video_heads(self._ctx, _ptr(hidden_rows), M, _ptr(out)), self._ctx, 'mmd_video_heads')
        return out

    # ---- fused fast paths used by mmduet_amd.inference (same results as __call__ + heads) -------------------------
    def frame_step(self, inputs_embeds: torch.Tensor, past_key_values, head_rows):
        """One LLM step over `inputs_embeds` [S,H] and the 4 video-head logits at `head_rows` (list of row indices).
        Returns (numpy-like list [[inf0, inf1, rel0, rel1], ...] as a float32 CPU tensor, new handle)."""
        x = inputs_embeds.reshape(-1, self.config.hidden_size).to(device=self.device, dtype=self.dtype).contiguous()
        S = x.shape[0]
        if S > self.max_step_tokens:
            raise ValueError(f'step of {S} tokens exceeds max_step_tokens={self.max_step_tokens}')
        rows = (C.c_int32 * len(head_rows))(*[int(r) for r in head_rows])
        res = (C.c_float * (4 * len(head_rows)))()
        with self._lock:
            arena, n = self._resolve_cache(past_key_values)
            self._bind_stream()
            check(lib().mmd_frame_step(self._ctx, arena.h, _ptr(x), S, rows, len(head_rows), res), self._ctx, 'mmd_frame_step')
            cache = KVCacheHandle(arena, n + S)
        return torch.tensor(list(res), dtype=torch.float32).view(-1, 4), cache

    def multi_step(self, segments, want_logits=True):
        """ONE causal forward over several video streams (mmd_frame_step_multi): the GEMMs run once over all rows, so a stream that
        is generating token by token rides on the other streams' frame chunks instead of streaming the weights for itself.

        segments: list of dicts with
            x          [S_i, H] / [1, S_i, H] input embeddings of this stream's rows
            cache      its KV handle (or None)
            head_rows  row indices (relative to the segment) whose 4 video-head logits are wanted
            hidden     'none' | 'last' | 'all': which final hidden rows to return (and, with want_logits, run lm_head on 'last')
        Returns, per segment, dict(heads=[n,4] fp32 CPU tensor | None, hidden=[r,H] device tensor | None,
        logits=[1,V] fp32 device tensor | None, cache=new handle)."""
        H = self.config.hidden_size
        xs, seg_rows, head_rows, hid_rows, hid_slices, arenas = [], [], [], [], [], []
        at = 0
        for sg in segments:
            x = sg['x'].reshape(-1, H).to(device=self.device, dtype=self.dtype)
            S = x.shape[0]
            if S == 0:
                raise ValueError('empty segment')
            xs.append(x); seg_rows.append(S)
            head_rows += [at + int(r) for r in sg.get('head_rows', ())]
            want = sg.get('hidden', 'none')
            rows = [at + S - 1] if want == 'last' else (list(range(at, at + S)) if want == 'all' else [])
            hid_slices.append((len(hid_rows), len(rows), want))
            hid_rows += rows
            at += S
        if at > self.max_step_tokens:
            raise ValueError(f'step of {at} tokens exceeds max_step_tokens={self.max_step_tokens}')
        x_all = torch.cat(xs, dim=0).contiguous() if len(xs) > 1 else xs[0].contiguous()
        n_seg = len(segments)
        hidden_out = torch.empty(max(1, len(hid_rows)), H, dtype=self.dtype, device=self.device)
        last_idx = [i for i, (o, n, w) in enumerate(hid_slices) if w == 'last']
        # lm_head runs over the 'last' rows only; when 'all' rows are mixed in, it is done per segment afterwards
        only_last = want_logits and last_idx and all(w in ('last', 'none') for (_, _, w) in hid_slices)
        logits = torch.empty(len(hid_rows), self.config.vocab_size, dtype=torch.float32, device=self.device) if only_last else None
        res = (C.c_float * (4 * max(1, len(head_rows))))()
        with self._lock:
            for sg in segments:
                arena, n = self._resolve_cache(sg.get('cache'))
                if any(a is arena for a, _ in arenas):
                    raise ValueError('a KV arena may appear once per multi_step')
                arenas.append((arena, n))
            self._bind_stream()
            streams = (C.c_void_p * n_seg)(*[a.h for a, _ in arenas])
            rows_c = (C.c_int32 * n_seg)(*seg_rows)
            hr = (C.c_int32 * max(1, len(head_rows)))(*head_rows)
            hd = (C.c_int32 * max(1, len(hid_rows)))(*hid_rows)
            check(lib().mmd_frame_step_multi(self._ctx, streams, rows_c, n_seg, _ptr(x_all), hr, len(head_rows), res, hd, len(hid_rows),
                                             _ptr(hidden_out), _ptr(logits) if logits is not None else None), self._ctx, 'mmd_frame_step_multi')
            caches = [KVCacheHandle(a, n + S) for (a, n), S in zip(arenas, seg_rows)]
        heads_all = torch.tensor(list(res)[:4 * len(head_rows)], dtype=torch.float32).view(-1, 4)
        out, hp = [], 0
        for i, sg in enumerate(segments):
            nh = len(sg.get('head_rows', ()))
            o, n, want = hid_slices[i]
            hid = hidden_out[o:o + n] if n else None
            lg = None
            if want == 'last' and want_logits:
                lg = logits[o:o + 1] if logits is not None else self.lm_head(hid)
            out.append(dict(heads=heads_all[hp:hp + nh] if nh else None, hidden=hid, logits=lg, cache=caches[i]))
            hp += nh
        return out

    def new_sampler(self):
        """Device-resident greedy sampling state of one stream (mmd_sampler): the token drawn last + the repetition-penalty list."""
        return SamplerHandle(self)

    def round_multi(self, segments):
        """ONE scheduler round over several video streams with the sampling on the device (mmd_round_multi): `multi_step` for the rows, plus -- for the streams that are
        talking -- lm_head, repetition penalty, arg-max and the next round's embedding gather without a host round trip.  One synchronisation per round.

        segments: list of dicts with
            x          [S_i, H] / [1, S_i, H] input rows, or None with feed=True
            cache      the stream's KV handle (or None)
            head_rows  row indices (relative to the segment) whose 4 video-head logits are wanted
            sampler    a SamplerHandle (needed for feed / sample)
            feed       the segment is the ONE row of the token `sampler` drew in an earlier round
            sample     draw the next token from the segment's last row
        Returns per segment dict(heads=[n,4] fp32 CPU tensor | None, token=int | None, cache=new handle)."""
        H = self.config.hidden_size
        n_seg = len(segments)
        seg_rows, head_rows, ptrs, flags, samplers, keep = [], [], [], [], [], []
        at = 0
        for sg in segments:
            feed, smp = bool(sg.get('feed')), sg.get('sampler')
            if feed:
                S = 1; ptrs.append(None)
            else:
                x = sg['x'].reshape(-1, H)
                if x.dtype != self.dtype or x.device != self.device or not x.is_contiguous():
                    x = x.to(device=self.device, dtype=self.dtype).contiguous()
                S = x.shape[0]
                if S == 0:
                    raise ValueError('empty segment')
                keep.append(x); ptrs.append(x.data_ptr())
            seg_rows.append(S)
            head_rows += [at + int(r) for r in sg.get('head_rows', ())]
            flags.append((1 if feed else 0) | (2 if sg.get('sample') else 0))
            samplers.append(smp.h if smp is not None else None)
            at += S
        if at > self.max_step_tokens:
            raise ValueError(f'round of {at} tokens exceeds max_step_tokens={self.max_step_tokens}')
        nh = len(head_rows)
        res = (C.c_float * (4 * max(1, nh)))()
        toks = (C.c_int64 * n_seg)()
        arenas = []
        with self._lock:
            for sg in segments:
                arena, n = self._resolve_cache(sg.get('cache'))
                if any(a is arena for a, _ in arenas):
                    raise ValueError('a KV arena may appear once per round')
                arenas.append((arena, n))
            self._bind_stream()
            check(lib().mmd_round_multi(self._ctx, (C.c_void_p * n_seg)(*[a.h for a, _ in arenas]), (C.c_int32 * n_seg)(*seg_rows), n_seg, (C.c_void_p * n_seg)(*ptrs),
                                        (C.c_void_p * n_seg)(*samplers), (C.c_int32 * n_seg)(*flags), (C.c_int32 * max(1, nh))(*head_rows), nh, res, toks),
                  self._ctx, 'mmd_round_multi')
            caches = [KVCacheHandle(a, n + S) for (a, n), S in zip(arenas, seg_rows)]
        heads_all = torch.tensor(res[:4 * nh], dtype=torch.float32).view(-1, 4) if nh else None
        out, hp = [], 0
        for i, sg in enumerate(segments):
            k = len(sg.get('head_rows', ()))
            out.append(dict(heads=heads_all[hp:hp + k] if k else None, token=int(toks[i]) if flags[i] & 2 else None, cache=caches[i]))
            hp += k
        return out

    def greedy_generate(self, inputs_embeds, past_key_values, eos_token_id, max_new_tokens, repetition_penalty=None,
                        generated_token_ids=None):
        x = inputs_embeds.reshape(-1, self.config.hidden_size).to(device=self.device, dtype=self.dtype).contiguous()
        S = x.shape[0]
        pen = float(repetition_penalty) if repetition_penalty is not None else 0.0
        prev = list(generated_token_ids) if (generated_token_ids is not None and pen > 0) else []
        cap = len(prev) + max_new_tokens + 1
        prev_arr = (C.c_int64 * cap)(*prev)
        n_prev = C.c_int(len(prev))
        out_ids = (C.c_int64 * max_new_tokens)()
        n_out = C.c_int(0)
        with self._lock:
            arena, n = self._resolve_cache(past_key_values)
            self._bind_stream()
            check(lib().mmd_greedy_generate(self._ctx, arena.h, _ptr(x), S, int(eos_token_id if eos_token_id is not None else -1), pen, prev_arr,
                                            C.byref(n_prev), cap, out_ids, int(max_new_tokens), C.byref(n_out)), self._ctx, 'mmd_greedy_generate')
            cache = KVCacheHandle(arena, arena.length())
        ids = [int(out_ids[i]) for i in range(n_out.value)]
        if generated_token_ids is not None and pen > 0:
            generated_token_ids[:] = [int(prev_arr[i]) for i in range(n_prev.value)]
        return ids, cache

    # ---- measurement --------------------------------------------------------------------------------------------------
    def prof_enable(self, classes=True):
        """classes: True = all kernel classes, False = off, or an iterable of class names from _lib.K_NAMES."""
        if classes is True:
            mask = (1 << len(_lib.K_NAMES)) - 1
        elif not classes:
            mask = 0
        else:
            mask = sum(1 << _lib.K_NAMES.index(k) for k in classes)
        check(lib().mmd_prof_enable(self._ctx, mask), self._ctx)

    def prof_set_stride(self, stride):
        check(lib().mmd_prof_set_stride(self._ctx, int(stride)), self._ctx)

    def prof_reset(self):
        check(lib().mmd_prof_reset(self._ctx), self._ctx)

    def prof_read(self):
        n = len(_lib.K_NAMES)
        ms, cnt, by, fl = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)()
        check(lib().mmd_prof_read(self._ctx, ms, cnt, by, fl), self._ctx)
        return {k: dict(ms=ms[i], launches=cnt[i], bytes=by[i], flops=fl[i]) for i, k in enumerate(_lib.K_NAMES)}


# ----------------------------------------------------------------------------------------------------------------
def fast_greedy_generate(*, model, inputs_embeds: torch.Tensor, past_key_values, eos_token_id: int, inplace_output_ids: torch.Tensor,
                         repetition_penalty=None, generated_token_ids=None):
    """models/modeling_live.py:51-77, same signature and return triple.  The token loop runs natively
    (mmd_greedy_generate): argmax (+ HF repetition penalty over `generated_token_ids`) on the device, EOS written but
    neither fed back nor penalised."""
    if repetition_penalty is not None:
        assert isinstance(repetition_penalty, float)
    if generated_token_ids is None:
        generated_token_ids = list()
    max_new = inplace_output_ids.size(1)
    if not hasattr(model, 'greedy_generate') or getattr(model, 'python_generate_loop', False):
        return _greedy_generate_by_calls(model, inputs_embeds, past_key_values, eos_token_id, inplace_output_ids,
                                         repetition_penalty, generated_token_ids)
    ids, cache = model.greedy_generate(inputs_embeds, past_key_values, eos_token_id, max_new, repetition_penalty, generated_token_ids)
    n = len(ids)
    inplace_output_ids[:, :n] = torch.tensor(ids, dtype=inplace_output_ids.dtype, device=inplace_output_ids.device)
    return inplace_output_ids[:, :n], cache, generated_token_ids


def _greedy_generate_by_calls(model, x, cache, eos_token_id, out_ids, penalty, seen):
    """Token loop through the public model call (any duck-typed model; also the cross-check of the native loop)."""
    n = 0
    for n in range(out_ids.size(1)):
        res = model(inputs_embeds=x, past_key_values=cache, use_cache=True, return_dict=True)
        cache = res.past_key_values
        scores = res.logits[0, -1].float().clone()
        if penalty is not None and seen:
            idx = torch.as_tensor(seen, dtype=torch.long, device=scores.device)
            picked = scores[idx]
            scores[idx] = torch.where(picked < 0, picked * penalty, picked / penalty)
        tok = int(scores.argmax(-1))
        if penalty is not None and tok != eos_token_id:
            seen.append(tok)
        out_ids[:, n] = tok
        if tok == eos_token_id:
            break
        x = model.get_input_embeddings()(torch.tensor([[tok]], device=out_ids.device))
    return out_ids[:, :n + 1], cache, seen


def build_live(*, is_training: bool, config_class=VideoHeadLiveLlavaQwenConfig, model_class=VideoHeadLiveLlavaQwenForCausalLM,
               llm_pretrained: str = None, lora_pretrained: str = None, finetune_modules=None, lora_modules: str = None,
               lora_r: int = None, lora_alpha: int = None, set_vision_inside: bool = False,
               attn_implementation: str = 'flash_attention_2', torch_dtype='auto', **kwargs):
    """models/modeling_live.py:80-129 for inference.  `llm_pretrained` is a local checkpoint directory / hub id of a
    LLaVA-OneVision-Qwen2 checkpoint, or 'synthetic:<seed>' for seeded random weights at the configured shape."""
    from .weights import load_pretrained_into, load_lora_into, synthetic_weights
    if is_training:
        raise NotImplementedError('training is out of scope of the MI355X inference implementation')
    if torch_dtype == 'auto' or torch_dtype is None:
        torch_dtype = torch.bfloat16
    runtime = {k: kwargs.pop(k) for k in ('max_vit_batch', 'max_step_tokens', 'kv_initial_tokens') if k in kwargs}
    # workspace sizing from the driver flags: rows of one LLM forward (frames_per_forward chunks) and the KV arena
    fpf = int(kwargs.get('frames_per_forward', 1) or 1)
    nt = int(kwargs.get('frame_num_tokens', 49) or 49)
    runtime.setdefault('max_step_tokens', max(1024, fpf * nt + 256))
    if kwargs.get('kv_capacity_tokens'):
        runtime.setdefault('kv_initial_tokens', int(kwargs['kv_capacity_tokens']))
    elif kwargs.get('max_num_frames'):
        runtime.setdefault('kv_initial_tokens', max(32768, int(kwargs['max_num_frames']) * nt + 4096))
    if llm_pretrained.startswith('synthetic'):
        config = config_class(**kwargs)
    else:
        config = config_class.from_pretrained(llm_pretrained, **kwargs)
    model = model_class(config, torch_dtype=torch_dtype, **runtime)
    tokenizer = build_live_tokenizer_and_update_config(llm_pretrained, model.config)
    if llm_pretrained.startswith('synthetic'):
        seed = int(llm_pretrained.split(':')[1]) if ':' in llm_pretrained and llm_pretrained.split(':')[1].isdigit() else 0
        for name, t in synthetic_weights(config, seed=seed, device=model.device, dtype=torch_dtype):
            model.load_tensor(name, t)
    else:
        load_pretrained_into(model, llm_pretrained)
    if lora_pretrained:
        load_lora_into(model, lora_pretrained)
    else:
        import warnings
        warnings.warn(f'!!! Fail to load lora from checkpoint: {lora_pretrained}. Return a new initialized model.')
    model.finalize()
    return model, tokenizer


def build_model_and_tokenizer(is_training, **kwargs):
    """models/__init__.py:8-13."""
    llm_pretrained = kwargs.get('llm_pretrained', None)
    if llm_pretrained is not None and ('llava' in llm_pretrained or llm_pretrained.startswith('synthetic')):
        return build_live(is_training=is_training, **kwargs)
    raise NotImplementedError(f'Not support {llm_pretrained}')
