"""Model configuration for the MI355X-native LiveLlava (video-head) model.

API surface mirrored from the reference: `VideoHeadLiveLlavaQwenConfig` (models/live_llava/video_head_live_llava_qwen.py:41-45)
with the fields of `VideoHeadLiveConfigMixin` (models/configuration_live.py:22-36).  The stream driver reads
`hidden_size, frame_resolution, frame_num_tokens, v_placeholder, eos_token_id` (test/inference.py:33-38,60); the
model additionally uses `video_pooling_stride, mm_spatial_pool_mode, v_placeholder_id, vocab_size`.

Like the reference (`config_class.from_pretrained(llm_pretrained, **kwargs)`, models/modeling_live.py:96-97) every
unknown keyword is accepted and stored as an attribute, so the whole LiveTestArguments dataclass can be splatted in.

The shape of the SigLIP tower is not in LLaVA's config.json (LLaVA-NeXT hard-codes it in its SigLipVisionConfig
[3P-recalled]); it is carried here as `vit_*` fields whose defaults are google/siglip-so400m-patch14-384 with the last
encoder layer removed (26 of 27 run) and no post_layernorm -- both are load-time options.
"""
from transformers import PretrainedConfig


class VideoHeadLiveLlavaQwenConfig(PretrainedConfig):
    model_type = 'llava_qwen'

    def __init__(self, *,
                 # Qwen2 decoder (defaults: lmms-lab/llava-onevision-qwen2-7b-ov)
                 vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                 num_attention_heads=28, num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6,
                 max_position_embeddings=32768, tie_word_embeddings=False,
                 # SigLIP tower
                 vit_hidden_size=1152, vit_intermediate_size=4304, vit_num_hidden_layers=27, vit_layers_removed=1,
                 vit_num_attention_heads=16, vit_image_size=384, vit_patch_size=14, vit_layer_norm_eps=1e-6,
                 vit_post_layernorm=False,
                 # connector / pooling
                 video_pooling_stride=4, video_head_stop_grad=False, mm_spatial_pool_mode='bilinear',
                 # live mixin
                 vision_pretrained=None, frame_resolution=None, frame_token_cls=None, frame_token_pooled=None,
                 frame_num_tokens=None, v_placeholder='<v>', v_placeholder_id=None, vision_hidden_size=1024,
                 bos_token_id=None, eos_token_id=None, pad_token_id=None,
                 **kwargs):
        rope_params = kwargs.pop('rope_parameters', None)          # transformers>=5 checkpoints
        if isinstance(rope_params, dict) and 'rope_theta' in rope_params:
            rope_theta = rope_params['rope_theta']
        kwargs.pop('rope_scaling', None)                             # forced to None by the reference (:73)
        super().__init__(tie_word_embeddings=tie_word_embeddings, **kwargs)
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.rope_theta = float(rope_theta)
        self.rms_norm_eps = rms_norm_eps
        self.max_position_embeddings = max_position_embeddings
        self.vit_hidden_size = vit_hidden_size
        self.vit_intermediate_size = vit_intermediate_size
        self.vit_num_hidden_layers = vit_num_hidden_layers
        self.vit_layers_removed = vit_layers_removed
        self.vit_num_attention_heads = vit_num_attention_heads
        self.vit_image_size = vit_image_size
        self.vit_patch_size = vit_patch_size
        self.vit_layer_norm_eps = vit_layer_norm_eps
        self.vit_post_layernorm = vit_post_layernorm
        self.video_pooling_stride = video_pooling_stride
        self.video_head_stop_grad = video_head_stop_grad
        self.mm_spatial_pool_mode = mm_spatial_pool_mode
        self.vision_pretrained = vision_pretrained
        self.frame_resolution = frame_resolution
        self.frame_token_cls = frame_token_cls
        self.frame_token_pooled = frame_token_pooled
        self.frame_num_tokens = frame_num_tokens
        self.vision_hidden_size = vision_hidden_size
        self.v_placeholder = v_placeholder
        self.v_placeholder_id = v_placeholder_id

    # derived shapes ---------------------------------------------------------------------------------------------
    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def vit_layers_run(self):
        return self.vit_num_hidden_layers - self.vit_layers_removed

    @property
    def vit_grid(self):
        return self.vit_image_size // self.vit_patch_size
