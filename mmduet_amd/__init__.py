"""mmduet_amd -- MI355X-native (gfx950) implementation of MMDuet's streaming video-text-duet forward path.

Drop-in surface (reference: models/__init__.py:3-20):
    from mmduet_amd import build_model_and_tokenizer, fast_greedy_generate, parse_args
The arithmetic lives in mmduet_amd/csrc (libmmduet_hip.so, C ABI in include/mmduet.h); see DESIGN.md / INTEGRATION.md.
"""
from .arguments_live import LiveTrainingArguments, LiveTestArguments, get_args_class, parse_args
from .configuration_live import VideoHeadLiveLlavaQwenConfig
from .modeling_live import (VideoHeadLiveLlavaQwenForCausalLM, VideoHeadCausalLMOutputWithPast, KVCacheHandle,
                            build_live, build_model_and_tokenizer, fast_greedy_generate)
from .tokenization_live import build_live_tokenizer_and_update_config

__all__ = ['LiveTrainingArguments', 'LiveTestArguments', 'get_args_class', 'parse_args', 'VideoHeadLiveLlavaQwenConfig',
           'VideoHeadLiveLlavaQwenForCausalLM', 'VideoHeadCausalLMOutputWithPast', 'KVCacheHandle', 'build_live',
           'build_model_and_tokenizer', 'fast_greedy_generate', 'build_live_tokenizer_and_update_config']
