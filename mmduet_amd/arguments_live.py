"""Command-line / keyword arguments of the stream driver.

Mirrors the flag set of the reference's `LiveTestArguments(LiveTrainingArguments(TrainingArguments))`
(models/arguments_live.py:5-55) for the inference path.  The reference inherits the ~150 fields of HF
`TrainingArguments`; only `bf16`, `fp16` and `output_dir` of those are read on this path (test/inference.py:22-23),
so they are declared here directly and the training-only machinery is left out (training is out of scope).
`parse_args` keeps the reference signature (models/__init__.py:15-20) and ignores unknown flags so the reference's
shell recipes (scripts/inference/*.sh) run unchanged.
"""
import argparse
from dataclasses import dataclass, field, fields, MISSING
from typing import Optional, List


@dataclass
class LiveTrainingArguments:
    live_version: str = 'live1+'
    dataset_config: Optional[str] = None
    stream_loss_weight: float = 1.0
    llm_pretrained: str = 'lmms-lab/llava-onevision-qwen2-7b-ov'
    vision_pretrained: str = 'google/siglip-large-patch16-384'
    lora_pretrained: Optional[str] = None
    lora_modules: str = r"model\.layers.*(q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj)$"
    lora_r: int = 16
    lora_alpha: int = 32
    finetune_modules: List[str] = field(default_factory=lambda: ['connector', 'mm_projector', 'response_head', 'related_head'])
    frame_fps: float = 2
    frame_token_cls: bool = False
    frame_token_pooled: List[int] = field(default_factory=lambda: [7, 7])
    frame_num_tokens: int = 49
    video_pooling_stride: int = 4
    frame_resolution: int = 384
    embed_mark: str = '2fps_384_1+3x3'
    v_placeholder: str = '<image>'
    max_num_frames: int = 100
    augmentation: bool = False
    attn_implementation: str = 'flash_attention_2'     # accepted for compatibility; the HIP attention kernel is always used
    output_dir: str = 'outputs/debug'
    bf16: bool = False
    fp16: bool = False


@dataclass
class LiveTestArguments(LiveTrainingArguments):
    system_prompt: str = (
        "A multimodal AI assistant is helping users with some activities."
        " Below is their conversation, interleaved with the list of video frames received by the assistant."
    )
    live_version: str = 'test'
    is_online_model: bool = True
    grounding_mode: bool = False
    input_dir: str = 'datasets/shot2story/videos/'
    test_fname: str = ''
    output_fname: str = ''
    repetition_penalty: Optional[float] = None
    stream_end_prob_threshold: Optional[float] = None
    response_min_interval_frames: Optional[int] = None
    threshold_z: Optional[float] = None
    first_n_frames_no_generate: int = 0
    consecutive_n_frames_threshold: int = 1
    running_list_length: int = 20
    start_idx: int = 0
    end_idx: Optional[int] = None
    time_instruction_format: Optional[str] = None
    stream_end_score_sum_threshold: Optional[float] = None
    remove_assistant_turns: bool = False
    score_heads: str = 'informative_score'
    # --- additions of this implementation (all default to the reference behaviour) -------------------------------
    frames_per_forward: int = 1          # speculative multi-frame causal chunks (DESIGN.md "chunked stepping")
    kv_capacity_tokens: int = 0          # 0 = size the KV arena from max_num_frames
    max_new_tokens: int = 200            # test/inference.py:42 uses a 200-wide output buffer
    overlap_vision: bool = True          # encode frames on a side HIP stream, overlapping the LLM steps
    num_workers: int = 4                 # clip loader threads of the CLI (the reference: DataLoader(num_workers=4), test/inference.py:341); 0 = load inline, no prefetch
    streams_per_gpu: int = 1             # > 1: that many videos share each LLM forward (mmduet_amd/multistream.py)
    evaluator_format: bool = False       # write debug_data in the shape test/evaluate.py reads (results.result_record)
    features_dir: Optional[str] = None   # entries of --test_fname that carry "features": "<file>" read a pre-extracted feature file from here (mmduet_amd/features.py)
    weight_dtype: Optional[str] = None   # 'fp8_e4m3': decoder linear layers stored as per-channel-scaled OCP e4m3 (bf16 activations, fp32 accumulate)
    tower_dtype: Optional[str] = None    # 'fp16': the vision tower computes in IEEE half as under the reference's torch.cuda.amp.autocast() (models/modeling_live.py:28); None = model dtype


def get_args_class(args_version: str):
    if args_version == 'train':
        return LiveTrainingArguments
    if args_version == 'test':
        return LiveTestArguments
    raise NotImplementedError(args_version)


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('true', '1', 'yes', 'y'):
        return True
    if v.lower() in ('false', '0', 'no', 'n'):
        return False
    raise argparse.ArgumentTypeError(f'expected a boolean, got {v!r}')


def _parser_for(cls):
    p = argparse.ArgumentParser(allow_abbrev=False)
    for f in fields(cls):
        default = f.default if f.default is not MISSING else f.default_factory()
        tp = f.type
        if tp is bool or isinstance(default, bool):
            p.add_argument(f'--{f.name}', type=_str2bool, nargs='?', const=True, default=default)
        elif isinstance(default, list):
            et = type(default[0]) if default else str
            p.add_argument(f'--{f.name}', type=et, nargs='+', default=default)
        else:
            s = str(tp)             # the annotation decides (`frame_fps: float = 2`), Optional[...] included
            base = float if 'float' in s else int if 'int' in s else str
            p.add_argument(f'--{f.name}', type=base, default=default)
    return p


def parse_args(live_version=None, argv=None):
    """models/__init__.py:15-20: pick the dataclass from --live_version (or the argument), parse, return the dataclass."""
    if live_version is None:
        pre = argparse.ArgumentParser(add_help=False, allow_abbrev=False)
        pre.add_argument('--live_version', default=LiveTrainingArguments.live_version)
        live_version = pre.parse_known_args(argv)[0].live_version
        if live_version not in ('train', 'test'):
            live_version = 'train'
    cls = get_args_class(live_version)
    ns, _unknown = _parser_for(cls).parse_known_args(argv)
    return cls(**vars(ns))
