"""The C-ABI library loads here (no GPU needed) and exports every symbol include/mmduet.h declares."""
import ctypes, os, re
import pytest
from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mmduet.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mmd_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from mmduet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, set(names) ^ set(_lib.EXPORTED_SYMBOLS)


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    from mmduet_amd import _lib
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MmduetError):
            VideoHeadLiveLlavaQwenForCausalLM(VideoHeadLiveLlavaQwenConfig(frame_num_tokens=49))
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmmduet_hip.so')
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.MmduetError):
        _lib.lib()


def test_parse_args_accepts_reference_flag_sets():
    from mmduet_amd import parse_args
    # scripts/inference/youcook2.sh-style flags plus an unknown HF TrainingArguments flag
    a = parse_args('test', ['--bf16', 'true', '--stream_end_score_sum_threshold', '2', '--remove_assistant_turns', 'true',
                            '--score_heads', 'informative_score,relevance_score', '--frame_fps', '0.5', '--max_num_frames', '400',
                            '--per_device_eval_batch_size', '1', '--lora_pretrained', 'outputs/x'])
    assert a.bf16 and a.remove_assistant_turns and a.stream_end_score_sum_threshold == 2.0 and a.frame_fps == 0.5
    assert a.stream_end_prob_threshold is None and a.lora_pretrained == 'outputs/x' and a.frame_num_tokens == 49
