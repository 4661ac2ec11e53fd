import json, os
from conftest import GOLDEN
from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
from mmduet_amd.tokenization_live import build_live_tokenizer_and_update_config, chat_ids


def test_chat_template_matches_reference_renders():
    g = json.load(open(os.path.join(GOLDEN, 'templates.json')))
    cfg = VideoHeadLiveLlavaQwenConfig(frame_num_tokens=g['frame_num_tokens'], v_placeholder=g['v_placeholder'])
    tok = build_live_tokenizer_and_update_config('synthetic:x', cfg)
    assert cfg.eos_token_id == tok.convert_tokens_to_ids('<|im_end|>') and cfg.v_placeholder_id == tok.convert_tokens_to_ids('<image>')
    for case in g['cases']:
        assert tok.apply_chat_template(case['messages'], tokenize=False, **case['flags']) == case['text']
        assert chat_ids(tok, case['messages'], **case['flags'])[0].tolist() == case['ids']


def test_decode_roundtrip_and_specials():
    cfg = VideoHeadLiveLlavaQwenConfig(frame_num_tokens=2, v_placeholder='<image>')
    tok = build_live_tokenizer_and_update_config('synthetic:x', cfg)
    ids = chat_ids(tok, [{'role': 'user', 'content': 'héllo wörld'}], add_stream_prompt=True)[0]
    assert tok.decode(ids, skip_special_tokens=True) == '\nuser\nhéllo wörld\nstream\n'
    assert tok.bos_token == '<|im_start|>' and tok.eos_token == '<|im_end|>'
