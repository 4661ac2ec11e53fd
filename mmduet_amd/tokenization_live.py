"""Tokenizer + duet chat template (host side of the drop-in boundary).

Behaviour mirrored from the reference's models/tokenization_live.py:
  * `<image>` (config.v_placeholder) is added as a special token, bos/eos are overridden to
    `<|im_start|>` / `<|im_end|>`, and `v_placeholder_id` / `eos_token_id` are written back into the model
    config (:115-124);
  * the chat template knows the roles system / user / assistant / stream and the four prompt switches
    `add_generation_prompt`, `add_stream_prompt`, `add_stream_query_prompt`, `add_stream_generation_prompt` (:34-63);
  * `tokenizer.get_learn_ranges` (character ranges of the assistant turns, :96-112) is attached for API parity.

The template below is written from the format specification (one turn = "\n<|im_start|>{role}\n{body}<|im_end|>",
the system turn has no leading newline, a stream turn's body is frame_num_tokens*num_frames placeholders); its
renders are checked string-for-string against the reference template in tests/test_tokenization.py.

There are no Qwen2 tokenizer files offline, so besides `AutoTokenizer.from_pretrained` this module can build a
self-contained byte-level tokenizer (`build_byte_level_tokenizer`) with the same special tokens; it is selected with
`llm_pretrained='synthetic:...'` and is what the benchmarks / tests use.
"""
from functools import partial


def duet_chat_template(v_placeholder: str, frame_num_tokens: int) -> str:
    """Jinja template of the video-text duet format."""
    frame = v_placeholder * frame_num_tokens
    return (
        "{%- macro turn(role, body) -%}{{ '\n' + bos_token + role + '\n' + body + eos_token }}{%- endmacro -%}"
        "{%- set ns = namespace(first=true) -%}"
        "{%- for m in messages -%}"
        "{%- if loop.first and m['role'] == 'system' -%}"
        "{{ bos_token + 'system\n' + m['content'] + eos_token }}"
        "{%- elif m['role'] == 'user' -%}"
        "{%- if add_stream_query_prompt -%}{{ eos_token }}{%- endif -%}"
        "{{ turn('user', m['content']) }}"
        "{%- elif m['role'] == 'assistant' -%}"
        "{{ turn('assistant', m['content']) }}"
        "{%- elif m['role'] == 'stream' and m['num_frames'] > 0 -%}"
        "{{ '\n' + bos_token + 'stream\n' }}{%- for _ in range(m['num_frames']) -%}{{ FRAME }}{%- endfor -%}{{ eos_token }}"
        "{%- endif -%}"
        "{%- endfor -%}"
        "{%- if add_generation_prompt -%}{{ '\n' + bos_token + 'assistant\n' }}"
        "{%- elif add_stream_prompt -%}{{ '\n' + bos_token + 'stream\n' }}"
        "{%- elif add_stream_generation_prompt -%}{{ eos_token + '\n' + bos_token + 'assistant\n' }}"
        "{%- endif -%}"
    ).replace('FRAME', repr(frame))


def transition_lengths(tokenizer) -> dict:
    """Character length of the text emitted between two consecutive roles (models/tokenization_live.py:66-84)."""
    bos, eos = tokenizer.bos_token, tokenizer.eos_token
    roles = ('system', 'user', 'assistant', 'stream')
    table = {(None, 'system'): len(f'{bos}system\n'), 'assistant': len(f'{bos}assistant\n'), 'eos_token': len(eos)}
    for prev in roles:
        for nxt in ('user', 'assistant', 'stream'):
            table[(prev, nxt)] = len(f'{eos}\n{bos}{nxt}\n')
    return table


def get_learn_ranges(conversation, *, chat_template_offsets, model_config):
    """Character ranges (in the rendered prompt) of assistant turns flagged `learn` (models/tokenization_live.py:96-112)."""
    pos, prev, ranges = 0, None, []
    for msg in conversation:
        role = msg['role']
        pos += chat_template_offsets[(prev, role)]
        prev = role
        if role == 'stream':
            pos += msg['num_frames'] * model_config.frame_num_tokens * len(model_config.v_placeholder)
            continue
        if role == 'assistant' and msg.get('learn', False):
            ranges.append(range(pos, pos + len(msg['content']) + chat_template_offsets['eos_token']))
        pos += len(msg['content'])
    return ranges


QWEN_SPECIALS = ('<|endoftext|>', '<|im_start|>', '<|im_end|>')


def build_byte_level_tokenizer():
    """A dependency-free stand-in for the Qwen2 tokenizer: 256 byte symbols + the Qwen specials (ids 256..258)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders
    from transformers import PreTrainedTokenizerFast
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    core = Tokenizer(models.BPE(vocab={c: i for i, c in enumerate(alphabet)}, merges=[]))
    core.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    core.decoder = decoders.ByteLevel()
    tok = PreTrainedTokenizerFast(tokenizer_object=core, padding_side='left', clean_up_tokenization_spaces=False)
    tok.add_special_tokens({'additional_special_tokens': list(QWEN_SPECIALS)})
    return tok


def build_live_tokenizer_and_update_config(llm_pretrained: str, model_config):
    """models/tokenization_live.py:115-134."""
    if llm_pretrained.startswith('synthetic'):
        tokenizer = build_byte_level_tokenizer()
    elif 'llava' in llm_pretrained:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(llm_pretrained, use_fast=True, padding_side='left')
    else:
        raise NotImplementedError(f'Not support {llm_pretrained}')
    tokenizer.add_special_tokens({'additional_special_tokens': [model_config.v_placeholder]})
    tokenizer.bos_token, tokenizer.eos_token = '<|im_start|>', '<|im_end|>'
    for key, val in dict(v_placeholder_id=tokenizer.convert_tokens_to_ids(model_config.v_placeholder),
                         eos_token_id=tokenizer.eos_token_id).items():
        setattr(model_config, key, val)
    tokenizer.chat_template = duet_chat_template(model_config.v_placeholder, model_config.frame_num_tokens)
    tokenizer.get_learn_ranges = partial(get_learn_ranges, chat_template_offsets=transition_lengths(tokenizer),
                                         model_config=model_config)
    return tokenizer


def chat_ids(tokenizer, messages, **flags):
    """apply_chat_template -> LongTensor [1, k].  transformers>=5 returns a BatchEncoding by default where 4.44 (the
    reference's pin) returned the tensor; accept both."""
    out = tokenizer.apply_chat_template(messages, return_tensors='pt', **flags)
    if hasattr(out, 'keys') and 'input_ids' in out:
        out = out['input_ids']
    return out
