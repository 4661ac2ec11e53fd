"""Run the REFERENCE's stream driver classes, unchanged, over the product's Python surface.  TEST INFRASTRUCTURE, build container only
(needs /root/reference; started as a subprocess by tests/test_reference_driver_conformance.py, prints one JSON document).

What is real and what is replaced:
  real      /root/reference/test/inference.py::LiveInferForBenchmark (its own __init__ included) and demo/liveinfer.py::LiveInferForDemo;
            the `models` package they import = the two lines INTEGRATION.md section 1 prescribes (read from that file);
            mmduet_amd as shipped: build_model_and_tokenizer -> build_live -> config.from_pretrained / safetensors loader / AutoTokenizer,
            VideoHeadLiveLlavaQwenForCausalLM, KVCacheHandle + arena pool, lazy logits, fast_greedy_generate (its Python token loop).
  replaced  libmmduet_hip.so by tests/cabi_oracle_shim.FakeLib (the same C entry points computed by the oracle on the CPU);
            torch.cuda.* / 'cuda' device strings (no GPU here); packages the reference imports but the image lacks (tests/golden/ref_harness stubs);
            the tokenizer returns plain tensors from apply_chat_template as the reference's pinned transformers 4.44.2 does (5.x returns a BatchEncoding).
  poked     driver.eos_token_id (the fixture's synthetic eos) and driver.inplace_output_ids (12 tokens; the reference hard-codes 200) -- the same two
            attributes tests/golden/make_golden.py set when it recorded cfgA_streams.json from the reference model."""
import json, os, sys, tempfile, types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, 'golden'))
import numpy as np
import torch


class CudaIsCpu(torch.overrides.TorchFunctionMode):
    """device='cuda' / 'cuda:0' / torch.device('cuda', i) -> cpu, in keyword and positional arguments."""

    @staticmethod
    def _fix(a):
        if isinstance(a, str) and a.startswith('cuda'):
            return 'cpu'
        if isinstance(a, torch.device) and a.type == 'cuda':
            return torch.device('cpu')
        return a

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = {k: self._fix(v) for k, v in (kwargs or {}).items()}
        return func(*tuple(self._fix(a) for a in args), **kwargs)


def main():
    import ref_harness as R
    R.install()                                             # stubs for peft / torchvision / cv2 / llava, /root/reference on sys.path
    # ---- no GPU: torch.cuda answers as a one-device box, the shared library is the oracle-backed shim ----
    import mmduet_amd, mmduet_amd._lib as L, mmduet_amd.modeling_live as ML
    from cabi_oracle_shim import FakeLib
    fake = FakeLib()
    L.lib = ML.lib = lambda: fake
    stream = types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None)
    torch.cuda.is_available = lambda: True
    torch.cuda.current_device = lambda: 0
    torch.cuda.current_stream = lambda device=None: stream
    torch.cuda.empty_cache = lambda: None
    torch.cuda.synchronize = lambda device=None: None
    # ---- `models` = INTEGRATION.md section 1, verbatim ----
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = text[text.index('```python\n# models/__init__.py'):]
    block = block[len('```python\n'):block.index('\n```')]
    models = types.ModuleType('models'); models.__path__ = []
    exec(compile(block, 'INTEGRATION.md#models/__init__.py', 'exec'), models.__dict__)
    assert models.build_model_and_tokenizer is mmduet_amd.build_model_and_tokenizer and models.fast_greedy_generate is mmduet_amd.fast_greedy_generate
    sys.modules['models'] = models
    import importlib
    TI = importlib.import_module('test.inference')          # the reference's driver module, from /root/reference
    DL = importlib.import_module('demo.liveinfer')
    assert TI.__file__.startswith('/root/reference') and TI.build_model_and_tokenizer is mmduet_amd.build_model_and_tokenizer

    # ---- a checkpoint directory of the tiny golden model: config.json + model.safetensors + tokenizer files (the product's real loaders read it) ----
    from conftest import load_golden_weights, GOLDEN
    from helpers import product_config
    from mmduet_amd.weights import save_checkpoint
    from mmduet_amd.tokenization_live import build_byte_level_tokenizer
    cfgd, w = load_golden_weights('A')
    ckpt = os.path.join(tempfile.mkdtemp(prefix='mmduet_conf_'), 'llava-tiny-cfgA')
    save_checkpoint(w, ckpt, product_config(cfgd))
    build_byte_level_tokenizer().save_pretrained(ckpt)
    meta = json.load(open(os.path.join(GOLDEN, 'cfgA_streams.json')))

    gen_log = []
    orig_fgg = TI.fast_greedy_generate

    def fgg_log(**kw):
        o = orig_fgg(**kw)
        gen_log.append(o[0][0].tolist())
        return o
    TI.fast_greedy_generate = fgg_log                       # (the name the reference driver calls; still the product's function underneath)

    def make(cls, case):
        opts = case['opts']
        args = mmduet_amd.LiveTestArguments(
            llm_pretrained=ckpt, frame_fps=case['fps'], system_prompt=meta['system_prompt'], bf16=False, fp16=False,
            frame_num_tokens=cfgd['frame_num_tokens'], video_pooling_stride=cfgd['video_pooling_stride'], frame_resolution=cfgd['frame_resolution'],
            stream_end_prob_threshold=opts.get('stream_end_prob_threshold'), stream_end_score_sum_threshold=opts.get('stream_end_score_sum_threshold'),
            score_heads=opts.get('score_heads', 'informative_score'), remove_assistant_turns=opts.get('remove_assistant_turns', False),
            repetition_penalty=opts.get('repetition_penalty'), running_list_length=opts.get('running_list_length', 20))
        d = cls(args)                                       # the reference's own constructor: build_model_and_tokenizer(**asdict(args)), eval, image_processor, config reads, templates
        assert type(d.model) is mmduet_amd.VideoHeadLiveLlavaQwenForCausalLM
        d.model.python_generate_loop = True                 # the token loop of mmduet_amd.modeling_live through the public model call (mmd_greedy_generate is its native twin)
        tok = d.tokenizer
        act = tok.apply_chat_template
        tok.apply_chat_template = lambda msgs, **kw: act(msgs, return_dict=False, **kw) if kw.get('return_tensors') else act(msgs, **kw)
        for name in ('_start_ids', '_added_stream_prompt_ids', '_added_stream_generation_ids'):          # built in __init__ before the 4.44 shim: redo them through it
            flags = {'_start_ids': {}, '_added_stream_prompt_ids': dict(add_stream_prompt=True), '_added_stream_generation_ids': dict(add_stream_generation_prompt=True)}[name]
            msgs = [{'role': 'system', 'content': d.system_prompt}] if name == '_start_ids' else [{}]
            setattr(d, name, tok.apply_chat_template(msgs, return_tensors='pt', **flags))
        d.eos_token_id = meta['eos_token_id']
        d.inplace_output_ids = torch.zeros(1, 12, dtype=torch.long)
        return d

    out = {'benchmark': {}, 'demo': {}, 'calls': None}
    with CudaIsCpu(), torch.no_grad():
        for name, case in meta['cases'].items():
            frames = torch.from_numpy(np.load(os.path.join(GOLDEN, f'stream_{name}_frames.npy')))
            # (a) LiveInferForBenchmark.inference()
            del gen_log[:]
            d = make(TI.LiveInferForBenchmark, case)
            d.input_video_stream(frames)
            d.input_query_stream(case['conversation'])
            responses = d.inference()
            out['benchmark'][name] = dict(debug_data=d.debug_data_list, responses=responses, generated=list(gen_log), final_kv_len=len(d.past_key_values),
                                          penalty_ids=[int(x) for x in d.generated_token_ids], handle_type=type(d.past_key_values).__name__)
            # (b) LiveInferForDemo: the page feeds one frame per call and user messages from another callback (demo/app.py); same order of events
            del gen_log[:]
            d = make(DL.LiveInferForDemo, case)
            d.input_video_stream(frames)
            queries = sorted((t['time'], t['content']) for t in case['conversation'] if t['role'] == 'user')
            rows = []
            while d.frame_embeds_queue:
                if queries and d.video_time >= queries[0][0]:
                    d.encode_given_query(queries.pop(0)[1])
                rows.append(d.input_one_frame())
            out['demo'][name] = dict(rows=rows, generated=list(gen_log), final_kv_len=len(d.past_key_values))
    out['calls'] = fake.calls
    print('CONFORMANCE_JSON ' + json.dumps(out))


if __name__ == '__main__':
    main()
