"""Whole-stream checker: the oracle behind the stream driver's duck-type, teacher-forced.  TEST INFRASTRUCTURE -- imported by tests/test_gpu_fullsize.py and by
bench.py's post-timing self-check (`resp_head_logit_delta.measured_in_run`), there only as the checker: nothing here is timed or shipped, the product never imports it.

`StreamOracle` is oracle.duet_oracle.OracleModel (device-agnostic torch: on the GPU it runs through torch's own fp32 / bf16 kernels -- an implementation independent of
libmmduet_hip) with two changes that do not touch the arithmetic being checked: lm_head only on the rows that are read (the reference's all-position lm_head is 1.4 TF
of fp32 per chunk that nothing reads, models/live_llava/video_head_live_llava_qwen.py:155), and responses TEACHER-FORCED with the product's token ids -- greedy decoding on
random-init weights is tie-fragile, so instead of comparing free-running ids the oracle scores the product's ids: one causal forward over prompt + response, row i of the
logits is what step i of models/modeling_live.py:51-77 would have seen."""
import time
import torch
from . import duet_oracle as O


class StreamOracle(O.OracleModel):
    def __init__(self, cfg, weights, device):
        super().__init__(cfg, weights)
        self.device = device
        self.forced, self.resp, self.tf = [], 0, []

    def __call__(self, inputs_embeds=None, past_key_values=None, logit_rows=1, **kw):
        h, cache = O.llm_forward(self.w, self.cfg, inputs_embeds[0].to(self.dtype), past_key_values)
        return O.OracleOutput(logits=O.linear(h[-logit_rows:], self.w['lm_head.weight']).float()[None],
                              informative_logits=O.linear(h, self.w['informative_head.weight']).float()[None],
                              relevance_logits=O.linear(h, self.w['relevance_head.weight']).float()[None], past_key_values=cache)

    def greedy_generate(self, inputs_embeds, past_key_values, eos_token_id, max_new_tokens, repetition_penalty=None, generated_token_ids=None):
        """models/modeling_live.py:51-77 with the token choice given: prompt + ids[:-1] in ONE causal forward (the last token is written, never fed,
        :68-75); row i of the logits is what the loop's step i would have seen."""
        assert repetition_penalty is None
        ids = list(self.forced[self.resp]); self.resp += 1
        x = inputs_embeds.reshape(1, -1, self.cfg.hidden_size).to(self.dtype)
        if len(ids) > 1:
            x = torch.cat([x, self._embed(torch.tensor([ids[:-1]], device=self.device))], 1)
        out = self(inputs_embeds=x, past_key_values=past_key_values, logit_rows=len(ids))
        lg = out.logits[0]
        t = torch.tensor(ids, device=self.device)
        top2 = lg.topk(2, dim=-1).values
        self.tf.append(dict(deficit=(top2[:, 0] - lg.gather(1, t[:, None])[:, 0]).tolist(), agree=(lg.argmax(-1) == t).tolist(),
                            top2_margin=(top2[:, 0] - top2[:, 1]).tolist()))
        return ids, out.past_key_values


class FreeRunningOracle(StreamOracle):
    """The same oracle choosing its own tokens (models/modeling_live.py:51-77, no repetition penalty): for runs whose arithmetic is exact enough that the ids must come out
    EQUAL (the fp32-mode build)."""

    def greedy_generate(self, inputs_embeds, past_key_values, eos_token_id, max_new_tokens, repetition_penalty=None, generated_token_ids=None):
        assert repetition_penalty is None
        x, cache, ids = inputs_embeds.reshape(1, -1, self.cfg.hidden_size).to(self.dtype), past_key_values, []
        for _ in range(max_new_tokens):
            out = self(inputs_embeds=x, past_key_values=cache)
            cache = out.past_key_values
            tok = int(out.logits[0, -1].argmax(-1))
            ids.append(tok)
            if tok == eos_token_id:
                break
            x = self._embed(torch.tensor([[tok]], device=self.device)).to(self.dtype)          # (the last token is written, never fed)
        return ids, cache


def head_logits(driver):
    """[T, 4] float64 of a driver run with record_head_logits = True."""
    return torch.tensor([x['head_logits'] for x in driver.debug_data_list], dtype=torch.float64)


def run_oracle_stream(driver, oracle, forced_ids, query, frames=None, feats=None):
    """Run `driver` (a stream driver constructed around `oracle`) over uint8 `frames` (end to end: the oracle's own PIL preprocess + tower) or over frame
    embeddings `feats` [T, tokens, C] (LLM side isolated), answering with `forced_ids`.  Returns seconds."""
    oracle.forced, oracle.resp, oracle.tf = forced_ids, 0, []
    driver.reset()
    if feats is not None:
        driver.input_feature_stream(feats)
    else:
        driver.input_video_stream(frames.cpu())
    driver.input_query_stream([{'role': 'user', 'content': query, 'time': 0.0}])
    t0 = time.perf_counter()
    driver.inference()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.perf_counter() - t0


LINEAR = ('q_proj', 'k_proj', 'v_proj', 'o_proj', 'gate_proj', 'up_proj', 'down_proj')


def quantise_e4m3_rows(v):
    """Per output channel: scale = amax / 448, q = e4m3fn(W / scale) with IEEE fp32 divisions (the scheme tests/test_gpu_fp8.py::test_quantiser_is_bit_exact_with_torch_e4m3fn
    pins the HIP quantiser to, on the CPU).  torch's fp32 division ON THE GPU is not correctly rounded (measured on MI355X, tools/probes/fp8_cast_probe.py: W / scale differs in
    the last ulp, which flips 0.09 % of the codes -- every exact tie -- by a whole quantisation step), so both divisions are taken in double and rounded to single: for a
    quotient that IS the correctly rounded fp32 result (53 >= 2 x 24 + 2 bits), on any device."""
    vf = v.float()
    amax = vf.abs().amax(dim=1)
    scale = torch.where(amax > 0, (amax.double() / 448.0).float(), torch.ones_like(amax))
    q = (vf.double() / scale.double()[:, None]).float().to(torch.float8_e4m3fn)
    return q, scale


def dequantised_fp8(w):
    """The values the fp8 build computes with: W' = q x scale of the decoder matrices (quantise_e4m3_rows).  fp32 tensors."""
    out = {}
    for k, v in w.items():
        if k.startswith('model.layers.') and k.endswith('.weight') and any(f'.{l}.' in k for l in LINEAR):
            q, scale = quantise_e4m3_rows(v)
            out[k] = q.float() * scale[:, None]
        else:
            out[k] = v.float()
    return out
