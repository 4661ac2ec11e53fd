"""Stream driver: the per-frame "when to speak" loop over a HIP-backed LiveLlava model.

Public surface and behaviour mirror the reference's `LiveInferForBenchmark` (test/inference.py:20-313): the same
attributes (the Gradio demo pokes thresholds directly, demo/app.py:143-150), `reset / set_fps / input_video_stream /
input_query_stream / inference`, the same prompt-prefix rules, threshold rules and result format.  What differs is
how the work is scheduled on the GPU:

  * frame embeddings stay in HBM (the reference copies every frame to the host and back, :212,:237);
  * preprocess runs on the device (mmd_preprocess_frames);
  * one native call per step returns the head logits of the step (one sync), lm_head runs only when text is generated;
  * `frames_per_forward = k > 1` feeds k frames in ONE causal forward (M = 49k rows instead of 49, which moves the
    LLM GEMMs from the weight-bandwidth regime toward the MFMA regime).  Causality makes frame j's scores independent
    of frames j+1.. in the same forward, so decisions are taken frame by frame on the host; when frame j asks for a
    response the KV arena is truncated to the end of frame j (O(1)), the response is generated, and frames j+1.. are
    replayed.  Results equal the one-frame-per-forward schedule up to floating-point reduction order.
"""
from __future__ import annotations
import collections
import math
import os
from dataclasses import asdict
import torch

from .modeling_live import build_model_and_tokenizer, fast_greedy_generate
from .tokenization_live import chat_ids

VIT_BATCH = 32          # test/inference.py:208 (frames per tower call; results do not depend on it)


def _tower_batch(model):
    """Frames per tower call.  The reference's 32 is not special; on MI355X the tower's 256x256 GEMM tiles quantise over 256 CUs and
    35 frames (M = 25 515 rows) fill the last block wave of qkv / fc1 / fc2 best (934 us per frame against 973 at 32,
    tools/vit_batch_sweep.py).  Bounded by the model's tower workspace."""
    cap = getattr(model, 'max_vit_batch', None)
    return VIT_BATCH if not cap else (35 if cap >= 35 else min(VIT_BATCH, cap))


def _p1(l0, l1):
    """softmax([l0, l1])[1] (test/inference.py:243-244), evaluated in double precision and stable for any logit gap."""
    d = l0 - l1
    return 1.0 / (1.0 + math.exp(d)) if d < 700.0 else 0.0


class LiveInferForBenchmark:
    def __init__(self, args, model=None, tokenizer=None) -> None:
        assert not (args.bf16 and args.fp16), "only one of --bf16 true and --fp16 true can be set"
        if args.fp16:
            raise ValueError('fp16 is not offered by the MI355X implementation; use --bf16 true (or fp32)')
        self.torch_dtype = torch.bfloat16 if args.bf16 else torch.float32
        if model is None:
            kw = asdict(args)
            fpf, kvcap = kw.pop('frames_per_forward', 1) or 1, kw.pop('kv_capacity_tokens', 0)
            for k in ('max_new_tokens', 'overlap_vision', 'evaluator_format', 'streams_per_gpu'):
                kw.pop(k, None)
            # workspace rows of one LLM forward follow the chunk size; the KV arena follows the flags (build_live sizes the rest)
            kw['max_step_tokens'] = max(1024, int(getattr(args, 'streams_per_gpu', 1) or 1) * (int(fpf) * int(kw.get('frame_num_tokens', 49) or 49) + 256))
            if kvcap:
                kw['kv_initial_tokens'] = int(kvcap)
            model, tokenizer = build_model_and_tokenizer(is_training=False, set_vision_inside=True, torch_dtype=self.torch_dtype, **kw)
        self.model, self.tokenizer = model, tokenizer
        self.model.eval()
        self.image_processor = self.model.get_vision_tower().image_processor
        self.device = getattr(self.model, 'device', torch.device('cpu'))

        # visual
        self.hidden_size = self.model.config.hidden_size
        if args.frame_fps > 0:
            self.set_fps(args.frame_fps)
        self.frame_resolution = self.model.config.frame_resolution
        self.frame_num_tokens = self.model.config.frame_num_tokens
        self.frame_v_placeholder = self.model.config.v_placeholder * self.frame_num_tokens

        # generation / decision rule
        self.system_prompt = args.system_prompt
        self.max_new_tokens = getattr(args, 'max_new_tokens', 200)
        # token ids live on the HOST in this driver (the reference keeps them on 'cuda'): prefixes are built with CPU ops and cross to the device inside the
        # embedding gather, generated ids come back from the native loop as host integers -- no device-side cat / index_put kernels between two forwards
        self.inplace_output_ids = torch.zeros(1, self.max_new_tokens, dtype=torch.long)
        self.stream_end_prob_threshold = args.stream_end_prob_threshold
        self.response_min_interval_frames = args.response_min_interval_frames
        self.threshold_z = args.threshold_z
        self.first_n_frames_no_generate = args.first_n_frames_no_generate
        self.running_list_length = args.running_list_length
        self.stream_end_score_sum_threshold = args.stream_end_score_sum_threshold
        self.score_heads = args.score_heads.split(',')
        self.consecutive_n_frames_threshold = args.consecutive_n_frames_threshold
        n_set = sum(x is not None for x in (self.threshold_z, self.stream_end_prob_threshold, self.stream_end_score_sum_threshold))
        if n_set != 1:
            raise ValueError('only one of --stream_end_prob_threshold, --threshold_z and --stream_end_score_sum_threshold can be set. '
                             f'However, they are: {self.stream_end_prob_threshold}, {self.threshold_z}, {self.stream_end_score_sum_threshold}')
        if self.threshold_z is not None and self.first_n_frames_no_generate is None:
            raise ValueError('--first_n_frames_no_generate must be set when --threshold_z is set')
        self.remove_assistant_turns = args.remove_assistant_turns
        self.repetition_penalty = args.repetition_penalty
        self.frames_per_forward = max(1, int(getattr(args, 'frames_per_forward', 1)))
        self.overlap_vision = bool(getattr(args, 'overlap_vision', True))
        self.record_head_logits = False          # diagnostics (parity tests, bench.py's self-check): debug_data entries also carry the 4 raw head logits of the frame
        self.reuse_chunk_tail = True             # remove_assistant_turns: keep the chunk's frames behind a response instead of replaying them (same context)
        self._vit_stream = None
        self.vit_lookahead_batches = None      # None: the whole video is queued on the tower stream at once (unless the burst schedule below is on)
        # Burst schedule of the tower: a response's token-by-token decoding streams weights (HBM-bound, no use for the matrix cores) while the tower
        # is MFMA-bound -- but a persistent tower GEMM that owns every CU serialises with the decode kernels instead of sharing the chip with them.
        # So the tower stays `vit_lookahead` batches ahead of the LLM, and when a response starts up to `vit_burst_batches` further batches are issued
        # with their GEMM grids capped at `vit_burst_blocks` workgroups (half the CUs): decode and tower then run side by side.  0 = off.
        self.vit_burst_batches = int(os.environ.get('MMDUET_VIT_BURST', 3))
        self.vit_burst_blocks = int(os.environ.get('MMDUET_VIT_BURST_BLOCKS', 128))
        # a burst is sized to the response it hides under: one tower batch per `vit_burst_tokens_per_batch` expected tokens (a 35-frame batch at half share
        # takes about as long as 10 decode steps beside it), expectation = running mean of this driver's earlier responses, `max_new_tokens` before the first
        self.vit_burst_tokens_per_batch = int(os.environ.get('MMDUET_VIT_BURST_TOKENS', 10))
        self._resp_tokens_mean = None

        self.eos_token_id = self.model.config.eos_token_id
        self._start_ids = chat_ids(self.tokenizer, [{'role': 'system', 'content': self.system_prompt}])
        self._added_stream_prompt_ids = chat_ids(self.tokenizer, [{}], add_stream_prompt=True)
        self._added_stream_generation_ids = chat_ids(self.tokenizer, [{}], add_stream_generation_prompt=True)
        self._xbuf = None                      # step buffer [max_step_tokens, hidden]: prefix embeddings are gathered straight into it, frame embeddings copied behind
        self.reset()

    # ------------------------------------------------------------------------------------------------------------
    def set_fps(self, fps=None, frame_interval=None):
        assert (fps is None) != (frame_interval is None)
        if fps is not None:
            self.frame_fps, self.frame_interval = fps, 1 / fps
        else:
            self.frame_interval, self.frame_fps = frame_interval, 1 / frame_interval

    def _no_ids(self):
        return torch.zeros(1, 0, dtype=torch.long)

    def reset(self):
        self.query_queue = collections.deque()
        self.frame_embeds_queue = collections.deque()
        self.video_time = 0
        self.frame_idx = 0
        self.last_role = 'system'
        self.video_tensor = None
        self.last_ids = self._no_ids()
        self.past_key_values = None
        self.debug_data_list = list()
        self.generated_token_ids = list()
        self.num_frames_no_reply = 0
        self.stream_end_prob_list = list()
        self.stream_end_score_sum = 0
        self.consecutive_n_frames = 0
        self._frame_batch = {}               # frame embedding (data_ptr) -> tower batch index, overlap mode
        self._vit_out = self._vit_pixels = self._vit_host = None
        self._vit_fused = False
        self._vit_batches, self._vit_events, self._vit_waited = [], [], set()
        self.forward_calls = 0              # LLM forwards issued (diagnostics of the chunked schedule)
        self.replayed_frames = 0

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def input_video_stream(self, video_frames):
        """All frames of the video at once (uint8 [T,3,R,R]); queues (time, [frame_num_tokens, hidden]) per frame."""
        overlap = self.overlap_vision and self.device.type == 'cuda'
        # preprocess fused into the patch-embed load when the model offers it (visual_embed_frames: no pixel_values tensor, same bits)
        fused = hasattr(self.model, 'visual_embed_frames') and torch.is_tensor(video_frames) and video_frames.dtype == torch.uint8
        if not overlap:
            vb = _tower_batch(self.model)
            if fused:
                for b0 in range(0, len(video_frames), vb):          # (host frames cross per tower batch; asynchronously when the caller pinned them)
                    embeds = self.model.visual_embed_frames(video_frames[b0:b0 + vb].to(self.device, non_blocking=True)).split(self.frame_num_tokens)
                    self.frame_embeds_queue.extend(((r + b0) / self.frame_fps, f) for r, f in enumerate(embeds))
                return
            pixel_values = self.image_processor.preprocess(video_frames, return_tensors='pt')['pixel_values']
            pixel_values = pixel_values.to(self.device).to(self.torch_dtype)
            for b0 in range(0, len(pixel_values), vb):
                embeds = self.model.visual_embed(pixel_values[b0:b0 + vb]).split(self.frame_num_tokens)
                self.frame_embeds_queue.extend(((r + b0) / self.frame_fps, f) for r, f in enumerate(embeds))
            return
        # The tower is MFMA-bound, the LLM steps (weight streaming, token-by-token decoding) are not: encode the frames on a
        # side HIP stream so that batch i+1 of the tower overlaps the LLM work on batch i.  Frame embeddings are views of one
        # pre-allocated buffer; batch b carries an event and the LLM stream waits on it right before a frame of b is
        # consumed.  `vit_lookahead_batches` = None issues every batch now; an integer n keeps the tower n batches ahead of the
        # LLM (several streams sharing one tower stream interleave their batches that way).  Results are unchanged.
        if self._vit_stream is None:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
            # lowest priority for the tower: the LLM's short latency-bound kernels go first, tower tiles fill the gaps
            self._vit_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('MMDUET_VIT_PRIO', lo)))
        main = torch.cuda.current_stream(self.device)
        side = self._vit_stream
        T, nt = len(video_frames), self.frame_num_tokens
        self._vit_out = torch.empty(T * nt, self.hidden_size, dtype=self.torch_dtype, device=self.device)
        side.wait_stream(main)
        self._vit_fused = fused
        self._vit_host = None
        if fused and video_frames.device.type == 'cpu' and self.device.type == 'cuda':
            # host frames (the CLI's loader threads hand over pinned clips; the reference: pixel_values.to('cuda'), test/inference.py:203): every tower batch uploads
            # its own frames on the tower's side stream right before it runs (_issue_vit) -- no synchronous whole-video copy in front of the first LLM step
            self._vit_host = video_frames
            self._vit_pixels = torch.empty(video_frames.shape, dtype=torch.uint8, device=self.device)
        elif fused:
            self._vit_pixels = video_frames.to(self.device)             # the raw uint8 frames stay resident; each tower batch resamples its own
        else:
            with torch.cuda.stream(side):
                self._vit_pixels = self.image_processor.preprocess(video_frames, return_tensors='pt')['pixel_values'].to(self.torch_dtype)
        vb = _tower_batch(self.model)
        self._vit_batches = [(b0, min(T, b0 + vb)) for b0 in range(0, T, vb)]
        self._vit_events, self._vit_waited = [], set()
        for r in range(T):
            f = self._vit_out[r * nt:(r + 1) * nt]
            self._frame_batch[f.data_ptr()] = r // vb
            self.frame_embeds_queue.append((r / self.frame_fps, f))
        self._issue_vit(len(self._vit_batches) if self._vit_ahead() is None else self._vit_ahead() + 1)

    def _vit_ahead(self):
        """Tower batches kept in flight beyond the ones the LLM needs now; None = everything at once."""
        if self.vit_lookahead_batches is not None:
            return self.vit_lookahead_batches
        if self.vit_burst_batches > 0 and hasattr(self.model, 'set_tower_share'):
            return int(os.environ.get('MMDUET_VIT_LOOKAHEAD', 1))
        return None

    def _issue_vit_burst(self):
        """A response is about to be decoded: let the next tower batches run beside it on half the CUs."""
        if not self._vit_batches or self.vit_burst_batches <= 0 or len(self._vit_events) >= len(self._vit_batches) or not hasattr(self.model, 'set_tower_share'):
            return
        expect = self._resp_tokens_mean if self._resp_tokens_mean is not None else float(getattr(self, 'max_new_tokens', 0) or 0)
        n = min(self.vit_burst_batches, int(expect // max(1, self.vit_burst_tokens_per_batch)))
        if n <= 0:
            return
        self.model.set_tower_share(self.vit_burst_blocks)
        try:
            self._issue_vit(len(self._vit_events) + n)
        finally:
            self.model.set_tower_share(0)

    def _issue_vit(self, upto):
        """Enqueue tower batches [issued, upto) on the side stream."""
        nt = self.frame_num_tokens
        while len(self._vit_events) < min(upto, len(self._vit_batches)):
            b0, b1 = self._vit_batches[len(self._vit_events)]
            with torch.cuda.stream(self._vit_stream):
                if getattr(self, '_vit_fused', False):
                    if getattr(self, '_vit_host', None) is not None:
                        self._vit_pixels[b0:b1].copy_(self._vit_host[b0:b1], non_blocking=True)
                    self.model.visual_embed_frames(self._vit_pixels[b0:b1], out=self._vit_out[b0 * nt:b1 * nt])
                else:
                    self.model.visual_embed(self._vit_pixels[b0:b1], out=self._vit_out[b0 * nt:b1 * nt])
                ev = torch.cuda.Event()
                ev.record(self._vit_stream)
            self._vit_events.append(ev)

    @torch.no_grad()
    def input_feature_stream(self, features, tokens_per_frame=None):
        """Phase A from a pre-extracted feature file instead of frames (the reference's feature files, data/utils.py:99-117,
        feed `visual_embed` at the connector when the model carries no tower, models/modeling_live.py:26-33).  `features`: a path (.pt / .npy) or a
        tensor [T, tokens, C]: tower-level features [T, vit_tokens, vit_hidden] go through the connector + pooling on the GPU, pooled embeddings
        [T, frame_num_tokens, hidden] are queued as they are.  The queue then holds exactly what `input_video_stream` would have produced."""
        from .features import load_frame_features, feature_level
        if isinstance(features, (str, os.PathLike)):
            features = load_frame_features(features, tokens_per_frame)
        if features.ndim == 2:
            features = features.reshape(-1, tokens_per_frame or self.frame_num_tokens, features.shape[-1])
        level = feature_level(self.model, features)
        T, nt = features.shape[0], self.frame_num_tokens
        feats = features.to(device=self.device, dtype=self.torch_dtype)
        if level == 'tower':
            out = torch.empty(T * nt, self.hidden_size, dtype=self.torch_dtype, device=self.device)
            vb = _tower_batch(self.model)
            for b0 in range(0, T, vb):
                self.model.connector_pool(feats[b0:b0 + vb], out=out[b0 * nt:min(T, b0 + vb) * nt])
        else:
            out = feats.reshape(T * nt, self.hidden_size).contiguous()
        self._vit_out = out
        for r in range(T):
            self.frame_embeds_queue.append((r / self.frame_fps, out[r * nt:(r + 1) * nt]))

    def input_query_stream(self, conversation):
        for turn in conversation:
            if turn['role'] == 'user':
                self.query_queue.append((turn['time'], turn['content']))

    # ------------------------------------------------------------------------------------------------------------
    def _prefix_ids_for_next_frame(self):
        """Text that precedes the next frame's tokens (test/inference.py:229-234): the system turn before the very
        first forward (note: no `stream` header before the first frame, and nothing at all if a t=0 query already
        filled the cache); last generated token + "\\n<|im_start|>stream\\n" after an assistant turn that stays in
        context; otherwise nothing."""
        if not self.past_key_values:
            return self._start_ids
        if self.last_role == 'assistant' and not self.remove_assistant_turns:
            return torch.cat([self.last_ids, self._added_stream_prompt_ids], dim=1)
        return self._no_ids()

    def _embed(self, ids):
        return self.model.get_input_embeddings()(ids.to(self.device)).view(1, -1, self.hidden_size)

    def _forward_frames(self, frames):
        """One causal forward over `frames` (list of [frame_num_tokens, hidden]); returns per-frame
        (informative_score, relevance_score) and the KV length at the end of each frame."""
        if self._frame_batch:
            need = {self._frame_batch[f.data_ptr()] for f in frames if f.data_ptr() in self._frame_batch}
            if need:
                self._issue_vit(max(need) + 1 + (self._vit_ahead() or 0))
                for b in sorted(need - self._vit_waited):
                    torch.cuda.current_stream(self.device).wait_event(self._vit_events[b])
                    self._vit_waited.add(b)
        self.last_ids = self._prefix_ids_for_next_frame()
        P = int(self.last_ids.shape[1])
        nt = self.frame_num_tokens
        x = self._step_input(self.last_ids, frames)
        n0 = len(self.past_key_values) if self.past_key_values else 0
        rows = [P + (j + 1) * nt - 1 for j in range(len(frames))]
        if hasattr(self.model, 'frame_step'):
            head_logits, cache = self.model.frame_step(x, self.past_key_values, rows)
        else:       # generic duck-typed model
            out = self.model(inputs_embeds=x, use_cache=True, past_key_values=self.past_key_values, return_dict=True)
            head_logits = torch.cat([out.informative_logits[0, rows], out.relevance_logits[0, rows]], dim=-1).float().cpu()
            cache = out.past_key_values
        self.forward_calls += 1
        # 2-way softmax of the head logits on the host, in plain Python: a CPU torch op here wakes torch's intra-op thread pool
        # (one spinning thread per visible core), which on a CPU-quota'd box gets the process throttled for tens of ms at a time
        probs_inf, probs_rel = [], []
        self._chunk_head_logits = head_logits.tolist()
        for l0, l1, r0, r1 in self._chunk_head_logits:
            probs_inf.append(_p1(l0, l1)); probs_rel.append(_p1(r0, r1))
        ends = [n0 + P + (j + 1) * nt for j in range(len(frames))]
        return list(zip(probs_inf, probs_rel)), ends, cache

    def _step_input(self, prefix_ids, frames):
        """[1, P + k*frame_num_tokens, hidden] input of one forward without a concatenation kernel: the frames of a chunk are consecutive rows of the tower's output
        buffer, so with no text prefix (every chunk but the first and the ones behind a response) the input IS that slice; otherwise the prefix embeddings are
        gathered straight into the step buffer and the frame rows are copied behind them (one device memcpy)."""
        P, nt, H = int(prefix_ids.shape[1]), self.frame_num_tokens, self.hidden_size
        run = None
        if self._vit_out is not None and frames and all(f.is_contiguous() for f in frames):
            p0, es = frames[0].data_ptr(), frames[0].element_size()
            if all(f.data_ptr() == p0 + j * nt * H * es and f.shape[0] == nt for j, f in enumerate(frames)):
                off = (p0 - self._vit_out.data_ptr()) // (H * es)
                if 0 <= off and off + len(frames) * nt <= self._vit_out.shape[0] and (p0 - self._vit_out.data_ptr()) % (H * es) == 0:
                    run = self._vit_out[off:off + len(frames) * nt]
        if run is not None and P == 0:
            return run.view(1, -1, H)
        emb = self.model.get_input_embeddings()
        dev = getattr(self.model, 'device', None)
        if dev is None or getattr(dev, 'type', 'cpu') != 'cuda' or not hasattr(self.model, 'frame_step'):          # generic duck-typed model (the oracle behind this driver)
            prefix = emb(prefix_ids.to(self.device)).view(1, -1, H)
            return torch.cat([prefix] + [f.view(1, -1, H).to(prefix.device) for f in frames], dim=1)
        rows = P + sum(int(f.shape[0]) for f in frames)
        if self._xbuf is None or self._xbuf.shape[0] < rows or self._xbuf.dtype != self.torch_dtype:
            self._xbuf = torch.empty(max(rows, int(getattr(self.model, 'max_step_tokens', 0) or 0)), H, dtype=self.torch_dtype, device=dev)
        if P:
            emb(prefix_ids.view(-1), out=self._xbuf[:P])
        if run is not None:
            self._xbuf[P:rows].copy_(run)
        else:
            at = P
            for f in frames:
                self._xbuf[at:at + f.shape[0]].copy_(f.view(-1, H)); at += f.shape[0]
        return self._xbuf[:rows].view(1, -1, H)

    def _encode_frame(self):
        """Single-frame step with the reference's return value (test/inference.py:221-246)."""
        if not self.frame_embeds_queue:
            return None, None
        video_time, frame_embeds = self.frame_embeds_queue.popleft()
        scores, _, cache = self._forward_frames([frame_embeds])
        self.past_key_values = cache
        self.frame_idx += 1
        self.num_frames_no_reply += 1
        self.last_role = 'stream'
        return {'informative_score': scores[0][0], 'relevance_score': scores[0][1]}

    def _encode_query(self):
        """test/inference.py:248-255."""
        query_time, query = self.query_queue.popleft()
        self.last_ids = chat_ids(self.tokenizer, [{'role': 'user', 'content': query}],
                                 add_stream_query_prompt=self.last_role == 'stream', add_stream_prompt=True)
        outputs = self.model(inputs_embeds=self._embed(self.last_ids), past_key_values=self.past_key_values, use_cache=True, return_dict=True)
        self.past_key_values = outputs.past_key_values
        self.forward_calls += 1
        # the reference takes argmax of the last logits here and overwrites it before any use (:254); skipped (lm_head is lazy)
        self.last_ids = self._no_ids()
        self.last_role = 'user'

    def _generate_response(self):
        """test/inference.py:257-274."""
        self.last_ids = self._added_stream_generation_ids
        self._issue_vit_burst()
        output_ids, past_key_values, self.generated_token_ids = fast_greedy_generate(
            model=self.model, inputs_embeds=self._embed(self.last_ids), past_key_values=self.past_key_values,
            eos_token_id=self.eos_token_id, inplace_output_ids=self.inplace_output_ids,
            repetition_penalty=self.repetition_penalty, generated_token_ids=self.generated_token_ids)
        self.last_generated_ids = output_ids[0].tolist()
        n_tok = len(self.last_generated_ids)
        self._resp_tokens_mean = n_tok if self._resp_tokens_mean is None else 0.7 * self._resp_tokens_mean + 0.3 * n_tok
        if not self.remove_assistant_turns:
            self.past_key_values = past_key_values
            self.last_ids = output_ids[:, -1:].clone()
        else:
            self.last_ids = self._no_ids()          # the generated turn's KV is dropped: we keep the older handle
        response = self.tokenizer.decode(output_ids[0], skip_special_tokens=True, clean_up_tokenization_spaces=True)
        self.num_frames_no_reply = 0
        self.last_role = 'assistant'
        return response

    # ------------------------------------------------------------------------------------------------------------
    def _decide(self, video_scores):
        """Threshold rule of test/inference.py:289-299; returns need_response."""
        stream_end_score = sum(v for k, v in video_scores.items() if k in self.score_heads)
        self.stream_end_prob_list.append(stream_end_score)
        self.stream_end_score_sum += stream_end_score
        if isinstance(self.running_list_length, int) and self.running_list_length > 0:
            self.stream_end_prob_list = self.stream_end_prob_list[-self.running_list_length:]
        need = False
        if self.stream_end_score_sum_threshold is not None and self.stream_end_score_sum > self.stream_end_score_sum_threshold:
            need = True
            self.stream_end_score_sum = 0
        if self.stream_end_prob_threshold is not None and stream_end_score > self.stream_end_prob_threshold:
            need = True
        return need

    def _chunk_size(self):
        """Frames that may share the next forward: up to frames_per_forward, never across a pending user query
        (a query due at frame j must be encoded before frame j, test/inference.py:281-282)."""
        k = min(self.frames_per_forward, len(self.frame_embeds_queue))
        cap = getattr(self.model, 'max_step_tokens', None)
        if cap:                                  # leave room for the text prefix of the step (system prompt / stream header)
            k = max(1, min(k, (cap - 128) // self.frame_num_tokens))
        if self.query_queue:
            # replay the clock exactly as the loop advances it (video_time += 1/fps per frame, test/inference.py:311): j/fps and
            # j additions of 1/fps differ in the last ulp for any fps that is not a power of two, which would move the query
            # by one frame against the one-frame-per-forward schedule
            q_time = self.query_queue[0][0]
            t = self.video_time
            for j in range(1, k):
                t += 1 / self.frame_fps
                if t >= q_time:
                    return j
        return k

    @torch.no_grad()
    def inference(self):
        model_response_list = [{'time': q[0], 'content': q[1], 'role': 'user'} for q in self.query_queue]
        self.response_token_ids = []
        while self.frame_embeds_queue:
            # 1. a user query due at the current time goes in first
            if self.query_queue and self.video_time >= self.query_queue[0][0]:
                self._encode_query()
            # 2. one forward over the next chunk of frames
            k = self._chunk_size()
            chunk = [self.frame_embeds_queue.popleft() for _ in range(k)]
            scores, ends, cache = self._forward_frames([f for _, f in chunk])
            arena_handle = cache
            for j in range(k):
                self.frame_idx += 1
                self.num_frames_no_reply += 1
                self.last_role = 'stream'
                video_scores = {'informative_score': scores[j][0], 'relevance_score': scores[j][1]}
                self.debug_data_list.append(dict(time=self.video_time, **video_scores))
                if self.record_head_logits:
                    self.debug_data_list[-1]['head_logits'] = self._chunk_head_logits[j]
                # 3./4. decide, respond
                if self._decide(video_scores):
                    last = j == k - 1
                    # remove_assistant_turns: the response never stays in the context, so the frames of this chunk behind frame j are still what a
                    # replay would compute -- set their KV aside, let the response use (and drop) those slots, bring them back, go on deciding
                    keep_tail = (not last) and self.remove_assistant_turns and self.reuse_chunk_tail and hasattr(self.model, 'kv_stash')
                    if keep_tail:
                        stash = self.model.kv_stash(cache, ends[j])
                    # context = everything up to the end of frame j; otherwise later frames of the chunk are replayed
                    self.past_key_values = cache if last else self.model.cache_prefix(arena_handle, ends[j])
                    if not last and not keep_tail:
                        for item in reversed(chunk[j + 1:]):
                            self.frame_embeds_queue.appendleft(item)
                        self.replayed_frames += k - 1 - j
                    response = self._generate_response()
                    self.response_token_ids.append(self.last_generated_ids)
                    model_response_list.append({'time': self.video_time, 'content': response, 'role': 'assistant'})
                    self.num_frames_no_reply = 0
                    self.consecutive_n_frames = 0
                    self.video_time += 1 / self.frame_fps
                    if keep_tail:
                        cache = arena_handle = self.model.kv_unstash(stash)
                        continue
                    break
                # 5. advance the clock
                self.video_time += 1 / self.frame_fps
            else:
                self.past_key_values = cache
        return sorted(model_response_list, key=lambda x: x['time'])


def round_numbers(data, n):
    """test/inference.py:322-329."""
    if isinstance(data, list):
        return [round_numbers(d, n) for d in data]
    if isinstance(data, dict):
        return {k: round_numbers(v, n) for k, v in data.items()}
    if isinstance(data, float):
        return round(data, n)
    return data
