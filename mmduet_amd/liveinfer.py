"""Demo-side driver: one frame per call, asynchronous user text.

Mirrors `LiveInferForDemo` (demo/liveinfer.py:60-105): `encode_given_query` (called from the Gradio handler thread,
demo/app.py:84-85) and `input_one_frame` (called from the generator thread).  The reference shares
`past_key_values` between those two threads without a lock; here both go through one re-entrant lock.
`load_video` (demo/liveinfer.py:8-57: OpenCV decode, time-based frame picking, letterbox to 384) lives in `mmduet_amd.video_input.load_video_frames`
(sampling plan and letterbox geometry as the reference computes them, the resize / pad on the GPU) on top of `mmduet_amd.video_decode` (Motion-JPEG /
uncompressed AVI; the image has no codec library); `input_video_stream` takes the resulting uint8 [T,3,R,R] frames.
"""
import threading
from .inference import LiveInferForBenchmark
from .tokenization_live import chat_ids


class LiveInferForDemo(LiveInferForBenchmark):
    def __init__(self, *a, **k):
        self._step_lock = threading.RLock()
        super().__init__(*a, **k)

    def encode_given_query(self, query):
        with self._step_lock:
            self.last_ids = chat_ids(self.tokenizer, [{'role': 'user', 'content': query}],
                                     add_stream_query_prompt=self.last_role == 'stream', add_stream_prompt=True)
            outputs = self.model(inputs_embeds=self._embed(self.last_ids), past_key_values=self.past_key_values, use_cache=True, return_dict=True)
            self.past_key_values = outputs.past_key_values
            self.last_ids = outputs.logits[:, -1:].argmax(dim=-1).cpu()          # (ids live on the host in this driver)
            self.last_role = 'user'

    def input_one_frame(self):
        """demo/liveinfer.py:69-105: steps 2-5 of the benchmark loop for exactly one frame."""
        with self._step_lock:
            video_scores = self._encode_frame()
            ret = dict(frame_idx=self.frame_idx, time=round(self.video_time, 1), **video_scores)
            response = None
            if self._decide(video_scores):
                response = self._generate_response()
                self.num_frames_no_reply = 0
                self.consecutive_n_frames = 0
            ret['response'] = response
            self.video_time += 1 / self.frame_fps
            return ret
