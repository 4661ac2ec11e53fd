"""Video input producer: the reference's `load_video` (test/datasets.py:32-85, demo/liveinfer.py:8-57) with the codec factored out.

The reference decodes with OpenCV on the host, and for every kept frame runs cv2.resize -> cv2.copyMakeBorder ->
cv2.cvtColor -> transpose on the CPU inside 4 DataLoader workers (test/inference.py:341).  Here the decoder is whatever the
caller has (this image ships none); everything after it is reproduced:

* `frame_sampling_plan` -- WHICH decoded frames the reference keeps (its float accumulation `cur_time += 1/input_fps`
  compared against `i / output_fps`, the ceil/floor frame budget, the max_num_frames cut), output fps, duration;
* `letterbox_geometry` -- resize target and pad widths;
* `letterbox_frames` -- the per-frame pixel work as ONE HIP kernel over the whole clip (mmd_letterbox_frames), frames
  resident in HBM: uint8 [T,H,W,3] BGR -> uint8 [T,3,R,R] RGB, the layout `LiveInferForBenchmark.input_video_stream` takes;
* `time_instruction` -- the `timechat` / `vtimellm` prompt prefixes;
* `load_video_frames` -- the composition, returning what the reference's `load_video` returns.
"""
import ctypes as C
import math
import torch
from ._lib import lib, check


def frame_sampling_plan(input_fps, frame_count, output_fps, max_num_frames, n_decodable=None, budget='ceil'):
    """Returns (kept, output_fps, video_duration, frame_sec).  `frame_count` is the container's header value
    (CAP_PROP_FRAME_COUNT, sets duration and budget); `n_decodable` the frames the decoder actually yields (defaults to it).
    budget='ceil' is test/datasets.py:44, 'floor' demo/liveinfer.py:23."""
    if n_decodable is None:
        n_decodable = int(frame_count)
    video_duration = frame_count / input_fps
    if not output_fps > 0:
        output_fps = max_num_frames / video_duration                     # the 'auto' mode of test/datasets.py:15-17,43
    budget_frames = (math.ceil if budget == 'ceil' else math.floor)(video_duration * output_fps)
    frame_sec = [i / output_fps for i in range(budget_frames)]
    step = 1 / input_fps
    kept, clock = [], 0
    for index in range(n_decodable):
        if len(kept) < budget_frames and clock >= frame_sec[len(kept)]:
            kept.append(index)
        if len(kept) >= max_num_frames:
            break
        clock += step                                                     # accumulated, not index * step: matches the reference bit for bit
    return kept, output_fps, video_duration, frame_sec


def letterbox_geometry(width, height, resolution):
    """(new_w, new_h, (top, bottom, left, right)) -- test/datasets.py:53-68, computed by the library so host and kernel agree."""
    v = [C.c_int() for _ in range(6)]
    rc = lib().mmd_letterbox_geometry(int(width), int(height), int(resolution), *[C.byref(x) for x in v])
    if rc:
        raise ValueError(f'letterbox_geometry({width}, {height}, {resolution}) is invalid')
    nw, nh, top, bottom, left, right = (x.value for x in v)
    return nw, nh, (top, bottom, left, right)


def letterbox_frames(model, frames, resolution=384, pad_color=(0, 0, 0), bgr_input=True):
    """frames: uint8 [T,H,W,3] (decoder layout; moved to the model's device if needed) -> uint8 [T,3,R,R] on the device."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError(f'letterbox_frames expects uint8 [T,H,W,3], got {frames.dtype} {tuple(frames.shape)}')
    frames = frames.to(model.device, non_blocking=True).contiguous()          # (a pinned host clip -- the CLI's loader threads pin theirs -- crosses asynchronously, in stream order)
    T, H, W, _ = frames.shape
    out = torch.empty((T, 3, resolution, resolution), dtype=torch.uint8, device=model.device)
    pad = (C.c_uint8 * 3)(*[int(p) for p in pad_color])
    model._bind_stream()
    check(lib().mmd_letterbox_frames(model._ctx, frames.data_ptr(), T, H, W, int(resolution), pad, 1 if bgr_input else 0, out.data_ptr()), model._ctx)
    return out


def time_instruction(fmt, video_duration, n_frames, frame_sec):
    """test/datasets.py:78-84; returns None when no time instruction is configured."""
    if fmt == 'timechat':
        stamps = ",".join("%.2fs" % s for s in frame_sec)
        return ("The video lasts for %.2f seconds, and %d frames are uniformly sampled from it. These frames are located at %s."
                "Please answer the following questions related to this video." % (video_duration, n_frames, stamps))
    if fmt == 'vtimellm':
        return "This is a video with %d frames." % n_frames
    if fmt is None:
        return None
    raise ValueError(f'unknown time_instruction_format {fmt!r}')


def load_video_frames(model, decoded_frames, input_fps, frame_count=None, output_fps=2, resolution=384, max_num_frames=100,
                      time_instruction_format=None, pad_color=(0, 0, 0), budget='ceil', presampled=None):
    """`load_video` given the decoder's output: decoded_frames uint8 [N,H,W,3] BGR (host or device), in decode order.
    Returns (frames uint8 [T,3,R,R] on the device, output_fps, video_duration[, time_instruction]).
    presampled = (kept, n_decodable): `decoded_frames` already holds ONLY the frames `kept` of a file with n_decodable frames (video_decode.read_avi(select=...),
    the CLI's loader threads): the schedule is recomputed and must agree, the picking is skipped."""
    n = int(decoded_frames.shape[0]) if presampled is None else int(presampled[1])
    kept, out_fps, duration, frame_sec = frame_sampling_plan(input_fps, n if frame_count is None else frame_count, output_fps,
                                                             max_num_frames, n_decodable=n, budget=budget)
    if not kept:
        raise ValueError('need at least one array to stack')          # np.stack([]) in the reference (test/datasets.py:85)
    if presampled is None:
        picked = decoded_frames[torch.as_tensor(kept, device=decoded_frames.device)]
    else:
        if list(presampled[0]) != kept or int(decoded_frames.shape[0]) != len(kept):
            raise ValueError('presampled frames do not follow the sampling schedule')
        picked = decoded_frames
    frames = letterbox_frames(model, picked, resolution, pad_color)
    if time_instruction_format is None:
        return frames, out_fps, duration
    return frames, out_fps, duration, time_instruction(time_instruction_format, duration, len(kept), frame_sec)


class sampling_selector:
    """The `select` callback of video_decode.read_avi: the frames the reference's sampling loop keeps, so that only those are decoded.  Remembers the file's
    frame count it was called with (`n_decodable`), which `load_video_frames(presampled=...)` needs to recompute the same schedule."""

    def __init__(self, output_fps, max_num_frames, budget='ceil'):
        self.output_fps, self.max_num_frames, self.budget, self.n_decodable = output_fps, max_num_frames, budget, None

    def __call__(self, input_fps, frame_count, n_decodable):
        self.n_decodable = int(n_decodable)
        return frame_sampling_plan(input_fps, frame_count, self.output_fps, self.max_num_frames, n_decodable=n_decodable, budget=self.budget)[0]
