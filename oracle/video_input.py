"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the reference's video
input producer `load_video` (test/datasets.py:32-85; demo/liveinfer.py:8-57), minus the codec.

Pinned: the sampling schedule, letterbox geometry, max_num_frames cut, output fps / duration and the time-instruction strings
are checked against tests/golden/video_input.json -- the reference's own loops run with a recording mock of cv2
(tests/golden/make_video_input_golden.py).
PARITY UNPINNED: `cv2_resize_linear_u8`.  OpenCV (4.x pinned by the reference's requirements) is a third-party dependency
that is absent from this image, so its pixels cannot be generated here; the function restates OpenCV's published 8-bit
INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp: float tap positions, 11-bit short weights, int horizontal pass,
`((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2` vertical pass, and the exact-2x -> INTER_AREA shortcut).
"""
import math
import numpy as np

INTER_RESIZE_COEF_BITS = 11
_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _round_half_even_to_short(v):
    r = int(np.rint(np.float32(v)))                 # cvRound: round half to even
    return max(-32768, min(32767, r))


def cv2_linear_taps(src_n, dst_n, x_axis):
    """resize.cpp (ResizeLinear setup): returns int arrays s0, s1, w0, w1 of length dst_n."""
    inv_scale = float(dst_n) / float(src_n)
    scale = 1.0 / inv_scale
    s0 = np.zeros(dst_n, np.int64); s1 = np.zeros(dst_n, np.int64); w0 = np.zeros(dst_n, np.int64); w1 = np.zeros(dst_n, np.int64)
    for d in range(dst_n):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if x_axis:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src_n - 1:
                f, s = np.float32(0), src_n - 1
            a, b = s, min(s + 1, src_n - 1)
        else:
            a, b = min(max(s, 0), src_n - 1), min(max(s + 1, 0), src_n - 1)
        s0[d], s1[d] = a, b
        w0[d] = _round_half_even_to_short(np.float32(np.float32(1) - f) * np.float32(_SCALE))
        w1[d] = _round_half_even_to_short(f * np.float32(_SCALE))
    return s0, s1, w0, w1


def cv2_resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) for uint8 HxWxC, default INTER_LINEAR.  PARITY UNPINNED (see module header)."""
    H, W = img.shape[:2]
    sx, sy = 1.0 / (float(new_w) / W), 1.0 / (float(new_h) / H)
    eps = np.finfo(np.float64).eps
    src = img.astype(np.int64)
    if abs(sx - 2.0) < eps and abs(sy - 2.0) < eps:                      # INTER_LINEAR at exactly 2x == fast INTER_AREA
        a = src[0:2 * new_h:2, 0:2 * new_w:2]; b = src[0:2 * new_h:2, 1:2 * new_w:2]
        c = src[1:2 * new_h:2, 0:2 * new_w:2]; d = src[1:2 * new_h:2, 1:2 * new_w:2]
        return ((a + b + c + d + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = cv2_linear_taps(W, new_w, True)
    y0, y1, b0, b1 = cv2_linear_taps(H, new_h, False)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [H, new_w, C], scale 2^11
    r0, r1 = hor[y0], hor[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def letterbox_geometry(W, H, R):
    """test/datasets.py:53-60 + :63-68."""
    if W > H:
        new_w, new_h = R, int((H / W) * R)
    else:
        new_h, new_w = R, int((W / H) * R)
    return new_w, new_h, ((R - new_h) // 2, (R - new_h + 1) // 2, (R - new_w) // 2, (R - new_w + 1) // 2)


def letterbox_frame(frame_hwc, R, pad_color=(0, 0, 0), flip_channels=True):
    """resize + copyMakeBorder(BORDER_CONSTANT) + cvtColor(BGR2RGB) + HWC->CHW of one decoded frame (test/datasets.py:61-71)."""
    H, W = frame_hwc.shape[:2]
    new_w, new_h, (top, bottom, left, right) = letterbox_geometry(W, H, R)
    canvas = np.empty((R, R, 3), np.uint8); canvas[:] = np.asarray(pad_color, np.uint8)
    canvas[top:top + new_h, left:left + new_w] = cv2_resize_linear_u8(frame_hwc, new_w, new_h)
    if flip_channels:
        canvas = canvas[..., ::-1]
    return np.ascontiguousarray(canvas.transpose(2, 0, 1))


def sample_schedule(input_fps, frame_count, n_decodable, output_fps, max_num_frames, floor_total=False):
    """The frame-picking loop of load_video, codec removed: returns (kept source indices, output_fps, video_duration, frame_sec).
    test/datasets.py:35-76 rounds the frame budget up (math.ceil), demo/liveinfer.py:23 down (floor_total=True)."""
    video_duration = frame_count / input_fps
    output_fps = output_fps if output_fps > 0 else max_num_frames / video_duration
    total = math.floor(video_duration * output_fps) if floor_total else math.ceil(video_duration * output_fps)
    frame_sec = [i / output_fps for i in range(total)]
    kept, cur_time, frame_index = [], 0, 0
    for src_index in range(n_decodable):                                   # `ret, frame = cap.read()` until it fails
        if frame_index < len(frame_sec) and cur_time >= frame_sec[frame_index]:
            kept.append(src_index)
            frame_index += 1
        if len(kept) >= max_num_frames:
            break
        cur_time += 1 / input_fps
    return kept, output_fps, video_duration, frame_sec


def time_instruction(fmt, video_duration, n_frames, frame_sec):
    """test/datasets.py:78-84 (the reference's strings, typos included)."""
    if fmt == 'timechat':
        frame_sec_str = ",".join(f"{i:.2f}s" for i in frame_sec)
        return (f"The video lasts for {video_duration:.2f} seconds, and {n_frames} frames are uniformly sampled from it. "
                f"These frames are located at {frame_sec_str}.Please answer the following questions related to this video.")
    if fmt == 'vtimellm':
        return f"This is a video with {n_frames} frames."
    return None
