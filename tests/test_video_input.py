"""Video input producer: schedule/geometry against the reference's own loops (golden), the
cv2 bilinear restatement against its defining properties, and the HIP letterbox kernel bit-exact against the oracle."""
import json, os
import numpy as np
import pytest
import torch
from oracle import video_input as ov
from mmduet_amd import video_input as pv

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'video_input.json')))['runs']
DATASET_RUNS = [r for r in GOLD if r['loop'] == 'datasets']
DEMO_RUNS = [r for r in GOLD if r['loop'] == 'demo']


def _hdr(props):
    return props['decodable'] if props.get('count_from_header') is None else props['count_from_header']


@pytest.mark.parametrize('impl', ['oracle', 'product'])
def test_sampling_schedule_and_geometry_match_reference_loops(impl):
    assert len(DATASET_RUNS) == 40 and len(DEMO_RUNS) == 24
    for r in GOLD:
        p = r['props']
        floor = r['loop'] == 'demo'
        mx = r.get('max_num_frames', 400)
        if impl == 'oracle':
            kept, ofps, dur, fsec = ov.sample_schedule(p['fps'], float(_hdr(p)), p['decodable'], r['output_fps_arg'], mx, floor_total=floor)
        else:
            kept, ofps, dur, fsec = pv.frame_sampling_plan(p['fps'], float(_hdr(p)), r['output_fps_arg'], mx, n_decodable=p['decodable'],
                                                           budget='floor' if floor else 'ceil')
        if 'error' in r:
            assert kept == [] and r['error'] == 'ValueError'
            continue
        assert kept == r['kept'], (r['video'], r['output_fps_arg'])
        R = r['resolution']
        assert r['out_shape'] == [len(kept), 3, R, R]
        if r['loop'] == 'datasets':
            assert ofps == r['output_fps'] and dur == r['video_duration']          # exact: same float operations
            ti = (ov.time_instruction if impl == 'oracle' else pv.time_instruction)(r['time_instruction_format'], dur, len(kept), fsec)
            assert ti == r['time_instruction']
        geo = (ov.letterbox_geometry if impl == 'oracle' else pv.letterbox_geometry)(p['w'], p['h'], R)
        assert [geo[0], geo[1]] == r['resize_to'] and list(geo[2]) == r['pads']


def test_cv2_bilinear_restatement_properties():
    """No cv2 in this image (parity unpinned, oracle/video_input.py header): check what the published algorithm guarantees."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(ov.cv2_resize_linear_u8(img, 53, 37), img)                                   # same size: identity
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    box = ((big[0::2, 0::2].astype(int) + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(ov.cv2_resize_linear_u8(big, 32, 24), box)                                   # exact 2x: area average
    flat = np.full((30, 50, 3), 201, np.uint8)
    for size in ((384, 230), (17, 9), (50, 61)):
        assert (ov.cv2_resize_linear_u8(flat, *size) == 201).all()                                     # weights sum to 2048
    # against float bilinear with the same half-pixel centres: within 1 LSB (11-bit weights + two truncating shifts)
    for (w, h) in ((384, 216), (100, 100), (29, 77)):
        out = ov.cv2_resize_linear_u8(img, w, h).astype(float)
        xs = np.clip((np.arange(w) + 0.5) * (53 / w) - 0.5, 0, 52); ys = np.clip((np.arange(h) + 0.5) * (37 / h) - 0.5, 0, 36)
        x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, 52); fx = (xs - x0)[None, :, None]
        y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, 36); fy = (ys - y0)[:, None, None]
        f = img.astype(float)
        ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
        assert np.abs(out - ref).max() <= 1.0
    s0, s1, w0, w1 = ov.cv2_linear_taps(640, 384, True)
    assert ((w0 + w1) == 2048).all() and s0.min() == 0 and s1.max() == 639


def test_letterbox_frame_layout():
    frame = np.zeros((90, 160, 3), np.uint8); frame[..., 0] = 10; frame[..., 1] = 20; frame[..., 2] = 30      # B, G, R
    out = ov.letterbox_frame(frame, 64, pad_color=(1, 2, 3))
    nw, nh, (top, bottom, left, right) = ov.letterbox_geometry(160, 90, 64)
    assert out.shape == (3, 64, 64) and (nw, nh) == (64, 36) and (top, bottom, left, right) == (14, 14, 0, 0)
    assert (out[:, top:top + nh] == np.array([30, 20, 10], np.uint8)[:, None, None]).all()                     # RGB after the flip
    assert (out[:, :top] == np.array([3, 2, 1], np.uint8)[:, None, None]).all() and (out[:, top + nh:] == np.array([3, 2, 1], np.uint8)[:, None, None]).all()


# ---- GPU: the HIP kernel against the oracle, bit-exact ------------------------------------------------------------------
@pytest.fixture(scope='module')
def tiny_model():
    from helpers import hip_model
    return hip_model('A')[0]


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,R', [(360, 640, 384), (640, 360, 384), (480, 480, 384), (200, 300, 384), (768, 768, 384), (1080, 1920, 384),
                                   (97, 131, 56), (131, 97, 56), (720, 1280, 336), (2, 3, 8)])
def test_hip_letterbox_bit_exact(tiny_model, H, W, R):
    rng = np.random.default_rng(H * 7 + W)
    T = 3
    frames = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    for pad, flip in (((0, 0, 0), True), ((9, 120, 255), True), ((5, 6, 7), False)):
        out = pv.letterbox_frames(tiny_model, torch.from_numpy(frames), R, pad_color=pad, bgr_input=flip)
        torch.cuda.synchronize()
        want = np.stack([ov.letterbox_frame(f, R, pad, flip) for f in frames])
        got = out.cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (H, W, R, pad, flip, int(np.abs(got.astype(int) - want).max()))


@pytest.mark.gpu
def test_hip_load_video_frames_matches_reference_schedule(tiny_model):
    for r in [x for x in DATASET_RUNS if x['video'] in ('c.mp4', 'e.mp4', 'g.mp4')]:
        p = r['props']
        n = p['decodable']
        decoded = torch.zeros((n, p['h'], p['w'], 3), dtype=torch.uint8)
        idx = torch.arange(n)
        decoded[..., 0] = (idx & 255).to(torch.uint8)[:, None, None]; decoded[..., 1] = (idx >> 8).to(torch.uint8)[:, None, None]; decoded[..., 2] = 7
        out = pv.load_video_frames(tiny_model, decoded.cuda(), p['fps'], frame_count=float(_hdr(p)), output_fps=r['output_fps_arg'],
                                   resolution=r['resolution'], max_num_frames=r['max_num_frames'], time_instruction_format=r['time_instruction_format'])
        frames = out[0].cpu()
        assert list(frames.shape) == r['out_shape'] and out[1] == r['output_fps'] and out[2] == r['video_duration']
        R = r['resolution']
        centre = frames[:, :, R // 2, R // 2].to(torch.int64)            # RGB = (7, idx >> 8, idx & 255): the mock's tagging
        assert (centre[:, 0] == 7).all() and (centre[:, 1] * 256 + centre[:, 2]).tolist() == r['kept']
        if r['time_instruction_format']:
            assert out[3] == r['time_instruction']
        top, bottom, left, right = r['pads']
        if top: assert (frames[:, :, :top] == 0).all()
        if left: assert (frames[:, :, :, :left] == 0).all()
