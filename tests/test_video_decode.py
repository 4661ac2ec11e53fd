"""Container reader (mmduet_amd/video_decode.py): what cv2.VideoCapture hands the reference's load_video (test/datasets.py:32-50) for Motion-JPEG and
uncompressed AVI files -- fps, header frame count, BGR frames in order."""
import io, struct
import numpy as np
import pytest
import torch
from mmduet_amd.video_decode import read_avi, write_mjpeg_avi, AviError


def test_mjpeg_avi_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 256, (9, 36, 52, 3)).astype(np.uint8)
    fr[:, 8:20, 10:30] = (fr[:, 8:20, 10:30] // 8) * 8          # some structure besides noise
    write_mjpeg_avi(tmp_path / 'a.avi', fr, 29.97, quality=92, header_frame_count=11)      # header count != decodable count, as cv2 sees it in the wild
    frames, fps, count = read_avi(tmp_path / 'a.avi')
    assert frames.shape == (9, 36, 52, 3) and frames.dtype == torch.uint8 and fps == pytest.approx(29.97) and count == 11
    # pixel values: exactly Pillow's decode of the very JPEG bytes in the file, channel order flipped to BGR
    for i in range(9):
        b = io.BytesIO(); Image.fromarray(fr[i]).save(b, format='JPEG', quality=92, subsampling=0)
        ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert('RGB'))[:, :, ::-1]
        assert np.array_equal(frames[i].numpy(), ref)
    assert np.abs(frames.numpy()[..., ::-1].astype(int) - fr).mean() < 8      # and close to the source
    assert read_avi(tmp_path / 'a.avi', max_frames=4)[0].shape[0] == 4


def test_uncompressed_avi_bottom_up_bgr(tmp_path):
    H, W, N = 5, 7, 3                      # W*3 = 21 -> rows padded to 24 bytes
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8)

    def chunk(cc, d):
        return cc + struct.pack('<I', len(d)) + d + (b'\x00' if len(d) & 1 else b'')
    stride = (W * 3 + 3) & ~3
    avih = struct.pack('<IIIIIIIIII4I', 40000, 0, 0, 0x10, N, 0, 1, stride * H, W, H, 0, 0, 0, 0)
    strh = b'vids' + b'DIB ' + struct.pack('<IHHIIIIIIII4H', 0, 0, 0, 0, 1, 25, 0, N, stride * H, 0xffffffff, 0, 0, 0, W, H)
    strf = struct.pack('<IiiHHIIiiII', 40, W, H, 1, 24, 0, stride * H, 0, 0, 0, 0)
    hdrl = b'hdrl' + chunk(b'avih', avih) + chunk(b'LIST', b'strl' + chunk(b'strh', strh) + chunk(b'strf', strf))
    movi = b'movi'
    for f in bgr:
        rows = np.zeros((H, stride), np.uint8); rows[:, :W * 3] = f[::-1].reshape(H, W * 3)     # bottom-up
        movi += chunk(b'00db', rows.tobytes())
    body = b'AVI ' + chunk(b'LIST', hdrl) + chunk(b'LIST', movi)
    (tmp_path / 'r.avi').write_bytes(b'RIFF' + struct.pack('<I', len(body)) + body)
    frames, fps, count = read_avi(tmp_path / 'r.avi')
    assert fps == 25.0 and count == N and np.array_equal(frames.numpy(), bgr)


def test_unsupported_codec_fails_loudly(tmp_path):
    rng = np.random.default_rng(3)
    write_mjpeg_avi(tmp_path / 'a.avi', rng.integers(0, 256, (2, 16, 16, 3)).astype(np.uint8), 10)
    data = (tmp_path / 'a.avi').read_bytes().replace(b'MJPG', b'H264')
    (tmp_path / 'h.avi').write_bytes(data)
    with pytest.raises(AviError, match='not supported without a codec library'):
        read_avi(tmp_path / 'h.avi')
    (tmp_path / 'x.avi').write_bytes(b'not a riff file at all')
    with pytest.raises(AviError):
        read_avi(tmp_path / 'x.avi')


@pytest.mark.gpu
def test_cli_video_entry_equals_decoded_entry(tmp_path, monkeypatch):
    """`"video": "clip.avi"` = read_avi + the reference's sampling / letterbox on the GPU; same record as feeding the decoder's output directly."""
    import json
    import mmduet_amd.inference as inf
    import mmduet_amd.__main__ as cli
    from helpers import hip_model, tokenizer_for
    model, cfgd, _ = hip_model('A')
    tok = tokenizer_for(model.config)
    monkeypatch.setattr(inf, 'build_model_and_tokenizer', lambda **kw: (model, tok))
    R = model.config.frame_resolution
    rng = np.random.default_rng(4)
    write_mjpeg_avi(tmp_path / 'c.avi', rng.integers(0, 256, (40, 30, 50, 3)).astype(np.uint8), 10.0)
    frames, fps, count = read_avi(tmp_path / 'c.avi')
    np.save(tmp_path / 'c_raw.npy', frames.numpy())
    conv = [{'role': 'user', 'content': 'describe', 'time': 0.0}]
    outs = []
    for tag, entry in (('v', {'question_id': 'q', 'video': 'c.avi', 'conversation': conv}), ('d', {'question_id': 'q', 'decoded': 'c_raw.npy', 'input_fps': fps, 'frame_count': count, 'conversation': conv})):
        json.dump([entry], open(tmp_path / f'{tag}.json', 'w'))
        cli.main(['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', str(tmp_path), '--test_fname', str(tmp_path / f'{tag}.json'),
                  '--output_fname', str(tmp_path / f'{tag}.jsonl'), '--frame_fps', '2', '--frame_resolution', str(R), '--max_num_frames', '6',
                  '--stream_end_prob_threshold', '0.5', '--max_new_tokens', '4'])
        outs.append([json.loads(l) for l in open(tmp_path / f'{tag}.jsonl')])
    assert outs[0] == outs[1] and len(outs[0][0]['debug_data']) == 6 and outs[0][0]['video_duration'] == 4.0


def test_sampled_decode_equals_decode_then_pick(tmp_path):
    """read_avi(select=...) decodes only the frames the reference's sampling loop keeps (test/datasets.py:41-50); the kept frames equal read_avi + pick."""
    from mmduet_amd.video_input import sampling_selector, frame_sampling_plan
    rng = np.random.default_rng(5)
    fr = rng.integers(0, 256, (40, 24, 32, 3)).astype(np.uint8)
    write_mjpeg_avi(tmp_path / 'c.avi', fr, 10.0, quality=90, header_frame_count=43)
    full, fps, count = read_avi(tmp_path / 'c.avi')
    for out_fps, cap in ((2.0, 100), (1.0, 3), (0, 7)):
        sel = sampling_selector(out_fps, cap)
        got, fps2, count2, kept = read_avi(tmp_path / 'c.avi', select=sel)
        ref_kept = frame_sampling_plan(fps, count, out_fps, cap, n_decodable=len(full))[0]
        assert kept == ref_kept and sel.n_decodable == 40 and (fps2, count2) == (fps, count)
        assert torch.equal(got, full[torch.as_tensor(kept)])


def test_clip_prefetcher_order_and_inline_mode():
    import threading, time
    from mmduet_amd.prefetch import ClipPrefetcher
    seen = []

    def load(i):
        time.sleep(0.01 * (5 - i % 5))           # later indices finish earlier: order must still hold
        seen.append((i, threading.current_thread().name))
        return None if i == 3 else {'index': i}
    for workers in (0, 1, 4):
        seen.clear()
        with ClipPrefetcher(load, range(9), workers=workers) as pf:
            first = pf.take()
            assert first == (0, {'index': 0})
            if workers:
                time.sleep(0.2)
                assert pf.next_ready()                       # the following clips were loaded while the consumer was busy
            else:
                assert not pf.next_ready() and len(seen) == 1   # inline: nothing is loaded before it is asked for
            rest = list(pf)
        assert [i for i, _ in rest] == list(range(1, 9)) and rest[2][1] is None and pf.take() is None
        assert all(n.startswith('mmduet-clip') for _, n in seen) == (workers > 0)
