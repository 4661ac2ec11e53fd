"""Container reader for the front of `load_video`: what `cv2.VideoCapture` provides to the reference (test/datasets.py:32-50,
demo/liveinfer.py:8-30) -- the stream's fps, its header frame count, and the decoded BGR frames in order.

This image ships no video codec library (no OpenCV, PyAV, decord or ffmpeg), so the reader covers the container / codec pair that needs
none: **AVI (RIFF) with Motion-JPEG or uncompressed frames** -- every frame is an independent JPEG (decoded with Pillow's libjpeg) or a
bottom-up BGR bitmap.  That closes the loop file -> frames -> GPU sampling / letterbox (video_input.load_video_frames) -> stream driver;
H.264 / VP9 clips need to be transcoded once (`ffmpeg -c:v mjpeg -q:v 2 -an`), as the reference's own data preparation already does with
ffmpeg for fps / resolution (data/utils.py:60-96 `ffmpeg_once`).

What cv2 reports for such a file and this reader reproduces: CAP_PROP_FPS = strh.dwRate / strh.dwScale (falls back to 1e6 / avih.dwMicroSecPerFrame),
CAP_PROP_FRAME_COUNT = strh.dwLength (falls back to avih.dwTotalFrames), frames in file order, colour order BGR.
PARITY UNPINNED for the pixel values of MJPEG: OpenCV decodes with its bundled libjpeg-turbo, Pillow with the system libjpeg; IDCT and chroma
upsampling may differ by +-1 per channel.  `write_mjpeg_avi` exists for tests and for producing transcoded fixtures.
"""
import io
import struct
import numpy as np
import torch


class AviError(ValueError):
    pass


def _chunks(buf, start, end):
    """(fourcc, data_start, size) of every chunk in buf[start:end]; chunks are word-aligned."""
    pos = start
    while pos + 8 <= end:
        cc = bytes(buf[pos:pos + 4])
        size = struct.unpack_from('<I', buf, pos + 4)[0]
        yield cc, pos + 8, size
        pos += 8 + size + (size & 1)


def read_avi(path, max_frames=None, select=None):
    """-> (frames uint8 [N,H,W,3] BGR host tensor, fps, header_frame_count).  Raises AviError on anything but MJPEG / uncompressed RGB24 video.

    `select(fps, header_frame_count, n_frames_in_file) -> iterable of frame indices`: decode ONLY those frames (in file order) and return
    (frames [len(kept),H,W,3], fps, header_frame_count, kept).  Every frame of the two supported codecs is coded independently, so a frame the
    sampling schedule drops need not be decoded at all -- cv2.VideoCapture.read() in the reference's loop (test/datasets.py:46-50) decodes every
    frame and throws 29 of 30 away at 1 fps; the kept pixels are the same either way."""
    buf = memoryview(open(path, 'rb').read())
    if len(buf) < 12 or bytes(buf[:4]) != b'RIFF' or bytes(buf[8:12]) != b'AVI ':
        raise AviError(f'{path}: not a RIFF AVI file')
    info = dict(us_per_frame=0, total_frames=0, width=0, height=0, rate=0, scale=0, length=0, handler=b'', compression=b'', bits=24, stream=None)
    movi = None

    def walk(start, end, n_stream):
        nonlocal movi
        for cc, at, size in _chunks(buf, start, end):
            if cc == b'LIST':
                kind = bytes(buf[at:at + 4])
                if kind == b'movi':
                    movi = (at + 4, at + size)
                elif kind in (b'hdrl', b'strl'):
                    walk(at + 4, at + size, n_stream)
                    if kind == b'strl':
                        n_stream[0] += 1
            elif cc == b'avih':
                info['us_per_frame'], = struct.unpack_from('<I', buf, at)
                info['total_frames'], = struct.unpack_from('<I', buf, at + 16)
                info['width'], info['height'] = struct.unpack_from('<II', buf, at + 32)
            elif cc == b'strh' and bytes(buf[at:at + 4]) == b'vids' and info['stream'] is None:
                info['stream'] = n_stream[0]
                info['handler'] = bytes(buf[at + 4:at + 8])
                info['scale'], info['rate'] = struct.unpack_from('<II', buf, at + 20)
                info['length'], = struct.unpack_from('<I', buf, at + 32)
            elif cc == b'strf' and info['stream'] == n_stream[0] and not info['compression']:
                w, h = struct.unpack_from('<ii', buf, at + 4)
                info['bits'], = struct.unpack_from('<H', buf, at + 14)
                info['compression'] = bytes(buf[at + 16:at + 20])
                info['width'], info['height'] = w, h

    walk(12, len(buf), [0])
    if movi is None or info['stream'] is None:
        raise AviError(f'{path}: no video stream / no movi list')
    comp = info['compression'].upper()
    mjpeg = comp in (b'MJPG', b'JPEG') or info['handler'].upper() in (b'MJPG',)
    raw = comp in (b'\x00\x00\x00\x00', b'DIB ', b'RGB ') and info['bits'] == 24
    if not (mjpeg or raw):
        raise AviError(f"{path}: video codec {info['compression']!r} / {info['handler']!r} is not supported without a codec library "
                       '(transcode to Motion-JPEG: ffmpeg -i in.mp4 -c:v mjpeg -q:v 2 -an out.avi)')
    tag = b'%02d' % info['stream']
    frames = []
    from PIL import Image
    H, W = abs(info['height']), info['width']

    def frames_in(start, end):
        for cc, at, size in _chunks(buf, start, end):
            if cc == b'LIST' and bytes(buf[at:at + 4]) == b'rec ':
                yield from frames_in(at + 4, at + size)
            elif cc[:2] == tag and cc[2:] in (b'dc', b'db') and size > 0:
                yield at, size

    fps = info['rate'] / info['scale'] if info['rate'] and info['scale'] else (1e6 / info['us_per_frame'] if info['us_per_frame'] else 0.0)
    located = list(frames_in(*movi))
    if max_frames:
        located = located[:max_frames]
    count = info['length'] or info['total_frames'] or len(located)
    kept = None
    if select is not None:
        kept = [int(i) for i in select(float(fps), int(count), len(located))]
        located = [located[i] for i in kept]
    for at, size in located:
        data = buf[at:at + size]
        if mjpeg:
            rgb = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
            frames.append(rgb[:, :, ::-1])                                   # cv2 hands out BGR
        else:
            stride = (W * 3 + 3) & ~3
            a = np.frombuffer(data, dtype=np.uint8, count=stride * H).reshape(H, stride)[:, :W * 3].reshape(H, W, 3)
            frames.append(a[::-1] if info['height'] > 0 else a)              # positive biHeight = bottom-up rows; stored order is already BGR
    if not frames and select is not None and not kept:
        return torch.empty((0, H, W, 3), dtype=torch.uint8), float(fps), int(count), kept          # (an empty schedule is the caller's error to raise, as np.stack([]) is the reference's)
    if not frames:
        raise AviError(f'{path}: no decodable frames')
    out = torch.from_numpy(np.ascontiguousarray(np.stack(frames)))
    if select is not None:
        return out, float(fps), int(count), kept
    return out, float(fps), int(count)


def write_mjpeg_avi(path, frames_rgb, fps, quality=95, header_frame_count=None):
    """frames_rgb uint8 [N,H,W,3] (RGB) -> a minimal Motion-JPEG AVI (hdrl with avih / strh / strf, movi with 00dc chunks, idx1)."""
    from PIL import Image
    fr = np.asarray(frames_rgb)
    N, H, W, _ = fr.shape
    jpgs = []
    for f in fr:
        b = io.BytesIO(); Image.fromarray(f).save(b, format='JPEG', quality=quality, subsampling=0); jpgs.append(b.getvalue())
    scale, rate = 1000, int(round(fps * 1000))
    n_hdr = N if header_frame_count is None else int(header_frame_count)

    def chunk(cc, data):
        return cc + struct.pack('<I', len(data)) + data + (b'\x00' if len(data) & 1 else b'')

    avih = struct.pack('<IIIIIIIIII4I', int(1e6 / fps), 0, 0, 0x10, n_hdr, 0, 1, max(len(j) for j in jpgs), W, H, 0, 0, 0, 0)
    strh = b'vids' + b'MJPG' + struct.pack('<IHHIIIIIIII4H', 0, 0, 0, 0, scale, rate, 0, n_hdr, max(len(j) for j in jpgs), 0xffffffff, 0, 0, 0, W, H)
    strf = struct.pack('<IiiHH4sIiiII', 40, W, H, 1, 24, b'MJPG', W * H * 3, 0, 0, 0, 0)
    strl = b'strl' + chunk(b'strh', strh) + chunk(b'strf', strf)
    hdrl = b'hdrl' + chunk(b'avih', avih) + chunk(b'LIST', strl)
    movi, idx, off = b'movi', b'', 4
    for j in jpgs:
        c = chunk(b'00dc', j)
        idx += b'00dc' + struct.pack('<III', 0x10, off, len(j))
        movi += c; off += len(c)
    body = b'AVI ' + chunk(b'LIST', hdrl) + chunk(b'LIST', movi) + chunk(b'idx1', idx)
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', len(body)) + body)


def load_video(model, path, output_fps=2, resolution=384, max_num_frames=100, time_instruction_format=None, pad_color=(0, 0, 0), budget='ceil'):
    """`load_video(file)` of the reference (test/datasets.py:32-85; budget='floor' + pad (0,0,0): demo/liveinfer.py:8-57) for the containers `read_avi`
    covers: decode on the host, then the reference's sampling schedule, letterbox resize, pad and BGR->RGB on the GPU (video_input.load_video_frames)."""
    from .video_input import load_video_frames
    frames, fps, count = read_avi(path)
    return load_video_frames(model, frames, fps, count, output_fps=output_fps, resolution=resolution, max_num_frames=max_num_frames,
                             time_instruction_format=time_instruction_format, pad_color=pad_color, budget=budget)
