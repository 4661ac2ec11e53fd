#!/usr/bin/env python
"""Generate tests/golden/video_input.json: the frame-sampling schedule and letterbox geometry of the reference's two
`load_video` loops (test/datasets.py:32-85, demo/liveinfer.py:8-57), run HERE with a recording mock of cv2.

cv2 is not installed in this image, so the pixel arithmetic of cv2.resize cannot be pinned; what IS pinned is every
decision the reference's own code makes around it: which decoded frames are kept (the float accumulation
`cur_time += 1 / input_fps` against `frame_sec`), the resize target, the four pad widths, the channel flip, output fps,
duration, the max_num_frames cut and the time-instruction strings.  The mock tags every decoded frame with its index so the
kept indices can be read back from the returned tensor.  Only inputs and outputs are stored.
"""
import ast, json, math, os, types
import numpy as np
import torch

DATASETS = '/root/reference/test/datasets.py'
DEMO = '/root/reference/demo/liveinfer.py'


class MockCv2(types.SimpleNamespace):
    CAP_PROP_FPS, CAP_PROP_FRAME_COUNT, CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT = 5, 7, 3, 4
    BORDER_CONSTANT, COLOR_BGR2RGB = 0, 4

    def __init__(self, videos):
        super().__init__()
        self.videos, self.calls = videos, []
        outer = self

        class VideoCapture:
            def __init__(self, path):
                self.v = outer.videos[os.path.basename(path)]; self.i = 0; self.open = True
            def get(self, prop):
                v = self.v
                return {5: v['fps'], 7: float(v['decodable'] if v.get('count_from_header') is None else v['count_from_header']),
                        3: float(v['w']), 4: float(v['h'])}[prop]
            def isOpened(self): return self.open
            def read(self):
                if self.i >= self.v['decodable']:
                    return False, None
                f = np.zeros((self.v['h'], self.v['w'], 3), np.uint8)
                f[..., 0] = self.i & 255; f[..., 1] = self.i >> 8; f[..., 2] = 7          # B, G carry the index; R = 7
                self.i += 1
                return True, f
            def release(self): self.open = False
        self.VideoCapture = VideoCapture

    def resize(self, frame, dsize):
        self.calls.append(('resize', [int(frame.shape[1]), int(frame.shape[0])], [int(dsize[0]), int(dsize[1])]))
        out = np.empty((dsize[1], dsize[0], 3), np.uint8); out[:] = frame[0, 0]
        return out

    def copyMakeBorder(self, img, top, bottom, left, right, borderType, value):
        self.calls.append(('pad', [int(top), int(bottom), int(left), int(right)], [int(v) for v in value]))
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8); out[:] = np.asarray(value, np.uint8)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    def cvtColor(self, img, code):
        assert code == self.COLOR_BGR2RGB
        return img[..., ::-1]


def _exec_defs(path, names, ns):
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns


def decode_kept(frames):
    """[T,3,R,R] RGB tensor -> kept source indices (from the centre pixel) and the pad/content mask summary."""
    R = frames.shape[-1]
    c = frames[:, :, R // 2, R // 2].numpy().astype(int)            # RGB = (7, idx>>8, idx&255)
    assert (c[:, 0] == 7).all()
    return (c[:, 1] * 256 + c[:, 2]).tolist()


def main():
    videos = {
        'a.mp4': dict(fps=30.0, decodable=900, w=640, h=360),
        'b.mp4': dict(fps=29.97002997002997, decodable=1000, w=1280, h=720),
        'c.mp4': dict(fps=25.0, decodable=333, w=360, h=640),
        'd.mp4': dict(fps=23.976023976023978, decodable=2400, w=480, h=480),
        'e.mp4': dict(fps=15.0, decodable=40, w=1920, h=1080, count_from_header=45),      # header over-reports: decode ends early
        'f.mp4': dict(fps=60.0, decodable=3000, w=854, h=480),
        'g.mp4': dict(fps=10.0, decodable=95, w=300, h=200),                              # upscale
        'h.mp4': dict(fps=30.0, decodable=300, w=768, h=768),                             # exact 2x
    }
    runs = []
    for name, v in videos.items():
        for output_fps, R, max_frames, fmt in [(2, 384, 100, None), (1, 384, 400, 'timechat'), (0.5, 336, 60, 'vtimellm'), (-1, 384, 32, None), (2, 384, 7, None)]:
            cv2 = MockCv2(videos)
            ns = _exec_defs(DATASETS, {'FastAndAccurateStreamingVideoQADataset'}, {'cv2': cv2, 'np': np, 'torch': torch, 'math': math, 'os': os, 'json': json, 'Dataset': object, 'random': None})
            ds = object.__new__(ns['FastAndAccurateStreamingVideoQADataset'])
            ds.video_base_folder = ''; ds.output_fps = output_fps; ds.output_resolution = R; ds.max_num_frames = max_frames
            ds.pad_color = (0, 0, 0); ds.time_instruction_format = fmt
            out = ds.load_video(name)
            frames, ofps, dur = out[:3]
            resize = [c for c in cv2.calls if c[0] == 'resize']; pad = [c for c in cv2.calls if c[0] == 'pad']
            assert all(r == resize[0] for r in resize) and all(p == pad[0] for p in pad)
            runs.append(dict(loop='datasets', video=name, props=v, output_fps_arg=output_fps, resolution=R, max_num_frames=max_frames,
                             time_instruction_format=fmt, kept=decode_kept(frames), out_shape=list(frames.shape), output_fps=ofps,
                             video_duration=dur, resize_to=resize[0][2] if resize else None, pads=pad[0][1] if pad else None,
                             time_instruction=out[3] if fmt else None))
        for output_fps in (2, 1, -1):
            cv2 = MockCv2(videos)
            ns = _exec_defs(DEMO, {'load_video'}, {'cv2': cv2, 'np': np, 'torch': torch, 'math': math})
            try:
                frames, originals = ns['load_video'](name, output_fps)
            except Exception as e:          # np.stack of an empty list when floor() leaves no frame
                runs.append(dict(loop='demo', video=name, props=v, output_fps_arg=output_fps, error=type(e).__name__)); continue
            resize = [c for c in cv2.calls if c[0] == 'resize']; pad = [c for c in cv2.calls if c[0] == 'pad']
            runs.append(dict(loop='demo', video=name, props=v, output_fps_arg=output_fps, resolution=384, max_num_frames=400,
                             kept=decode_kept(frames), out_shape=list(frames.shape), resize_to=resize[0][2], pads=pad[0][1],
                             n_originals=len(originals)))
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'video_input.json')
    json.dump({'runs': runs}, open(out_path, 'w'))
    print('wrote', out_path, os.path.getsize(out_path), 'bytes,', len(runs), 'runs')


if __name__ == '__main__':
    main()
