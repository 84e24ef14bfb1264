"""Video in / out either side of the hot path: the loader / writer classes ``project/runml/predict.py`` drives for video
sources, restated over what this image has.

  VideoLoader / VideoLoaderCV      src/mon/core/data/dataset/video.py:35-300   (frames as uint8 HWC, ``{"frame", "meta"}`` datapoints)
  VideoWriter / VideoWriterCV      src/mon/core/video/io.py:109-315            (write / write_batch, ``denormalize``, ``save_image``)

The reference decodes with ``cv2.VideoCapture`` / encodes with ``cv2.VideoWriter``; cv2 (and ffmpeg) are absent from this image, so the
``*CV`` classes are gated: they work where cv2 imports and raise ``IEError`` where it does not.  ``FrameSequenceLoader`` /
``FrameSequenceWriter`` give the same protocol over what PIL reads and writes -- a directory of frame images, or a multi-frame file
(animated GIF / APNG / WebP / multi-page TIFF) -- and are what the tests and ``predict_video`` exercise here.  Decode / encode stay on
the host like in the reference; the frames cross to the device as uint8 HWC and come back the same way (``ie_zdce_forward_u8`` for
Zero-DCE, ``to_float`` -> ``infer`` -> ``to_uint8`` for the other models)."""
from __future__ import annotations

import abc
import pathlib
from typing import Sequence

import numpy as np
import torch

from . import capi

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")
MULTI_EXT = (".gif", ".apng", ".webp", ".tif", ".tiff", ".png")
VIDEO_EXT = (".mp4", ".avi", ".mov", ".mkv", ".m4v", ".mpg", ".mpeg", ".wmv", ".flv")


def _cv2():
    try:
        import cv2
        return cv2
    except ImportError as e:                          # gated, not faked: the product has no silent stand-in for a codec
        raise capi.IEError("cv2 is not installed in this image: VideoLoaderCV / VideoWriterCV need it (use FrameSequenceLoader / "
                           "FrameSequenceWriter for frame directories and multi-frame image files)") from e


def _frame_u8(frame, denormalize: bool = False) -> np.ndarray:
    """torch.Tensor (CHW / HWC, float [0, 1] or uint8) or ndarray -> HWC uint8 (core/image/utils.py:766-798: ``to_image_nparray``)."""
    if isinstance(frame, torch.Tensor):
        t = frame.detach()
        if t.dim() == 4 and t.shape[0] == 1:
            t = t[0]
        if t.dim() != 3:
            raise ValueError(f"`frame` must be a 3-D image, but got {tuple(frame.shape)}.")
        if t.shape[0] in (1, 3, 4) and t.shape[-1] not in (1, 3, 4):
            t = t.permute(1, 2, 0)
        a = t.cpu().numpy()
    elif isinstance(frame, np.ndarray):
        a = frame
    else:
        raise ValueError(f"`image` must be a `torch.Tensor` or `numpy.ndarray`, but got {type(frame)}.")
    if a.dtype != np.uint8:
        # the reference's rule is the explicit flag (utils.py:782-795: ``denormalize`` -> x * 255, then the uint8 cast); float frames without it are
        # taken to be in [0, 255] already.  No guessing from the data: a [0, 1] frame with one pixel at 1.01 must not turn black.
        a = a.astype(np.float32)
        if denormalize:
            a = np.clip(a, 0.0, 1.0) * 255.0
        a = np.clip(a, 0.0, 255.0).astype(np.uint8)     # utils.py:795: truncation (clipped first: a float above 255 would wrap)
    return np.ascontiguousarray(a)


def _frame_u8_ffmpeg(frame, denormalize: bool = False) -> np.ndarray:
    """The ffmpeg writer's rule (core/video/io.py:78-101, ``write_video_ffmpeg``), which differs by container type: a torch.Tensor goes
    through ``to_image_nparray`` with the explicit flag (like every other writer here), an ndarray is **auto-detected** --
    ``is_normalized_image`` (core/image/utils.py:165-170: ``abs(max) <= 1.0``) -> x * 255 -- whatever the flag says, then made channel-last
    and cast.  A float ndarray in [0, 1] written without the flag therefore is a picture, not a black frame."""
    if isinstance(frame, np.ndarray) and frame.dtype != np.uint8:
        a = frame
        if a.ndim == 4 and a.shape[0] == 1:
            a = a[0]
        if a.ndim != 3:
            raise ValueError(f"`frame` must be a 3-D image, but got {tuple(frame.shape)}.")
        if a.size and abs(float(np.amax(a))) <= 1.0:                  # is_normalized_image
            a = np.clip(a.astype(np.float32), 0.0, 1.0) * 255.0      # denormalize_image
        if a.shape[0] in (1, 3, 4) and a.shape[-1] not in (1, 3, 4):  # is_channel_first_image -> to_channel_last_image
            a = np.transpose(a, (1, 2, 0))
        return np.ascontiguousarray(np.clip(a, 0.0, 255.0).astype(np.uint8))
    return _frame_u8(frame, denormalize)


# ---- loaders -------------------------------------------------------------------------------------------------------------------
class VideoLoader(abc.ABC):
    """Frames of one source, indexable and iterable; ``loader[i]`` -> ``{"frame": HWC uint8 ndarray (or CHW float tensor with
    ``to_tensor=True``), "meta": {...}}`` (video.py:82-105)."""

    def __init__(self, root, to_tensor: bool = False, verbose: bool = False, *args, **kwargs):
        self.root, self.to_tensor, self.verbose = pathlib.Path(root), to_tensor, verbose
        self.num_frames, self.index = 0, 0
        self.get_data()
        if self.num_frames <= 0 and not getattr(self, "is_stream", False):
            raise RuntimeError("No datapoints in the dataset.")          # video.py:129-131

    def __len__(self) -> int:
        return max(int(self.num_frames), 0)               # a stream reports -1 frames (video.py:251): len() must not raise for it

    def __iter__(self):
        self.reset()
        return self

    def __next__(self) -> dict:
        if 0 <= self.num_frames <= self.index:
            raise StopIteration
        item = self[self.index]
        self.index += 1
        return item

    def __getitem__(self, index: int) -> dict:
        frame = self.get_frame(index)
        meta = {"id": index, "name": f"{self.root.stem}_{index:06d}", "path": self.root, "shape": tuple(frame.shape)}
        if self.to_tensor:                                                # to_image_tensor(normalize=True): CHW float [0, 1]
            frame = torch.from_numpy(frame).permute(2, 0, 1).float() / 255.0
        return {"frame": frame, "meta": meta}

    def batches(self, batch_size: int):
        """Stacks of up to ``batch_size`` frames as one (N, H, W, 3) uint8 tensor + their metas (predict.py:70-90 batching).  Walks the
        source front to back through ``get_frame`` until it ends -- a stream (``num_frames`` = -1) has no length to count to."""
        buf, metas = [], []
        i = 0
        while self.num_frames < 0 or i < self.num_frames:
            try:
                frame = self.get_frame(i)
            except IndexError:
                break                                   # end of a stream
            buf.append(torch.from_numpy(frame))
            metas.append({"id": i, "name": f"{self.root.stem}_{i:06d}", "path": self.root, "shape": tuple(frame.shape)})
            i += 1
            if len(buf) == batch_size:
                yield torch.stack(buf), metas
                buf, metas = [], []
        if buf:
            yield torch.stack(buf), metas

    # frame geometry: every loader sets frame_height / frame_width in get_data()
    frame_height = frame_width = 0

    @property
    def shape(self):
        return self.frame_height, self.frame_width, 3

    @property
    def imgsz(self):
        return self.frame_height, self.frame_width

    def reset(self):
        self.index = 0

    def close(self):
        pass

    @abc.abstractmethod
    def get_data(self):
        ...

    @abc.abstractmethod
    def get_frame(self, index: int) -> np.ndarray:
        ...


class FrameSequenceLoader(VideoLoader):
    """A directory of frame images (sorted by name) or one multi-frame image file, decoded with PIL."""

    def get_data(self):
        from PIL import Image
        self._files, self._multi = None, None
        if self.root.is_dir():
            self._files = sorted(p for p in self.root.iterdir() if p.suffix.lower() in IMAGE_EXT)
            self.num_frames = len(self._files)
        elif self.root.is_file() and self.root.suffix.lower() in MULTI_EXT:
            self._multi = Image.open(self.root)
            self.num_frames = int(getattr(self._multi, "n_frames", 1))
        else:
            raise IOError("Error when reading input stream or video file!")      # video.py:239
        f0 = self.get_frame(0) if self.num_frames else np.zeros((0, 0, 3), np.uint8)
        self.frame_height, self.frame_width = int(f0.shape[0]), int(f0.shape[1])

    def get_frame(self, index: int) -> np.ndarray:
        from PIL import Image
        if not 0 <= index < self.num_frames:
            raise IndexError(index)
        if self._files is not None:
            with Image.open(self._files[index]) as im:
                return np.array(im.convert("RGB"))
        self._multi.seek(index)
        return np.array(self._multi.convert("RGB"))

    def close(self):
        if self._multi is not None:
            self._multi.close()
            self._multi = None


class VideoLoaderCV(VideoLoader):
    """cv2.VideoCapture over a video file or stream (video.py:138-300).  Frames come back RGB (the reference converts BGR -> RGB)."""

    def get_data(self):
        cv2 = _cv2()
        if not (self.root.suffix.lower() in VIDEO_EXT or "://" in str(self.root)):
            raise IOError("Error when reading input stream or video file!")
        self.video_capture = cv2.VideoCapture(str(self.root), cv2.CAP_FFMPEG)
        self.is_stream = "://" in str(self.root)
        self.num_frames = -1 if self.is_stream else int(self.video_capture.get(cv2.CAP_PROP_FRAME_COUNT))
        self.frame_height = int(self.video_capture.get(cv2.CAP_PROP_FRAME_HEIGHT))
        self.frame_width = int(self.video_capture.get(cv2.CAP_PROP_FRAME_WIDTH))
        self.fps = int(self.video_capture.get(cv2.CAP_PROP_FPS))

    def get_frame(self, index: int) -> np.ndarray:
        cv2 = _cv2()
        if int(self.video_capture.get(cv2.CAP_PROP_POS_FRAMES)) != index:
            self.video_capture.set(cv2.CAP_PROP_POS_FRAMES, index)
        ok, frame = self.video_capture.read()
        if not ok:
            raise IndexError(index)
        return np.ascontiguousarray(frame[:, :, ::-1])

    def reset(self):
        super().reset()
        self.video_capture.set(_cv2().CAP_PROP_POS_FRAMES, 0)

    def close(self):
        if getattr(self, "video_capture", None) is not None:
            self.video_capture.release()
            self.video_capture = None


# ---- writers -------------------------------------------------------------------------------------------------------------------
class VideoWriter(abc.ABC):
    """io.py:109-193: ``dst``, ``image_size`` [H, W], ``frame_rate``, ``save_image`` (also keep every frame as an image file),
    ``denormalize`` (float [0, 1] frames -> [0, 255])."""

    def __init__(self, dst, image_size: int | Sequence[int] = (480, 640), frame_rate: float = 10, save_image: bool = False,
                 denormalize: bool = False, verbose: bool = False, *args, **kwargs):
        self.dst, self.denormalize, self.index = pathlib.Path(dst), denormalize, 0
        self.image_size = (int(image_size), int(image_size)) if isinstance(image_size, int) else tuple(int(v) for v in image_size)[-2:]
        self.frame_rate, self.save_image, self.verbose = frame_rate, save_image, verbose
        self.init()

    def __len__(self) -> int:
        return self.index

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @abc.abstractmethod
    def init(self):
        ...

    @abc.abstractmethod
    def close(self):
        ...

    @abc.abstractmethod
    def write(self, frame, path=None, denormalize: bool = False):
        ...

    def write_batch(self, frames, paths=None, denormalize: bool = False):
        paths = paths if paths is not None else [None] * len(frames)
        for f, p in zip(frames, paths):
            self.write(f, p, denormalize)


class FrameSequenceWriter(VideoWriter):
    """Frames as ``<dst>/<stem>_<index>.png`` files; ``dst`` with a multi-frame suffix (.gif / .apng / .webp / .tiff) additionally gets
    the whole sequence as one animated file on ``close()`` (PIL)."""

    def init(self):
        self._anim = self.dst if self.dst.suffix.lower() in (".gif", ".apng", ".webp", ".tif", ".tiff") else None
        self._dir = self.dst.parent / self.dst.stem if self._anim else self.dst
        self._dir.mkdir(parents=True, exist_ok=True)
        self._frames = []

    def write(self, frame, path=None, denormalize: bool = False):
        from PIL import Image
        a = _frame_u8(frame, denormalize or self.denormalize)
        im = Image.fromarray(a if a.shape[2] != 1 else a[:, :, 0])
        if self._anim is None or self.save_image:
            name = pathlib.Path(path).stem if path is not None else f"{self._dir.name}_{self.index:06d}"
            im.save(self._dir / f"{name}.png")
        if self._anim is not None:
            self._frames.append(im)
        self.index += 1

    def close(self):
        if getattr(self, "_anim", None) is not None and self._frames:
            first, rest = self._frames[0], self._frames[1:]
            first.save(self._anim, save_all=True, append_images=rest, duration=int(1000 / max(self.frame_rate, 1e-3)), loop=0)
            self._frames = []


class VideoWriterCV(VideoWriter):
    """cv2.VideoWriter (io.py:196-315): mp4v into ``<dst>.mp4``; frames arrive RGB and are written BGR."""

    def init(self):
        cv2 = _cv2()
        path = self.dst if self.dst.suffix else self.dst.with_suffix(".mp4")
        path.parent.mkdir(parents=True, exist_ok=True)
        self.video_writer = cv2.VideoWriter(str(path), cv2.VideoWriter_fourcc(*"mp4v"), float(self.frame_rate),
                                            (self.image_size[1], self.image_size[0]))

    def write(self, frame, path=None, denormalize: bool = False):
        a = _frame_u8(frame, denormalize or self.denormalize)
        self.video_writer.write(np.ascontiguousarray(a[:, :, ::-1]))
        self.index += 1

    def close(self):
        if getattr(self, "video_writer", None) is not None:
            self.video_writer.release()
            self.video_writer = None


class VideoWriterFFmpeg(VideoWriter):
    """Raw rgb24 frames piped into an ``ffmpeg`` process (io.py:318-470: ``<dst>/result.mp4`` for a directory, ``<dst stem>.mp4``
    otherwise, ``pix_fmt`` yuv420p, extra kwargs handed to ffmpeg as ``-key value``).  The reference drives the binary through
    ffmpeg-python; this is the same command line through ``subprocess``.  Gated like the cv2 classes: no binary, no writer."""

    def __init__(self, dst, image_size: int | Sequence[int] = (480, 640), frame_rate: float = 10, pix_fmt: str = "yuv420p",
                 save_image: bool = False, denormalize: bool = False, verbose: bool = False, *args, **kwargs):
        self.pix_fmt, self.ffmpeg_process, self.ffmpeg_kwargs = pix_fmt, None, kwargs
        super().__init__(dst, image_size, frame_rate, save_image, denormalize, verbose)

    def command(self, video_file) -> list:
        cmd = ["ffmpeg", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{self.image_size[1]}x{self.image_size[0]}", "-i", "pipe:",
               "-pix_fmt", self.pix_fmt]
        for k, v in self.ffmpeg_kwargs.items():
            cmd += [f"-{k}", str(v)]
        cmd += [str(video_file)]
        if not self.verbose:
            cmd += ["-loglevel", "quiet"]
        return cmd + ["-y"]

    def init(self):
        import shutil
        import subprocess
        if shutil.which("ffmpeg") is None:
            raise capi.IEError("the ffmpeg binary is not installed in this image: VideoWriterFFmpeg needs it (use FrameSequenceWriter)")
        video_file = self.dst / "result.mp4" if self.dst.is_dir() else self.dst.parent / f"{self.dst.stem}.mp4"
        video_file.parent.mkdir(parents=True, exist_ok=True)
        self.ffmpeg_process = subprocess.Popen(self.command(video_file), stdin=subprocess.PIPE)

    def write(self, frame, path=None, denormalize: bool = False):
        denormalize = denormalize or self.denormalize
        if self.save_image:
            from PIL import Image
            a = _frame_u8(frame, denormalize)
            d = self.dst if self.dst.is_dir() else self.dst.parent
            Image.fromarray(a if a.shape[2] != 1 else a[:, :, 0]).save(d / f"{pathlib.Path(path).stem}.png")
        self.ffmpeg_process.stdin.write(_frame_u8_ffmpeg(frame, denormalize).tobytes())
        self.index += 1

    def close(self):
        if getattr(self, "ffmpeg_process", None) is not None:
            self.ffmpeg_process.stdin.close()
            self.ffmpeg_process.wait()
            self.ffmpeg_process = None


def open_video(source, **kw) -> VideoLoader:
    """The loader for ``source``: a frame directory / multi-frame image (PIL) or a video file / stream (cv2)."""
    p = pathlib.Path(source)
    if p.is_dir() or p.suffix.lower() in MULTI_EXT:
        return FrameSequenceLoader(p, **kw)
    return VideoLoaderCV(p, **kw)


def predict_video(model, source, dst, batch_size: int = 8, frame_rate: float = 10) -> dict:
    """predict.py:58-120 for a video source: frames -> uint8 batches on the device -> model -> uint8 frames -> writer.
    Zero-DCE takes the frames as they are (``enhance_u8``); other models go through ``to_float`` / ``infer`` / ``to_uint8``."""
    from . import image as I
    loader = open_video(source)
    writer_cls = VideoWriterCV if pathlib.Path(dst).suffix.lower() in VIDEO_EXT else FrameSequenceWriter
    writer = writer_cls(dst, image_size=loader.imgsz, frame_rate=frame_rate)
    n = 0
    try:
        for frames, metas in loader.batches(batch_size):
            x = frames.to(model.device)
            if hasattr(model, "enhance_u8"):
                out = model.enhance_u8(x)["u8"]
            else:
                res = model.infer({"image": I.to_float(x)})
                out = I.to_uint8(res["enhanced"], "save_image")
            out = out.cpu()
            writer.write_batch([out[i].numpy() for i in range(out.shape[0])], [m["name"] for m in metas])
            n += int(out.shape[0])
    finally:
        loader.close()
        writer.close()
    return {"frames": n, "dst": str(dst)}
