"""uint8 <-> float image edges (src/mon/core/image/utils.py:766-847, core/image/io.py:127-145)
executed by the HIP library on device tensors."""
from __future__ import annotations

import torch

from . import capi


def to_uint8(x: torch.Tensor, rule: str = "save_image") -> torch.Tensor:
    """NCHW float [0,1] (cuda) -> NHWC uint8.  ``rule="save_image"``: torchvision's
    ``clamp(x*255+0.5, 0, 255)`` used by ``write_image`` (io.py:139-140); ``"truncate"``:
    ``to_image_nparray(denormalize=True)``'s ``(x*255).astype(uint8)`` (utils.py:795) on the
    clamped value."""
    if rule not in ("save_image", "truncate"):
        raise ValueError(f"`rule` must be 'save_image' or 'truncate', but got {rule}.")
    if not x.is_cuda:
        raise capi.IEError("to_uint8 runs on the HIP device only; move the tensor to cuda.")
    x = x.to(torch.float32).contiguous()
    if x.dim() == 3:
        x = x[None]
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_nchw_f32_to_u8_nhwc(capi.ptr(x), capi.ptr(out), n, c, h, w,
                                                     0 if rule == "save_image" else 1, capi.current_stream()),
                   "ie_nchw_f32_to_u8_nhwc")
    return out


def to_float(x: torch.Tensor) -> torch.Tensor:
    """NHWC uint8 (cuda) -> NCHW float32 /255 (``to_image_tensor(normalize=True)``, utils.py:801-847)."""
    if not x.is_cuda or x.dtype != torch.uint8:
        raise capi.IEError("to_float expects a cuda uint8 NHWC tensor.")
    x = x.contiguous()
    if x.dim() == 3:
        x = x[None]
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_u8_nhwc_to_nchw_f32(capi.ptr(x), capi.ptr(out), n, c, h, w, capi.current_stream()),
                   "ie_u8_nhwc_to_nchw_f32")
    return out


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """10*log10(1/MSE) on [0,1] floats (the build's own PSNR definition)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * __import__("math").log10(1.0 / mse)


# ---- file I/O at the edges (src/mon/core/image/io.py:37-145): decode / encode on the host (PIL), conversion on the device --
IMAGE_FILE_FORMATS = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp", ".ppm")


def read_image(path, device="cuda") -> torch.Tensor:
    """``read_image(to_tensor=True, normalize=True)`` (io.py:37-83): file -> (1,3,H,W) float32 in [0,1] on ``device``.
    The uint8 HWC pixels go to the device as they are; the /255 and the HWC->NCHW transpose run in the library."""
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        arr = np.array(im.convert("RGB"), dtype=np.uint8)          # a writable copy
    return to_float(torch.from_numpy(np.ascontiguousarray(arr))[None].to(device))


def write_image(path, image: torch.Tensor):
    """``write_image`` for tensors (io.py:127-145) == torchvision ``save_image``: clamp(x*255+0.5, 0, 255) -> uint8 -> file.
    ``image``: (3,H,W) or (1,3,H,W) float on the device, or a uint8 NHWC tensor produced by the library."""
    import os
    from PIL import Image
    if image.dtype != torch.uint8:
        if image.dim() == 4 and image.shape[0] != 1:
            raise ValueError("write_image saves one image per call (the reference tiles batches into a grid; predict.py never does).")
        image = to_uint8(image, "save_image")
    if image.dim() == 4:
        image = image[0]
    os.makedirs(os.path.dirname(os.path.abspath(str(path))) or ".", exist_ok=True)
    Image.fromarray(image.cpu().numpy()).save(str(path))


def list_images(source) -> list:
    """Image files of a directory (sorted) or the single file (the part of ``parse_io_worker`` predict.py relies on)."""
    import os
    source = str(source)
    if os.path.isdir(source):
        return sorted(os.path.join(source, f) for f in os.listdir(source) if f.lower().endswith(IMAGE_FILE_FORMATS))
    if os.path.isfile(source) and source.lower().endswith(IMAGE_FILE_FORMATS):
        return [source]
    raise ValueError(f"`source` must be an image file or a directory of images, but got {source}.")


def predict(model_config: dict, source, save_dir, image_size=512, resize: bool = False, device="cuda", save_debug: bool = False) -> dict:
    """The inference driver of ``project/runml/predict.py:23-125``: build the model from the registry config, ``infer`` every
    image of ``source`` (resize to a multiple of 32 and back), write ``<stem>.png`` with the save_image rule into
    ``save_dir/<data_name>/``.  Returns {"save_dir", "files", "avg_time"}."""
    import os
    from .registry import MODELS
    model = MODELS.build(config=dict(model_config)).to(device).eval()
    files = list_images(source)
    data_name = os.path.basename(os.path.normpath(str(source))) if os.path.isdir(str(source)) else os.path.splitext(os.path.basename(str(source)))[0]
    out_dir = os.path.join(str(save_dir), data_name)
    times, written = [], []
    for f in files:
        stem = os.path.splitext(os.path.basename(f))[0]
        outputs = model.infer({"image": read_image(f, device)}, image_size=image_size, resize=resize)
        t = outputs.pop("time", None)
        if t:
            times.append(t)
        _, output = outputs.popitem()                       # "enhanced" is the last key (predict.py:97)
        path = os.path.join(out_dir, f"{stem}.png")
        write_image(path, output)
        written.append(path)
        if save_debug:
            for k, v in outputs.items():
                if isinstance(v, torch.Tensor) and v.dim() == 4 and v.shape[1] == 3:
                    write_image(os.path.join(str(save_dir), f"{data_name}_debug", f"{stem}_{k}.png"), v)
    return {"save_dir": out_dir, "files": written, "avg_time": float(sum(times) / len(times)) if times else 0.0}
