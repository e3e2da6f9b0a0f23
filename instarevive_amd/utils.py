"""Host-side plumbing of the CLI: the helper functions test_scripts/inference.py imports from the reference's `utils`
package, re-implemented here with the same names, arguments and results (pinned bit-for-bit by tests/golden/glue.npz and
tests/test_host_cpu.py):

    instantiate_from_config, get_obj_from_str, load_state_dict   <- utils/common.py:7-18,35-51
    list_image_files, get_file_name_parts                        <- utils/file.py:20-47
    center_crop_arr, auto_resize, pad                            <- utils/image/common.py:12-36,229-249

Nothing here touches the GPU.
"""
import itertools
import math
import os
import pkgutil
from typing import Any, Iterator, List, Mapping, Sequence, Tuple

import numpy as np
from PIL import Image

# `target:` strings of the reference's YAML files -> the classes of this package that stand in for them
TARGET_ALIASES = {"diffusion.model.swinir.SwinIR": "instarevive_amd.models.SwinIR"}
IMAGE_EXTENSIONS = (".jpg", ".png", ".jpeg", ".arw")
_DDP_PREFIX = "module."


# ------------------------------------------------------------------------------------------------ config -> object
def get_obj_from_str(string: str, reload: bool = False) -> Any:
    """Dotted path -> Python object. `reload` is accepted for signature compatibility and ignored (nothing is re-imported)."""
    dotted = TARGET_ALIASES.get(string, string)
    module_name, _, attr = dotted.rpartition(".")
    if not module_name:
        raise ValueError(f"'{string}' is not a dotted 'package.module.Name' path")
    return pkgutil.resolve_name(f"{module_name}:{attr}")


def instantiate_from_config(config: Mapping[str, Any]) -> Any:
    """{'target': 'pkg.mod.Class', 'params': {...}} -> Class(**params); a missing 'target' is a KeyError like the reference's."""
    try:
        target = config["target"]
    except KeyError:
        raise KeyError("Expected key `target` to instantiate.") from None
    kwargs = config.get("params") or {}
    return get_obj_from_str(target)(**kwargs)


def load_yaml(path: str) -> dict:
    """Stands in for OmegaConf.load on plain configs such as configs/swinir.yaml (omegaconf is not a dependency here)."""
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh)


# ------------------------------------------------------------------------------------------------ checkpoints
def _has_ddp_prefix(keys: Sequence[str]) -> bool:
    return bool(keys) and keys[0].startswith(_DDP_PREFIX)


def load_state_dict(model, state_dict: Mapping[str, Any], strict: bool = False) -> None:
    """Load a checkpoint that may be wrapped in {'state_dict': ...} and may or may not carry DistributedDataParallel's
    'module.' key prefix: the prefix is added or removed so that it matches what `model` itself reports (decided, like the
    reference, from the FIRST key on either side)."""
    weights = state_dict.get("state_dict", state_dict)
    want = _has_ddp_prefix(list(model.state_dict().keys()))
    have = _has_ddp_prefix(list(weights.keys()))
    if want != have:
        if want:
            weights = {_DDP_PREFIX + k: v for k, v in weights.items()}
        else:
            weights = {k[len(_DDP_PREFIX):]: v for k, v in weights.items()}
    model.load_state_dict(weights, strict=strict)


# ------------------------------------------------------------------------------------------------ files
def _walk_images(root: str, exts: Tuple[str, ...], follow_links: bool) -> Iterator[str]:
    for folder, _subdirs, names in os.walk(root, followlinks=follow_links):
        for name in names:
            if os.path.splitext(name)[1].lower() in exts:
                yield os.path.join(folder, name)


def list_image_files(img_dir: str, exts: Tuple[str, ...] = IMAGE_EXTENSIONS, follow_links: bool = False, log_progress: bool = False,
                     log_every_n_files: int = 10000, max_size: int = -1) -> List[str]:
    """Image files under img_dir in os.walk order (the reference's order: no sorting), filtered by lower-cased extension;
    at most max_size of them when max_size >= 0."""
    found = _walk_images(img_dir, tuple(exts), follow_links)
    if max_size >= 0:
        found = itertools.islice(found, max_size)
    files: List[str] = []
    for path in found:
        files.append(path)
        if log_progress and len(files) % log_every_n_files == 0:
            print(f"find {len(files)} images in {img_dir}")
    return files


def get_file_name_parts(file_path: str) -> Tuple[str, str, str]:
    """'a/b/name.ext' -> ('a/b', 'name', '.ext')."""
    stem, ext = os.path.splitext(os.path.basename(file_path))
    return os.path.dirname(file_path), stem, ext


# ------------------------------------------------------------------------------------------------ image geometry
def _scaled_size(size: Tuple[int, int], factor: float, rounding) -> Tuple[int, int]:
    return tuple(int(rounding(edge * factor)) for edge in size)


def auto_resize(img: Image.Image, size: int) -> Image.Image:
    """Bicubic upscale so that the SHORT edge reaches `size` (edges rounded up); images that are large enough are copied."""
    short = min(img.size)
    if short >= size:
        return img.copy()
    return img.resize(_scaled_size(img.size, size / short, math.ceil), Image.BICUBIC)


def pad(img: np.ndarray, scale: int) -> np.ndarray:
    """Zero-pad an HWC array at the bottom / right up to the next multiples of `scale`."""
    rows, cols = img.shape[:2]
    return np.pad(img, ((0, -rows % scale), (0, -cols % scale), (0, 0)), mode="constant", constant_values=0)


def center_crop_arr(pil_image: Image.Image, image_size: int) -> np.ndarray:
    """ADM-style centre crop: halve with a box filter while the short edge is at least twice the target, bicubic-resize the
    short edge to the target (edges rounded to nearest), cut the central image_size x image_size window."""
    while min(pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize(_scaled_size(pil_image.size, 0.5, math.floor), resample=Image.BOX)
    pil_image = pil_image.resize(_scaled_size(pil_image.size, image_size / min(pil_image.size), round), resample=Image.BICUBIC)
    pixels = np.array(pil_image)
    top, left = ((edge - image_size) // 2 for edge in pixels.shape[:2])
    return pixels[top:top + image_size, left:left + image_size]
