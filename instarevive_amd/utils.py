"""Host-side helpers of the CLI with the reference's names (utils/common.py:7-18,35-51, utils/file.py:20-47,
utils/image/common.py:12-36,229-249). Pure Python / PIL / numpy plumbing around the GPU path."""
import importlib
import math
import os
from typing import Any, List, Mapping, Tuple

import numpy as np
from PIL import Image

# the reference config names `diffusion.model.swinir.SwinIR`; map reference targets onto this package's classes
_TARGET_ALIASES = {"diffusion.model.swinir.SwinIR": "instarevive_amd.models.SwinIR"}


def get_obj_from_str(string: str, reload: bool = False) -> object:
    string = _TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config: Mapping[str, Any]) -> object:
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_state_dict(model, state_dict: Mapping[str, Any], strict: bool = False) -> None:
    state_dict = state_dict.get("state_dict", state_dict)
    is_model_key_starts_with_module = list(model.state_dict().keys())[0].startswith("module.")
    is_state_dict_key_starts_with_module = list(state_dict.keys())[0].startswith("module.")
    if is_model_key_starts_with_module and not is_state_dict_key_starts_with_module:
        state_dict = {f"module.{key}": value for key, value in state_dict.items()}
    if not is_model_key_starts_with_module and is_state_dict_key_starts_with_module:
        state_dict = {key[len("module."):]: value for key, value in state_dict.items()}
    model.load_state_dict(state_dict, strict=strict)


def load_yaml(path: str) -> dict:
    """OmegaConf.load replacement for configs/swinir.yaml-style files (omegaconf is not a dependency here)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def list_image_files(img_dir: str, exts: Tuple[str, ...] = (".jpg", ".png", ".jpeg", ".arw"), follow_links: bool = False,
                     log_progress: bool = False, log_every_n_files: int = 10000, max_size: int = -1) -> List[str]:
    files = []
    for dir_path, _, file_names in os.walk(img_dir, followlinks=follow_links):
        early_stop = False
        for file_name in file_names:
            if os.path.splitext(file_name)[1].lower() in exts:
                if max_size >= 0 and len(files) >= max_size:
                    early_stop = True
                    break
                files.append(os.path.join(dir_path, file_name))
                if log_progress and len(files) % log_every_n_files == 0:
                    print(f"find {len(files)} images in {img_dir}")
        if early_stop:
            break
    return files


def get_file_name_parts(file_path: str) -> Tuple[str, str, str]:
    parent_path, file_name = os.path.split(file_path)
    stem, ext = os.path.splitext(file_name)
    return parent_path, stem, ext


def auto_resize(img: Image.Image, size: int) -> Image.Image:
    short_edge = min(img.size)
    if short_edge < size:
        r = size / short_edge
        return img.resize(tuple(math.ceil(x * r) for x in img.size), Image.BICUBIC)
    return img.copy()


def pad(img: np.ndarray, scale: int) -> np.ndarray:
    h, w = img.shape[:2]
    ph = 0 if h % scale == 0 else math.ceil(h / scale) * scale - h
    pw = 0 if w % scale == 0 else math.ceil(w / scale) * scale - w
    return np.pad(img, pad_width=((0, ph), (0, pw), (0, 0)), mode="constant", constant_values=0)


def center_crop_arr(pil_image: Image.Image, image_size: int) -> np.ndarray:
    while min(*pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize(tuple(x // 2 for x in pil_image.size), resample=Image.BOX)
    scale = image_size / min(*pil_image.size)
    pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
    arr = np.array(pil_image)
    crop_y = (arr.shape[0] - image_size) // 2
    crop_x = (arr.shape[1] - image_size) // 2
    return arr[crop_y: crop_y + image_size, crop_x: crop_x + image_size]
