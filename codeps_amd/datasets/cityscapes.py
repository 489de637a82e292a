"""Cityscapes: drop-in for datasets/cityscapes.py:17-323 (file layout, frame listing, `__getitem__` contract) with the per-sample
work on the device (codeps_amd.datasets.preprocessing): the PNGs are inflated on the host and copied once; resize, label
conversion, instance bookkeeping, centre / offset maps, depth, augmentation and normalisation are kernels.  Returns DEVICE
tensors where the reference returns numpy arrays / tensors (same keys, shapes and dtypes otherwise)."""
import json
from pathlib import Path
from typing import Any, Dict, List

import torch

from ..misc.camera_model import CameraModel
from . import preprocessing as P

CODEPS_THINGS, CITYSCAPES_THINGS = [9, 10, 11, 12, 13], [11, 12, 13, 14, 15, 16, 17, 18]


class Cityscapes(torch.utils.data.Dataset):

    def __init__(self, mode: str, cfg, return_depth: bool = False, return_only_rgb: bool = False, label_mode: str = "codeps",
                 device=None):
        super().__init__()
        assert mode in ["train", "val"], f"Unsupported mode: {mode}"
        assert label_mode in ["codeps", "cityscapes"], f"Unsupported label mode: {label_mode}"
        assert cfg.name == "cityscapes"
        self.mode, self.return_depth, self.return_only_rgb, self.label_mode = mode, return_depth, return_only_rgb, label_mode
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.path_base = Path(cfg.path)
        self.image_size = list(cfg.feed_img_size)                       # [H, W]
        self.offsets = [-x for x in reversed(cfg.offsets)] + list(cfg.offsets)
        self.sigma = cfg.center_heatmap_sigma
        self.small_instance_weight = cfg.small_instance_weight
        self.small_instance_area_full_res = cfg.small_instance_area_full_res
        self.augmentation_cfg, self.normalization_cfg = cfg.augmentation, cfg.normalization
        self.remove_classes = list(cfg.remove_classes)
        if mode == "val":
            self.augmentation_cfg = type(cfg.augmentation)({"active": False})
        self.frame_paths = self._get_frames()

    @property
    def thing_classes(self) -> List[int]:
        return CODEPS_THINGS if self.label_mode == "codeps" else CITYSCAPES_THINGS

    def _get_frames(self) -> List[Dict[str, Path]]:
        """datasets/cityscapes.py:31-67."""
        frames = []
        for semantic in sorted((self.path_base / "gtFine" / self.mode).glob("*/*_gtFine_labelIds.png")):
            file_base = semantic.stem.replace("_gtFine_labelIds", "")
            city = file_base.split("_")[0]
            frame = {"rgb": self.path_base / "leftImg8bit_sequence" / self.mode / city / f"{file_base}_leftImg8bit.png",
                     "semantic": semantic, "instance": semantic.parent / semantic.name.replace("label", "instance"),
                     "camera": self.path_base / "camera" / self.mode / city / f"{file_base}_camera.json",
                     "disparity": (self.path_base / "disparity_sequence" / self.mode / city / f"{file_base}_disparity.png")
                     if self.return_depth else None}
            for path in frame.values():
                if path is not None:
                    assert path.exists(), f"File does not exist: {path}"
            frames.append(frame)
        return frames

    def __len__(self) -> int:
        return len(self.frame_paths)

    def _mapping(self):
        m = P.CITYSCAPES_TO_CODEPS if self.label_mode == "codeps" else P.CITYSCAPES_TO_CITYSCAPES
        return [(a, b) for a, b in m if b not in self.remove_classes]          # Dataset._rm_classes_mapping

    def __getitem__(self, index: int, do_network_preparation: bool = True, do_augmentation: bool = True,
                    return_only_rgb: bool = False) -> Dict[str, Any]:
        """datasets/cityscapes.py:113-270."""
        fp = self.frame_paths[index]
        image_path = fp["rgb"]
        center = P.decode_png(image_path, self.device)
        full_h, full_w = center.shape[:2]
        center_number = image_path.stem.split("_")[2]
        frames = {0: center}
        for offset in self.offsets:
            name = image_path.name.replace(center_number, str(int(center_number) + offset).zfill(len(center_number)))
            path = image_path.parent / name
            assert path.exists(), f"Offset file does not exist: {path}"
            frames[offset] = P.decode_png(path, self.device)
        keys = list(frames)
        resized = P.resize_lanczos(torch.stack([frames[k] for k in keys]), self.image_size)      # the triplet in one launch pair
        images = {k: resized[i] for i, k in enumerate(keys)}
        with open(fp["camera"], "r", encoding="utf-8") as f:
            camera_data = json.load(f)
        intr = camera_data["intrinsic"]
        cam = CameraModel(full_w, full_h, intr["fx"], intr["fy"], intr["u0"], intr["v0"])
        height, width = self.image_size
        output: Dict[str, Any] = {"rgb": images, "camera_model": cam.get_scaled_model_image_size(width, height).to_tensor()}
        if not (self.return_only_rgb or return_only_rgb):
            sem = P.decode_png(fp["semantic"], self.device, None)
            inst = P.decode_png(fp["instance"], self.device, None)
            output.update(P.cityscapes_labels(sem, inst, self.image_size, (full_h, full_w), self._mapping(), self.thing_classes,
                                              self.small_instance_area_full_res, self.small_instance_weight, self.sigma))
        if self.return_depth:
            disp = P.decode_png(fp["disparity"], self.device, None)
            output["depth"] = P.disparity_to_depth(disp, intr["fx"], camera_data["extrinsic"]["baseline"], self.image_size)
        if do_augmentation:
            P.augment_data(output, self.augmentation_cfg)
        if do_network_preparation:
            P.prepare_for_network(output, self.normalization_cfg)
        return output
