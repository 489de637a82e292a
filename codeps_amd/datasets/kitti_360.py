"""Kitti360: drop-in for datasets/kitti_360.py:16-397 (modes train / val from the frame lists, "sequence" over raw drives; camera
from calibration/perspective.txt; optional Velodyne depth) with the per-sample work on the device
(codeps_amd.datasets.preprocessing).  Device tensors where the reference returns numpy arrays."""
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from ..misc.camera_model import CameraModel
from . import preprocessing as P
from .cityscapes import CITYSCAPES_THINGS, CODEPS_THINGS


class Kitti360(torch.utils.data.Dataset):
    SEQUENCES = ["00", "02", "03", "04", "05", "06", "07", "09", "10"]

    def __init__(self, mode: str, cfg, return_depth: bool = False, return_only_rgb: bool = False, sequences: Optional[List[str]] = None,
                 sequence_reference_mode: str = "rgb", label_mode: str = "codeps", device=None):
        super().__init__()
        assert mode in ["train", "val", "sequence"], f"Unsupported mode: {mode}"
        assert label_mode in ["codeps", "cityscapes"], f"Unsupported label mode: {label_mode}"
        assert cfg.name == "kitti_360"
        if mode == "sequence":
            assert sequence_reference_mode in ["semantic", "rgb"], f"Unsupported sequence reference mode: {sequence_reference_mode}"
            assert sequences is not None and len(sequences) > 0, "In 'sequence' mode, sequences have to be given."
            for seq in sequences:
                assert seq in self.SEQUENCES, f"Passed invalid sequence: {seq}"
        self.mode, self.return_depth, self.return_only_rgb, self.label_mode = mode, return_depth, return_only_rgb, label_mode
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.path_base = Path(cfg.path)
        self.image_size = list(cfg.feed_img_size)
        self.offsets = [-x for x in reversed(cfg.offsets)] + list(cfg.offsets)
        self.sigma = cfg.center_heatmap_sigma
        self.small_instance_weight = cfg.small_instance_weight
        self.small_instance_area_full_res = cfg.small_instance_area_full_res
        self.augmentation_cfg, self.normalization_cfg = cfg.augmentation, cfg.normalization
        self.remove_classes = list(cfg.remove_classes)
        if mode == "val":
            self.augmentation_cfg = type(cfg.augmentation)({"active": False})
        self.sequence_reference_mode = sequence_reference_mode
        self.semantic_seq_mode_image_skipped: List[bool] = []
        self.sequences = sequences if mode == "sequence" else None
        self.frame_paths = self._get_frames()
        if self.return_only_rgb:
            assert self.mode != "sequence", "Not implemented"
            self.frame_paths = self._get_frames_only_rgb()
        self.full_size = None
        self.camera_model = self._read_calibration()
        self.lidar_to_rect, self.camera_matrix = (self._read_lidar_to_rect() if return_depth else (None, None))

    @property
    def thing_classes(self) -> List[int]:
        return CODEPS_THINGS if self.label_mode == "codeps" else CITYSCAPES_THINGS

    def _get_frames(self) -> List[Dict[str, Optional[Path]]]:
        """datasets/kitti_360.py:51-120."""
        frames = []
        if self.sequences is None:
            filename = self.path_base / "data_2d_semantics" / "train" / f"2013_05_28_drive_{self.mode}_frames.txt"
            for line in filename.read_text(encoding="utf-8").splitlines():
                if self.mode == "val" and "0000004391.png" in line:
                    continue                                   # no preceding file for the image triplet
                rgb, semantic = self.path_base / line.split(" ")[0], self.path_base / line.split(" ")[1]
                instance = semantic.parents[1] / "instance" / semantic.name
                depth = (self.path_base / "data_3d_raw" / semantic.parents[2].name / "velodyne_points" / "data" / f"{semantic.stem}.bin") \
                    if self.return_depth else None
                frame = {"rgb": rgb, "semantic": semantic, "instance": instance, "depth": depth}
                for path in frame.values():
                    if path is not None:
                        assert path.exists(), f"File does not exist: {path}"
                frames.append(frame)
            return frames
        rgb_files: List[Path] = []
        for sequence in self.sequences:
            files = sorted((self.path_base / "data_2d_raw" / f"2013_05_28_drive_00{sequence}_sync" / "image_00" / "data_rect").glob("*.png"))
            files = files[max(self.offsets):-max(self.offsets)]
            if sequence == "09":
                files = files[4999:8499]                       # the reference's manually selected subset (kitti_360.py:93-97)
            rgb_files += files
        for rgb in rgb_files:
            sequence = rgb.parents[2].name
            base = self.path_base / "data_2d_semantics" / "train" / sequence / "image_00"
            semantic, instance = base / "semantic" / rgb.name, base / "instance" / rgb.name
            if self.sequence_reference_mode == "semantic":
                if not semantic.exists() or not instance.exists():
                    self.semantic_seq_mode_image_skipped.append(True)
                    continue
                self.semantic_seq_mode_image_skipped.append(False)
            else:
                semantic = semantic if semantic.exists() else None
                instance = instance if instance.exists() else None
            depth = (self.path_base / "data_3d_raw" / sequence / "velodyne_points" / "data" / f"{rgb.stem}.bin") if self.return_depth else None
            frames.append({"rgb": rgb, "semantic": semantic, "instance": instance, "depth": depth})
        return frames

    def _get_frames_only_rgb(self) -> List[Dict[str, Path]]:
        frames, max_offset = [], max(self.offsets)
        for sequence in sorted((self.path_base / "data_2d_raw").glob("*")):
            files = sorted(sequence.glob("image_00/data_rect/*.png"))
            frames += [{"rgb": f} for f in files[max_offset:-max_offset]]
        return frames

    def _perspective(self, key: str, shape):
        for line in (self.path_base / "calibration" / "perspective.txt").read_text(encoding="utf-8").splitlines():
            if line.split(" ")[0] == key:
                return np.array(line.replace(key + " ", "").split(), dtype=float).reshape(shape)
        return None

    def _read_calibration(self) -> CameraModel:
        """datasets/kitti_360.py:140-160."""
        from PIL import Image
        p = self._perspective("P_rect_00:", (3, 4))
        p = np.zeros((3, 4)) if p is None else p
        w, h = Image.open(self.frame_paths[0]["rgb"]).size
        self.full_size = (h, w)
        height, width = self.image_size
        return CameraModel(w, h, p[0, 0], p[1, 1], p[0, 2], p[1, 2]).get_scaled_model_image_size(width, height)

    def _read_lidar_to_rect(self):
        """datasets/kitti_360.py:162-181."""
        cam_to_velo = np.concatenate((np.loadtxt(self.path_base / "calibration" / "calib_cam_to_velo.txt").reshape(3, 4),
                                      np.array([0, 0, 0, 1]).reshape(1, 4)))
        rect, k = np.eye(4), np.eye(3, 4)
        r = self._perspective("R_rect_00:", (3, 3))
        if r is not None:
            rect[:3, :3] = r
        p = self._perspective("P_rect_00:", (3, 4))
        k = k if p is None else p
        return rect @ np.linalg.inv(cam_to_velo), k

    def __len__(self) -> int:
        return len(self.frame_paths)

    def _mapping(self):
        m = P.KITTI360_TO_CODEPS if self.label_mode == "codeps" else P.KITTI360_TO_CITYSCAPES
        return [(a, b) for a, b in m if b not in self.remove_classes]

    def __getitem__(self, index: int, do_network_preparation: bool = True, do_augmentation: bool = True,
                    return_only_rgb: bool = False) -> Dict[str, Any]:
        """datasets/kitti_360.py:183-338."""
        fp = self.frame_paths[index]
        image_path = fp["rgb"]
        frames = {0: P.decode_png(image_path, self.device)}
        full_h, full_w = frames[0].shape[:2]
        for offset in self.offsets:
            path = image_path.parent / f"{str(int(image_path.stem) + offset).zfill(len(image_path.stem))}.png"
            assert path.exists(), f"Offset file does not exist: {path}"
            frames[offset] = P.decode_png(path, self.device)
        keys = list(frames)
        resized = P.resize_lanczos(torch.stack([frames[k] for k in keys]), self.image_size)
        output: Dict[str, Any] = {"rgb": {k: resized[i] for i, k in enumerate(keys)}, "camera_model": self.camera_model.to_tensor()}
        if not (self.return_only_rgb or return_only_rgb):
            if fp.get("semantic") is not None and fp.get("instance") is not None:
                sem = P.decode_png(fp["semantic"], self.device, None)
                inst = P.decode_png(fp["instance"], self.device, None)
                output.update(P.kitti360_labels(sem, inst, self.image_size, (full_h, full_w), self._mapping(), self.thing_classes,
                                                self.small_instance_area_full_res, self.small_instance_weight, self.sigma))
            if self.return_depth:
                pts = torch.from_numpy(np.fromfile(fp["depth"], dtype=np.float32).reshape(-1, 4)).to(self.device, non_blocking=True)
                output["depth"] = P.lidar_depth(pts, self.lidar_to_rect, self.camera_matrix, (full_h, full_w), self.image_size)
        if do_augmentation:
            P.augment_data(output, self.augmentation_cfg)
        if do_network_preparation:
            P.prepare_for_network(output, self.normalization_cfg)
        return output
