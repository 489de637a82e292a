"""Synthetic inputs of the benchmark / smoke runs: device-side generation, seeded per rank."""
import torch


def _smooth(b, c, h, w, gen, device):
    coarse = torch.rand(b, c, max(h // 8, 2), max(w // 8, 2), generator=gen, device=device)
    img = torch.nn.functional.interpolate(coarse, (h, w), mode="bilinear", align_corners=False)
    return (img + 0.02 * torch.randn(b, c, h, w, generator=gen, device=device)).clamp_(0, 1)


def triplet_item(b, h, w, gen, device, labels=False, num_classes=14):
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
    f0 = (_smooth(b, 3, h, w, gen, device) - mean) / std
    fm = torch.roll(f0, -2, 3) + 0.01 * torch.randn(f0.shape, generator=gen, device=device)
    fp = torch.roll(f0, 2, 3) + 0.01 * torch.randn(f0.shape, generator=gen, device=device)
    intr = torch.tensor([2262.52 * w / 2048, 2265.30 * h / 1024, 1096.98 * w / 2048, 513.137 * h / 1024], device=device)
    d = {"rgb": {0: f0.contiguous(), -1: fm.contiguous(), 1: fp.contiguous()}, "camera_model": intr.repeat(b, 1)}
    if labels:
        blk = 32
        lab = torch.randint(0, num_classes, (b, -(-h // blk), -(-w // blk)), generator=gen, device=device)
        lab = lab.repeat_interleave(blk, 1).repeat_interleave(blk, 2)[:, :h, :w]
        void = torch.rand(lab.shape, generator=gen, device=device) < 0.05
        d["semantic"] = torch.where(void, torch.full_like(lab, 255), lab).to(torch.uint8).contiguous()
        d["semantic_weights"] = (1 + 2 * torch.randint(0, 2, (b, h, w), generator=gen, device=device)).to(torch.uint8)
        yy, xx = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                                torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
        ctr = torch.zeros(b, 1, h, w, device=device)
        for i in range(b):
            for _ in range(12):
                cy, cx = torch.rand(2, generator=gen, device=device) * torch.tensor([h, w], device=device)
                ctr[i, 0] = torch.maximum(ctr[i, 0], torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 128.0))
        d["center"] = ctr
        d["offset"] = (torch.rand(b, 2, h, w, generator=gen, device=device) * 128 - 64)
        d["center_weights"] = torch.ones(b, 1, h, w, dtype=torch.uint8, device=device)
        d["offset_weights"] = torch.ones(b, 1, h, w, dtype=torch.uint8, device=device)
        d["thing_mask"] = (lab >= 9).to(torch.uint8).unsqueeze(1).contiguous()
    return d


def adapt_batch(h, w, device, seed, n_target=1, n_replay=3, n_source=2, n_mixup=2, target_hw=None):
    """Per-GPU adaptation batch of BASELINE config 3: 1 target + 3 target_replay + 2 source + 2 cut_mixup
    (pre-mixed) centre images at (h, w); target_hw = a different size for the target-domain keys (the YAML's
    384x1408 KITTI-360 crops)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    th, tw = target_hw if target_hw is not None else (h, w)
    data = {}
    for key, n, (hh, ww), lab in (("target", n_target, (th, tw), False), ("target_replay", n_replay, (th, tw), False),
                                  ("source", n_source, (h, w), True)):
        if n > 0:                  # a rank of a batch-sharded step may hold no item of a key (BASELINE configs[3])
            data[key] = triplet_item(n, hh, ww, gen, device, labels=lab)
    if n_mixup > 0:
        mix = triplet_item(n_mixup, h, w, gen, device, labels=True)
        data["cut_mixup"] = {"rgb": {0: mix["rgb"][0]}, "semantic": mix["semantic"]}
    return data, n_target + n_replay + n_source + n_mixup


def mixup_raw_item(n, h, w, th, tw, device, seed, num_classes=14):
    """The raw `mixup_data` of one `*mixup` key as the replay buffer hands it over (datasets/replay_buffer.py:281-305):
    a labelled source image at (h, w) and a target image at (th, tw); the teacher's predictions are added by
    codeps.pre_step, the mix itself by datasets.Mixup.do_mixup."""
    gen = torch.Generator(device=device).manual_seed(seed)
    src = triplet_item(n, h, w, gen, device, labels=True, num_classes=num_classes)
    tgt = triplet_item(n, th, tw, gen, device)
    sem = src["semantic"]
    blk = 32
    k = torch.randint(1, 6, (n, -(-h // blk), -(-w // blk)), generator=gen, device=device)
    k = k.repeat_interleave(blk, 1).repeat_interleave(blk, 2)[:, :h, :w]
    inst = torch.where((sem >= 9) & (sem < 255), sem.to(torch.int32) * 1000 + k.to(torch.int32), torch.zeros_like(k, dtype=torch.int32))
    return {"rgb_src": {0: src["rgb"][0]}, "rgb_tgt": {0: tgt["rgb"][0]}, "semantic_src": sem.unsqueeze(1).contiguous(),
            "instance_src": inst.unsqueeze(1).contiguous(), "camera_model_src": src["camera_model"],
            "camera_model_tgt": tgt["camera_model"], "tgt_is_replay": torch.zeros(n, dtype=torch.long, device=device)}
