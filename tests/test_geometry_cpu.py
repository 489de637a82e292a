"""CameraModel host class (misc/camera_model.py:9-71) against the reference's outputs in tests/golden/geometry.npz."""
import numpy as np
import torch

import inputs as gi
from helpers import load


def test_camera_model_matches_the_reference_bit_for_bit():
    from codeps_amd.misc import CameraModel
    z, meta = load("geometry")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    intr = gi.intrinsics(b, h, w, seed)
    cm = [CameraModel.from_tensor(w, h, intr[i]) for i in range(b)]
    sc = cm[0].get_scaled_model_image_size(w // 2, h // 2)
    assert np.array_equal(np.array(list(sc.intrinsics.values()), np.float64), z["scaled_intr"])
    assert sc.image_size == {"width": w // 2, "height": h // 2}
    u = torch.arange(w).float().expand(h, w).contiguous()
    v = torch.arange(h).float().view(h, 1).expand(h, w).contiguous()
    rx, ry, rz = cm[1].get_viewing_ray(u.clone(), v.clone())
    assert np.array_equal(torch.stack([rx, ry, rz]).numpy(), z["ray"])
    pu, pv = cm[1].get_image_point(rx * 5, ry * 5, rz * 5)
    assert np.array_equal(torch.stack([pu, pv]).numpy(), z["reproj"])
    # round trips of the tensor form
    assert torch.equal(cm[1].to_tensor(), intr[1])
    assert torch.equal(CameraModel.batch_tensor(cm), intr)
    m = cm[0].get_scaled_model(0.5, 0.25)
    assert m.image_size == {"width": w * 0.5, "height": h * 0.25}
    i0 = cm[0].intrinsics
    assert [m.intrinsics[k] for k in ("fx", "fy", "cx", "cy")] == [i0["fx"] * 0.5, i0["fy"] * 0.25, i0["cx"] * 0.5, i0["cy"] * 0.25]
