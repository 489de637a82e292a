"""Host-side mirrors of the reference's `datasets` package that sit on the adapt path."""
from .mixup import Mixup  # noqa: F401
from .replay_buffer import DiversityReplayBuffer  # noqa: F401
from . import preprocessing  # noqa: F401
from .cityscapes import Cityscapes  # noqa: F401
from .kitti_360 import Kitti360  # noqa: F401
