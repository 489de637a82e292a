"""Host-side mirror of the reference's `codeps` package: CodepsNet (codeps/online_adap.py) and
gen_models (codeps/model_setup.py)."""
from .frame import OnlineAdapter, pre_step
from .model_setup import gen_models
from .online_adap import CodepsNet

__all__ = ["CodepsNet", "OnlineAdapter", "gen_models", "pre_step"]
