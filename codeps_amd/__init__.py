"""codeps_amd -- MI355X-native (gfx950) implementation of the CoDEPS dense-vision hot path.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels + the C ABI (-> libcodeps_hip.so, declared in include/codeps_hip.h)
  _lib.py          ctypes binding of that C ABI (loads the in-tree .so; raises if it is missing)
  functional.py    torch.autograd.Function wrappers over the C ABI (device memory / streams = plumbing)
  models/ algos/ misc/ codeps/   host-side mirror of the reference's packages of the same names
                   (same class names, ctor/forward signatures, state_dict keys)
  distributed.py   batch-sharded step: one flat-gradient RCCL all-reduce
There is NO CPU fallback: every op fails loudly when the HIP library or a GPU is absent.
"""
__version__ = "0.1.0"
