"""CPU oracle for the CoDEPS dense-vision hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a from-scratch CPU restatement (plain ``torch`` CPU ops + numpy) of the reference
algorithms on the hot path named in BASELINE.json / SURVEY.md §8a.  Every function cites the
reference ``file:line`` it follows (paths relative to the reference repo root).

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
    anything from here, and only as the *checker* / reported CPU baseline;
  * nothing under ``codeps_amd/`` (the product) imports it -- the product path raises when the HIP
    extension is missing, it never falls back to this code;
  * nothing here reads /root/reference at run time.

Parity pinning:
  * heads, camera/warp geometry, all losses, panoptic fusion, mix-up (oracle/mixup.py), evaluation metrics
    (oracle/metrics.py): PINNED against golden vectors produced
    by importing and running the real reference in the development container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
  * ResNet trunk arithmetic: the reference delegates it to torchvision==0.13.1 (requirements.txt:2),
    which is neither vendored under the reference nor installed here -> ``oracle/resnet.py`` restates
    the published torchvision v0.13 ResNet (v1.5 bottleneck) definition and is **parity unpinned**
    w.r.t. a reference run; it is pinned only by self-consistency checks (state-dict key set/shapes,
    published conv-MAC / parameter counts, per-block composition from torch.nn.functional).
  * replay-buffer diversity sampler (oracle/replay.py): the search runs inside faiss-gpu==1.7.2, absent here ->
    **parity unpinned**; restates the reference's call sites on top of faiss's published IndexFlatIP / normalize_L2
    behaviour (self-consistency test only).
"""
