"""The diagnostics of csrc/diag.hip (measured rooflines; bench.py and tools/peaks.py / tools/feed_rate.py report
them): every mode launches, returns a positive time, rejects bad arguments, and the numbers are physically sensible."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from codeps_amd import _lib
    return _lib


def test_mfma_rate_and_triad_are_sensible():
    L = _lib()
    lib = L.diag_lib()
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    scratch = torch.empty(cus * 8 * 256, device=dev)
    flops, ms = ctypes.c_double(), ctypes.c_float()
    for mode in (0, 1, 2):
        for rnd in (0, 1):
            L.check_diag(lib.codeps_diag_mfma_rate(mode, rnd, cus * 8, 2000, 2, L.ptr(scratch), ctypes.byref(flops), ctypes.byref(ms), L.stream_ptr()),
                    "codeps_diag_mfma_rate")
            tf = flops.value / (ms.value * 1e-3) / 1e12
            assert ms.value > 0 and 20 < tf < 3000, (mode, rnd, tf)                 # between a tenth of the f32 peak and the bf16 vendor peak
    n = 1 << 26
    a, b, c = (torch.ones(n, device=dev) for _ in range(3))
    L.check_diag(lib.codeps_diag_stream_triad(L.ptr(a), L.ptr(b), L.ptr(c), n, 5, ctypes.byref(ms), L.stream_ptr()), "codeps_diag_stream_triad")
    gbs = 12.0 * n / (ms.value * 1e-3) / 1e9
    assert 500 < gbs < 9000, gbs
    assert torch.all(a == 2.5)                                                      # a = b + 1.5 c
    assert lib.codeps_diag_mfma_rate(7, 0, cus, 10, 1, L.ptr(scratch), ctypes.byref(flops), ctypes.byref(ms), L.stream_ptr()) != 0


def test_feed_rate_modes_run_and_dma_is_not_slower_than_vgpr_staging():
    L = _lib()
    lib = L.diag_lib()
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    window = torch.randn((2 << 20) // 4, device=dev)
    scratch = torch.empty(cus * 256, device=dev)
    ms = ctypes.c_float()
    t = {}
    for mode, kb in ((0, 0), (1, 32), (2, 32), (5, 32), (6, 32)):
        L.check_diag(lib.codeps_diag_feed_rate(mode, kb, cus, 500, 3, L.ptr(window), window.numel() * 4, L.ptr(scratch), ctypes.byref(ms), L.stream_ptr()),
                "codeps_diag_feed_rate")
        assert ms.value > 0
        t[mode] = ms.value
    assert t[2] <= t[1] * 1.05 and t[6] <= t[5] * 1.05, t                          # LDS DMA beats the VGPR round trip (measured: -15 %)
    # bad arguments are refused, not launched
    assert lib.codeps_diag_feed_rate(3, 32, cus, 10, 1, L.ptr(window), window.numel() * 4, L.ptr(scratch), ctypes.byref(ms), L.stream_ptr()) != 0
    assert lib.codeps_diag_feed_rate(6, 4, cus, 10, 1, L.ptr(window), window.numel() * 4, L.ptr(scratch), ctypes.byref(ms), L.stream_ptr()) != 0
    assert lib.codeps_diag_feed_rate(2, 30, cus, 10, 1, L.ptr(window), window.numel() * 4, L.ptr(scratch), ctypes.byref(ms), L.stream_ptr()) != 0
