"""Per-layer timing of the 1 x 1 weight gradient (codeps_conv2d_wgrad) -- run once per CODEPS_WGRAD_1X1S setting."""
import ctypes
import sys
sys.path.insert(0, ".")
import torch
from codeps_amd import functional as F
from codeps_amd._lib import lib

dev = "cuda:0"
lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
cases = [(4, 128, 256, 64, 256, 1), (4, 128, 256, 256, 64, 1), (4, 64, 128, 128, 512, 1), (4, 64, 128, 512, 128, 1), (4, 32, 64, 256, 1024, 1),
         (4, 32, 64, 1024, 256, 1), (4, 16, 32, 512, 2048, 1), (4, 16, 32, 2048, 512, 1), (4, 64, 128, 256, 512, 2), (4, 32, 64, 512, 1024, 2)]
for b, h, w, ci, co, s in cases:
    x = torch.randn(b, h, w, ci, device=dev)
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    dz = torch.randn(b, ho, wo, co, device=dev)
    dw = torch.empty(co, ci, 1, 1, device=dev)
    desc = F._conv_desc(x, None, co, 1, 1, (s, s), (0, 0), 0, 0, False)
    wsb = lib().codeps_conv2d_wgrad_workspace_bytes(ctypes.byref(desc))
    ws = torch.empty(wsb // 4, device=dev)

    def run():
        F.check(lib().codeps_conv2d_wgrad(ctypes.byref(desc), F.ptr(x), None, F.ptr(dz), F.ptr(dw), co, ci, 0, F.ptr(ws), ws.numel() * 4, F.stream_ptr()), "wgrad")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    fl = 2.0 * b * ho * wo * ci * co
    by = 4.0 * b * ho * wo * (ci + co)
    print(f"{lib().codeps_conv2d_last_kernel().decode():20s} B{b} {h}x{w} {ci}->{co} s{s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s  {by / us / 1e3:6.0f} GB/s (incl. reduce)")
