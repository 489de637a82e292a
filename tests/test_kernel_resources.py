"""Register / scratch budget of the HIP kernels, from the resource remarks the build leaves in codeps_amd/csrc/build/*.hip.res
(-Rpass-analysis=kernel-resource-usage; no GPU needed).  A kernel of the conv engine that starts spilling, or an instantiation that
falls to a lower occupancy than the one it was tuned at, costs whole percents of the step and shows up in no parity test."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "codeps_amd", "csrc", "build")

# kernels allowed to carry scratch: (name fragment, bytes per lane)
SCRATCH_OK = {"photo_fwd_kernel": 64,          # 36 bytes at 3 waves per SIMD, measured faster than 2 waves without (photometric.hip)
              "wgrad_halo64s_kernel": 96,      # 80 bytes since round 4: the generic single-buffered body's spills (the double-buffered body the step runs has none)
              "dwconv5_wgrad_tile_kernel": 8,  # one loop-invariant staging offset at the 256-register budget of two blocks per CU (25 tap sums x 4 channels = 100 registers)
              "match_lut_kernel": 8192}        # one thread per channel builds a 256-entry CDF + LUT in private memory (preprocess.hip, not on the step's path)


def _kernels():
    out = {}
    for f in glob.glob(os.path.join(BUILD, "*.hip.res")):
        t = open(f).read()
        for blk in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
            name = blk.split(" ")[0].split("\n")[0]

            def g(k, blk=blk):
                m = re.search(k + r": (\d+)", blk)
                return int(m.group(1)) if m else -1
            out[name] = dict(vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]"),
                             file=os.path.basename(f))
    return out


def _demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60)
        return dict(zip(names, r.stdout.splitlines()))
    except (OSError, subprocess.SubprocessError):
        return {n: n for n in names}


def test_no_kernel_spills_and_tuned_occupancies_hold():
    ks = _kernels()
    if not ks:
        pytest.skip("no build/*.hip.res (the library was built without the Makefile of this tree)")
    assert len(ks) > 150, f"only {len(ks)} kernels found in the resource remarks"
    nice = _demangle(list(ks))
    bad = []
    for n, r in ks.items():
        allowed = max((v for k, v in SCRATCH_OK.items() if k in n), default=0)
        if r["scratch"] > allowed:
            bad.append(f"{nice[n][:120]}: {r['scratch']} bytes of scratch per lane ({r['file']})")
    assert not bad, "kernels that spill:\n" + "\n".join(bad)
    # the occupancies the hot kernels were tuned at (waves per SIMD by registers; LDS decides the rest): plain instantiations only
    want = {"conv_halo3r_kernel<0, 2, false, 3, 4>": 1, "conv_halo3r_kernel<0, 1, false, 16, 8>": 2,
            "conv_halo3_kernel<0, 1, 2, 2, false, 3>": 3, "conv_halo3_kernel<0, 1, 2, 2, false, 16>": 3,
            "conv_igemm_kernel<0, 2, 2, 2, 2, 32, 1, 3, false>": 3,
            "conv_igemm_kernel<0, 2, 2, 2, 1, 32, 1, 3, false>": 4, "conv_igemm_kernel<1, 2, 2, 2, 1, 32, 1, 3, false>": 4,
            "conv_igemm_kernel<0, 2, 2, 2, 1, 32, 1, 2, false>": 4, "conv_igemm_kernel<1, 2, 2, 2, 1, 32, 1, 2, false>": 4,
            "chain_expand_reduce_kernel": 1, "chain_expand_reduce_h_kernel<256>": 2, "chain_expand_reduce_h_kernel<128>": 4, "chain_expand_reduce_h_kernel<64>": 4}
    for frag, occ in want.items():
        hits = [r for n, r in ks.items() if frag in nice[n].replace("(anonymous namespace)::", "")]
        assert hits, f"kernel {frag} not found in the resource remarks"
        assert all(h["occ"] >= occ for h in hits), f"{frag}: occupancy {[h['occ'] for h in hits]} < {occ}"
