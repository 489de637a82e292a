"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a compile).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c conv.hip -o /tmp/x.o 2> res.txt ; python tools/kernel_resources.py res.txt [filter ...]"""
import re
import sys


def main():
    t = open(sys.argv[1]).read()
    filt = sys.argv[2:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
    seen = set()
    for b in blocks:
        name = b.split("\n")[0]
        if name in seen or (filt and not any(f in name for f in filt)):
            continue
        seen.add(name)

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{name[:110]:110s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):4d} scratch={scratch:5d} occ={occ:2d} lds={lds}")


if __name__ == "__main__":
    main()
