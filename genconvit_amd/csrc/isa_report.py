#!/usr/bin/env python3
"""Print per-kernel register / LDS / scratch usage from a hipcc -save-temps .s file."""
import re, sys, subprocess
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
for path in sys.argv[1:]:
    s = open(path).read()
    for b in re.split(r'^\s+- \.agpr_count', s, flags=re.M)[1:]:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, b).group(1)
        nm = demangle(g('name'))
        nm = re.sub(r'\(.*', '', nm)[:110]
        print(f"{nm:110s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
