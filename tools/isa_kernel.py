#!/usr/bin/env python3
"""Cut one kernel out of a device assembly listing: tools/isa_kernel.py <listing.s> <mangled-name-substring> > out.s
(listing: hipcc ... -S --cuda-device-only -o listing.s csrc/rtx_device.hip)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sys.argv[2] in l]
if not starts:
    sys.exit("no kernel matches")
start = starts[0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
valu = sum(1 for l in body if re.match(r"\s+v_", l))
salu = sum(1 for l in body if re.match(r"\s+s_", l))
print("; %s: %d lines, %d VALU, %d SALU" % (lines[start].split(":")[0], len(body), valu, salu))
print("\n".join(body))
