#!/usr/bin/env python3
"""Build guard: kernels whose name contains <pattern> load weights into registers with un-waited asm loads and
must not use scratch (a spill of such a register would store data that has not arrived yet).
   check_scratch.py <hipcc -Rpass-analysis=kernel-resource-usage stderr> <pattern>"""
import re
import sys

text = open(sys.argv[1]).read()
errors = [l for l in text.splitlines() if " error: " in l]
if errors:
    sys.exit("\n".join(errors))
bad, seen = [], 0
for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", text, re.S):
    if sys.argv[2] in m.group(1):
        seen += 1
        if int(m.group(2)):
            bad.append("%s: %s bytes of scratch per lane" % (m.group(1), m.group(2)))
if bad:
    sys.exit("scratch in no-spill kernels:\n" + "\n".join(bad))
print("check_scratch: %d %s kernels, no scratch" % (seen, sys.argv[2]))
