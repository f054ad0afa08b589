#!/usr/bin/env python3
"""Where a kernel waits for memory: the global / LDS memory instructions and s_waitcnt of one kernel of a device assembly listing, with the
vector instructions between them counted.  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S csrc/x.hip -o x.s;
tools/asm_waits.py x.s <mangled-name substring> [first line] [last line]"""
import sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0]][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
body = lines[start:end]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
print(lines[start][:100], len(body), "lines")
valu = salu = 0
for i, l in enumerate(body):
    t = l.strip()
    if not t or t.startswith(";"): continue
    op = t.split()[0]
    if op.startswith("v_"): valu += 1
    elif op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_cbranch") and not op.startswith("s_barrier"): salu += 1
    if lo <= i < hi and (op.startswith("s_waitcnt") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_") or op.startswith("ds_") or op.startswith("s_barrier") or op.startswith("s_load") or "Loop Header" in t or op.startswith("s_cbranch")):
        print("%5d [v %3d s %3d] %s" % (i, valu, salu, t[:100])); valu = salu = 0
