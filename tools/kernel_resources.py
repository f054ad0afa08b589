#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch / code-size table of one HIP translation unit, compiled for gfx950 on this machine
(no GPU needed).  Usage: tools/kernel_resources.py hyrise_amd/csrc/aggregate.hip [name filter] [-- extra hipcc flags]

What the numbers mean for occupancy on gfx950 (512 VGPRs per SIMD lane, allocated in blocks of 8): waves per SIMD =
floor(512 / roundup8(vgpr)), capped at 8; 160 KiB of LDS per CU bound the workgroups per CU; any scratch bytes mean
spills.  The assembly is left in the temporary directory that is printed (grep it for the hot loops)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        extra = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    if not args:
        raise SystemExit(__doc__)
    source = os.path.abspath(args[0])
    name_filter = args[1] if len(args) > 1 else ""
    out = tempfile.mkdtemp(prefix="kernel_resources_")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = os.path.splitext(os.path.basename(source))[0]
    # the sources include "../../include/hyrise_amd.h" relative to csrc/: compile in place, outputs elsewhere
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-save-temps=obj", source, "-o", os.path.join(out, base + ".o")] + extra
    subprocess.check_call(cmd, cwd=os.path.dirname(source), stderr=subprocess.DEVNULL)
    assembly = os.path.join(out, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.s")
    text = open(assembly).read()
    sizes = {}
    binary = os.path.join(out, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.out")
    if os.path.exists(binary):
        listing = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", binary], stdout=subprocess.PIPE, text=True).stdout
        for line in listing.splitlines():
            fields = line.split()
            if len(fields) >= 8 and fields[3] == "FUNC":
                sizes[fields[7]] = int(fields[2])
    demangle = lambda n: subprocess.run(["c++filt", n], stdout=subprocess.PIPE, text=True).stdout.strip()
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>6s} {'scratch':>8s} {'lds':>7s} {'code':>8s} {'waves/SIMD':>10s}")
    for block in re.findall(r"  - \.agpr_count:.*?\.wavefront_size: +\d+", text, flags=re.S):
        get = lambda key: int(re.search(rf"\.{key}: +(\d+)", block).group(1)) if re.search(rf"\.{key}: +(\d+)", block) else 0
        mangled = re.search(r"\.name: +(\S+)", block).group(1)
        name = demangle(mangled)
        if name_filter and name_filter not in name:
            continue
        vgpr = get("vgpr_count")
        waves = min(8, 512 // max(8, (vgpr + 7) // 8 * 8))
        print(f"{name[:70]:70s} {vgpr:5d} {get('agpr_count'):5d} {get('sgpr_count'):5d} {get('vgpr_spill_count'):6d} {get('private_segment_fixed_size'):8d} "
              f"{get('group_segment_fixed_size'):7d} {sizes.get(mangled, 0):8d} {waves:10d}")
    print("assembly:", assembly)


if __name__ == "__main__":
    main()
