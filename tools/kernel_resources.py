#!/usr/bin/env python3
"""Register / occupancy table of the device code (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
usage: kernel_resources.py [substring ...]   (default: the kernels the default paths launch)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "fractalrenderer_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                      "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-c", os.path.join(csrc, "fr_device.hip"),
                      "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("EXTRA_HIPFLAGS", "").split(),
                     capture_output=True, text=True)
if out.returncode:
    sys.exit(out.stderr[-3000:])
usage, cur = {}, None
for line in out.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = usage.setdefault(m.group(1), {})
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
want = sys.argv[1:] or ["tile_kernelIdLi0ELi3ELb0ELb0ELb0", "tile_kernelIfLi1ELi3ELb0ELb0ELb0", "pool_kernelIdLi0ELb0",
                        "pool_kernelIdLi0ELb1", "pool_kernelIfLi1ELb0", "tile_lean_kernelIdLi0ELb0ELi2", "tile_lean_kernelIfLi1ELb0ELi2", "deep_zoom", "colorize_kernelIdLi0", "export"]
print(f"{'kernel':84s} {'VGPR':>5s} {'SGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for name in sorted(usage):
    if any(w in name for w in want):
        u = usage[name]
        print(f"{name[:84]:84s} {u.get('VGPRs',0):5d} {u.get('SGPRs',0):5d} {u.get('SGPRs Spill',0):6d} {u.get('VGPRs Spill',0):6d} "
              f"{u.get('ScratchSize [bytes/lane]',0):7d} {u.get('Occupancy [waves/SIMD]',0):4d} {u.get('LDS Size [bytes/block]',0):6d}")
