#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the built library, from the code-object metadata.

    python tools/kernel_resources.py [substring ...] [--spills]

Reads the gfx950 code object out of every octopuszk_amd/_obj/*.o (llvm-objcopy the .hip_fatbin section,
clang-offload-bundler --unbundle, llvm-readelf --notes) and prints one line per kernel:
VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, static LDS bytes, max workgroup size.
`--spills` lists only kernels with a non-zero spill count or scratch (exit code 1 if any has more than one wave
per workgroup: the throughput kernels must not spill)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n")


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "x.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
        if not os.path.exists(fb) or os.path.getsize(fb) == 0:
            return []
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    ks, cur = [], None
    for line in notes.split("\n"):
        if re.match(r"\s+- \.agpr_count:", line) or re.match(r"\s+- \.args:", line):
            cur = {}
            ks.append(cur)
        m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)\s*$", line)
        if m and cur is not None and m.group(1) in ("agpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                    "group_segment_fixed_size", "private_segment_fixed_size",
                                                    "max_flat_workgroup_size", "name"):
            cur[m.group(1)] = m.group(2)
    return [k for k in ks if "name" in k]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_spills = "--spills" in sys.argv
    objdir = os.path.join(ROOT, "octopuszk_amd", "_obj")
    rows = []
    for f in sorted(os.listdir(objdir)):
        if f.endswith(".hip.o"):
            for k in kernels_of(os.path.join(objdir, f)):
                rows.append((f[:-len(".hip.o")], k))
    names = demangle([k["name"] for _, k in rows])
    bad = 0
    print("%-14s %5s %5s %6s %8s %7s %5s  %s" % ("TU", "VGPR", "AGPR", "spill", "scratch", "LDS", "wg", "kernel"))
    for (tu, k), nm in zip(rows, names):
        nm = re.sub(r"\(.*$", "", nm).replace("ozk::", "")
        if args and not any(a in nm for a in args):
            continue
        spill, scratch = int(k.get("vgpr_spill_count", 0)), int(k.get("private_segment_fixed_size", 0))
        if only_spills and not (spill or scratch):
            continue
        wg = int(k.get("max_flat_workgroup_size", 0))
        if (spill or scratch) and wg > 64:
            bad += 1
        print("%-14s %5s %5s %6d %8d %7s %5d  %s" % (tu, k.get("vgpr_count"), k.get("agpr_count"), spill, scratch,
                                                      k.get("group_segment_fixed_size"), wg, nm))
    return 1 if (only_spills and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
