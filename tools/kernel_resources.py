"""VGPR / spill / LDS figures of the compiled gfx950 kernels (from the code-object metadata of an object file).

usage: python tools/kernel_resources.py nonlinear_optimizer_for_slam_amd/csrc/nos_core.o [name-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(obj):
    """→ list of dicts {name (demangled), agpr, vgpr, spill, scratch, lds, sgpr} for every gfx950 kernel in `obj`
    (an object file, shared library or executable with a .hip_fatbin section)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--unbundle"])
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    blocks = re.split(r"\n\s+- \.agpr_count:", txt)[1:]
    names = [re.search(r"\.name:\s+(\S+)", b).group(1) for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = []
    for b, name, d in zip(blocks, names, dem):
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))  # noqa: E731
        out.append({"name": d.strip() or name, "agpr": int(re.match(r"\s*(\d+)", b).group(1)), "vgpr": g("vgpr_count"),
                    "spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                    "lds": g("group_segment_fixed_size"), "sgpr": g("sgpr_count")})
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernel_resources(sys.argv[1]):
        if flt in k["name"]:
            print("agpr %3d vgpr %3d spill %3d scratch %5d lds %6d sgpr %3d  %s" % (
                k["agpr"], k["vgpr"], k["spill"], k["scratch"], k["lds"], k["sgpr"], k["name"][:150]))
