"""VGPR / spill / LDS figures of the compiled gfx950 kernels (from the code-object metadata of an object file).

usage: python tools/kernel_resources.py nonlinear_optimizer_for_slam_amd/csrc/nos_core.o [name-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fatbin"), os.path.join(d, "co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--unbundle"])
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for b in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))  # noqa: E731
    print("agpr %3d vgpr %3d spill %3d scratch %5d lds %6d sgpr %3d  %s" % (
        int(re.match(r"\s*(\d+)", b).group(1)), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
        g("group_segment_fixed_size"), g("sgpr_count"), dem[:150]))
