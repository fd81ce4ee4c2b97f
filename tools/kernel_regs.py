#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels in one compiled object of the build (no GPU needed):
    python tools/kernel_regs.py conv_kernels [name-filter]
Reads the gfx950 code object out of sdeflow_light_amd/build/<name>.o (llvm-objcopy + clang-offload-bundler) and prints the
amdhsa kernel metadata — the check for spills (scratch > 0) and for the occupancy a register count allows."""
import os, re, subprocess, sys, tempfile

LL = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdeflow_light_amd", "build", sys.argv[1] + ".o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as T:
    subprocess.check_call([f"{LL}/llvm-objcopy", f"--dump-section=.hip_fatbin={T}/fat.bin", obj, f"{T}/copy.o"])   # an output file: objcopy rewrites its input otherwise
    subprocess.check_call([f"{LL}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={T}/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={T}/k.co"])
    txt = subprocess.check_output([f"{LL}/llvm-readelf", "--notes", f"{T}/k.co"], text=True)
for blk in txt.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or flt not in name.group(1):
        continue
    g = lambda k: re.search(k + r":\s+(\d+)", blk).group(1)
    ag = re.match(r":\s+(\d+)", blk).group(1)
    print("%-72s agpr %3s vgpr %3s sgpr %3s scratch %4s lds %s" % (name.group(1)[:72], ag, g(r"\.vgpr_count"), g(r"\.sgpr_count"),
                                                                 g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
