#!/usr/bin/env python3
"""VGPR / SGPR / scratch / spill counts of the library's kernels whose name contains a pattern (dev tool).
   python tools/kernel_regs.py [pattern=ntt_pass_kernel] [path=stark-anatomy_amd/libstarkcore.so]"""
import glob, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "ntt_pass_kernel"
so = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "stark-anatomy_amd", "libstarkcore.so"))
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as tmp:
    link = os.path.join(tmp, "lib.so")
    os.symlink(so, link)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", link], capture_output=True, cwd=tmp)     # writes lib.so.N.<target> beside the link
    for f in sorted(glob.glob(link + ".*gfx950")):
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or pat not in name.group(1):
                continue
            g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-70s vgpr %3s sgpr %3s scratch %4s vgpr_spill %3s" % (dem[-70:], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
