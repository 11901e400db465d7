"""Instruction census of the main loop of each kernel in a hipcc --save-temps .s file (dev tool).
usage: python tools/microbench/isa_loop_census.py file.s kernel_name [kernel_name ...]
The main loop = the longest backward s_cbranch span of the kernel.  Prints VALU totals split into the gfx950 issue classes of
profiles/r01/valu_ubench.txt: plain VOP1/VOP2 32-bit (about 2.6 cycles per wave-instruction) and everything else (64-bit, carry,
multiplier, f64: about 4.3-4.9)."""
import collections
import re
import sys

FAST = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32",
        "v_accvgpr", "v_bfi_b32", "v_and_or_b32", "v_or3_b32", "v_lshl_or_b32", "v_perm_b32", "v_bfe_u32")


def census(text, name):
    m = re.search(r"^(_Z\d+%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % name, text, re.S | re.M)
    lines = m.group(2).split("\n")
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            j = labels[mm.group(1)]
            if best is None or i - j > best[1] - best[0]:
                best = (j, i)
    c = collections.Counter()
    for l in lines[best[0]:best[1] + 1]:
        t = l.strip().split()
        if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
            c[t[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    fast = sum(v for k, v in c.items() if k.startswith(FAST))
    vg = re.search(r"\.amdhsa_kernel %s.*?\.amdhsa_next_free_vgpr (\d+)" % re.escape(m.group(1)), text, re.S)
    sc = re.search(r"; ScratchSize: (\d+)", text[m.start():])
    return c, valu, fast, int(vg.group(1)), int(sc.group(1)) if sc else None


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        c, valu, fast, vgpr, scratch = census(text, name)
        print("%s: main loop %d VALU = %d four-cycle-class + %d plain 32-bit; %d s_nop; %d VGPRs, scratch %s" % (name, valu, valu - fast, fast, c["s_nop"], vgpr, scratch))
        print("    " + ", ".join("%s %d" % kv for kv in c.most_common(14)))
