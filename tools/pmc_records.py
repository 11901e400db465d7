#!/usr/bin/env python3
"""What the committed PMC passes (profiles/rNN/pmc_summary.json, traffic.json; collected by tools/gpu_record.sh in separate rocprofv3 --pmc runs)
say about one launch of ntt_pass_kernel, for bench.py's `roofline` object; and the digest that ties those figures to the kernel sources."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measured_valu(log2n):
    """wave-level VALU instructions per ntt_pass_kernel launch from the committed PMC runs (profiles/*/pmc_summary.json), or None"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "pmc_summary.json"))):
        try:
            d = json.load(open(f))
            for run, kernels in d.items():
                if not run.endswith("_%s" % log2n):
                    continue
                for name, ctrs in kernels.items():
                    if "ntt_pass" in name and "SQ_INSTS_VALU" in ctrs:
                        best = ctrs["SQ_INSTS_VALU"]["avg_per_dispatch"]
        except Exception:
            pass
    return best


KERNEL_SOURCES = ("ntt_tile.cuh", "ntt_plan.h", "field.cuh", "field_asm.cuh")


def kernel_source_digest():
    """SHA-256 over the sources that define ntt_pass_kernel: tools/gpu_record.sh stores it next to the PMC figures it collects
    (profiles/rNN/traffic.json), and a bench line says so when the figures it quotes predate a change to those files"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(REPO, "stark-anatomy_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_figures_are_current():
    """True / False / None (no record): were the latest committed PMC figures collected with today's kernel sources?"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic.json")))
    if not files:
        return None
    try:
        recorded = json.load(open(files[-1])).get("kernel_source_sha256_16")
    except Exception:
        return None
    return None if recorded is None else recorded == kernel_source_digest()


def measured_traffic(log2n):
    """HBM bytes per ntt_pass_kernel launch from the committed rocprofv3 PMC runs (profiles/*/traffic.json), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic.json"))):
        try:
            rec = json.load(open(f)).get(str(log2n))
            if rec:
                best = rec["hbm_bytes_per_launch"]
        except Exception:
            pass
    return best
