#!/usr/bin/env python3
"""Which kernels did an edit of the device header change?  Pairs the code objects of the in-tree cache by configuration name
(NAME-gfx950-<key of the source text>.hsaco: two keys per name after an edit, older file first) and compares their machine
code — kmc.kernel_code_sha256's sections (.text, .rodata, .note) — and, where it differs, the instruction histograms of
kmc_expand_*.  No GPU needed.
usage: tools/compare_code_objects.py [cache_dir] > profiles/rNN_code_identity.txt"""
import collections
import glob
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_specification_amd.checker import elf_sections  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def ident(path):
    sec = elf_sections(open(path, "rb").read())
    h = hashlib.sha256()
    for n in (".text", ".rodata", ".note"):
        h.update(n.encode() + len(sec[n]).to_bytes(8, "little") + sec[n])
    return h.hexdigest()


def histogram(path):
    text = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
    h, on = collections.Counter(), False
    for line in text.splitlines():
        if line.endswith(">:"):
            on = "<kmc_expand_" in line
        elif on:
            m = re.match(r"\s+([a-z_0-9]+)", line)
            if m:
                h[m.group(1)] += 1
    return h


def vgprs(path):
    notes = subprocess.run([OBJDUMP.replace("objdump", "readelf"), "--notes", path], capture_output=True, text=True).stdout
    m = re.search(r"\.name:\s+kmc_expand_\S+.*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", notes, re.S)
    return (int(m.group(1)), int(m.group(2))) if m else None


def main():
    """Files older than the newest build's start (the largest gap between consecutive mtimes) are the OLD generation.  A name
    can have several builds per generation (KMC_VERIFY's pair, forced layouts, profiling defines: other keys, same name): a
    new file is `same` if ANY old file of its name holds identical machine code, otherwise it is compared with the old file
    whose kmc_expand has the closest instruction count."""
    cache = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "kafka_specification_amd", "kmc_cache")
    files = sorted(glob.glob(os.path.join(cache, "*.hsaco")), key=os.path.getmtime)
    times = [os.path.getmtime(f) for f in files]
    cut = max(range(1, len(files)), key=lambda i: times[i] - times[i - 1])
    old_by, new_by = collections.defaultdict(list), collections.defaultdict(list)
    for k, f in enumerate(files):
        (old_by if k < cut else new_by)[os.path.basename(f).rsplit("-gfx950-", 1)[0]].append(f)
    same = changed = single = 0
    lines = []
    for name in sorted(new_by):
        olds = {ident(f): f for f in old_by.get(name, [])}
        for new in new_by[name]:
            if not olds:
                single += 1
                continue
            idn = ident(new)
            if idn in olds:
                same += 1
                lines.append(f"same     {name}  {idn[:16]}")
                continue
            changed += 1
            hn = histogram(new)
            old = min(olds.values(), key=lambda f: abs(sum(histogram(f).values()) - sum(hn.values())))
            ho = histogram(old)
            diff = {k: hn[k] - ho[k] for k in set(ho) | set(hn) if hn[k] != ho[k]}
            lines.append(f"CHANGED  {name}  kmc_expand: {sum(ho.values())} -> {sum(hn.values())} instructions, (VGPRs, spilled) "
                         f"{vgprs(old)} -> {vgprs(new)}; " + ", ".join(f"{k} {v:+d}" for k, v in sorted(diff.items())))
    print(f"# {len(files) - cut} code objects of the newest build against the {cut} before it: {same} with machine code identical to "
          f"an earlier build of the same configuration, {changed} changed, {single} without a predecessor")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
