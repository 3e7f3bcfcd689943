"""Registers, spills, scratch and LDS of every kernel in the built librankfm_hip.so, read from the code objects' own metadata
(the AMDGPU msgpack note of each gfx950 ELF inside the library's .hip_fatbin bundles).  Runs anywhere (no GPU, no ROCm tools).

    python tools/kernel_resources.py [--all] [--filter sgd_] [--csv profiles/rNN_kernel_resources.csv]

Without --all only kernels whose name contains one of the production entry points are listed.  Measurement tooling, not product."""
import argparse
import os
import re
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """every device ELF of every offload bundle in a host object / shared library"""
    at = 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size and "amdgcn" in triple:
                yield triple, blob[at + off:at + off + size]
        at += len(MAGIC)


def elf_notes(elf):
    """(name, type, desc) of every note of a little-endian ELF64"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0").decode()
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def kernels(path):
    rows = []
    with open(path, "rb") as f:
        blob = f.read()
    for triple, elf in code_objects(blob):
        for name, ntype, desc in elf_notes(elf):
            if name == "AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    rows.append(dict(triple=triple, symbol=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                                     vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                     scratch=k.get(".private_segment_fixed_size", 0), lds_static=k.get(".group_segment_fixed_size", 0),
                                     max_wg=k.get(".max_flat_workgroup_size", 0)))
    names = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["kernel"] = re.sub(r"\(rfm::SgdArgs\)|\(.*\)$", "", names[r["symbol"]]).replace("void ", "").replace("rfm::", "")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "rankfm_amd", "librankfm_hip.so"))
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--filter", default="")
    ap.add_argument("--csv", default="")
    a = ap.parse_args()
    rows = kernels(a.lib)
    if not rows:
        sys.exit("no gfx950 code objects found in " + a.lib)
    if a.filter:
        rows = [r for r in rows if a.filter in r["kernel"]]
    rows.sort(key=lambda r: r["kernel"])
    cols = ("kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch", "lds_static", "max_wg")
    if a.csv:
        with open(a.csv, "w") as f:
            f.write(",".join(cols) + "\n")
            for r in rows:
                f.write(",".join('"%s"' % r[c] if c == "kernel" else str(r[c]) for c in cols) + "\n")
    w = max(len(r["kernel"]) for r in rows)
    print("%-*s %5s %5s %5s %7s %7s %8s %8s %6s" % (w, "kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch", "lds", "max wg"))
    for r in rows:
        print("%-*s %5d %5d %5d %7d %7d %8d %8d %6d" % (w, r["kernel"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch"], r["lds_static"], r["max_wg"]))
    print("%d kernels, %d with VGPR spills" % (len(rows), sum(1 for r in rows if r["vgpr_spill"])))


if __name__ == "__main__":
    main()
