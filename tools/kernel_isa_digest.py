"""A digest of every kernel's instruction stream in the built librankfm_hip.so: llvm-objdump on each gfx950 code object, the mnemonics
and operands of each kernel symbol hashed (addresses and encodings left out; branch targets kept as the offset from the kernel's start).
Two builds whose digests agree for a kernel run the same instructions there -- what a refactoring that should not touch a kernel is
checked with (round 6: the stripe sampler's removal).  Runs anywhere (no GPU).

    python tools/kernel_isa_digest.py [--out digests.json] [--diff before.json]

Measurement tooling, not product."""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digests(lib):
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    with open(lib, "rb") as f:
        blob = f.read()
    out = {}
    for _, elf in kernel_resources.code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(elf)
            tmp.flush()
            text = subprocess.run([objdump, "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True, check=True).stdout
        cur, start, h, n = None, 0, None, 0
        for line in text.splitlines():
            m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
            if m:
                if cur:
                    out[cur] = (h.hexdigest()[:16], n)
                cur, start, h, n = m.group(2), int(m.group(1), 16), hashlib.sha256(), 0
                continue
            if cur is None or "//" not in line:
                continue
            ins = line.split("//")[0].strip()
            # branch targets: "s_cbranch_scc1 65212" style operands are relative already; symbolic "<sym+0x1c4>" forms are normalised
            ins = re.sub(r"<[^>+]+\+(0x[0-9a-f]+)>", r"<+\1>", ins)
            h.update(ins.encode() + b"\n")
            n += 1
        if cur:
            out[cur] = (h.hexdigest()[:16], n)
    names = kernel_resources.demangle(list(out))
    return {re.sub(r"\(rfm::SgdArgs\)$", "", names[k]).replace("void ", ""): v for k, v in out.items() if not k.endswith(".kd")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "rankfm_amd", "librankfm_hip.so"))
    ap.add_argument("--out", default="")
    ap.add_argument("--diff", default="")
    a = ap.parse_args()
    d = digests(a.lib)
    if a.out:
        json.dump(d, open(a.out, "w"), indent=0, sort_keys=True)
    if a.diff:
        old = {k: tuple(v) for k, v in json.load(open(a.diff)).items()}
        same = [k for k in d if k in old and tuple(d[k]) == old[k]]
        changed = [k for k in d if k in old and tuple(d[k]) != old[k]]
        print("%d kernels identical to the instruction, %d changed, %d gone, %d new" % (
            len(same), len(changed), len(set(old) - set(d)), len(set(d) - set(old))))
        for k in sorted(changed):
            print("  changed: %s  (%d -> %d instructions)" % (k, old[k][1], d[k][1]))
        for k in sorted(set(old) - set(d)):
            print("  gone:    %s" % k)
        for k in sorted(set(d) - set(old)):
            print("  new:     %s" % k)
    else:
        print("%d kernels" % len(d))


if __name__ == "__main__":
    main()
