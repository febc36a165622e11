#!/usr/bin/env python3
"""Where do a kernel's loops lie relative to the 64-byte lines of its code?   usage: tools/loop_lines.py <lib.so> <kernel substring> [line bytes]

Written for profiles/r06_step_forms.txt (5): rcw_fill256_cast_kernel is 25-30 us slower at 16384 x 512 / 8192 x 1024 view columns in builds whose fill
half's chunk loop (540 / 684 bytes, three / four 16-byte stores) starts 4 or 36 bytes into a line than in builds where it starts at 0 or 28; what the
casting half's march loops (60 / 68 bytes) span does not differ between them.  -falign-loops=64 for rcw_cast.hip (csrc/Makefile) puts every loop on a line."""
import re
import subprocess
import sys
import tempfile

OBJDUMP, OBJCOPY, READELF = (f"/opt/rocm/lib/llvm/bin/llvm-{t}" for t in ("objdump", "objcopy", "readelf"))


def code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".fatbin") as f:
        subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", lib, f.name], check=True)
        blob = open(f.name, "rb").read()
    i = 0
    while True:
        i = blob.find(b"\x7fELF", i)
        if i < 0:
            return
        yield blob[i:]
        i += 4


def loops(lib, pat):
    """{kernel symbol: [(start address, bytes, ds_read_u8 in the body, 16-byte stores in the body)]} of every backward branch of the kernels whose
    symbol holds `pat`."""
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            syms = subprocess.run([READELF, "-sW", f.name], capture_output=True, text=True).stdout
            names = sorted({l.split()[-1] for l in syms.splitlines() if pat in l and " FUNC " in l})
            for name in names:
                dis = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={name}", f.name], capture_output=True, text=True).stdout
                ins = []                                        # (address, mnemonic, operand text)
                for l in dis.splitlines():
                    m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", l)
                    if m:
                        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
                rows = []
                for addr, mn, ops in ins:
                    if not mn.startswith("s_cbranch") and mn != "s_branch":
                        continue
                    off = int(ops.split()[0])
                    if off < 32768:
                        continue                                # forward
                    target, end = addr + 4 + (off - 65536) * 4, addr + 4
                    body = [i for i in ins if target <= i[0] < end]
                    rows.append((target, end - target, sum(1 for i in body if i[1] == "ds_read_u8"),
                                 sum(1 for i in body if i[1].startswith("global_store_dwordx4"))))
                out[name] = rows
    return out


def chunk_loops(lib, pat):
    """The fill half's chunk loops (16-byte stores, no ds_read_u8, under 1200 bytes) of the kernels whose symbol holds `pat`: {symbol: [(bytes, start % 64)]}."""
    return {k: [(r[1], r[0] % 64) for r in v if r[3] and not r[2] and r[1] < 1200] for k, v in loops(lib, pat).items()}


def main():
    lib, pat = sys.argv[1], sys.argv[2]
    line = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    for name, rows in loops(lib, pat).items():
        chunk = [r for r in rows if r[3] and not r[2] and r[1] < 1200]
        march = [r for r in rows if r[2] == 1 and r[1] <= 96]
        spans = [(r[0] + r[1] - 1) // line - r[0] // line + 1 for r in march]
        print(f"{name[:70]}: {len(rows)} loops; chunk loops (16-byte stores, no ds_read_u8): " + ", ".join(f"{r[1]} B from byte {r[0] % line} of a line" for r in chunk) +
              f"; march loops (one ds_read_u8, <= 96 B): {len(march)}, bytes {sorted({r[1] for r in march})}, spanning 1 / 2 / 3 lines of {line} B: {[spans.count(n) for n in (1, 2, 3)]}")


if __name__ == "__main__":
    main()
