#!/usr/bin/env python3
"""make check-isa -- what geodesic_dma_kernel (srh_dense.hip) asks of the COMPILER, checked in the ISA it produced.

The kernel requests the next tile by LDS-DMA in assembler statements of its own (s_mov_b32 m0 / global_load_lds_dword[x4]),
hides those LDS writes from the compiler on purpose and stands on hand-placed waits.  Two things would
break it silently, and a ROCm upgrade could bring either: the compiler keeping a value of its own in m0 across the statements
(the clobber on the reserved register is ignored: the build warns about exactly that), and a tile's LDS reads scheduled before
the s_waitcnt vmcnt(0) that covers the tile's requests.  So, on the disassembly of the kernel:

  A. every instruction that names m0 is one of the statements' own `s_mov_b32 m0, s<N>`, followed by `s_nop 0` and a
     `global_load_lds_dword` / `_dwordx4`; every global_load_lds has that prologue; there are at least two of each kind;
  B. behind every global_load_lds, in program text, an `s_waitcnt vmcnt(0)` comes before the next `s_barrier` (the hand-over
     of the tile buffers) and before the end of the program;
  C. the first LDS read of the kernel comes after the first `s_waitcnt vmcnt(0)` (the first tile: nothing hides that wait).

usage: check_isa.py kernel.s   (exit 0: holds, 1: broken -- the build fails)"""
import re
import sys


def instructions(path, prefix):
    body, on = [], False
    for line in open(path):
        if line.startswith(prefix):
            on = True
            continue
        if not on:
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        body.append(t.split(";")[0].strip())
        if t.startswith("s_endpgm"):
            break
    return body


def main():
    ins = instructions(sys.argv[1], "_ZN3srh19geodesic_dma_kernel")
    if len(ins) < 1000:
        print("check-isa: geodesic_dma_kernel not found in", sys.argv[1]); return 1
    bad = []
    m0 = [k for k, t in enumerate(ins) if re.search(r"\bm0\b", t)]
    dma = [k for k, t in enumerate(ins) if t.startswith("global_load_lds_dword")]
    for k in m0:
        ok = re.fullmatch(r"s_mov_b32 m0, s\d+", ins[k]) and ins[k + 1] == "s_nop 0" and ins[k + 2].startswith("global_load_lds_dword")
        if not ok:
            bad.append("A: m0 is touched outside the request statements: '%s' / '%s' / '%s'" % (ins[k], ins[k + 1], ins[k + 2]))
    for k in dma:
        if k < 2 or not re.fullmatch(r"s_mov_b32 m0, s\d+", ins[k - 2]) or ins[k - 1] != "s_nop 0":
            bad.append("A: a global_load_lds without its m0 prologue: '%s' / '%s' / '%s'" % (ins[k - 2], ins[k - 1], ins[k]))
    if len(dma) < 2 or len(m0) != len(dma) or sum(t.startswith("global_load_lds_dwordx4") for t in ins) < 2:
        bad.append("A: expected the tile requests (dwordx4) and the mask requests (dword), each at the kernel's head and in its loop: %d m0 writes, %d requests" % (len(m0), len(dma)))
    for k in dma:
        for t in ins[k + 1:]:
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                break
            if t.startswith("s_barrier") or t.startswith("s_endpgm"):
                bad.append("B: '%s' reached behind request %d without an s_waitcnt vmcnt(0)" % (t, k)); break
    first_wait = next((k for k, t in enumerate(ins) if t.startswith("s_waitcnt") and "vmcnt(0)" in t), None)
    first_read = next((k for k, t in enumerate(ins) if t.startswith("ds_read")), None)
    if first_wait is None or first_read is None or first_read < first_wait:
        bad.append("C: the first LDS read (instruction %s) is not behind the first s_waitcnt vmcnt(0) (%s)" % (first_read, first_wait))
    for b in bad:
        print("check-isa:", b)
    if not bad:
        print("check-isa: geodesic_dma_kernel: %d request statements own m0, every request is waited for before the tile hand-over (%d instructions)" % (len(dma), len(ins)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
