#!/usr/bin/env python3
"""Static check of a gfx950 kernel's vector-memory bookkeeping (no GPU): walk the disassembly, keep the in-order queue of
outstanding vector-memory operations (loads with their destination registers, stores), retire all but the youngest N at every
`s_waitcnt vmcnt(N)`, and report every instruction that touches a register whose load is still in the queue.  The hardware does
not interlock on outstanding loads -- such an instruction reads (or clobbers) whatever the register held before.

Every path through the control-flow graph is followed (both sides of every conditional branch, loops until the queue at their
head repeats).

    python tools/isa_pending.py matcha-tts-24k_amd/build/tblock_chain.o [kernel-substring]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = "/opt/rocm/lib/llvm/bin"
INSN = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):.*?(?:<([^>+]+)(?:\+0x([0-9a-f]+))?>)?\s*$")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def disassemble(obj):
    d = Path(tempfile.mkdtemp())
    fat, co = d / "fat.bin", d / "dev.co"
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(\S+)>:$", line)
        if m:
            base, cur = int(m.group(1), 16), []
            kernels[m.group(2)] = cur
            continue
        if cur is None:
            continue
        m = INSN.match(line)
        if not m:
            continue
        mnem, ops, addr, _, toff = m.groups()
        is_branch = mnem.startswith("s_cbranch") or mnem == "s_branch"
        cur.append((int(addr, 16) - base, mnem, ops, int(toff, 16) if (is_branch and toff is not None) else (0 if is_branch else None)))
    return kernels


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def is_vmem(mnem):
    return mnem.startswith(("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
                            "scratch_load", "scratch_store", "flat_load", "flat_store", "flat_atomic"))


def pending_violations(insns, max_visits=400000):
    """[(offset, mnemonic, operands, registers)]: instructions that touch the destination of a load still in the queue, on ANY
    path through the control-flow graph (states = the in-order queue of outstanding operations; a (block, state) pair is walked
    once, so loops are followed until the state at their head repeats)."""
    n = len(insns)
    at = {off: i for i, (off, *_r) in enumerate(insns)}
    leaders = {0}
    succ = {}
    for i, (off, mnem, ops, tgt) in enumerate(insns):
        if tgt is not None:
            t = at.get(tgt)
            if t is not None:
                leaders.add(t)
            if i + 1 < n:
                leaders.add(i + 1)
            succ[i] = ([t] if t is not None else []) + ([i + 1] if (mnem != "s_branch" and i + 1 < n) else [])
        elif mnem == "s_endpgm":
            succ[i] = []
            if i + 1 < n:
                leaders.add(i + 1)
    found = {}
    seen = set()
    work = [(0, ())]
    visits = 0
    regs_of = [vregs(ops) if not mnem.startswith("s_") else frozenset() for (off, mnem, ops, tgt) in insns]
    while work:
        i, state = work.pop()
        if (i, state) in seen:
            continue
        seen.add((i, state))
        visits += 1
        if visits > max_visits:
            raise RuntimeError("isa_pending: state space too large")
        queue = list(state)
        while True:
            off, mnem, ops, tgt = insns[i]
            if mnem == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    k = int(m.group(1))
                    del queue[:max(0, len(queue) - k)]
            elif queue or is_vmem(mnem):
                r = regs_of[i]
                if is_vmem(mnem) and "_load" in mnem:         # a load over a pending load's register lands after it (in-order return):
                    r = vregs(",".join(ops.split(",")[1:]))   # only its address operands are read now
                if r and queue:
                    busy = frozenset().union(*queue)
                    hit = r & busy
                    if hit:
                        found.setdefault(off, (off, mnem, ops, sorted(hit)))
                if is_vmem(mnem):
                    queue.append(frozenset(vregs(ops.split(",")[0])) if ("_load" in mnem and "lds" not in mnem) else frozenset())
                    if len(queue) > 64:
                        del queue[0]
            if i in succ:
                for t in succ[i]:
                    work.append((t, tuple(queue)))
                break
            i += 1
            if i >= n:
                break
            if i in leaders:
                work.append((i, tuple(queue)))
                break
    return [found[k] for k in sorted(found)]


def main():
    obj = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    for name, insns in disassemble(obj).items():
        if pat not in name:
            continue
        v = pending_violations(insns)
        print(f"{name}: {len(insns)} instructions, {len(v)} touch a register with its load outstanding")
        for off, mnem, ops, regs in v[:40]:
            print(f"    +0x{off:x}  {mnem} {ops}    <- v{regs}")
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
