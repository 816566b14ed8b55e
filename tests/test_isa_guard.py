"""Build-time guard of the instruction streams the kernels' hand-counted `s_waitcnt vmcnt(N)` depend on (CPU only: the gfx950 code
objects of the in-tree build are disassembled with llvm-objdump, nothing runs).

The check that matters most is `test_chain_kernel_never_touches_a_register_with_its_load_outstanding`: the round-3 determinism
failure (a whole chain launch a few per cent off, once in a few hundred launches) was hipcc copying ring registers between two
phases of tblock_chain_kernel BEFORE the inline-asm loads into them had landed.

Two kernels count vector-memory operations by hand:
  * gemm_p16_kernel's LDS-DMA ring (csrc/gemm_p16.hip): tile D-1 .. D requests with (D-1) * PER_TILE (+ the residual prefetch) loads
    in flight across a raw s_barrier;
  * tblock_chain_kernel's weight ring (csrc/tblock_chain.hip): inline-asm global loads the compiler does not count at all.
Either goes silently wrong if the compiler adds, removes, hoists or spills a load inside the counted region (round-2 verdict
item 8 / advisor finding).  The tests assert, per instantiation, what the counts assume: the loads per loop iteration, the wait
constants present, no scratch (spill) traffic and no foreign vector-memory loads inside the hot loops."""
import re
import subprocess
from collections import Counter

import pytest

from conftest import ROOT, sub

LLVM = "/opt/rocm/lib/llvm/bin"
PKG = ROOT / "matcha-tts-24k_amd"
INSN = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):.*?(?:<([^>+]+)(?:\+0x([0-9a-f]+))?>)?\s*$")


def disassemble(stem, tmp_path_factory):
    """{kernel symbol: [(offset, mnemonic, operands, branch target offset or None)]} of one translation unit's device code."""
    sub("_hip").build()
    obj = PKG / "build" / f"{stem}.o"
    assert obj.exists(), obj
    d = tmp_path_factory.mktemp("isa_" + stem)
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


def loops(insns):
    """[(first index, last index)] of the bodies closed by a backward branch, outermost last."""
    at = {off: i for i, (off, *_r) in enumerate(insns)}
    out = []
    for i, (off, mnem, ops, tgt) in enumerate(insns):
        if tgt is not None and tgt <= off and tgt in at:
            out.append((at[tgt], i))
    return out


def hot_loops(insns, min_mfma=8):
    """Innermost loops that contain matrix instructions.  Divergent control flow inside a body gives several backward branches
    to (almost) the same head: those are one loop, its body ends at the last of them."""
    ls = [(a, b) for a, b in loops(insns) if sum(1 for x in insns[a:b + 1] if x[1].startswith("v_mfma")) >= min_mfma]
    merged = []
    for a, b in sorted(ls):
        if merged and a - merged[-1][0] <= 8:
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    return [(a, b) for a, b in merged if not any((c, d) != (a, b) and a <= c and d <= b for c, d in merged)]


def vm_waits(body):
    return Counter(int(m.group(1)) for x in body if x[1] == "s_waitcnt" for m in [re.search(r"vmcnt\((\d+)\)", x[2])] if m)


@pytest.fixture(scope="module")
def chain_isa(tmp_path_factory):
    return disassemble("tblock_chain", tmp_path_factory)


@pytest.fixture(scope="module")
def p16_isa(tmp_path_factory):
    return disassemble("gemm_p16", tmp_path_factory)


def test_chain_kernel_never_touches_a_register_with_its_load_outstanding(chain_isa):
    """tblock_chain_kernel<C, QB, CH>, every instantiation, every path of its control-flow graph: between a vector-memory load and
    the `s_waitcnt vmcnt(N)` that retires it, no instruction reads or writes the load's destination (tools/isa_pending.py).  The
    hardware has no interlock there and the compiler does not know the inline-asm loads are asynchronous: a live-range split, a
    spill or a register reuse inside that window compiles, passes every test on a warm L2 and reads the previous fragment on a
    cold one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_pending", ROOT / "tools" / "isa_pending.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    names = [n for n in chain_isa if "tblock_chain_kernel" in n]
    assert len(names) >= 6
    for name in names:
        bad = mod.pending_violations(chain_isa[name])
        assert not bad, (name, [(hex(o), m, ops) for o, m, ops, _ in bad[:6]])


def test_chain_kernel_rings_are_not_spilled_and_waits_are_the_written_ones(chain_isa):
    """tblock_chain_kernel<C, QB, CH>: the fragment ring lives in registers that inline-asm loads fill; the compiler treats an asm
    output as available at once, so a spill (or any compiler-made copy) of a ring register ahead of its wait would store
    garbage.  Inside every loop with matrix instructions: no scratch traffic, no vector-memory loads except the ring's
    global_load_dwordx4, only the hand-written vmcnt constants (0 = the drain that ends every k-loop)."""
    names = [n for n in chain_isa if "tblock_chain_kernel" in n]
    assert len(names) >= 6
    for name in names:
        C, QB, CH = (int(v) for v in re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)E", name).groups())
        NT, NT1, MT, KG, KG2 = C // 128, CH // 128, QB // 16, C // 32, CH // 32
        R = 12 if C == 384 else 8
        FW, F1S = 2 * NT, 2 * NT1
        insns = chain_isa[name]
        hot = hot_loops(insns)
        if C != 384:
            # the narrow widths of the test suite (the compiler unrolls some of their loops completely and adds a full drain of its
            # own in one of them -- slower, not wrong): no spill anywhere between the first and the last matrix instruction
            mf = [i for i, x in enumerate(insns) if x[1].startswith("v_mfma")]
            assert not any(x[1].startswith("scratch_") for x in insns[mf[0]:mf[-1]]), (name, "scratch traffic among the k-loops")
            continue
        assert len(hot) >= 3, (name, hot)                     # out-projection, hidden chunks, q|k|v passes
        allowed = {0, R - FW, R - F1S, FW + 1}                 # 0: the drain that ends every k-loop
        seen_chunk = seen_qkv = False
        for a, b in hot:
            body = insns[a:b + 1]
            ops = Counter(x[1] for x in body)
            assert not any(k.startswith("scratch_") for k in ops), (name, "scratch traffic inside a k-loop")
            foreign = [k for k in ops if (k.startswith("global_load") and k != "global_load_dwordx4") or k.startswith("buffer_load")
                       or k.startswith("flat_load")]
            assert not foreign, (name, foreign)
            waits = vm_waits(body)
            assert set(waits) <= allowed, (name, dict(waits), allowed)
            n_mfma = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
            n_load = ops["global_load_dwordx4"]
            if n_mfma == 3 * MT * (KG * NT1 + KG2 * NT) and ops.get("s_barrier", 0):      # one hidden chunk: FF1 + FF2
                seen_chunk = True
                assert n_load == KG * F1S + KG2 * FW, (name, n_load)
                assert waits[R - F1S] >= KG and waits[R - FW] >= (KG2 if F1S != FW else 0), (name, dict(waits))
                assert ops["s_barrier"] == 2
            elif n_mfma == 3 * MT * KG * NT and ops.get("global_store_dwordx2", 0):   # a q|k|v pass
                seen_qkv = True
                assert n_load == KG * FW, (name, n_load)
        assert seen_chunk and seen_qkv, name


def test_p16_ring_loop_holds_exactly_its_tile_requests(p16_isa):
    """gemm_p16_kernel<BM, LN, NST >= 3, ...>: the k-loop's counted wait `vmcnt((NST-2) * PER_TILE)` (plus 2 / 4 / 8 while the
    residual prefetch is still in the queue) is exact only if an iteration issues PER_TILE = BM/32 + 4 LDS-DMA requests and
    nothing else that counts: no spill, no other load.  Ahead of the loop the prefetch's plain loads must sit behind the last
    tile request, never between two of them."""
    names = [n for n in p16_isa if "gemm_p16_kernel" in n]
    assert names
    checked = 0
    for name in names:
        m = re.search(r"gemm_p16_kernelILi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)E", name)
        BM, LN, NST, MODE, M16, GN, KS = (int(v) for v in m.groups())
        if NST < 3:
            continue
        per_tile, D = BM // 32 + 4, NST - 1
        insns = p16_isa[name]
        hot = hot_loops(insns, 2)
        assert hot, name
        ring = [(a, b) for a, b in hot if any(x[1].startswith("global_load_lds") for x in insns[a:b + 1])]
        assert len(ring) == 1, (name, ring)
        a, b = ring[0]
        body = insns[a:b + 1]
        ops = Counter(x[1] for x in body)
        assert not any(k.startswith("scratch_") for k in ops), (name, "spill inside the ring loop")
        assert ops["global_load_lds_dwordx4"] == per_tile, (name, ops["global_load_lds_dwordx4"], per_tile)
        other = [k for k in ops if (k.startswith("global_load") and not k.startswith("global_load_lds")) or k.startswith("buffer_load")
                 or k.startswith("flat_load") or k.startswith("global_store") or k.startswith("global_atomic")]
        assert not other, (name, other)
        steady = (D - 1) * per_tile
        allowed = {0, steady, steady + 2, steady + 4, steady + 8}
        waits = vm_waits(body)
        assert steady in waits and set(waits) <= allowed, (name, dict(waits), allowed)
        # prologue: D tiles requested ahead of the loop; plain global loads between the first and the last of those requests
        # would be counted as if they were behind them
        pre = insns[:a]
        lds_at = [i for i, x in enumerate(pre) if x[1].startswith("global_load_lds")]
        assert len(lds_at) >= D * per_tile or KS > 1, (name, len(lds_at))
        if lds_at:
            lo = lds_at[-D * per_tile] if len(lds_at) >= D * per_tile else lds_at[0]
            between = [x[1] for x in pre[lo:lds_at[-1]] if x[1].startswith("global_load") and not x[1].startswith("global_load_lds")]
            assert not between, (name, between)
        checked += 1
    assert checked >= 8, checked
