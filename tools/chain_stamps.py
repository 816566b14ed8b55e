#!/usr/bin/env python3
"""Phase stamps of workgroup 0 of the chain kernel (diagnostic build: tools/build_variant.sh chain_stamp "-DMTTS_CHAIN_STAMP"
tblock_chain.hip model.hip; run with MTTS_HIP_LIB=$PWD/tools/ab/chain_stamp.so)."""
import importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hip = importlib.import_module("matcha-tts-24k_amd._hip")
NAMES = {0: "start", 1: "prologue done (x tile, constants, ring, first attention rows)", 2: "out-projection k-loop", 3: "its epilogue + LayerNorm moments",
         4: "chunk 0: FF1 k-loop", 5: "chunk 0: SnakeBeta epilogue + barrier", 6: "chunk 0: FF2 k-loop + barrier",
         7: "chunk 1: FF1 k-loop", 8: "chunk 1: epilogue + barrier", 9: "chunk 1: FF2 k-loop + barrier",
         10: "remaining chunks", 11: "FF epilogue + output rows", 12: "q|k|v (LayerNorm moments, passes, stores)"}


def main():
    C, inner, nq = 384, 384, 1152
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    w = dict(w_out=r(C, inner, sc=inner ** -0.5), b_out=r(C), w1=r(4 * C, C, sc=C ** -0.5), b1=r(4 * C), p0=torch.exp(r(4 * C, sc=0.2)),
             p1=1.0 / (torch.exp(r(4 * C, sc=0.2)) + 1e-9), w2=r(C, 4 * C, sc=(4 * C) ** -0.5), b2=r(C))
    wq, bq = r(nq, C, sc=C ** -0.5), r(nq)
    for cfg in sys.argv[1:] or ["64:128:64", "64:128:10304", "32:128:5152", "48:256:10304"]:
        pair = cfg.endswith(":pair")                           # "48:256:5152:pair": the pair form (two workgroups per row tile)
        qb, ch, M = (int(v) for v in cfg.split(":")[:3])
        att, x = r(M, inner).cuda(), (r(M, C) * 2 + 0.3).cuda()
        for rep in range(2):                                   # second call: caches warm
            _, q = hip.tblock_chain(att, x, w["w_out"], w["b_out"], w["w1"], w["b1"], w["p0"], w["p1"], w["w2"], w["b2"], w_qkv=wq, b_qkv=bq, qb=qb, ch=ch, pair=pair)
        st = q.view(-1).view(torch.int64)[:16].cpu().tolist()
        print(f"== qb {qb} ch {ch} rows {M}{' PAIR form' if pair else ''} ({(M + qb - 1) // qb} row tiles): total {st[12] - st[0]} cycles")
        prev = st[0]
        for i in range(1, 12):
            print(f"   {st[i] - prev:8d}  {NAMES[i]}")
            prev = st[i]
        if pair:
            xs = q.view(-1).view(torch.int64)[4000:4008].cpu().tolist()
            names = ["partial stores issued", "stores landed (vmcnt 0)", "barrier", "release fence", "partner's flag seen", "acquire fence", "barrier", "partner's partial read, x2 formed"]
            print("   inside the exchange (cycles since its start):", "; ".join(f"{n} {xs[k] - xs[0]}" for k, n in enumerate(names) if k))
        nwg = (M + qb - 1) // qb
        allst = q.view(-1).view(torch.int64)[16:16 + 2 * nwg].cpu().view(nwg, 2)
        t0 = int(allst[:, 0].min())
        life = (allst[:, 1] - allst[:, 0]).double()
        end = (allst[:, 1] - t0).double()
        startd = (allst[:, 0] - t0).double()
        print(f"   all {nwg} workgroups: lifetime min / median / max {int(life.min())} / {int(life.median())} / {int(life.max())} cycles; "
              f"start spread {int(startd.max())}; last end {int(end.max())} after the first start")
        import os
        npf = int(os.environ.get("MTTS_CHAIN_PF", "8"))
        if npf:
            pfs = q.view(-1).view(torch.int64)[16 + 2 * nwg:16 + 2 * (nwg + npf)].cpu().view(npf, 2)
            print("   prefetch workgroups' lifetimes (cycles):", (pfs[:, 1] - pfs[:, 0]).tolist())
        rt = q.view(-1).view(torch.int64)[16 + 2 * (nwg + npf):18 + 2 * (nwg + npf)].cpu().tolist()
        print(f"   workgroup 0: {st[12] - st[0]} shader cycles in {rt[1] - rt[0]} ticks of the 100 MHz real-time counter = {(rt[1] - rt[0]) / 100.0:.1f} us "
              f"-> {(st[12] - st[0]) / max(1, rt[1] - rt[0]) * 0.1:.2f} GHz")
        for x in range(8):
            sel = torch.arange(nwg) % 8 == x
            st_x, en_x = allst[sel, 0], allst[sel, 1]
            order = torch.argsort(st_x)
            gaps = (st_x[order][1:] - st_x[order][:-1]).tolist()
            print(f"     XCD {x}: {int(sel.sum())} workgroups, lifetime median {int(life[sel].median())} max {int(life[sel].max())}; "
                  f"first start -> last start {int(st_x.max() - st_x.min())}, first start -> last end {int(en_x.max() - st_x.min())} cycles; start gaps {gaps[:8]}")
        for i, name in ((13, "q|k|v: LayerNorm moments + constants + pass 0 k-loop"), (14, "pass 0 epilogue + pass 1 k-loop"),
                        (15, "pass 1 epilogue + pass 2 k-loop"), (12, "pass 2 epilogue")):
            print(f"   {st[i] - prev:8d}  {name}")
            prev = st[i]


if __name__ == "__main__":
    main()
