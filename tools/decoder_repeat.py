#!/usr/bin/env python3
"""GPU diagnostic: ONE estimator evaluation (mtts_decoder_forward) repeated on the same inputs -- bitwise repeatable?  Where not:
    python tools/decoder_repeat.py [--batch 32] [--frames 320] [--runs 200]"""
import argparse, importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def analyse_snapshot(lib, M, C):
    import ctypes
    nbytes = M * C * 4
    ha, hb = torch.empty(nbytes, dtype=torch.uint8), torch.empty(nbytes, dtype=torch.uint8)
    sid = lib.mtts_debug_verify_snapshot(ctypes.c_void_p(ha.data_ptr()), ctypes.c_void_p(hb.data_ptr()), ctypes.c_size_t(nbytes))
    if not sid:
        return
    def unpack(t):
        h = t.view(torch.float16).view(M, C // 32, 2, 32).float()
        return (h[:, :, 0, :] + h[:, :, 1, :] / 2048.0).reshape(M, C)
    a, b = unpack(ha), unpack(hb)
    if hasattr(lib, "mtts_debug_probe_read"):
        nw = (M + 47) // 48
        pa, pb = (ctypes.c_uint * (8 * nw))(), (ctypes.c_uint * (8 * nw))()
        got = lib.mtts_debug_probe_read(pa, pb, nw)
        if got:
            dd = (a - b).abs().amax(1)
            rows = []
            for w in range(got):
                bad = bool((dd[48 * w:48 * w + 48] > 0).any())
                ra, rb = pa[8 * w:8 * w + 8], pb[8 * w:8 * w + 8]
                rows.append((w, bad, ra, rb))
            def hw(v):          # HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
                return f"se{(v >> 13) & 7}.sh{(v >> 12) & 1}.cu{(v >> 8) & 15}"
            nbad = sum(1 for r in rows if r[1])
            ct_diff = sum(1 for w, bad, ra, rb in rows if ra[2] != rb[2])
            st_diff = sum(1 for w, bad, ra, rb in rows if ra[4] != rb[4])
            ka_diff = sum(1 for w, bad, ra, rb in rows if ra[6] != rb[6] or ra[3] != 0 or ra[5] not in (0, 8))
            print(f"  probe: {nbad} bad workgroups of {got}; constants hash differs (exec1 vs exec2) in {ct_diff}, row statistics hash in {st_diff}, "
                  f"kernel-argument hash anomalies in {ka_diff}")
            both = sum(1 for w, bad, ra, rb in rows if bad and ra[2] != rb[2])
            print(f"  bad workgroups with a differing constants hash: {both}; good workgroups with a differing constants hash: {ct_diff - both}")
            by_xcc = {}
            for w, bad, ra, rb in rows:
                k = ra[1] & 15
                by_xcc.setdefault(k, [0, 0])[1 if bad else 0] += 1
            print("  exec1 workgroups per XCC (good, bad):", {k: tuple(v) for k, v in sorted(by_xcc.items())})
            print("  first bad workgroups (wg, exec1 place, exec2 place):", [(w, f"x{ra[1] & 15}." + hw(ra[0]), f"x{rb[1] & 15}." + hw(rb[0])) for w, bad, ra, rb in rows if bad][:12])
            print("  first good workgroups:", [(w, f"x{ra[1] & 15}." + hw(ra[0]), f"x{rb[1] & 15}." + hw(rb[0])) for w, bad, ra, rb in rows if not bad][:12])
    d = (a - b)
    bad_rows = torch.nonzero(d.abs().amax(1) > 0).flatten()
    print(f"  snapshot of slot {sid}: {bad_rows.numel()} rows differ; |exec2| rms {float(b.pow(2).mean().sqrt()):.3f}; diff rms over bad rows "
          f"{float(d[bad_rows].pow(2).mean().sqrt()):.3e} max {float(d.abs().max()):.3e}")
    db = d[bad_rows]
    per_tile = db.abs().view(-1, C // 16, 16).mean(dim=(0, 2))
    print("  mean |diff| per 16-channel tile (wave = tile // 3):", " ".join(f"{v:.1e}" for v in per_tile.tolist()))
    rit = bad_rows % 48
    prof = [float(d[bad_rows[rit == k]].abs().mean()) if bool((rit == k).any()) else 0.0 for k in range(48)]
    print("  mean |diff| per row-in-workgroup:", " ".join(f"{v:.1e}" for v in prof))
    colmean, colstd = db.mean(0), db.std(0)
    print(f"  per-column diff: |mean over rows| avg {float(colmean.abs().mean()):.3e}, std over rows avg {float(colstd.mean()):.3e}")
    wg = (bad_rows // 48)
    cnt = torch.bincount(wg, minlength=(M + 47) // 48)
    print("  bad rows per workgroup (first 40):", cnt[:40].tolist())
    for r in bad_rows[:3].tolist():
        print(f"  row {r}: exec1 {[round(v, 4) for v in a[r, :6].tolist()]} exec2 {[round(v, 4) for v in b[r, :6].tolist()]}")
    nz = (d[bad_rows] != 0).float().mean()
    print(f"  fraction of elements differing within bad rows: {float(nz):.3f}; relative diff rms {float(d[bad_rows].pow(2).mean().sqrt() / b[bad_rows].pow(2).mean().sqrt()):.3e}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--runs", type=int, default=200)
    ap.add_argument("--verify", action="store_true", help="a -DMTTS_CHAIN_VERIFY build (tools/build_variant.sh): read its per-launch comparison")
    args = ap.parse_args()
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=1)
    sd = synthetic.make_state_dict(hp, seed=7)
    m = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    hip = m.hip
    B, T = args.batch, args.frames
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 100, T, generator=g).to(dev)
    mu = torch.randn(B, 100, T, generator=g).to(dev)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :].to(dev)
    first, bad = None, 0
    import ctypes
    vbuf = (ctypes.c_int * (8 * 64))()
    for r in range(args.runs):
        out = hip.decoder_forward(x, mask, mu, 0.37)
        torch.cuda.synchronize()
        if args.verify:
            sect = None
            if hasattr(hip.lib, "mtts_debug_verify_sections"):
                sbuf = (ctypes.c_int * (8 * 8))()
                if hip.lib.mtts_debug_verify_sections(sbuf, 8):
                    sect = list(sbuf)
            n = hip.lib.mtts_debug_verify_read(vbuf, 64)
            if sect is not None:
                for k in range(n // 2):
                    if any(sect[8 * k:8 * k + 6]):
                        print(f"run {r}: chain launch {k}: LDS dumps of executions 1 and 2 differ in 16-byte chunks per section "
                              f"[x0, ct, x1, srow, h0, x2] = {sect[8 * k:8 * k + 6]}; first ct chunk {sect[8 * k + 6]}, first srow chunk {sect[8 * k + 7]}", flush=True)
            for k in range(n):
                cnt, r0, r1, c0, c1 = vbuf[8 * k:8 * k + 5]
                if cnt:
                    print(f"run {r}: chain launch {k // 2} of {n // 2}, executions {1 + (k & 1)} and {2 + (k & 1)} differ in {cnt} 16-byte chunks, rows {r0}..{r1} "
                          f"(workgroups {r0 // 48}..{r1 // 48}), chunks {c0}..{c1}", flush=True)
        if args.verify and n and any(vbuf[8 * k] for k in range(0, n, 2)):
            analyse_snapshot(hip.lib, B * T, 384)
        if first is None:
            first = out.clone()
            print(f"run 0: finite {bool(torch.isfinite(out).all())} flags {hip.range_flags().tolist()}", flush=True)
            continue
        d = (out - first).abs()
        if not bool((d > 0).any()):
            continue
        bad += 1
        per_b = d.amax(dim=(1, 2))
        ub = torch.nonzero(per_b > 0).flatten().tolist()
        desc = []
        for b in ub[:10]:
            fr = torch.nonzero(d[b].amax(dim=0) > 0).flatten()
            desc.append(f"b={b} (len {int(lens[b])}) frames {int(fr.min())}..{int(fr.max())} ({fr.numel()}) rows {b * T + int(fr.min())}.. max {float(per_b[b]):.2e}")
        print(f"run {r}: DIFFERS in {len(ub)} utterances, {int((d > 0).sum())} values; " + "; ".join(desc), flush=True)
    print(f"result: {bad} of {args.runs - 1} repeats differ", flush=True)


if __name__ == "__main__":
    main()
