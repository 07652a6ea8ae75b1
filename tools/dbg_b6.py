"""Compare gocc / dadj of the bf16x6 backward with the FP32-MFMA backward on one test shape (debug)."""
import os, sys, subprocess, torch
sys.path.insert(0, '.')
def run(variant):
    os.environ["MARIUS_SCORES"] = variant
    import importlib
    import tests.test_gpu_parity as T
    from marius_amd import hip as H
    dev = torch.device("cuda:0")
    B, C, N, d = 1000, 10, 500, 100
    U, R = max(40, B), 11
    emb, state, edges, dst_neg, src_neg, rel, inv = T.make_batch("DISTMULT", B, C, N, d, U, R, seed=B + d, zipf=False)
    W = T.run_hip_lp(H, dev, "DISTMULT", emb, edges, dst_neg, src_neg, rel, inv, True, "sum")
    torch.cuda.synchronize()
    return W.gocc()[:, :d].cpu().clone(), W.neg(0).cpu().clone()
if len(sys.argv) > 1:
    g, s = run(sys.argv[1])
    torch.save((g, s), "/tmp/dbg_%s.pt" % sys.argv[1])
else:
    for v in ("p", "b"):
        subprocess.check_call([sys.executable, __file__, v])
    gp, sp = torch.load("/tmp/dbg_p.pt"); gb, sb = torch.load("/tmp/dbg_b.pt")
    print("S max abs diff", (sp - sb).abs().max().item())
    diff = (gp - gb).abs()
    tol = 1e-4 * gp.abs().max()
    bad = (diff > tol).nonzero()
    print("gocc shape", tuple(gp.shape), "bad entries", bad.shape[0], "max diff", diff.max().item(), "scale", gp.abs().max().item())
    rows = sorted(set(bad[:, 0].tolist()))
    print("bad rows (first 40):", rows[:40])
    for r, c in bad[:10].tolist():
        print(r, c, gp[r, c].item(), gb[r, c].item())
